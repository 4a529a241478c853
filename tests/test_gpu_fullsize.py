"""Full-size GPU checks: the reference-derived pins reproduced by the HIP path
itself at 640x480 and 1920x1080, BASELINE config 2 through the graph seam, and size-independent
properties at 1080p (determinism, chain == stream, rasters partition every frame)."""
import hashlib

import numpy as np
import pytest

import oracle_lib as ol
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsg():
    import video_segment_amd as v
    from video_segment_amd import _lib
    _lib.build()
    assert _lib.lib().vsg_device_count() > 0
    return v


def hip_probe(vsg, W, H, N, flow, frame_fn=synth.probe_frame, chunk=20):
    s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=flow)
    fl = synth.const_flow(W, H) if flow else None
    planes, sha = [], []
    for k in range(N):
        n = s.process_frame(frame_fn(W, H, k), fl if (flow and k > 0) else None, flush=(k == N - 1))
        for i in range(n):
            planes.append(s.result_id_image(i))
            sha.append(hashlib.sha256(s.result_bytes(i)).hexdigest())
    s.close()
    return planes, sha


def test_hip_reproduces_reference_pin_vga(vsg):
    planes, _ = hip_probe(vsg, 640, 480, 22, False)
    assert len(planes) == 22 and synth.fnv1a32_fast(planes) == 0x8D5C857C


def test_hip_reproduces_reference_pin_1080p(vsg):
    planes, _ = hip_probe(vsg, 1920, 1080, 22, True)
    assert len(planes) == 22 and synth.fnv1a32_fast(planes) == 0xDF411091
    assert len(np.unique(planes[0])) == 90      # 90 Region2D in frame 0


def test_config2_spatial_only_graph(vsg):
    """BASELINE configs[1]: 640x480, 32-slice window, spatial-only dense graph through the
    DenseSegGraphInterface seam; reference-derived pin 288 regions / 1184 neighbour links, and
    identical region index images / sizes as the oracle."""
    W, H, F = 640, 480, 32
    g = vsg.DenseSegGraph(W, H, F)
    o = ol.OracleGraph(W, H, F)
    for k in range(F):
        frame = synth.probe_frame(W, H, k)
        g.add_frame_bgr(frame)
        o.add_frame(ol.preprocess(frame))
    g.segment(983, False)
    o.segment(983, False)
    assert np.array_equal(g.merge_stats(), o.merge_stats())
    g.obtain_results(use_flows=False)
    o.obtain_results(None, True, True)
    assert (g.num_regions(), g.num_neighbor_links()) == (288, 1184)
    gs, gc = g.region_sizes()
    os_, oc = o.region_sizes()
    assert np.array_equal(gs, os_) and np.array_equal(gc, oc)
    for t in (0, 15, 31):
        assert np.array_equal(g.index_image(t), o.index_image(t))


def test_1080p_bench_properties(vsg):
    """No oracle at this size: determinism, chain == continuous stream, and every output is a
    partition of the frame with consistent sizes."""
    import torch
    from video_segment_amd.multi_gpu import local_transport, run_chain
    W, H, N, chunk = 1920, 1080, 41, 20
    _, sha1 = hip_probe(vsg, W, H, N, True, synth.bench_frame, chunk)
    _, sha2 = hip_probe(vsg, W, H, N, True, synth.bench_frame, chunk)
    assert sha1 == sha2 and len(sha1) == N
    fl = synth.const_flow(W, H)
    dev = torch.device("cuda", 0)
    got = run_chain(
        lambda: vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True),
        lambda k: synth.bench_frame(W, H, k), lambda k: fl, N, chunk, W, H, 0, 1,
        local_transport(W, H, dev))
    assert [hashlib.sha256(b).hexdigest() for _, b in got] == sha1
    # partition property on a decoded frame of the second chunk
    from test_proto_wire import build_schema
    m = build_schema()()
    m.ParseFromString(got[25][1])
    cover = np.zeros((H, W), np.int32)
    for r in m.region:
        area = 0
        for iv in r.raster.scan_inter:
            cover[iv.y, iv.left_x:iv.right_x + 1] += 1
            area += iv.right_x - iv.left_x + 1
        assert r.shape_moments.size == float(area)
    assert (cover == 1).all()
