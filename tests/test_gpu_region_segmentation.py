"""BASELINE configs[4] end to end: the dense over-segmentation on the MI355X (vsg_stream_*) feeding
the hierarchical RegionSegmentation (vsg_regionseg_*, host), against the CPU oracle running both
stages -- every hierarchical SegmentationDesc byte for byte -- and the C++ unit tree
(seg_tree_synth --region_segmentation: source -> DenseSegmentationUnit -> RegionSegmentationUnit ->
sink, threaded pipeline) against the same oracle run."""
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
import synth
from test_region_segmentation import check_structure

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "video_segment_amd", "host")


@pytest.fixture(scope="module")
def vsg():
    import video_segment_amd as v
    from video_segment_amd import _lib
    _lib.build()
    assert _lib.lib().vsg_device_count() > 0
    return v


def oracle_pipeline(W, H, N, chunk, opts, frame_fn=synth.soft_frame):
    fl = synth.const_flow(W, H)
    o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
    r = ol.OracleRegionSegmentation(W, H, ol.region_options(**opts))
    frames = [frame_fn(W, H, k) for k in range(N)]
    out, fed = [], 0
    for k in range(N):
        n = o.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
        segs = [o.result_bytes(i) for i in range(n)]
        for j, seg in enumerate(segs):
            last = k == N - 1 and j == len(segs) - 1
            m = r.process_frame(seg, frames[fed], fl if fed > 0 else None, flush=last)
            assert m >= 0
            out += [r.result_bytes(i) for i in range(m)]
            fed += 1
    assert fed == N and len(out) == N
    return out


@pytest.mark.parametrize("W,H,N,chunk,opts", [
    (96, 64, 60, 8, dict(chunk_set_size=3, chunk_set_overlap=1, constraint_chunks=1, min_region_num=3)),
    (320, 240, 61, 20, dict(chunk_set_size=2, chunk_set_overlap=1, min_region_num=5)),
])
def test_gpu_overseg_plus_hierarchy_matches_oracle(vsg, W, H, N, chunk, opts):
    want = oracle_pipeline(W, H, N, chunk, opts)
    fl = synth.const_flow(W, H)
    d = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
    r = vsg.RegionSegmentation(W, H, vsg.default_region_options(**opts))
    frames = [synth.soft_frame(W, H, k) for k in range(N)]
    got, fed = [], 0
    for k in range(N):
        n = d.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
        segs = [d.result_bytes(i) for i in range(n)]
        for j, seg in enumerate(segs):
            last = k == N - 1 and j == len(segs) - 1
            m = r.process_frame(seg, frames[fed], fl if fed > 0 else None, flush=last)
            got += [r.result_bytes(i) for i in range(m)]
            fed += 1
    d.close()
    r.close()
    assert len(got) == N
    for k in range(N):
        assert got[k] == want[k], "hierarchical SegmentationDesc %d differs" % k
    assert check_structure(got, W, H) >= 2


def test_config4_shape_3840x2160_hierarchy_properties(vsg):
    """configs[4] at its own frame size: 3840x2160 + flow, 20 frames (one flushed over-segmentation
    chunk) through both stages; the oracle needs minutes for the hierarchy at this size, so the
    full-size run is checked through the size-independent properties (partition, forest, sizes) and
    determinism, and the over-segmentation half against the oracle in test_gpu_configs."""
    import torch
    W, H, N, chunk = 3840, 2160, 20, 20
    dev = torch.device("cuda", 0)
    fl_h = synth.const_flow(W, H)
    fl = torch.from_numpy(fl_h).to(dev)
    frames_h = [synth.soft_frame(W, H, k) for k in range(N)]
    runs = []
    for _ in range(2):
        d = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
        r = vsg.RegionSegmentation(W, H, vsg.default_region_options(min_region_num=5))
        segs = []
        for k in range(N):
            n = d.process_frame(torch.from_numpy(frames_h[k]).to(dev), fl if k > 0 else None, flush=(k == N - 1))
            segs += [d.result_bytes(i) for i in range(n)]
        d.close()
        out = []
        for k, seg in enumerate(segs):
            m = r.process_frame(seg, frames_h[k], fl_h if k > 0 else None, flush=(k == N - 1))
            out += [r.result_bytes(i) for i in range(m)]
        r.close()
        runs.append(out)
    assert len(runs[0]) == N and runs[0] == runs[1]
    assert check_structure(runs[0][:2] + runs[0][-1:], W, H) == 1


def test_unit_tree_with_region_segmentation_unit(vsg):
    """seg_tree_synth --region_segmentation (C++ VideoUnit tree, threaded pipeline): the label planes of
    the hierarchical output equal the oracle pipeline's."""
    subprocess.check_call(["make", "-C", HOST, "-s"])
    W, H, N, chunk = 96, 64, 60, 8
    opts = dict(chunk_set_size=3, chunk_set_overlap=1, min_region_num=3)
    want = oracle_pipeline(W, H, N, chunk, opts)
    from test_proto_wire import build_schema
    Msg = build_schema()
    planes, total_regions, nbytes = [], 0, 0
    for b in want:
        m = Msg()
        m.ParseFromString(b)
        img = np.full((H, W), -1, np.int32)
        for reg in m.region:
            for iv in reg.raster.scan_inter:
                img[iv.y, iv.left_x:iv.right_x + 1] = reg.id
        planes.append(img)
        total_regions += len(m.region)
        nbytes += len(b)
    m0 = Msg()
    m0.ParseFromString(want[0])
    for extra in (["--use_pipeline"], ["--nouse_pipeline"]):
        p = subprocess.run([os.path.join(HOST, "seg_tree_synth"), "--width", str(W), "--height", str(H),
                            "--frames", str(N), "--chunk_size", str(chunk), "--input", "soft", "--flow",
                            "--region_segmentation", "--chunk_set_size", "3", "--chunk_set_overlap", "1",
                            "--min_region_num", "3"] + extra, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        m = re.search(r"frames=(\d+) first_frame_regions=(\d+) total_regions=(\d+) label_fnv1a32=(\w+) bytes=(\d+)"
                      r".* hierarchy_levels=(\d+)", p.stdout)
        assert m, p.stdout
        assert int(m.group(1)) == N and int(m.group(3)) == total_regions and int(m.group(5)) == nbytes
        assert int(m.group(4), 16) == synth.fnv1a32_fast(planes)
        assert int(m.group(6)) == len(m0.hierarchy) >= 2


def test_bench_checker_through_both_stages_reports_invalid(vsg):
    """The headline input (un-softened checker) behind the GPU dense unit: neighbouring cells have
    disjoint Lab histograms, the reference aborts on its CHECK (region_segmentation_graph.cpp:165).
    The product's answer is the documented one, VSG_ERR_INVALID with the diagnosis -- from the same
    frame on which the oracle reports the abort -- and never a made-up hierarchy."""
    from video_segment_amd._lib import VsgError, VSG_ERR_INVALID
    W, H, N, chunk = 96, 64, 16, 8
    fl = synth.const_flow(W, H)
    d = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
    o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
    r = vsg.RegionSegmentation(W, H, vsg.default_region_options(min_region_num=3))
    ro = ol.OracleRegionSegmentation(W, H, ol.region_options(min_region_num=3))
    frames = [synth.bench_frame(W, H, k) for k in range(N)]
    fed, aborted = 0, False
    for k in range(N):
        n = d.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
        assert n == o.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
        for i in range(n):
            seg = d.result_bytes(i)
            assert seg == o.result_bytes(i)
            last = k == N - 1 and i == n - 1
            code = ro.process_frame(seg, frames[fed], fl if fed > 0 else None, flush=last)
            if code == -2:
                with pytest.raises(VsgError, match="the reference aborts") as e:
                    r.process_frame(seg, frames[fed], fl if fed > 0 else None, flush=last)
                assert e.value.code == VSG_ERR_INVALID
                aborted = True
                break
            assert r.process_frame(seg, frames[fed], fl if fed > 0 else None, flush=last) == code
            fed += 1
        if aborted:
            break
    assert aborted
    d.close()
    o.close()


def test_config4_first_chunk_set_against_oracle(vsg):
    """configs[4] at its own size against the oracle running BOTH stages: 3840x2160 + flow, twelve
    frames = one flushed over-segmentation chunk = one chunk set, every hierarchical
    SegmentationDesc byte for byte (40 s, most of it the CPU oracle)."""
    import torch
    W, H, N, chunk = 3840, 2160, 12, 20
    want = oracle_pipeline(W, H, N, chunk, dict(min_region_num=5))
    dev = torch.device("cuda", 0)
    fl_h = synth.const_flow(W, H)
    fl = torch.from_numpy(fl_h).to(dev)
    frames_h = [synth.soft_frame(W, H, k) for k in range(N)]
    d = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
    r = vsg.RegionSegmentation(W, H, vsg.default_region_options(min_region_num=5))
    segs = []
    for k in range(N):
        n = d.process_frame(torch.from_numpy(frames_h[k]).to(dev), fl if k > 0 else None, flush=(k == N - 1))
        segs += [d.result_bytes(i) for i in range(n)]
    d.close()
    got = []
    for k, seg in enumerate(segs):
        m = r.process_frame(seg, frames_h[k], fl_h if k > 0 else None, flush=(k == N - 1))
        got += [r.result_bytes(i) for i in range(m)]
    r.close()
    assert len(got) == N == len(want)
    for k in range(N):
        assert got[k] == want[k], "hierarchical SegmentationDesc %d differs at 3840x2160" % k
