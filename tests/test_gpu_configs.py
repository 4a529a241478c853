"""GPU parity at the sizes of BASELINE.json's configs ("next round" item 1).

* configs[2] (the bench workload): 1920x1080 bench generator + flow, chunk 20, 41 frames = the
  unconstrained first chunk, one steady-state constrained chunk and a flushed tail -- every
  SegmentationDesc byte and the merge statistics of every chunk against the CPU oracle.
* configs[3] shape: 1920x1080, chunk 32, >= 3 chunks through the chunk chain (run_chain, one
  device) against the continuous stream, plus the size-independent partition property.
* configs[0] stand-in: 272x480x120 stream with the source's row stride (width_step 816, SURVEY
  A.7-9) against the oracle.
* configs[4] over-segmentation half: 3840x2160 two-chunk property run (determinism, partition).
"""
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


def _stream_both(vsg, W, H, N, chunk, frame_fn, flow, pad_to=None):
    """Feeds the same frames to the HIP stream and the oracle; returns the number of compared
    messages.  pad_to: row stride in bytes of the frame buffer handed to both."""
    g = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=flow)
    o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=flow)
    fl = synth.const_flow(W, H) if flow else None
    total = 0
    for k in range(N):
        frame = frame_fn(W, H, k)
        if pad_to is not None:
            buf = np.zeros((H, pad_to), np.uint8)
            buf[:, :W * 3] = frame.reshape(H, W * 3)
            frame = np.lib.stride_tricks.as_strided(buf, (H, W, 3), (pad_to, 3, 1))
        f = fl if (flow and k > 0) else None
        ng = g.process_frame(frame, f, flush=(k == N - 1))
        no = o.process_frame(frame, f, flush=(k == N - 1))
        assert ng == no, (k, ng, no)
        if ng:
            assert np.array_equal(g.last_merge_stats(), o.last_merge_stats()), k
        for i in range(ng):
            assert g.result_bytes(i) == o.result_bytes(i), "SegmentationDesc differs at %d/%d" % (k, i)
        total += ng
    g.close()
    o.close()
    return total


def test_bench_workload_1080p_vs_oracle(vsg):
    """The bench's own workload at full size, two chunk boundaries + flush, byte for byte."""
    assert _stream_both(vsg, 1920, 1080, 41, 20, synth.bench_frame, True) == 41


def test_config0_standin_272x480_vs_oracle(vsg):
    """configs[0] stand-in (test_video.MOV is 272x480, rows of 816 bytes, 120 frames; no H.264
    decoder in the image, so the bench generator supplies the pixels)."""
    assert _stream_both(vsg, 272, 480, 120, 20, synth.bench_frame, False, pad_to=816) == 120


def test_config0_standin_padded_rows(vsg):
    """A width whose BGR24 rows need padding to a multiple of 4 bytes."""
    assert _stream_both(vsg, 271, 96, 30, 10, synth.bench_frame, True, pad_to=816) == 30


def _decode(bytes_):
    from test_proto_wire import build_schema
    m = build_schema()()
    m.ParseFromString(bytes_)
    return m


def _check_partition(msg, W, H):
    cover = np.zeros((H, W), np.int32)
    for r in msg.region:
        area = 0
        for iv in r.raster.scan_inter:
            cover[iv.y, iv.left_x:iv.right_x + 1] += 1
            area += iv.right_x - iv.left_x + 1
        assert r.shape_moments.size == float(area)
    assert (cover == 1).all()


def test_config3_shape_chain_vs_stream(vsg):
    """configs[3] shape on one device: 1080p, chunk 32, 3 full chunks + tail; the chunk chain
    (what N GPUs run, here world = 1) reproduces the continuous stream byte for byte."""
    import torch
    from video_segment_amd.multi_gpu import product_halo, run_chain
    W, H, chunk = 1920, 1080, 32
    N = 31 * 3 + 6
    fl = synth.const_flow(W, H)
    s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
    sha = []
    for k in range(N):
        n = s.process_frame(synth.bench_frame(W, H, k), fl if k > 0 else None, flush=(k == N - 1))
        sha += [hashlib.sha256(s.result_bytes(i)).hexdigest() for i in range(n)]
    s.close()
    assert len(sha) == N
    dev = torch.device("cuda", 0)
    got = run_chain(
        lambda: vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True),
        lambda k: synth.bench_frame(W, H, k), lambda k: fl, N, chunk, W, H, 0, 1, None,
        from_engine_halo=lambda e: product_halo(e, W, H, dev))
    assert [hashlib.sha256(b).hexdigest() for _, b in got] == sha
    for idx in (0, 40, 95):
        _check_partition(_decode(got[idx][1]), W, H)
    m = _decode(got[31][1])          # first frame of the second chunk carries its hierarchy
    assert m.chunk_size == 31 and m.hierarchy_frame_idx == 31 and len(m.hierarchy) == 1


def test_config4_overseg_3840x2160_properties(vsg):
    """configs[4], over-segmentation half: 4K + flow, two chunk boundaries; determinism and the
    partition property (the oracle needs minutes at this size)."""
    import torch
    W, H, N, chunk = 3840, 2160, 24, 12
    dev = torch.device("cuda", 0)
    fl = torch.from_numpy(synth.const_flow(W, H)).to(dev)
    frames = [torch.from_numpy(synth.bench_frame(W, H, k)).to(dev) for k in range(N)]
    runs = []
    for _ in range(2):
        s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
        out = []
        for k in range(N):
            n = s.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
            out += [s.result_bytes(i) for i in range(n)]
        s.close()
        runs.append(out)
    assert len(runs[0]) == N and runs[0] == runs[1]
    _check_partition(_decode(runs[0][0]), W, H)
    _check_partition(_decode(runs[0][15]), W, H)
