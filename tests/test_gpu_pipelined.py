"""PipelinedDenseSegmentation (video_segment_amd/pipelined.py): one video over two chunk engines on
one GPU -- every SegmentationDesc byte-identical to a single stream (and through it to the oracle,
tests/test_gpu_parity.py), whatever the interleaving of the two engine threads."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsg():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import video_segment_amd as v
    from video_segment_amd import _lib
    _lib.build()
    return v


def single(vsg, frames, flow, W, H, chunk):
    s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=flow is not None)
    out = []
    for k, f in enumerate(frames):
        n = s.process_frame(f, flow if (flow is not None and k > 0) else None, flush=(k == len(frames) - 1))
        out += [s.result_bytes(i) for i in range(n)]
    s.close()
    return out


@pytest.mark.parametrize("W,H,N,chunk,kind,flow", [
    (96, 64, 50, 8, "bench", True),      # 7 chunks, the last one short
    (64, 48, 29, 8, "probe", True),      # the video ends exactly with a chunk
    (80, 60, 23, 8, "noise", False),     # no flow stream
    (64, 48, 5, 8, "bench", True),       # shorter than one chunk
])
def test_pipelined_is_byte_identical_to_a_single_stream(vsg, W, H, N, chunk, kind, flow):
    import torch
    dev = torch.device("cuda")
    if kind == "probe":
        frames = [torch.from_numpy(synth.probe_frame(W, H, k)).to(dev) for k in range(N)]
    else:
        frames = [synth.frame_torch(kind, W, H, k, dev) for k in range(N)]
    fl = torch.from_numpy(synth.const_flow(W, H)).to(dev) if flow else None
    want = single(vsg, frames, fl, W, H, chunk)
    assert len(want) == N
    for wait in (False, True):
        p = vsg.PipelinedDenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=flow)
        got = []
        for k, f in enumerate(frames):
            n = p.process_frame(f, fl if (flow and k > 0) else None, flush=(k == N - 1), wait=wait)
            got += [p.result_bytes(i) for i in range(n)]
        p.close()
        assert len(got) == N
        for k in range(N):
            assert got[k] == want[k], "frame %d differs (wait=%s)" % (k, wait)
