"""PipelinedDenseSegmentation with two ORACLE engines (no GPU): the frame routing (overlap frames go
to two engines, the flush frame to one), the late import of the hand-off, the ordered hand-out of
the results and the shutdown paths of video_segment_amd/pipelined.py -- byte-identical to one
oracle stream.  The product engines run the same code on the GPU (tests/test_gpu_pipelined.py)."""
import numpy as np
import pytest

import oracle_lib as ol
import synth
from video_segment_amd.pipelined import PipelinedDenseSegmentation


def frames_of(kind, W, H, N):
    if kind == "probe":
        return [synth.probe_frame(W, H, k) for k in range(N)]
    rng = np.random.default_rng(4)
    return [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(N)]


def single(frames, flow, W, H, chunk):
    s = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=flow is not None)
    out, counts = [], []
    for k, f in enumerate(frames):
        n = s.process_frame(f, flow if (flow is not None and k > 0) else None, flush=(k == len(frames) - 1))
        out += [s.result_bytes(i) for i in range(n)]
        counts.append(n)
    s.close()
    return out, counts


@pytest.mark.parametrize("W,H,N,chunk,kind,flow,wait", [
    (48, 36, 30, 8, "probe", True, False),     # 4 full chunks + a short one
    (48, 36, 29, 8, "probe", True, True),      # the video ends exactly with a chunk; results in step
    (40, 30, 17, 8, "noise", False, False),    # no flow stream
    (40, 30, 5, 8, "probe", True, False),      # shorter than a chunk
    (40, 30, 8, 8, "noise", True, False),      # exactly one chunk
    (40, 30, 9, 8, "noise", True, True),       # one frame into the second chunk
])
def test_two_oracle_engines_equal_one_stream(W, H, N, chunk, kind, flow, wait):
    frames = frames_of(kind, W, H, N)
    fl = synth.const_flow(W, H) if flow else None
    want, counts = single(frames, fl, W, H, chunk)
    opts = ol.default_options(chunk_size=chunk)
    p = PipelinedDenseSegmentation(
        W, H, opts, has_flow=flow,
        engine_factory=lambda: ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=flow),
        halo_of=lambda e: e.export_halo())
    got, seen_early = [], 0
    for k, f in enumerate(frames):
        n = p.process_frame(f, fl if (flow and k > 0) else None, flush=(k == N - 1), wait=wait)
        if wait:   # what a single stream does: the results of a chunk with the frame that completes it
            assert n == counts[k], (k, n, counts[k])
        seen_early += n if k < N - 1 else 0
        got += [p.result_bytes(i) for i in range(n)]
    p.close()
    assert got == want


def test_close_in_the_middle_of_a_video():
    W, H, chunk = 40, 30, 8
    frames = frames_of("probe", W, H, 12)
    p = PipelinedDenseSegmentation(
        W, H, ol.default_options(chunk_size=chunk), has_flow=False,
        engine_factory=lambda: ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=False),
        halo_of=lambda e: e.export_halo())
    for f in frames:
        p.process_frame(f)
    p.close()      # engines may be waiting for frames or for a hand-off: must not hang


class _FailingEngine:
    """Stream interface whose third frame raises: what a device fault in one engine looks like."""

    def __init__(self):
        self.frames = 0

    def process_frame(self, frame, flow, flush=False):
        self.frames += 1
        if self.frames == 3:
            raise RuntimeError("engine failed")
        return 0

    def restart(self):
        pass

    def expect_halo(self):
        pass

    def import_halo(self, *a):
        pass

    def result_bytes(self, i):
        return b""

    def close(self):
        pass


@pytest.mark.timeout(60)
def test_failed_engine_reaches_a_fast_caller_and_close_does_not_hang():
    """A caller that feeds faster than the engines consume sits in a full input queue most of the
    time; when an engine dies it has to see the error there (not block for ever), and closing the
    unit afterwards has to return.  The engine threads do not keep the unit alive either."""
    import gc
    import weakref
    W, H, chunk = 40, 30, 8
    frames = frames_of("probe", W, H, 2)
    p = PipelinedDenseSegmentation(W, H, ol.default_options(chunk_size=chunk), has_flow=False,
                                   engine_factory=_FailingEngine, halo_of=lambda e: (None, None, None))
    with pytest.raises(RuntimeError, match="engine failed"):
        for k in range(400):
            p.process_frame(frames[k % 2])
    p.close()
    q = PipelinedDenseSegmentation(W, H, ol.default_options(chunk_size=chunk), has_flow=False,
                                   engine_factory=_FailingEngine, halo_of=lambda e: (None, None, None))
    ref = weakref.ref(q)
    threads = list(q._threads)
    del q
    gc.collect()
    assert ref() is None            # collected although nobody called close() ...
    for t in threads:
        t.join(10)
        assert not t.is_alive()     # ... and its __del__ stopped the engine threads
