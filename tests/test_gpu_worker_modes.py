"""GPU parity of the merge worker's mechanisms and of awkward stream shapes.

The round-based wave worker has debug switches that disable its mechanisms
one by one (VSG_WAVE_DBG bit mask), a self check that replays every committed chain with the
generic edge code.  Every mode must give the oracle's bytes.  Shapes: odd sizes, padded rows, tiny frames, random (out of range) flow."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import synth
from test_gpu_parity import rand_frame, run_streams, vsg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

CASES = [(96, 64, 26, "smooth", 10), (128, 96, 44, "bench", 20), (64, 48, 30, "noise", 8)]


@pytest.mark.parametrize("mode", ["1", "4", "8", "32", "64", "9"])
def test_worker_mechanisms_off_one_by_one(vsg, monkeypatch, mode):
    monkeypatch.setenv("VSG_WAVE_DBG", mode)
    for (W, H, N, kind, chunk) in CASES:
        run_streams(vsg, W, H, N, kind, True, chunk)


def test_chain_self_check_is_silent(vsg, monkeypatch, capfd):
    """Bit 16: every committed chain is replayed with DecideEdge inside the kernel and compared
    (means bit for bit, sizes, flags, constraints, merge statistics class)."""
    monkeypatch.setenv("VSG_WAVE_DBG", "16")
    for (W, H, N, kind, chunk) in CASES + [(256, 144, 44, "bench", 20)]:
        run_streams(vsg, W, H, N, kind, True, chunk)
    err = capfd.readouterr().err
    assert "self check" not in err, err


@pytest.mark.parametrize("env", [{"VSG_RLE": "0"}, {"VSG_WINDOWS": "1"},
                                 {"VSG_WINDOWS": "5", "VSG_WINDOW_MIN": "1"},
                                 {"VSG_WINDOWS": "3", "VSG_WINDOW_MIN": "1", "VSG_RLE": "0"},
                                 {"VSG_WINDOWS": "4", "VSG_WINDOW_MIN": "1", "VSG_FORCE_ROLLBACK": "1"},
                                 {"VSG_GROUP_BUCKETS": "0"},
                                 {"VSG_SPINE_MIN": "0"},
                                 {"VSG_SPINE_MIN": "32", "VSG_SPINE_CHECK": "1"},
                                 {"VSG_SPINE_MIN": "32", "VSG_SPINE_CHECK": "1", "VSG_WINDOWS": "1",
                                  "VSG_GROUP_BUCKETS": "0"},
                                 {"VSG_SPINE_MIN": "48", "VSG_FORCE_ROLLBACK": "1"},
                                 {"VSG_SPINE_MIN": "32", "VSG_SPINE_MAX_EDGES": "4096"},
                                 # the streamed spine (prep / scan / chain / verify; two passes with one
                                 # batch of k_spine between them) on every tree replay, however small
                                 {"VSG_SPINE_MIN": "32", "VSG_SPINE_FAST_MIN": "0", "VSG_SPINE_CHECK": "1"},
                                 {"VSG_SPINE_MIN": "64", "VSG_SPINE_FAST_MIN": "0", "VSG_SPINE_FAST": "1"},
                                 {"VSG_SPINE_MIN": "32", "VSG_SPINE_FAST_MIN": "0", "VSG_SPINE_FAST": "3",
                                  "VSG_WINDOWS": "1", "VSG_GROUP_BUCKETS": "0"},
                                 {"VSG_SPINE_MIN": "48", "VSG_SPINE_FAST": "0"},
                                 # a pool of zeroed counters so small that it changes halves within a chunk
                                 {"VSG_SPINE_MIN": "32", "VSG_ZERO_POOL": "4096", "VSG_SPINE_CHECK": "1"},
                                 # Euler tours ranked by sampling (every 64th arc) whatever their length
                                 {"VSG_SPINE_MIN": "32", "VSG_RANK_SPLIT_MIN": "0", "VSG_SPINE_CHECK": "1"},
                                 # every round of the spanning forest answered before the next is launched
                                 {"VSG_SPINE_MIN": "32", "VSG_BOR_AHEAD": "0", "VSG_SPINE_CHECK": "1"},
                                 # the forest's edge list compacted whenever the rounds allow it, however
                                 # short it is: one round ahead of the host (after every round, its length
                                 # known on the device only) and synchronously (when half the list is settled)
                                 {"VSG_SPINE_MIN": "32", "VSG_BOR_COMPACT_MIN": "16", "VSG_SPINE_CHECK": "1"},
                                 {"VSG_SPINE_MIN": "32", "VSG_BOR_COMPACT_MIN": "16", "VSG_BOR_AHEAD": "0",
                                  "VSG_SPINE_CHECK": "1", "VSG_ZERO_POOL": "4096"},
                                 # the arrays that hold a stage's active edges start far too small and
                                 # grow inside the stages (the compaction is repeated)
                                 {"VSG_ACTIVE_CAP": "64"},
                                 {"VSG_ACTIVE_CAP": "64", "VSG_SPINE_MIN": "32", "VSG_FORCE_ROLLBACK": "1"},
                                 # the wide worker (merge_wide.hip: several wavefronts per component in
                                 # lock-step rounds; off by default) on every component it can take
                                 # hub regions off; violated stages never cut (rerun with the exclusion list / edge by
                                 # edge) or cut once; every stage cut whatever it costs
                                 {"VSG_HUBS": "0"},
                                 {"VSG_HUB_SPLITS": "0"},
                                 {"VSG_HUB_SPLITS": "1", "VSG_SPINE_MIN": "32"},
                                 {"VSG_HUB_SPLITS": "0", "VSG_FORCE_ROLLBACK": "1"},
                                 # the hand-written radix sort for every size / for none (csrc/radix_sort.hip)
                                 {"VSG_SORT_HAND": "0:2000000000", "VSG_SPINE_MIN": "32", "VSG_SPINE_CHECK": "1"},
                                 {"VSG_SORT_HAND": "1:0"},
                                 {"VSG_WIDE_MIN": "25"},
                                 {"VSG_WIDE_MIN": "40", "VSG_WIDE_WAVES": "2", "VSG_SPINE_MIN": "0"},
                                 {"VSG_WIDE_MIN": "25", "VSG_FORCE_ROLLBACK": "1", "VSG_SPINE_MIN": "64"}])
def test_stage_decomposition_variants(vsg, monkeypatch, env):
    """The stage driver's two decompositions are exact whatever their parameters: a bucket split
    into consecutive rank windows (each its own filter -> components -> replay), runs of equal
    root pairs replayed through their leader only (with the rollback of a constrained split), the
    buckets above the force-merge weight replayed as one edge sequence, and the large components
    replayed along their Kruskal tree (from which size, with its structure checked against a
    sequential replay on the host, with a scratch pool that is too small, with forced rollbacks)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for (W, H, N, kind, chunk) in CASES + [(256, 144, 44, "bench", 20)]:
        run_streams(vsg, W, H, N, kind, True, chunk)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_parity as sp
    for idx in range(8):   # (27 variants x 8 random cases; the GPU suite's budget: tests/conftest.py)
        sp.one_case(np.random.default_rng([77, idx]), idx)


@pytest.mark.parametrize("small", ["1", "128", "100000"])
def test_lane_worker_threshold(vsg, monkeypatch, small):
    """The size up to which a component is replayed by a single lane (k_merge_small) instead of a
    wavefront is a tuning knob: every value gives the same bytes (1: nearly everything on the wave
    worker, 100000: everything on single lanes)."""
    monkeypatch.setenv("VSG_SMALL_SEG", small)
    for (W, H, N, kind, chunk) in CASES:
        run_streams(vsg, W, H, N, kind, True, chunk)


@pytest.mark.parametrize("W,H,N,kind,flow,chunk", [
    (67, 45, 20, "noise", True, 8),       # odd sizes
    (33, 31, 19, "smooth", True, 8),
    (17, 9, 12, "noise", True, 8),        # tiny
    (9, 17, 9, "smooth", False, 8),
    (131, 71, 23, "bench", True, 10),
    (256, 144, 44, "bench", True, 20),    # large components: wave worker with constraints
])
def test_stream_shapes(vsg, W, H, N, kind, flow, chunk):
    run_streams(vsg, W, H, N, kind, flow, chunk)


def test_padded_rows_and_random_flow(vsg):
    """Frames handed over as views of wider buffers (width_step > 3 W, like the reference's padded
    VideoFrames) and a different random flow field per frame, including far out-of-range and
    non-finite vectors (x86 float->int semantics of the displaced pixel)."""
    W, H, N, chunk = 70, 50, 21, 8
    rng = np.random.default_rng(11)
    gs = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
    os_ = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
    for k in range(N):
        wide = np.zeros((H, W + 6, 3), np.uint8)
        wide[:] = rng.integers(0, 256, wide.shape, dtype=np.uint8)   # garbage in the padding
        wide[:, :W] = rand_frame(rng, W, H, "smooth")
        view = wide[:, :W]
        assert not view.flags["C_CONTIGUOUS"]
        fl = rng.normal(0, 6.0, (H, W, 2)).astype(np.float32)
        fl[0, 0] = (1e9, -1e9)
        fl[1, 2] = (np.nan, 3.0)
        fl[2, 1] = (np.inf, -np.inf)
        f = fl if k > 0 else None
        last = k == N - 1
        ng = gs.process_frame(view, f, flush=last)
        no = os_.process_frame(np.ascontiguousarray(view), f, flush=last)
        assert ng == no
        for i in range(no):
            assert gs.result_bytes(i) == os_.result_bytes(i), (k, i)
    gs.close()
    os_.close()


def test_invalid_arguments_are_rejected(vsg):
    from video_segment_amd._lib import VsgError
    with pytest.raises(VsgError):
        vsg.DenseSegmentation(0, 48, vsg.default_options(), has_flow=True)
    with pytest.raises(VsgError):
        vsg.DenseSegmentation(64, 48, vsg.default_options(chunk_size=5), has_flow=True)   # overlap needs >= 8
    s = vsg.DenseSegmentation(64, 48, vsg.default_options(), has_flow=True)
    s.process_frame(synth.probe_frame(64, 48, 0), None)
    with pytest.raises(VsgError):      # a stream created with flow needs a flow field from frame 1 on
        s.process_frame(synth.probe_frame(64, 48, 1), None)
    s.close()


def test_randomised_streams_against_oracle(vsg):
    """tools/stress_parity.py: random sizes, chunk sizes, content kinds, flows and lengths.
    Case (11, 30) is the regression for a region that acquires a constraint and is merged away
    within one batch of the wave worker (its own constraint field must still reach memory:
    MergeConstrainedRegions reads it per node)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_parity as sp
    sp.one_case(np.random.default_rng([11, 30]), 30)
    for idx in range(40):
        sp.one_case(np.random.default_rng([5, idx]), idx)


def test_randomised_streams_larger_sizes(vsg):
    """The same differential sweep at sizes up to about 600 x 390 (components large enough for the
    wave worker's chains, the Kruskal-tree replay and its streamed spine with the default
    thresholds), with two_stage_oversegment in a third of the cases: presmoothing on / off, L1 / L2,
    random / constant / no flow, chunk sizes 8 - 20.  Streams are cut to 2.5 M pixel-frames so that
    the oracle's share stays around a second per case."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_parity as sp
    seen = set()
    for idx in range(44):
        desc = sp.one_case(np.random.default_rng([2024, idx]), idx, scale=3.0, max_px_frames=2.5e6, two_stage=True)
        seen.add((desc[4], desc[5], tuple(sorted(desc[6].items()))))
    assert len(seen) >= 25, "the sweep is supposed to cover many option combinations"
