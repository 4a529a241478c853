"""Randomised differential test of the HIP stream against the oracle (run on a GPU box).

    python tools/stress_parity.py [num_cases] [seed]

Random frame sizes, chunk sizes, content kinds (noise / smooth / blocks / bench), flow fields and
stream lengths; every serialized SegmentationDesc must be byte-identical.  With VSG_DEBUG_STATS=1 the
library prints the merge worker's counters (rounds, chain merges, chain cuts), which shows that
the rare paths (failed chain tests, finalized hot regions, constrained partners) are exercised."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
import synth  # noqa: E402
import video_segment_amd as vsg  # noqa: E402


def make_frame(rng, W, H, kind, k):
    if kind == "noise":
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if kind == "smooth":
        x = np.linspace(0, 255, W)[None, :, None]
        y = np.linspace(0, 255, H)[:, None, None]
        return np.clip(0.5 * x + 0.5 * y + rng.normal(0, 2.0, (H, W, 3)), 0, 255).astype(np.uint8)
    if kind == "blocks":   # flat blocks of random colours + mild noise, moving
        bw, bh = max(4, W // 6), max(4, H // 5)
        gx = (np.arange(W)[None, :] + 3 * k) // bw
        gy = np.arange(H)[:, None] // bh
        base = ((gx * 37 + gy * 91) % 7) * 36
        img = np.stack([base, (base * 2) % 255, 255 - base], -1).astype(np.float64)
        return np.clip(img + rng.normal(0, 1.5, (H, W, 3)), 0, 255).astype(np.uint8)
    if kind == "twotone":  # two large flat regions with a contrast edge: failed chain tests
        img = np.zeros((H, W, 3), np.float64)
        img[:, : W // 2] = 60
        img[:, W // 2:] = 60 + rng.integers(8, 40)
        return np.clip(img + rng.normal(0, 0.7, (H, W, 3)), 0, 255).astype(np.uint8)
    return synth.bench_frame(W, H, k)


def one_case(rng, idx, scale=None, max_px_frames=None, two_stage=False):
    """scale: sizes up to 200 x 130 times this; max_px_frames: the stream is cut to this many
    pixel-frames (the CPU oracle does about 3 M per second); two_stage: a third of the cases run
    with two_stage_oversegment."""
    if scale is None:
        scale = float(os.environ.get("STRESS_SCALE", "1"))
    W = int(rng.integers(24, int(200 * scale)))
    H = int(rng.integers(16, int(130 * scale)))
    chunk = int(rng.choice([8, 9, 10, 13, 20]))
    N = int(rng.integers(1, 3 * chunk + 3))
    if max_px_frames:
        N = max(1, min(N, int(max_px_frames // (W * H))))
    kind = str(rng.choice(["noise", "smooth", "blocks", "twotone", "bench"]))
    flow_kind = str(rng.choice(["none", "const", "random"]))
    has_flow = flow_kind != "none"
    # (drawn last so that the cases of earlier rounds keep their inputs)
    extra = {}
    if os.environ.get("STRESS_OPTIONS", "1") != "0":
        orng = np.random.default_rng([int(rng.integers(0, 1 << 30)), 7])
        if orng.random() < 0.35:
            extra["presmoothing"] = 0          # unfiltered features: many failed tests, finalized regions
        if orng.random() < 0.25:
            extra["color_distance"] = 0        # L1
        if two_stage and orng.random() < 0.34:
            extra["two_stage_oversegment"] = 1
    go = vsg.default_options(chunk_size=chunk, **extra)
    oo = ol.default_options(chunk_size=chunk, **extra)
    gs = vsg.DenseSegmentation(W, H, go, has_flow=has_flow)
    os_ = ol.OracleStream(W, H, oo, has_flow=has_flow)
    total = 0
    for k in range(N):
        frame = make_frame(rng, W, H, kind, k)
        fl = None
        if has_flow and k > 0:
            fl = synth.const_flow(W, H) if flow_kind == "const" else \
                rng.normal(0, 3.0, (H, W, 2)).astype(np.float32)
        last = k == N - 1
        ng = gs.process_frame(frame, fl, flush=last)
        no = os_.process_frame(frame, fl, flush=last)
        assert ng == no, (idx, k, ng, no)
        for i in range(no):
            if gs.result_bytes(i) != os_.result_bytes(i):
                gi, oi = gs.result_id_image(i), os_.result_id_image(i)
                raise AssertionError("case %d (%dx%d N=%d chunk=%d %s flow=%s): frame result %d of call %d "
                                     "differs: %d px differ, len %d vs %d, merge stats %s vs %s" %
                                     (idx, W, H, N, chunk, kind, flow_kind, i, k, int((gi != oi).sum()),
                                      len(gs.result_bytes(i)), len(os_.result_bytes(i)),
                                      gs.last_merge_stats(), os_.last_merge_stats()))
        total += no
    assert total == N
    gs.close()
    os_.close()
    return W, H, N, chunk, kind, flow_kind, extra


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    failures = 0
    for idx in range(n):
        # every case gets its own generator so that a failure does not shift the later cases
        crng = np.random.default_rng([seed, idx])
        try:
            desc = one_case(crng, idx)
            print("case %3d ok %s" % (idx, desc), flush=True)
        except AssertionError as e:
            failures += 1
            print("FAIL %s" % e, flush=True)
    print("%d of %d CASES IDENTICAL in %.1f s" % (n - failures, n, time.time() - t0))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
