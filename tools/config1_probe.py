"""BASELINE configs[1] on its own: 640x480, 32-slice window, spatial-only graph through seam 3 (what
bench.py's `configs[1]` leg times): ms per window over a few repetitions, each on a fresh graph."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import synth  # noqa: E402
import video_segment_amd as vsg  # noqa: E402

cw, chh, cf = 640, 480, 32
dev = torch.device("cuda", 0)
frames = [synth.frame_torch("bench", cw, chh, k, dev) for k in range(cf)]


def run_graph():
    g = vsg.DenseSegGraph(cw, chh, cf, device=0)
    for f in frames:
        g.add_frame_bgr(f)
    g.finish_building()
    t0 = time.perf_counter()
    g.segment(983, False)
    t1 = time.perf_counter()
    g.obtain_results(use_flows=False)
    n = g.num_regions()
    t2 = time.perf_counter()
    d = g.diagnostics()
    g.close()
    return n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, d


if "--after-stream" in sys.argv:   # the state bench.py's leg runs in: a 1080p stream has come and gone
    W, H, chunk = 1920, 1080, 20
    fl = torch.from_numpy(synth.const_flow(W, H)).to(dev)
    fr = [synth.frame_torch("bench", W, H, k, dev) for k in range(chunk + 19 * 2)]
    st = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
    for k, f in enumerate(fr):
        st.process_frame(f, fl if k > 0 else None)
    st.close()
    del fr, st
    print("(after a 1080p stream of three chunks)")

reps = 10
for a in sys.argv[1:]:
    if a.startswith("--reps="):
        reps = int(a.split("=")[1])
all_ms = []
for rep in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n, seg_ms, ro_ms, d = run_graph()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    all_ms.append(ms)
    print("window %d: %.1f ms (segment %.1f, read-out %.1f), %d regions | stages %d, slab growths %d (%.1f ms), "
          "spine growths %d (%.1f ms), hipMalloc %d (%.1f ms), hipFree %d (%.1f ms), cache hits %d, device syncs %d "
          "(%.2f ms), mail waits %d (%.1f ms, longest %.2f), mode %d" % (
              rep, ms, seg_ms, ro_ms, n, d["stages"], d["slab_growths"], d["slab_growth_ms"],
              d["spine_pool_growths"], d["spine_pool_growth_ms"], d["runtime_mallocs"], d["runtime_malloc_ms"],
              d["runtime_frees"], d["runtime_free_ms"], d["cache_hits"], d["device_syncs"], d["device_sync_ms"],
              d["mail_waits"], d["mail_wait_ms"], d["mail_wait_longest_ms"], d["mail_mode"]))
s_ = sorted(all_ms[1:])
print("windows 1..%d: min %.1f median %.1f max %.1f ms -> %.0f frames/s at the median; memory %s" % (
    reps - 1, s_[0], s_[len(s_) // 2], s_[-1], cf / (s_[len(s_) // 2] * 1e-3), vsg.memory_stats(0)))
