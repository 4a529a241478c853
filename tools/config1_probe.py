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
    g.close()
    return n, (t1 - t0) * 1e3, (t2 - t1) * 1e3


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

for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n, seg_ms, ro_ms = run_graph()
    torch.cuda.synchronize()
    print("window %d: %.1f ms (segment %.1f, read-out %.1f), %d regions" % (rep, (time.perf_counter() - t0) * 1e3, seg_ms, ro_ms, n))
