"""Long streams (24 to 62 chunks) at small frame sizes against the oracle, byte for byte: what only shows
after many chunks -- constraints that accumulate, constrained splits, stages cut at the edges that break
their assumptions -- at sizes the oracle replays in a minute.  ~4 minutes on a GPU box;
with VSG_DEBUG_STATS=1 the library reports how often each mechanism ran."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import video_segment_amd as vsg
from test_gpu_parity import run_streams
import stress_parity as sp
t0 = time.time()
for (W, H, N, kind, chunk) in [(320, 240, 260, "noise", 10), (256, 144, 400, "noise", 20), (320, 200, 300, "smooth", 10),
                               (384, 216, 240, "bench", 20), (200, 150, 500, "noise", 8)]:
    t1 = time.time()
    run_streams(vsg, W, H, N, kind, True, chunk)
    print("ok", W, H, N, kind, chunk, "%.1f s" % (time.time() - t1), flush=True)
# long streams of the stress generator's kinds (blocks / twotone: constrained splits)
rng = np.random.default_rng(9)
for kind in ("blocks", "twotone", "noise"):
    W, H, N, chunk = 240, 160, 240, 10
    frames = [sp.make_frame(rng, W, H, kind, k) for k in range(N)]
    t1 = time.time()
    run_streams(vsg, W, H, N, kind, True, chunk, frames=frames)
    print("ok long", kind, "%.1f s" % (time.time() - t1), flush=True)
print("all identical, %.1f s" % (time.time() - t0))
