"""Debug helper: runs a few graph parity cases under the VSG_WAVE_DBG toggles of the wave worker
(1 no chain, 4 no hot region, 8 one generic lane per round, 16 chain self check, 32 no jumping, 64 one chain lane per round) and reports merge
statistics / partition equality against the oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(96, 64, 4, "smooth", False), (64, 48, 6, "probe", True), (48, 40, 5, "noise", True)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import video_segment_amd as vsg
    import test_gpu_parity as tp
    for (W, H, F, kind, flow) in CASES:
        gg, og, minsz, flows = tp.build_pair(vsg, W, H, F, kind, flow, seed=3)
        gg.segment(minsz, False)
        og.segment(minsz, False)
        same = np.array_equal(tp.canon_partition(gg.node_roots()), tp.canon_partition(og.node_roots()))
        print("  %s: stats %s vs %s partition_equal=%s" % ((W, H, F, kind, flow), gg.merge_stats(),
                                                           og.merge_stats(), same))
    sys.exit(0)

for dbg in (sys.argv[1:] or ["0", "1", "4", "8", "16", "32", "64"]):
    env = dict(os.environ, VSG_WAVE_DBG=dbg)
    print("VSG_WAVE_DBG=%s" % dbg, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=300)
