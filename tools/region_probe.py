"""Host timing of the hierarchical stage (vsg_regionseg_*) on an over-segmentation from the CPU
oracle (no GPU needed): python tools/region_probe.py W H N chunk  ->  ms per frame, and the
oracle's own time next to it with --oracle; --cache=<file> keeps the over-segmentation between runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
import synth
import video_segment_amd as vsg
from video_segment_amd import _lib

W, H, N, chunk = (int(a) for a in sys.argv[1:5])
_lib.build()
fl = synth.const_flow(W, H)
ol.set_threads(8)
frames = [synth.soft_frame(W, H, k) for k in range(N)]
cache = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--cache=")]
if cache and os.path.exists(cache[0]):
    import pickle
    segs = pickle.load(open(cache[0], "rb"))
else:
    o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
    segs = []
    t0 = time.time()
    for k in range(N):
        n = o.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
        segs += [o.result_bytes(i) for i in range(n)]
    o.close()
    print("oracle dense: %.1f ms/frame" % ((time.time() - t0) * 1e3 / N))
    if cache:
        import pickle
        pickle.dump(segs, open(cache[0], "wb"))
for name, mk in (("product", lambda: vsg.RegionSegmentation(W, H, vsg.default_region_options())),
                 ("oracle", lambda: ol.OracleRegionSegmentation(W, H, ol.region_options()))):
    if name == "oracle" and "--oracle" not in sys.argv:
        continue
    r = mk()
    t0 = time.time()
    outs = 0
    per = []
    for k in range(N):
        t1 = time.time()
        outs += r.process_frame(segs[k], frames[k], fl if k > 0 else None, flush=(k == N - 1))
        per.append((time.time() - t1) * 1e3)
    print("%s hierarchy: %.1f ms/frame (median frame %.1f, slowest %.1f), %d results"
          % (name, (time.time() - t0) * 1e3 / N, float(np.median(per)), max(per), outs))
    r.close()
