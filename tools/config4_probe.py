"""BASELINE configs[4] on one GPU (debug / profiling aid): the dense unit on the device, the
hierarchical unit on the host, one after the other per frame, with the time of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
import video_segment_amd as vsg

W, H, N, chunk = (int(a) for a in sys.argv[1:5])
dev = torch.device("cuda")
frames = [synth.frame_torch("soft", W, H, k, dev) for k in range(N)]
host = [f.cpu().numpy() for f in frames]
flh = synth.const_flow(W, H)
fl = torch.from_numpy(flh).to(dev)
d = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
r = vsg.RegionSegmentation(W, H, vsg.default_region_options())
td = tr = 0.0
fed = 0
for k in range(N):
    t0 = time.perf_counter()
    n = d.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
    segs = [d.result_bytes(i) for i in range(n)]
    t1 = time.perf_counter()
    td += t1 - t0
    for j, seg in enumerate(segs):
        r.process_frame(seg, host[fed], flh if fed > 0 else None, flush=(k == N - 1 and j == len(segs) - 1))
        fed += 1
    tr += time.perf_counter() - t1
print("dense %.1f ms/frame, hierarchy %.1f ms/frame" % (td / N * 1e3, tr / N * 1e3))
d.close()
r.close()
