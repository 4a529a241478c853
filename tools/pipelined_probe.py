"""Chunk completion intervals of PipelinedDenseSegmentation next to a single stream (debug aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import synth
import video_segment_amd as vsg

W, H, N, chunk = (int(a) for a in sys.argv[1:5])
kind = sys.argv[5] if len(sys.argv) > 5 else "bench"
dev = torch.device("cuda")
frames = [synth.frame_torch(kind, W, H, k, dev) for k in range(N)]
fl = torch.from_numpy(synth.const_flow(W, H)).to(dev)
torch.cuda.synchronize()
p = vsg.PipelinedDenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
t0 = time.perf_counter()
got = 0
for k in range(N):
    got += p.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
dt = time.perf_counter() - t0
st = np.array(p.stamps)
print("pipelined: %d frames in %.3f s -> %.1f fps; chunk intervals (ms): %s"
      % (got, dt, N / dt, np.round(np.diff(st) * 1e3, 1)))
p.close()
s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
t0 = time.perf_counter()
stamps = []
for k in range(N):
    if s.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1)):
        stamps.append(time.perf_counter())
dt = time.perf_counter() - t0
print("single:    %d frames in %.3f s -> %.1f fps; chunk intervals (ms): %s"
      % (N, dt, N / dt, np.round(np.diff(np.array(stamps)) * 1e3, 1)))
s.close()
