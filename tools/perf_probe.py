"""Per-chunk stage timings of the HIP path on the bench input (debug / profiling aid)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import synth
import video_segment_amd as vsg

W, H, N, chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kind = sys.argv[5] if len(sys.argv) > 5 else "bench"
# "varflow": the headline frames with the spatially varying backward flow of synth.var_flow (a field per frame)
var = kind == "varflow"
if var:
    kind = "bench"
s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
fl = torch.from_numpy(synth.const_flow(W, H)).cuda()
flows = [synth.flow_torch("var", W, H, k, torch.device("cuda")) for k in range(N)] if var else None
frames = [synth.frame_torch(kind, W, H, k, torch.device("cuda")) if kind in synth.FRAME_FNS
          else torch.from_numpy(synth.probe_frame(W, H, k)).cuda() for k in range(N)]
torch.cuda.synchronize()
t0 = time.time(); tl = t0
for k in range(N):
    n = s.process_frame(frames[k], (flows[k] if var else fl) if k > 0 else None, flush=(k == N - 1))
    if n:
        t = s.last_timings()
        now = time.time()
        print("k=%d out=%d wall=%.3fs | pre %.1f edges %.1f merge %.1f (filter %.1f wave %.1f spine %.1f) readout %.1f host %.1f ms | edges %d merges %d stats %s"
              % (k, n, now - tl, t.preprocess_ms, t.edges_ms, t.merge_ms, t.filter_kernel_ms, t.wave_kernel_ms,
                 t.spine_kernel_ms, t.readout_ms, t.host_post_ms, t.edges_total, t.merges, s.last_merge_stats()),
              flush=True)
        tl = now
print("total %.3fs -> %.2f fps" % (time.time() - t0, N / (time.time() - t0)))
