"""3840x2160 stream: wall / merge time and device memory per chunk (debugging aid)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
import video_segment_amd as vsg
W, H, chunk = 3840, 2160, 20
N = int(sys.argv[1]) if len(sys.argv) > 1 else 39
dev = torch.device("cuda")
s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
fl = torch.from_numpy(synth.const_flow(W, H)).cuda()
tl = time.time()
for k in range(N):
    frame = synth.frame_torch("bench", W, H, k, dev)
    torch.cuda.synchronize()   # (the handle reads device inputs on its own stream: the producer has to be complete)
    n = s.process_frame(frame, fl if k > 0 else None)
    if n:
        now = time.time()
        d = s.last_diagnostics()
        print("4K k=%d wall %.1f ms merge %.1f in_use %.2f GB stages %d rollbacks %d slab growths %d spine growths %d" % (
            k, (now - tl) * 1e3, s.last_timings().merge_ms, vsg.memory_stats(0)["bytes_in_use"] / 1e9, d["stages"],
            d["rollbacks"], d["slab_growths"], d["spine_pool_growths"]), flush=True)
        tl = time.time()
