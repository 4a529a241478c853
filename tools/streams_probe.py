"""S streams on one GPU from S host threads: per-stream stage times (mean per chunk) next to the one-stream
numbers, host cores, mailbox waits.  python tools/streams_probe.py S [chunks]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
import video_segment_amd as vsg

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H, chunk = 1920, 1080, 20
N = 20 + 19 * CH
dev = torch.device("cuda")
frames = [synth.frame_torch("bench", W, H, k, dev) for k in range(N)]
fl = torch.from_numpy(synth.const_flow(W, H)).cuda()
torch.cuda.synchronize()
print("host cores:", os.cpu_count(), "usable:", len(os.sched_getaffinity(0)), flush=True)
streams = [vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True) for _ in range(S)]
rows = [[] for _ in range(S)]


def run(si):
    s = streams[si]
    t_last = time.perf_counter()
    for k in range(N):
        n = s.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
        if n:
            for i in range(n):
                s.result_bytes(i)
            t = s.last_timings()
            d = s.last_diagnostics()
            now = time.perf_counter()
            rows[si].append(((now - t_last) * 1e3, t.preprocess_ms, t.edges_ms, t.merge_ms, t.readout_ms, t.host_post_ms,
                             t.filter_kernel_ms, t.wave_kernel_ms, t.spine_kernel_ms, d["mail_waits"], d["mail_wait_ms"],
                             d["mail_mode"]))
            t_last = now


t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i,)) for i in range(S)]
[t.start() for t in th]
[t.join() for t in th]
wall = time.perf_counter() - t0
print("S=%d: %.1f frames/s/GPU" % (S, S * N / wall))
print("stream  wall  pre edges merge readout host | filter wave spine | waits wait_ms mode   (mean of the steady chunks)")
for si in range(S):
    r = rows[si][1:] or rows[si]
    m = [sum(x[j] for x in r) / len(r) for j in range(len(r[0]))]
    print("%6d %6.1f %4.1f %5.1f %5.1f %6.1f %5.1f | %6.1f %5.1f %5.1f | %5.0f %6.1f %4.0f" % ((si,) + tuple(m)))
