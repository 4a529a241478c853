import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import video_segment_amd as vsg
import synth
for (W, H, N, chunk) in [(320, 240, 260, 10), (200, 150, 300, 8)]:
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(N)]
    fl = synth.const_flow(W, H)
    s = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True)
    t0 = time.time(); merge = 0; nb = 0
    for k in range(N):
        n = s.process_frame(frames[k], fl if k > 0 else None, flush=(k == N - 1))
        if n:
            merge += s.last_timings().merge_ms; nb += 1
    print(W, H, N, chunk, "GPU path %.2f s, merge %.1f ms per chunk over %d chunks" % (time.time() - t0, merge / nb, nb), flush=True)
