"""Localises GPU/oracle divergences on a constrained chunk graph (debug aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol, synth
import video_segment_amd as vsg
from test_gpu_parity import rand_frame, canon_partition

W, H, chunk, seed = 64, 48, 8, 5
kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
rng = np.random.default_rng(seed)
frames = [rand_frame(rng, W, H, kind) for _ in range(16)]
fl = synth.const_flow(W, H)
s = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
ids = []
for k in range(chunk):
    n = s.process_frame(frames[k], fl if k > 0 else None)
    ids += [s.result_id_image(i) for i in range(n)]
print("chunk0 outputs", len(ids))
L_virt, L_cons = ids[-2], ids[-1]
minsz = int(np.float32(0.01) * np.float32(W) * np.float32(0.01) * np.float32(H) * np.float32(chunk))
print("minsz", minsz)

def build():
    gg = vsg.DenseSegGraph(W, H, chunk + 1)
    og = ol.OracleGraph(W, H, chunk + 1)
    gg.add_virtual_frame(L_virt); og.add_virtual_frame(L_virt)
    f = ol.preprocess(frames[6])
    gg.add_frame_bgr(frames[6], constraint_ids=L_cons); og.add_frame(f, L_cons)
    gg.add_temporal(fl, is_virtual=True); og.add_temporal(None, None, fl, True)
    prev = f
    flows = [None, fl]
    for k in range(7, 7 + chunk - 1):
        f = ol.preprocess(frames[k])
        gg.add_frame_bgr(frames[k]); og.add_frame(f)
        gg.add_temporal(fl); og.add_temporal(f, prev, fl)
        flows.append(fl)
        prev = f
    return gg, og, flows

for force in (False, True):
    gg, og, flows = build()
    gg.segment(minsz, force); og.segment(minsz, force)
    print("force", force, "stats gpu", gg.merge_stats(), "oracle", og.merge_stats())
    a, b = canon_partition(gg.node_roots()), canon_partition(og.node_roots())
    print("  partition equal:", np.array_equal(a, b), "ndiff", int((a != b).sum()),
          "regions", a.max() + 1, b.max() + 1)
    if force:
        gg.obtain_results(use_flows=True); og.obtain_results(flows)
        print("  regions", gg.num_regions(), og.num_regions(), "links", gg.num_neighbor_links(), og.num_neighbor_links())
        gs, gc = gg.region_sizes(); os_, oc = og.region_sizes()
        n = min(len(gs), len(os_))
        print("  sizes equal", np.array_equal(gs[:n], os_[:n]), "cons equal", np.array_equal(gc[:n], oc[:n]))
        if not np.array_equal(gs[:n], os_[:n]):
            d = np.nonzero(gs[:n] != os_[:n])[0]
            print("   first size diffs idx", d[:10], gs[d[:10]], os_[d[:10]])
        if not np.array_equal(gc[:n], oc[:n]):
            d = np.nonzero(gc[:n] != oc[:n])[0]
            print("   first cons diffs idx", d[:10], gc[d[:10]], oc[d[:10]])
        for t in range(chunk + 1):
            gi, oi = gg.index_image(t), og.index_image(t)
            print("   slice", t, "index image equal", np.array_equal(gi, oi), int((gi != oi).sum()))
