"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Integer / index results and the f32 feature planes must be bit-exact; the serialized
SegmentationDesc messages (ids, scan intervals, shape moments, hierarchy, neighbours, chunk
fields) must be byte-identical.
"""
import numpy as np
import pytest

import oracle_lib as ol
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vsg():
    import video_segment_amd as v
    from video_segment_amd import _lib
    _lib.build()
    assert _lib.lib().vsg_device_count() > 0, "GPU tests need a HIP device"
    return v


def rand_frame(rng, W, H, kind):
    if kind == "noise":
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if kind == "const":
        return np.full((H, W, 3), 77, np.uint8)
    if kind == "smooth":
        x = np.linspace(0, 255, W)[None, :, None]
        y = np.linspace(0, 255, H)[:, None, None]
        img = 0.5 * x + 0.5 * y + rng.normal(0, 2.0, (H, W, 3))
        return np.clip(img, 0, 255).astype(np.uint8)
    raise ValueError(kind)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def canon_partition(labels):
    """Relabels by first occurrence so that two partitions can be compared exactly."""
    _, first, inv = np.unique(labels, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first))
    return order[inv]


@pytest.mark.parametrize("W,H,kind,pad", [(64, 48, "noise", 0), (161, 97, "smooth", 5),
                                          (70, 20, "const", 2), (320, 240, "smooth", 0)])
def test_bilateral_bit_exact(vsg, W, H, kind, pad):
    rng = np.random.default_rng(1)
    buf = np.zeros((H, W * 3 + pad), np.uint8)
    frame = rand_frame(rng, W, H, kind)
    buf[:, :W * 3] = frame.reshape(H, W * 3)
    view = np.lib.stride_tricks.as_strided(buf, (H, W, 3), (buf.strides[0], 3, 1))
    g = vsg.DenseSegGraph(W, H, 2)
    g.add_frame_bgr(view, presmoothing=2)
    got = g.smoothed(0)
    want = ol.preprocess(view, 2)
    assert np.array_equal(bits(got), bits(want))
    g2 = vsg.DenseSegGraph(W, H, 2)
    g2.add_frame_bgr(view, presmoothing=0)
    assert np.array_equal(bits(g2.smoothed(0)), bits(ol.preprocess(view, 0)))


@pytest.mark.parametrize("W,H,l1", [(64, 48, False), (161, 97, False), (64, 48, True)])
def test_edge_buckets_bit_exact(vsg, W, H, l1):
    rng = np.random.default_rng(2)
    f0 = rng.random((H, W, 3), dtype=np.float32)
    f1 = (f0 + rng.normal(0, 0.02, (H, W, 3))).astype(np.float32)
    flow = rng.normal(0, 3.0, (H, W, 2)).astype(np.float32)
    flow[0, 0] = (1e12, -1e12)      # absurd values: x86 float->int semantics
    flow[1, 1] = (np.nan, 0.5)
    g = vsg.DenseSegGraph(W, H, 3, l1=l1)
    g.add_frame_features(f0)
    g.add_frame_features(f1)
    g.add_temporal(flow)
    g.add_frame_features(f0)
    g.add_temporal(None)
    assert np.array_equal(g.spatial_buckets(0), ol.spatial_buckets(f0, l1))
    assert np.array_equal(g.spatial_buckets(1), ol.spatial_buckets(f1, l1))
    tb, pidx = g.temporal_buckets(1)
    wtb, wpidx = ol.temporal_buckets(f1, f0, flow, l1)
    assert np.array_equal(pidx, wpidx)
    assert np.array_equal(tb, wtb)
    tb2, pidx2 = g.temporal_buckets(2)
    wtb2, wpidx2 = ol.temporal_buckets(f0, f1, None, l1)
    assert np.array_equal(pidx2, wpidx2) and np.array_equal(tb2, wtb2)


def build_pair(vsg, W, H, F, kind, flow, seed, chunk_for_min=20):
    rng = np.random.default_rng(seed)
    gg = vsg.DenseSegGraph(W, H, F)
    og = ol.OracleGraph(W, H, F)
    fl = synth.const_flow(W, H) if flow else None
    prev = None
    flows = [None]
    for t in range(F):
        frame = rand_frame(rng, W, H, kind) if kind != "probe" else synth.probe_frame(W, H, t)
        gg.add_frame_bgr(frame)
        feat = ol.preprocess(frame)
        og.add_frame(feat)
        if t > 0:
            gg.add_temporal(fl)
            og.add_temporal(feat, prev, fl)
            flows.append(fl)
        prev = feat
    minsz = int(np.float32(0.01) * np.float32(W) * np.float32(0.01) * np.float32(H) *
                np.float32(chunk_for_min))
    return gg, og, minsz, (flows if flow else None)


@pytest.mark.parametrize("W,H,F,kind,flow", [(64, 48, 4, "probe", False), (64, 48, 6, "probe", True),
                                             (48, 40, 5, "noise", True), (96, 64, 4, "smooth", False),
                                             (40, 30, 3, "const", False)])
def test_graph_merge_and_readout(vsg, W, H, F, kind, flow):
    gg, og, minsz, flows = build_pair(vsg, W, H, F, kind, flow, seed=3)
    gg.segment(minsz, False)
    og.segment(minsz, False)
    assert np.array_equal(gg.merge_stats(), og.merge_stats())
    assert np.array_equal(canon_partition(gg.node_roots()), canon_partition(og.node_roots()))
    gg.obtain_results(use_flows=flow)
    og.obtain_results(flows)
    assert gg.num_regions() == og.num_regions()
    gs, gc = gg.region_sizes()
    os_, oc = og.region_sizes()
    assert np.array_equal(gs, os_) and np.array_equal(gc, oc)
    for t in range(F):
        assert np.array_equal(gg.index_image(t), og.index_image(t)), t
    assert gg.num_neighbor_links() == og.num_neighbor_links()


def assert_region_lists_equal(gg, og, F):
    """The whole RegionInfoList through the C ABI (vsg_graph_get_regions / _get_intervals) against the
    oracle's: index, size, constrained id, frame span, sorted neighbour lists, per-frame scan
    intervals in rasterization order (dense_seg_graph_interface.h:147-158)."""
    gr, gp, gi = gg.get_regions()
    orr, op, oi = og.get_regions()
    assert gr.shape == orr.shape and np.array_equal(gr, orr)
    assert np.array_equal(gp, op) and np.array_equal(gi, oi)
    n = len(gr)
    assert np.array_equal(gr[:, 0], np.arange(n))
    for i in range(n):   # neighbour lists: sorted, unique, symmetric
        nb = gi[gp[i]:gp[i + 1]]
        assert np.all(np.diff(nb) > 0)
    sym = set()
    for i in range(n):
        for j in gi[gp[i]:gp[i + 1]]:
            sym.add((i, int(j)))
    assert all((j, i) in sym for (i, j) in sym)
    for t in range(F):
        a, b = gg.get_intervals(t), og.get_intervals(t)
        assert a.shape == b.shape and np.array_equal(a, b), t
        if len(a):   # consistent with the region table's frame spans
            spans = gr[a[:, 0]]
            assert np.all(spans[:, 3] <= t) and np.all(spans[:, 4] >= t)


@pytest.mark.parametrize("W,H,F,kind,flow", [(64, 48, 6, "probe", True), (48, 40, 5, "noise", True),
                                             (96, 64, 4, "smooth", False), (40, 30, 3, "const", False)])
def test_seam3_region_list_through_c_abi(vsg, W, H, F, kind, flow):
    gg, og, minsz, flows = build_pair(vsg, W, H, F, kind, flow, seed=11)
    gg.segment(minsz, False)
    og.segment(minsz, False)
    gg.obtain_results(use_flows=flow)
    og.obtain_results(flows)
    assert_region_lists_equal(gg, og, F)


def test_seam3_region_list_constrained_chunk(vsg):
    """A second-chunk graph (virtual slice + constrained slice + virtual temporal edges) driven
    through seam 3 the way DenseSegmentation drives it (dense_segmentation.cpp:291-331): regions
    carry constrained ids, the virtual slice has no rasterization."""
    W, H = 64, 48
    rng = np.random.default_rng(4)
    labels_v = (np.arange(W * H).reshape(H, W) // (W * 6) * 4 + (np.arange(W)[None, :] // 16)).astype(np.int32)
    labels_c = np.roll(labels_v, 2, axis=1).astype(np.int32)
    gg = vsg.DenseSegGraph(W, H, 5)
    og = ol.OracleGraph(W, H, 5)
    gg.add_virtual_frame(labels_v)
    og.add_virtual_frame(labels_v)
    fl = synth.const_flow(W, H)
    prev = None
    for t in range(4):
        frame = synth.probe_frame(W, H, t)
        feat = ol.preprocess(frame)
        gg.add_frame_bgr(frame, constraint_ids=labels_c if t == 0 else None)
        og.add_frame(feat, labels_c if t == 0 else None)
        gg.add_temporal(fl, is_virtual=(t == 0))
        og.add_temporal(feat if t else None, prev, fl, is_virtual=(t == 0))
        prev = feat
    gg.segment(30, True)
    og.segment(30, True)
    assert np.array_equal(gg.merge_stats(), og.merge_stats())
    gg.obtain_results(use_flows=True)
    og.obtain_results([None, fl, fl, fl, fl])
    assert_region_lists_equal(gg, og, 5)
    regs, _, _ = gg.get_regions()
    assert (regs[:, 2] >= 0).any()
    assert len(gg.get_intervals(0)) == 0   # the virtual slice is never rasterized


def run_streams(vsg, W, H, N, kind, flow, chunk, seed=5, frames=None, **options):
    rng = np.random.default_rng(seed)
    go = vsg.default_options(chunk_size=chunk, **options)
    oo = ol.default_options(chunk_size=chunk, **options)
    gs = vsg.DenseSegmentation(W, H, go, has_flow=flow)
    os_ = ol.OracleStream(W, H, oo, has_flow=flow)
    fl = synth.const_flow(W, H) if flow else None
    total = 0
    for k in range(N):
        if frames is not None:
            frame = frames[k]
        elif kind == "probe":
            frame = synth.probe_frame(W, H, k)
        elif kind == "bench":
            frame = synth.bench_frame(W, H, k)
        else:
            frame = rand_frame(rng, W, H, kind)
        f = fl if (flow and k > 0) else None
        last = k == N - 1
        ng = gs.process_frame(frame, f, flush=last)
        no = os_.process_frame(frame, f, flush=last)
        assert ng == no, (k, ng, no)
        for i in range(no):
            gb, ob = gs.result_bytes(i), os_.result_bytes(i)
            if gb != ob:
                gi, oi = gs.result_id_image(i), os_.result_id_image(i)
                raise AssertionError(
                    "frame result %d of call %d differs: %d px differ, len %d vs %d" %
                    (i, k, int((gi != oi).sum()), len(gb), len(ob)))
        if no:
            assert np.array_equal(gs.last_merge_stats(), os_.last_merge_stats())
        total += no
    assert total == N
    gs.close()
    os_.close()


@pytest.mark.parametrize("W,H,N,kind,flow,chunk", [
    (64, 48, 8, "probe", False, 20),
    (64, 48, 8, "probe", True, 20),
    (64, 48, 45, "probe", True, 20),      # 3 chunks: virtual + constrained slices
    (64, 48, 30, "noise", True, 8),       # many chunks, heavy min-size merging
    (50, 36, 20, "const", False, 8),      # single region
    (64, 48, 1, "probe", False, 20),      # one frame
    (96, 64, 26, "smooth", True, 10),
    (128, 96, 24, "bench", True, 20),
])
def test_stream_byte_identical(vsg, W, H, N, kind, flow, chunk):
    run_streams(vsg, W, H, N, kind, flow, chunk)


@pytest.mark.parametrize("W,H,N,kind,chunk", [
    (64, 48, 12, "bench", 8),     # the golden case that caught an unsafe pass in round 3
    (96, 64, 30, "bench", 10),
    (80, 56, 27, "smooth", 8),
    (128, 96, 45, "bench", 20),
])
def test_unfiltered_features_kept_edges(vsg, W, H, N, kind, chunk):
    """presmoothing = NONE: noisy features, many failed tests -> finalized regions and *kept* edges
    everywhere.  The wave worker commits kept edges ahead of earlier kept edges (merge_wave.hip,
    res2 / res3): only past those that are committed in the same round -- a kept edge that waits
    can turn into a merge once its other region has changed."""
    run_streams(vsg, W, H, N, kind, True, chunk, presmoothing=0)


def test_stream_320x240_probe(vsg):
    run_streams(vsg, 320, 240, 22, "probe", True, 20)


def test_stream_pins_direct(vsg):
    """The HIP path itself reproduces the reference-derived pins."""
    W, H, N = 64, 48, 45
    s = vsg.DenseSegmentation(W, H, vsg.default_options(), has_flow=True)
    fl = synth.const_flow(W, H)
    planes = []
    for k in range(N):
        n = s.process_frame(synth.probe_frame(W, H, k), fl if k > 0 else None, flush=(k == N - 1))
        planes += [s.result_id_image(i) for i in range(n)]
    assert synth.fnv1a32_fast(planes) == 0x5EF008E2


def test_device_resident_inputs(vsg):
    """Frames and flow handed over as device pointers give the same bytes as host pointers."""
    import torch
    W, H, N = 64, 48, 24
    fl = synth.const_flow(W, H)
    a = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=10), has_flow=True)
    b = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=10), has_flow=True)
    fl_dev = torch.from_numpy(fl).cuda()
    for k in range(N):
        frame = synth.bench_frame(W, H, k)
        fr_dev = torch.from_numpy(frame).cuda()
        na = a.process_frame(frame, fl if k > 0 else None, flush=(k == N - 1))
        nb = b.process_frame(fr_dev, fl_dev if k > 0 else None, flush=(k == N - 1))
        assert na == nb
        for i in range(na):
            assert a.result_bytes(i) == b.result_bytes(i)


@pytest.mark.parametrize("W,H,N,kind,chunk", [(64, 48, 30, "noise", 8), (96, 64, 26, "smooth", 10),
                                              (128, 96, 44, "bench", 20)])
def test_optimistic_stage_rollback_is_exact(vsg, monkeypatch, W, H, N, kind, chunk):
    """Constrained chunks settle 'kept' edges optimistically and roll a stage back when a marked
    region changes its constraint.  Forcing the rollback of every optimistic stage, and disabling
    the optimisation altogether, must give the same bytes as the oracle."""
    monkeypatch.setenv("VSG_FORCE_ROLLBACK", "1")
    run_streams(vsg, W, H, N, kind, True, chunk)
    monkeypatch.delenv("VSG_FORCE_ROLLBACK")
    monkeypatch.setenv("VSG_INERT_MODE", "0")
    run_streams(vsg, W, H, N, kind, True, chunk)


def test_chain_protocol_on_gpu(vsg):
    """Fresh HIP engine per chunk + label-plane halo hand-off == one continuous oracle stream
    (the multi-GPU chain mode, run on one device)."""
    import torch
    from video_segment_amd.multi_gpu import local_transport, run_chain
    W, H, N, chunk = 64, 48, 40, 8
    dev = torch.device("cuda", 0)
    fl = synth.const_flow(W, H)
    got = run_chain(
        lambda: vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk), has_flow=True),
        lambda k: synth.bench_frame(W, H, k), lambda k: fl, N, chunk, W, H, 0, 1,
        local_transport(W, H, dev))
    o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
    want = []
    for k in range(N):
        n = o.process_frame(synth.bench_frame(W, H, k), fl if k > 0 else None, flush=(k == N - 1))
        want += [o.result_bytes(i) for i in range(n)]
    assert [k for k, _ in got] == list(range(N))
    assert [b for _, b in got] == want


def test_two_stage_oversegment_stream_and_graph(vsg):
    """two_stage_oversegment (dense_segmentation.h:71): SegmentGraphSpatially before
    SegmentFullGraph -- through the stream option and through the graph seam
    (vsg_graph_segment_spatially, dense_seg_graph_interface.h:138)."""
    for (W, H, N, kind, chunk) in [(96, 64, 26, "smooth", 10), (128, 96, 30, "bench", 12)]:
        g = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk, two_stage_oversegment=1),
                                  has_flow=True)
        o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk, two_stage_oversegment=1),
                            has_flow=True)
        plain = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
        rng = np.random.default_rng(3)
        fl = synth.const_flow(W, H)
        differs = False
        for k in range(N):
            frame = synth.bench_frame(W, H, k) if kind == "bench" else rand_frame(rng, W, H, kind)
            f = fl if k > 0 else None
            last = k == N - 1
            ng, no, npl = g.process_frame(frame, f, flush=last), o.process_frame(frame, f, flush=last), \
                plain.process_frame(frame, f, flush=last)
            assert ng == no == npl
            if ng:
                assert np.array_equal(g.last_merge_stats(), o.last_merge_stats())
            for i in range(ng):
                assert g.result_bytes(i) == o.result_bytes(i), (k, i)
                differs = differs or o.result_bytes(i) != plain.result_bytes(i)
        g.close()
        o.close()
        plain.close()
        assert differs, "the two-stage option had no effect on this input"
    # graph seam
    W, H, F = 96, 64, 5
    g = vsg.DenseSegGraph(W, H, F)
    o = ol.OracleGraph(W, H, F)
    feats = []
    for k in range(F):
        frame = synth.bench_frame(W, H, k)
        feat = ol.preprocess(frame)
        feats.append(feat)
        g.add_frame_bgr(frame)
        o.add_frame(feat)
        if k > 0:
            g.add_temporal(None, False)
            o.add_temporal(feats[k], feats[k - 1], None, False)
    g.finish_building()
    g.segment_spatially()
    o.segment_spatially()
    assert np.array_equal(g.merge_stats(), o.merge_stats())
    g.segment(50, False)
    o.segment(50, False)
    assert np.array_equal(g.merge_stats(), o.merge_stats())
    g.obtain_results(use_flows=False)
    o.obtain_results(None, True, True)
    assert g.num_regions() == o.num_regions()
    assert g.num_neighbor_links() == o.num_neighbor_links()
    for t in range(F):
        assert np.array_equal(g.index_image(t), o.index_image(t))
