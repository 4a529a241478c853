"""Hierarchical RegionSegmentation (BASELINE configs[4]): the product's host
implementation (vsg_regionseg_*, video_segment_amd/csrc/region_segmentation.cpp) against the
oracle's class-by-class restatement (oracle/vs_oracle_region.inc) -- every emitted
SegmentationDesc byte for byte: Region2D ids and rasters, all hierarchy levels with sizes,
neighbours, parents, children, frame spans, chunk-set fields, vector data.

Host code on both sides, so the parity tests run without a GPU: the over-segmentation fed to both
comes from the oracle stream (the GPU tests run the dense unit on the device first,
tests/test_gpu_region_segmentation.py).  Parity for this row is unpinned: the reference
ships no fixture for it, cv::cvtColor and the hash-map iteration orders come from outside its tree."""
import numpy as np
import pytest

import oracle_lib as ol
import synth
from test_proto_wire import build_schema


@pytest.fixture(scope="module")
def vsg():
    import video_segment_amd as v
    from video_segment_amd import _lib
    _lib.build()
    return v


def overseg(W, H, N, chunk, frame_fn, flow):
    """[(frame, flow or None, serialized SegmentationDesc)] from the CPU oracle stream."""
    o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=flow is not None)
    frames = [frame_fn(W, H, k) for k in range(N)]
    out = []
    for k in range(N):
        n = o.process_frame(frames[k], flow if (flow is not None and k > 0) else None, flush=(k == N - 1))
        out += [o.result_bytes(i) for i in range(n)]
    o.close()
    assert len(out) == N
    return [(frames[k], flow if (flow is not None and k > 0) else None, out[k]) for k in range(N)]


def run_both(vsg, W, H, feed, opt_kw):
    p = vsg.RegionSegmentation(W, H, vsg.default_region_options(**opt_kw))
    o = ol.OracleRegionSegmentation(W, H, ol.region_options(**opt_kw))
    got, want = [], []
    for k, (frame, fl, seg) in enumerate(feed):
        last = k == len(feed) - 1
        no = o.process_frame(seg, frame, fl, flush=last)
        assert no >= 0, "the reference aborts on this input"
        npd = p.process_frame(seg, frame, fl, flush=last)
        assert npd == no, (k, npd, no)
        for i in range(no):
            want.append(o.result_bytes(i))
            got.append(p.result_bytes(i))
    p.close()
    o.close()
    return got, want


def check_structure(msgs, W, H, expect_levels_at_least=2):
    """Size-independent properties of the output: every frame is partitioned by its Region2Ds, the
    hierarchy is a forest level by level (children / parents consistent, sizes add up)."""
    Msg = build_schema()
    hier_frames = 0
    for b in msgs:
        m = Msg()
        m.ParseFromString(b)
        cover = np.zeros((H, W), np.int32)
        for r in m.region:
            for iv in r.raster.scan_inter:
                cover[iv.y, iv.left_x:iv.right_x + 1] += 1
        assert (cover == 1).all()
        if len(m.hierarchy) == 0:
            continue
        hier_frames += 1
        assert len(m.hierarchy) >= expect_levels_at_least
        ids0 = {c.id for c in m.hierarchy[0].region}
        assert {r.id for r in m.region} <= ids0
        for l in range(len(m.hierarchy) - 1):
            lower = {c.id: c for c in m.hierarchy[l].region}
            upper = {c.id: c for c in m.hierarchy[l + 1].region}
            kids = {}
            for c in lower.values():
                assert c.parent_id in upper
                kids.setdefault(c.parent_id, []).append(c.id)
            for pid, c in upper.items():
                assert sorted(c.child_id) == sorted(kids.get(pid, []))
                assert c.size == sum(lower[k].size for k in c.child_id)
        sizes = [len(h.region) for h in m.hierarchy]
        assert sizes == sorted(sizes, reverse=True)
    return hier_frames


def test_lab_conversion_matches_oracle(vsg):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(vsg.bgr_to_lab(img), ol.bgr_to_lab(img))
    # every grey level and the primaries
    ramp = np.zeros((4, 256, 3), np.uint8)
    ramp[0] = np.arange(256)[:, None]
    ramp[1, :, 0] = np.arange(256)
    ramp[2, :, 1] = np.arange(256)
    ramp[3, :, 2] = np.arange(256)
    lab = vsg.bgr_to_lab(ramp)
    assert np.array_equal(lab, ol.bgr_to_lab(ramp))
    assert lab[0, 0, 0] == 0 and lab[0, 255, 0] == 255            # L of black / white
    assert abs(int(lab[0, 128, 1]) - 128) <= 1 and abs(int(lab[0, 128, 2]) - 128) <= 1   # greys are neutral


@pytest.mark.parametrize("W,H,N,chunk,flow,opts", [
    # several chunk sets with overlap and constraints (the hand-over between Segmentation objects)
    (96, 64, 60, 8, True, dict(chunk_set_size=3, chunk_set_overlap=1, constraint_chunks=1, min_region_num=3)),
    # default chunk-set geometry on a short video: one flushed set; no flow stream
    (80, 60, 24, 8, False, dict(use_flow=0, min_region_num=4)),
    # lookahead chunks beyond the constraints, no vector data, no size penalizer
    (96, 64, 70, 8, True, dict(chunk_set_size=4, chunk_set_overlap=2, constraint_chunks=1, min_region_num=3,
                               compute_vectorization=0, use_size_penalizer=0)),
    # flow only
    (64, 48, 40, 8, True, dict(chunk_set_size=3, chunk_set_overlap=1, use_appearance=0, min_region_num=3)),
    # the first level has to be cut down to max_region_num
    (96, 64, 30, 10, True, dict(chunk_set_size=2, chunk_set_overlap=1, max_region_num=20, min_region_num=3)),
    # save_descriptors: SegmentationDesc.features on the hierarchy frames
    (96, 64, 60, 8, True, dict(chunk_set_size=3, chunk_set_overlap=1, constraint_chunks=1, min_region_num=3,
                               save_descriptors=1)),
])
def test_region_segmentation_bytes_match_oracle(vsg, W, H, N, chunk, flow, opts):
    fl = synth.const_flow(W, H) if flow else None
    feed = overseg(W, H, N, chunk, synth.soft_frame, fl)
    got, want = run_both(vsg, W, H, feed, opts)
    assert len(want) == N
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "hierarchical SegmentationDesc %d differs" % k
    assert check_structure(got, W, H) >= 1
    Msg = build_schema()
    with_features = 0
    for g in got:
        m = Msg()
        m.ParseFromString(g)
        if opts.get("save_descriptors") and len(m.hierarchy) > 0:
            # one RegionFeatures { id } per region of the first level (segmentation.cpp:490-501)
            assert sorted(f.id for f in m.features) == sorted(r.id for r in m.hierarchy[0].region)
            with_features += 1
        else:
            assert len(m.features) == 0
    assert with_features >= 2 if opts.get("save_descriptors") else with_features == 0


def test_threaded_descriptor_accumulation_is_identical(vsg, monkeypatch):
    """The pixels of a frame's regions are visited on several host threads (dense scratch table
    per thread, bins written back in first-touch order): forced on at a small size."""
    monkeypatch.setenv("VSG_PARALLEL_MIN_WORK", "1")
    W, H, N, chunk = 96, 64, 40, 8
    fl = synth.const_flow(W, H)
    feed = overseg(W, H, N, chunk, synth.soft_frame, fl)
    got, want = run_both(vsg, W, H, feed, dict(chunk_set_size=3, chunk_set_overlap=1, min_region_num=3))
    assert got == want
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (70, 53, 3), dtype=np.uint8)
    assert np.array_equal(vsg.bgr_to_lab(img), ol.bgr_to_lab(img))


def test_reference_abort_is_reported_not_reproduced(vsg):
    """On the bench input neighbouring checker cells have disjoint Lab histograms: distance exactly
    1.0, which RegionAgglomerationGraph files under the virtual edges and then refuses to merge
    (glog CHECK, region_segmentation_graph.cpp:165).  The oracle reports the abort, the product
    returns VSG_ERR_INVALID with the same diagnosis -- neither invents a result."""
    from video_segment_amd._lib import VsgError
    W, H, N, chunk = 96, 64, 16, 8
    fl = synth.const_flow(W, H)
    feed = overseg(W, H, N, chunk, synth.bench_frame, fl)
    o = ol.OracleRegionSegmentation(W, H, ol.region_options(min_region_num=3))
    p = vsg.RegionSegmentation(W, H, vsg.default_region_options(min_region_num=3))
    codes = []
    for k, (frame, f, seg) in enumerate(feed):
        last = k == N - 1
        codes.append(o.process_frame(seg, frame, f, flush=last))
        if codes[-1] == -2:
            with pytest.raises(VsgError, match="the reference aborts"):
                p.process_frame(seg, frame, f, flush=last)
            break
        assert p.process_frame(seg, frame, f, flush=last) == codes[-1]
    assert codes[-1] == -2


def test_options_contract(vsg):
    from video_segment_amd._lib import VsgError
    for bad in (dict(chunk_set_size=1), dict(chunk_set_overlap=0), dict(chunk_set_size=2, chunk_set_overlap=2),
                dict(constraint_chunks=3), dict(use_appearance=0, use_flow=0)):
        with pytest.raises(VsgError):
            vsg.RegionSegmentation(64, 48, vsg.default_region_options(**bad))
    p = vsg.RegionSegmentation(64, 48)
    with pytest.raises(VsgError):
        p.process_frame(b"\xff\xff\xff", np.zeros((48, 64, 3), np.uint8))   # malformed message
    p.close()


def test_foreign_or_corrupt_oversegmentation_is_rejected(vsg):
    """The rasters of the dense unit's message index the frame, the flow field and the id image of
    the vectorisation: a message of another frame size, or one whose scan intervals leave the
    frame, is refused (VSG_ERR_INVALID) before anything is indexed with it."""
    from video_segment_amd._lib import VsgError
    W, H, chunk = 64, 48, 8
    frame = np.zeros((H, W, 3), np.uint8)
    other = overseg(80, 60, 1, chunk, synth.soft_frame, None)[0][2]     # a unit of another size
    p = vsg.RegionSegmentation(W, H)
    with pytest.raises(VsgError, match="another frame size"):
        p.process_frame(other, frame)
    good = overseg(W, H, 1, chunk, synth.soft_frame, None)[0][2]
    Msg = build_schema()
    for field, value in (("right_x", W + 5), ("y", H), ("left_x", -1)):
        m = Msg()
        m.ParseFromString(good)
        iv = m.region[0].raster.scan_inter[0]
        setattr(iv, field, value)
        if field == "left_x":
            iv.right_x = 0
        with pytest.raises(VsgError, match="outside the frame"):
            vsg.RegionSegmentation(W, H).process_frame(m.SerializeToString(), frame)
    inverted = Msg()
    inverted.ParseFromString(good)
    iv = inverted.region[0].raster.scan_inter[0]
    iv.left_x, iv.right_x = 7, 3
    with pytest.raises(VsgError, match="outside the frame"):
        vsg.RegionSegmentation(W, H).process_frame(inverted.SerializeToString(), frame)
    assert vsg.RegionSegmentation(W, H).process_frame(good, frame, flush=True) == 1   # the intact one is fine


def _descriptor_stress_frame(W, H, k):
    """Low-contrast blocks (neighbouring regions have to share histogram bins, or the reference
    aborts) around a grey whose 8-bit L is exactly on a luminance bin (L = 85: the weight of the upper
    bin is 0 and both weights go to one bin), one block of that exact grey without noise, a noisy
    gradient; intervals longer than one accumulation batch (128 px)."""
    rng = np.random.default_rng(100 + k)
    ramp = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    L = ol.bgr_to_lab(ramp)[0, :, 0]
    grey = int(np.argmin(np.abs(L.astype(int) - 85)))
    assert L[grey] == 85
    img = np.empty((H, W, 3), np.int64)
    x = np.arange(W)[None, :]
    y = np.arange(H)[:, None]
    img[..., 0] = grey - 6 + (x * 12) // W
    img[..., 1] = grey - 4 + (y * 8) // H
    img[..., 2] = grey + 5 * (((x + 3 * k) // (W // 3)) % 2)
    img += rng.integers(-1, 2, (H, W, 3))
    img[: H // 4, : W // 2] = grey                      # exact grey, no noise: L on a bin
    return np.clip(img, 0, 255).astype(np.uint8)


def _descriptor_stress_flow(W, H, k):
    rng = np.random.default_rng(500 + k)
    fl = rng.normal(0, 2.0, (H, W, 2)).astype(np.float32)
    fl[: H // 5] = 0.0                                   # atan2(0, 0)
    fl[H // 5: 2 * H // 5, :, 0] = -0.0
    fl[H // 5: 2 * H // 5, :, 1] = np.where(rng.random((H // 5 * 1, W)) < 0.5, 0.0, -0.0)[: fl[H // 5: 2 * H // 5].shape[0]]
    fl[2 * H // 5: 3 * H // 5, : W // 2] = (-2.0, 0.0)   # on the last bin's upper edge
    fl[2 * H // 5: 3 * H // 5, W // 2:] = (0.0, 3.0)     # axis aligned
    fl[3 * H // 5: 4 * H // 5, ::2] = (1.5, -1.5)        # diagonal, alternating with random vectors
    return fl


def test_descriptor_passes_on_varied_colours_and_flow(vsg, monkeypatch):
    """The descriptor passes of AddOverSegmentation beyond what the small constant-flow cases reach:
    per-pixel flow in every direction (zero vectors, negative zeros, axes, bin edges), colours on bin
    positions, intervals longer than an accumulation batch, several host threads."""
    monkeypatch.setenv("VSG_PARALLEL_MIN_WORK", "1")
    W, H, N, chunk = 300, 100, 16, 8
    o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
    frames = [_descriptor_stress_frame(W, H, k) for k in range(N)]
    flows = [None] + [_descriptor_stress_flow(W, H, k) for k in range(1, N)]
    segs = []
    for k in range(N):
        n = o.process_frame(frames[k], flows[k], flush=(k == N - 1))
        segs += [o.result_bytes(i) for i in range(n)]
    o.close()
    feed = [(frames[k], flows[k], segs[k]) for k in range(N)]
    got, want = run_both(vsg, W, H, feed, dict(chunk_set_size=2, chunk_set_overlap=1, min_region_num=3))
    assert len(want) == N
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, "hierarchical SegmentationDesc %d differs" % k


def test_lab_conversion_against_opencv_when_present(vsg):
    """BgrToLab8 restates cv::cvtColor(BGR2Lab) for 8-bit images (fixed-point RGB2Lab_b of the
    OpenCV 2.4 line the reference links; un-vendored third-party arithmetic, parity unpinned in this
    image: no cv2).  Where an OpenCV is importable this pins it: exhaustive over every value of each
    channel against mid-grey partners, the grey ramp, and random colours.  OpenCV >= 3.3 changed the
    coefficient rounding and the cube-root tables of this conversion (values may differ by 1 there);
    the version the comparison ran against is part of the failure message."""
    cv2 = pytest.importorskip("cv2")
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, (64, 257, 3), dtype=np.uint8),
            np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)]
    for ch in range(3):
        im = np.full((3, 256, 3), 128, np.uint8)
        im[0, :, ch] = np.arange(256)
        im[1, :, ch] = np.arange(256)
        im[1, :, (ch + 1) % 3] = 30
        im[2, :, ch] = np.arange(256)
        im[2, :, (ch + 2) % 3] = 220
        imgs.append(im)
    for im in imgs:
        want = cv2.cvtColor(im, cv2.COLOR_BGR2Lab)
        got = vsg.bgr_to_lab(im)
        diff = np.abs(got.astype(int) - want.astype(int))
        major = int(cv2.__version__.split(".")[0])
        if major < 3:
            assert diff.max() == 0, "differs from cv2 %s (the 2.4 line is what the reference links)" % cv2.__version__
        else:
            assert diff.max() <= 1, "differs by more than 1 from cv2 %s" % cv2.__version__
