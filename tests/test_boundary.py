"""Boundary vectorisation (`seg_tree_sample --over_segment` sets
compute_vectorization): the host C++ implementation in the product library against

* the C++ oracle (oracle/vs_oracle_boundary.inc, an independent restatement that follows the
  reference's BoundaryComputation member by member, with cv::approxPolyDP restated in its other
  published form): the serialized SegmentationDesc BYTE FOR BYTE -- polygon start points, hole
  order (the iteration order of the reference's hash map), vector mesh order included;
* the plain-Python oracle (tests/boundary_oracle.py), as geometry up to polygon rotation.

Parity stays unpinned for this row (approxPolyDP and the hash-map order come from outside the
reference tree); what these tests pin is that two independent restatements agree.
The CPU tests go through the host-only C entry points vsg_vectorize_id_image /
vso_vectorize_id_image; the GPU test runs the dense unit with compute_vectorization against the
oracle stream with the same option."""
import ctypes as C

import numpy as np
import pytest

import boundary_oracle as bo
import synth
from test_proto_wire import build_schema


def _canon(poly):
    """Closed polygon -> rotation-independent form (hole polygons start where a hash map's
    iteration happened to reach them first)."""
    pts = list(poly[:-1]) if len(poly) > 1 and poly[0] == poly[-1] else list(poly)
    n = len(pts)
    best = min(range(n), key=lambda i: pts[i:] + pts[:i])
    rot = pts[best:] + pts[:best]
    return tuple(rot + [rot[0]])


def decoded_geometry(msg):
    mesh = list(msg.vector_mesh.coord)
    out = {}
    for r in msg.region:
        polys = []
        for p in r.vectorization.polygon:
            pts = tuple((int(mesh[i]), int(mesh[i + 1])) for i in p.coord_idx)
            polys.append((bool(p.hole), _canon(pts)))
        if polys:
            out[r.id] = sorted(polys)
    return out


def oracle_geometry(ids):
    return {k: sorted((h, _canon(p)) for h, p in v) for k, v in bo.vectorize(ids).items()}


def product_vectorize(ids):
    from video_segment_amd import _lib
    ids = np.ascontiguousarray(ids, np.int32)
    p, n = C.c_void_p(), C.c_size_t()
    _lib.check(_lib.lib().vsg_vectorize_id_image(ids.ctypes.data_as(C.c_void_p), ids.shape[1],
                                                 ids.shape[0], C.byref(p), C.byref(n)))
    m = build_schema()()
    m.ParseFromString(C.string_at(p, n.value))
    return m


def _voronoi(rng, W, H, k):
    """Random N4-connected regions: nearest seed in the L1 metric, then connected components."""
    sx, sy = rng.integers(0, W, k), rng.integers(0, H, k)
    yy, xx = np.mgrid[0:H, 0:W]
    lab = np.argmin(np.abs(xx[None] - sx[:, None, None]) + np.abs(yy[None] - sy[:, None, None]), axis=0)
    from scipy import ndimage
    out = np.zeros((H, W), np.int32)
    nxt = 0
    for v in np.unique(lab):
        comp, n = ndimage.label(lab == v)   # 4-connectivity
        for c in range(1, n + 1):
            out[comp == c] = nxt
            nxt += 1
    return out


CASES = {
    "single": np.zeros((5, 7), np.int32),
    "two_halves": np.repeat(np.array([[0, 0, 0, 1, 1, 1]], np.int32), 4, axis=0),
    "hole": np.pad(np.ones((4, 5), np.int32), 3, constant_values=0),
    "small_hole_dropped": np.pad(np.ones((1, 1), np.int32), 3, constant_values=0),
    "nested": np.pad(np.pad(np.full((3, 3), 2, np.int32), 3, constant_values=1), 3, constant_values=0),
    "four_corner": np.block([[np.zeros((4, 4), np.int32), np.ones((4, 4), np.int32)],
                             [np.full((4, 4), 2, np.int32), np.full((4, 4), 3, np.int32)]]),
    "diagonal_touch": np.array([[0, 0, 0, 0, 0, 0], [0, 1, 1, 0, 0, 0], [0, 1, 1, 0, 0, 0],
                                [0, 0, 0, 2, 2, 0], [0, 0, 0, 2, 2, 0], [0, 0, 0, 0, 0, 0]], np.int32),
}


def product_bytes(ids):
    from video_segment_amd import _lib
    ids = np.ascontiguousarray(ids, np.int32)
    p, n = C.c_void_p(), C.c_size_t()
    _lib.check(_lib.lib().vsg_vectorize_id_image(ids.ctypes.data_as(C.c_void_p), ids.shape[1],
                                                 ids.shape[0], C.byref(p), C.byref(n)))
    return C.string_at(p, n.value)


def _holes_case(rng, W, H, k):
    """A partition with islands: rectangles of new ids dropped into a Voronoi partition, some of
    them nested, some touching -- regions with several holes, holes with several neighbours."""
    ids = _voronoi(rng, W, H, k)
    nxt = int(ids.max()) + 1
    for _ in range(int(rng.integers(3, 9))):
        w, h = int(rng.integers(2, 9)), int(rng.integers(2, 8))
        x, y = int(rng.integers(1, W - w - 1)), int(rng.integers(1, H - h - 1))
        ids[y:y + h, x:x + w] = nxt
        nxt += 1
    # components of equal id must be N4-connected regions of their own: relabel
    from scipy import ndimage
    out = np.zeros_like(ids)
    n = 0
    for v in np.unique(ids):
        comp, c = ndimage.label(ids == v)
        for q in range(1, c + 1):
            out[comp == q] = n
            n += 1
    return out


@pytest.mark.parametrize("name", sorted(CASES))
def test_vectorization_bytes_match_cpp_oracle_small(name):
    import oracle_lib as ol
    ids = CASES[name]
    assert product_bytes(ids) == ol.vectorize_id_image(ids)


@pytest.mark.parametrize("seed", range(12))
def test_vectorization_bytes_match_cpp_oracle_random(seed):
    """Byte equality, not rotation-canonicalised geometry: which vertex a hole polygon starts at and
    the order of the holes follow the iteration order of the segment hash map, and the vector mesh
    lists the points in order of first use."""
    import oracle_lib as ol
    rng = np.random.default_rng(100 + seed)
    W, H = int(rng.integers(24, 90)), int(rng.integers(20, 70))
    ids = _holes_case(rng, W, H, int(rng.integers(4, 20))) if seed % 2 else _voronoi(rng, W, H, int(rng.integers(3, 30)))
    got, want = product_bytes(ids), ol.vectorize_id_image(ids)
    assert got == want
    m = build_schema()()
    m.ParseFromString(got)
    if seed in (1, 3, 9):   # (these partitions are known to contain holes: the hole path is covered)
        assert any(p.hole for r in m.region for p in r.vectorization.polygon)
    assert decoded_geometry(m) == oracle_geometry(ids)


@pytest.mark.parametrize("name", sorted(CASES))
def test_vectorization_small_cases(name):
    ids = CASES[name]
    m = product_vectorize(ids)
    assert decoded_geometry(m) == oracle_geometry(ids)
    assert m.HasField("vector_mesh")
    # the mesh holds every point once
    pts = list(zip(m.vector_mesh.coord[0::2], m.vector_mesh.coord[1::2]))
    assert len(pts) == len(set(pts))


def test_vectorization_known_answers():
    """Hand-checked: a 6x4 frame split in two halves -> two rectangles sharing the middle edge;
    a 4x5 hole in a 10x11 frame -> outer frame rectangle + hole polygon, and the inner region."""
    g = decoded_geometry(product_vectorize(CASES["two_halves"]))
    assert g == {0: [(False, ((0, 0), (0, 4), (3, 4), (3, 0), (0, 0)))],
                 1: [(False, ((3, 0), (3, 4), (6, 4), (6, 0), (3, 0)))]}
    g = decoded_geometry(product_vectorize(CASES["hole"]))
    assert g[1] == [(False, ((3, 3), (3, 7), (8, 7), (8, 3), (3, 3)))]
    assert (False, ((0, 0), (0, 10), (11, 10), (11, 0), (0, 0))) in g[0]
    holes = [p for h, p in g[0] if h]
    assert holes == [((3, 3), (8, 3), (8, 7), (3, 7), (3, 3))]   # the hole runs the other way round
    assert decoded_geometry(product_vectorize(CASES["small_hole_dropped"])).keys() == {0}


@pytest.mark.parametrize("seed", range(6))
def test_vectorization_random_partitions(seed):
    rng = np.random.default_rng(seed)
    W, H = int(rng.integers(12, 40)), int(rng.integers(10, 30))
    ids = _voronoi(rng, W, H, int(rng.integers(3, 14)))
    assert decoded_geometry(product_vectorize(ids)) == oracle_geometry(ids)


def test_approx_poly_dp_properties():
    """The restated cv::approxPolyDP: end points of an open curve are kept, every dropped point is
    within eps of the polyline, straight runs collapse."""
    rng = np.random.default_rng(4)
    for _ in range(50):
        n = int(rng.integers(3, 60))
        steps = rng.integers(0, 4, n)
        pts = [(0, 0)]
        for s in steps:
            x, y = pts[-1]
            pts.append((x + bo.DX[2 * int(s)], y + bo.DY[2 * int(s)]))
        if pts[0] == pts[-1]:
            continue
        out = bo.approx_poly_dp(pts, 1.0, False)
        assert out[0] == pts[0] and out[-1] == pts[-1] and len(out) <= len(pts)
    assert bo.approx_poly_dp([(0, 0), (1, 0), (2, 0), (3, 0)], 1.0, False) == [(0, 0), (3, 0)]


@pytest.mark.gpu
def test_dense_unit_with_compute_vectorization():
    import oracle_lib as ol
    import video_segment_amd as vsg
    W, H, N, chunk = 96, 64, 20, 8
    g = vsg.DenseSegmentation(W, H, vsg.default_options(chunk_size=chunk, compute_vectorization=1),
                              has_flow=True)
    o = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk), has_flow=True)
    ov = ol.OracleStream(W, H, ol.default_options(chunk_size=chunk, compute_vectorization=1), has_flow=True)
    fl = synth.const_flow(W, H)
    Msg = build_schema()
    frames = 0
    for k in range(N):
        f = fl if k > 0 else None
        frame = synth.bench_frame(W, H, k)
        ng = g.process_frame(frame, f, flush=(k == N - 1))
        no = o.process_frame(frame, f, flush=(k == N - 1))
        assert ng == no and ov.process_frame(frame, f, flush=(k == N - 1)) == no
        for i in range(ng):
            # the whole message, vector data included, against the oracle with the same option
            assert g.result_bytes(i) == ov.result_bytes(i)
            m = Msg()
            m.ParseFromString(g.result_bytes(i))
            assert m.HasField("vector_mesh") and len(m.vector_mesh.coord) > 0
            assert decoded_geometry(m) == oracle_geometry(g.result_id_image(i))
            # without the vector data the message is the oracle's, byte for byte
            m.ClearField("vector_mesh")
            for r in m.region:
                r.ClearField("vectorization")
            ref = Msg()
            ref.ParseFromString(o.result_bytes(i))
            assert m == ref
            frames += 1
    assert frames == N
