"""ctypes binding of the CPU oracle (oracle/libvs_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "libvs_oracle.so")


class VsoOptions(C.Structure):
    _fields_ = [
        ("presmoothing", C.c_int),
        ("frac_min_region_size", C.c_float),
        ("chunk_size", C.c_int),
        ("chunk_overlap_ratio", C.c_float),
        ("num_constraint_frames", C.c_int),
        ("enforce_n4_connectivity", C.c_int),
        ("enforce_spatial_connectedness", C.c_int),
        ("color_distance", C.c_int),
        ("two_stage_oversegment", C.c_int),
        ("compute_vectorization", C.c_int),
    ]


class VsoRegionOptions(C.Structure):
    _fields_ = [
        ("min_region_num", C.c_int), ("max_region_num", C.c_int),
        ("level_cutoff_fraction", C.c_float), ("small_region_penalizer", C.c_float),
        ("luminance_bins", C.c_int), ("color_bins", C.c_int), ("flow_bins", C.c_int),
        ("chunk_set_size", C.c_int), ("chunk_set_overlap", C.c_int), ("constraint_chunks", C.c_int),
        ("use_appearance", C.c_int), ("use_flow", C.c_int), ("use_size_penalizer", C.c_int),
        ("compute_vectorization", C.c_int), ("save_descriptors", C.c_int),
    ]


def build_oracle(force=False):
    src = os.path.join(ORACLE_DIR, "vs_oracle.cpp")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    L.vso_default_options.argtypes = [C.POINTER(VsoOptions)]
    L.vso_stream_create.restype = vp
    L.vso_stream_create.argtypes = [C.POINTER(VsoOptions), C.c_int, C.c_int]
    L.vso_stream_destroy.argtypes = [vp]
    L.vso_stream_process_frame.restype = C.c_int
    L.vso_stream_process_frame.argtypes = [vp, C.c_int, vp, C.c_size_t, vp, C.c_int]
    L.vso_stream_num_results.restype = C.c_int
    L.vso_stream_num_results.argtypes = [vp]
    L.vso_stream_result_bytes.restype = C.c_int
    L.vso_stream_result_bytes.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.vso_stream_result_id_image.restype = C.c_int
    L.vso_stream_result_id_image.argtypes = [vp, C.c_int, vp]
    L.vso_stream_result_num_regions.restype = C.c_int
    L.vso_stream_result_num_regions.argtypes = [vp, C.c_int]
    L.vso_stream_result_hierarchy_regions.restype = C.c_int
    L.vso_stream_result_hierarchy_regions.argtypes = [vp, C.c_int]
    L.vso_stream_result_first_region.restype = C.c_int
    L.vso_stream_result_first_region.argtypes = [vp, C.c_int, C.POINTER(C.c_int), vp]
    L.vso_stream_last_merge_stats.argtypes = [vp, vp]
    L.vso_stream_last_smoothed.restype = C.c_int
    L.vso_stream_last_smoothed.argtypes = [vp, vp]
    L.vso_stream_export_halo.restype = C.c_int
    L.vso_stream_export_halo.argtypes = [vp, vp, vp, vp]
    L.vso_stream_import_halo.argtypes = [vp, vp, vp, vp]
    L.vso_preprocess.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp]
    L.vso_bilateral_tables.restype = C.c_float
    L.vso_bilateral_tables.argtypes = [C.c_float, C.c_float, vp, vp]
    L.vso_spatial_buckets.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    L.vso_temporal_buckets.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.vso_vectorize_id_image.restype = C.c_int
    L.vso_vectorize_id_image.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.vso_region_default_options.argtypes = [C.POINTER(VsoRegionOptions)]
    L.vso_region_create.restype = vp
    L.vso_region_create.argtypes = [C.POINTER(VsoRegionOptions), C.c_int, C.c_int]
    L.vso_region_destroy.argtypes = [vp]
    L.vso_region_process_frame.restype = C.c_int
    L.vso_region_process_frame.argtypes = [vp, C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp]
    L.vso_region_result_bytes.restype = C.c_int
    L.vso_region_result_bytes.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.vso_bgr_to_lab.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp]
    L.vso_graph_create.restype = vp
    L.vso_graph_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.vso_graph_destroy.argtypes = [vp]
    L.vso_graph_add_frame.argtypes = [vp, vp, vp]
    L.vso_graph_add_virtual_frame.argtypes = [vp, vp]
    L.vso_graph_add_temporal.argtypes = [vp, vp, vp, vp, C.c_int]
    L.vso_set_threads.argtypes = [C.c_int]
    L.vso_graph_segment_spatially.argtypes = [vp]
    L.vso_graph_segment.argtypes = [vp, C.c_int, C.c_int]
    L.vso_graph_obtain_results.argtypes = [vp, vp, C.c_int, C.c_int]
    L.vso_graph_num_regions.restype = C.c_int
    L.vso_graph_num_regions.argtypes = [vp]
    L.vso_graph_num_neighbor_links.restype = C.c_int64
    L.vso_graph_num_neighbor_links.argtypes = [vp]
    L.vso_graph_node_roots.argtypes = [vp, vp]
    L.vso_graph_index_image.argtypes = [vp, C.c_int, vp]
    L.vso_graph_region_sizes.argtypes = [vp, vp, vp]
    L.vso_graph_merge_stats.argtypes = [vp, vp]
    L.vso_graph_get_regions.restype = C.c_int64
    L.vso_graph_get_regions.argtypes = [vp, vp, vp, vp]
    L.vso_graph_get_intervals.restype = C.c_int64
    L.vso_graph_get_intervals.argtypes = [vp, C.c_int, vp]
    L.vso_graph_bucket_census.argtypes = [vp, vp]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_threads(n):
    """1: single threaded; n > 1: the reference's default threading (see vs_oracle.h)."""
    lib().vso_set_threads(int(n))


def default_options(**kw):
    o = VsoOptions()
    lib().vso_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class OracleStream:
    """DenseSegmentation::ProcessFrame restatement."""

    def __init__(self, width, height, options=None, has_flow=False):
        self.W, self.H = width, height
        self.has_flow = has_flow
        self.opts = options if options is not None else default_options()
        self.h = lib().vso_stream_create(C.byref(self.opts), width, height)

    def close(self):
        if self.h:
            lib().vso_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def expect_halo(self):
        """Overlapped chain order (vsg_stream_expect_halo): the CPU restatement has nothing to
        build ahead, so frames fed before the halo are held back and replayed after import_halo."""
        self._held = []

    def restart(self):
        lib().vso_stream_destroy(self.h)
        self.h = lib().vso_stream_create(C.byref(self.opts), self.W, self.H)
        self._held = None

    def process_frame(self, bgr, flow=None, flush=False):
        """bgr: HxWx3 uint8 (any row stride) or None.  Returns number of results."""
        if getattr(self, "_held", None) is not None:
            self._held.append((None if bgr is None else np.array(bgr, copy=True),
                               None if flow is None else np.array(flow, copy=True), flush))
            return 0
        return self._process_frame(bgr, flow, flush)

    def _process_frame(self, bgr, flow=None, flush=False):
        if bgr is not None:
            assert bgr.dtype == np.uint8 and bgr.shape == (self.H, self.W, 3)
            assert bgr.strides[2] == 1 and bgr.strides[1] == 3
            stride = bgr.strides[0]
        else:
            stride = 0
        if flow is not None:
            flow = np.ascontiguousarray(flow, dtype=np.float32)
            assert flow.shape == (self.H, self.W, 2)
        return lib().vso_stream_process_frame(self.h, int(flush), _ptr(bgr), stride, _ptr(flow),
                                              int(self.has_flow))

    def num_results(self):
        return lib().vso_stream_num_results(self.h)

    def result_bytes(self, i):
        p = C.c_void_p()
        n = C.c_size_t()
        rc = lib().vso_stream_result_bytes(self.h, i, C.byref(p), C.byref(n))
        assert rc == 0
        return C.string_at(p, n.value)

    def result_id_image(self, i):
        out = np.empty((self.H, self.W), np.int32)
        assert lib().vso_stream_result_id_image(self.h, i, _ptr(out)) == 0
        return out

    def result_num_regions(self, i):
        return lib().vso_stream_result_num_regions(self.h, i)

    def result_hierarchy_regions(self, i):
        return lib().vso_stream_result_hierarchy_regions(self.h, i)

    def result_first_region(self, i):
        rid = C.c_int()
        m = np.zeros(6, np.float32)
        assert lib().vso_stream_result_first_region(self.h, i, C.byref(rid), _ptr(m)) == 0
        return rid.value, m

    def last_merge_stats(self):
        s = np.zeros(3, np.int64)
        lib().vso_stream_last_merge_stats(self.h, _ptr(s))
        return s

    def last_smoothed(self):
        out = np.empty((self.H, self.W, 3), np.float32)
        assert lib().vso_stream_last_smoothed(self.h, _ptr(out)) == 0
        return out

    def export_halo(self):
        a = np.empty((self.H, self.W), np.int32)
        b = np.empty((self.H, self.W), np.int32)
        s = np.zeros(4, np.int64)
        assert lib().vso_stream_export_halo(self.h, _ptr(a), _ptr(b), _ptr(s)) == 0
        return a, b, s

    def import_halo(self, labels_virtual, labels_constrained, scalars):
        a = np.ascontiguousarray(labels_virtual, np.int32)
        b = np.ascontiguousarray(labels_constrained, np.int32)
        s = np.ascontiguousarray(scalars, np.int64)
        lib().vso_stream_import_halo(self.h, _ptr(a), _ptr(b), _ptr(s))
        held, self._held = getattr(self, "_held", None), None
        for (bgr, flow, flush) in held or []:
            n = self._process_frame(bgr, flow, flush)
            assert n == 0, "the frame that completes the chunk has to follow the halo"


def region_options(**kw):
    o = VsoRegionOptions()
    lib().vso_region_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def bgr_to_lab(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty_like(bgr)
    lib().vso_bgr_to_lab(_ptr(bgr), bgr.strides[0], bgr.shape[1], bgr.shape[0], _ptr(out))
    return out


class OracleRegionSegmentation:
    """RegionSegmentation restatement (hierarchical stage on top of the over-segmentation)."""

    def __init__(self, width, height, options=None):
        self.W, self.H = width, height
        self.opts = options if options is not None else region_options()
        self.h = lib().vso_region_create(C.byref(self.opts), width, height)

    def close(self):
        if self.h:
            lib().vso_region_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process_frame(self, seg_bytes, bgr, flow=None, flush=False):
        """seg_bytes: serialized SegmentationDesc of the frame (None with bgr None: pure flush)."""
        if seg_bytes is None:
            return lib().vso_region_process_frame(self.h, int(flush), None, 0, None, 0, None)
        bgr = np.ascontiguousarray(bgr, np.uint8)
        if flow is not None:
            flow = np.ascontiguousarray(flow, np.float32)
        buf = C.create_string_buffer(seg_bytes, len(seg_bytes))
        n = lib().vso_region_process_frame(self.h, int(flush), C.cast(buf, C.c_void_p), len(seg_bytes),
                                           _ptr(bgr), bgr.strides[0], _ptr(flow))
        assert n != -1, "malformed SegmentationDesc"
        return n   # -2: the reference aborts on this input (see vs_oracle.h)

    def result_bytes(self, i):
        p, n = C.c_void_p(), C.c_size_t()
        assert lib().vso_region_result_bytes(self.h, i, C.byref(p), C.byref(n)) == 0
        return C.string_at(p, n.value)


def vectorize_id_image(ids):
    """Serialized SegmentationDesc (Region2D list sorted by id + vectorization + vector mesh) of a
    frame given as an H x W region-id image (mirror of vsg_vectorize_id_image)."""
    ids = np.ascontiguousarray(ids, np.int32)
    p, n = C.c_void_p(), C.c_size_t()
    rc = lib().vso_vectorize_id_image(_ptr(ids), ids.shape[1], ids.shape[0], C.byref(p), C.byref(n))
    assert rc == 0
    return C.string_at(p, n.value)


def preprocess(bgr, presmoothing=2):
    H, W, _ = bgr.shape
    out = np.empty((H, W, 3), np.float32)
    lib().vso_preprocess(_ptr(bgr), bgr.strides[0], W, H, presmoothing, _ptr(out))
    return out


def bilateral_tables(min_val, max_val):
    lut = np.empty(12288, np.float32)
    sw = np.empty(81, np.float32)
    scale = lib().vso_bilateral_tables(float(min_val), float(max_val), _ptr(lut), _ptr(sw))
    return scale, lut, sw[:49].copy()


def spatial_buckets(feat, l1=False):
    H, W, _ = feat.shape
    feat = np.ascontiguousarray(feat, np.float32)
    out = np.empty((4, H, W), np.uint16)
    lib().vso_spatial_buckets(_ptr(feat), W, H, int(l1), _ptr(out))
    return out


def temporal_buckets(cur, prev, flow=None, l1=False):
    H, W, _ = cur.shape
    cur = np.ascontiguousarray(cur, np.float32)
    prev = np.ascontiguousarray(prev, np.float32)
    if flow is not None:
        flow = np.ascontiguousarray(flow, np.float32)
    out = np.empty((9, H, W), np.uint16)
    pidx = np.empty((H, W), np.int32)
    lib().vso_temporal_buckets(_ptr(cur), _ptr(prev), _ptr(flow), W, H, int(l1), _ptr(out),
                               _ptr(pidx))
    return out, pidx


class OracleGraph:
    """DenseSegGraphInterface restatement (seam 3)."""

    def __init__(self, width, height, max_frames, l1=False):
        self.W, self.H, self.max_frames = width, height, max_frames
        self.h = lib().vso_graph_create(width, height, max_frames, int(l1))
        self._keep = []
        self.num_frames = 0

    def close(self):
        if self.h:
            lib().vso_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add_frame(self, feat, constraint_ids=None):
        feat = np.ascontiguousarray(feat, np.float32)
        self._keep.append(feat)
        if constraint_ids is not None:
            constraint_ids = np.ascontiguousarray(constraint_ids, np.int32)
        lib().vso_graph_add_frame(self.h, _ptr(feat), _ptr(constraint_ids))
        self.num_frames += 1

    def add_virtual_frame(self, constraint_ids):
        constraint_ids = np.ascontiguousarray(constraint_ids, np.int32)
        lib().vso_graph_add_virtual_frame(self.h, _ptr(constraint_ids))
        self.num_frames += 1

    def add_temporal(self, cur, prev, flow=None, is_virtual=False):
        if flow is not None:
            flow = np.ascontiguousarray(flow, np.float32)
            self._keep.append(flow)
        lib().vso_graph_add_temporal(self.h, _ptr(cur), _ptr(prev), _ptr(flow), int(is_virtual))

    def segment_spatially(self):
        lib().vso_graph_segment_spatially(self.h)

    def segment(self, min_region_size, force_constraints):
        lib().vso_graph_segment(self.h, min_region_size, int(force_constraints))

    def node_roots(self):
        out = np.empty(self.W * self.H * self.num_frames, np.int32)
        lib().vso_graph_node_roots(self.h, _ptr(out))
        return out

    def obtain_results(self, flows=None, enforce_n4=True, enforce_spatial_connectedness=True):
        arr = None
        if flows is not None:
            arr = (C.c_void_p * len(flows))()
            for i, f in enumerate(flows):
                if f is not None:
                    f = np.ascontiguousarray(f, np.float32)
                    self._keep.append(f)
                    arr[i] = f.ctypes.data
                else:
                    arr[i] = None
        lib().vso_graph_obtain_results(self.h, arr, int(enforce_n4),
                                       int(enforce_spatial_connectedness))

    def num_regions(self):
        return lib().vso_graph_num_regions(self.h)

    def num_neighbor_links(self):
        return lib().vso_graph_num_neighbor_links(self.h)

    def index_image(self, t):
        out = np.empty((self.H, self.W), np.int32)
        lib().vso_graph_index_image(self.h, t, _ptr(out))
        return out

    def region_sizes(self):
        n = self.num_regions()
        s = np.empty(n, np.int32)
        c = np.empty(n, np.int32)
        lib().vso_graph_region_sizes(self.h, _ptr(s), _ptr(c))
        return s, c

    def merge_stats(self):
        s = np.zeros(3, np.int64)
        lib().vso_graph_merge_stats(self.h, _ptr(s))
        return s

    def get_regions(self):
        """RegionInfoList: (regions [n,5] = index, size, constrained_id, first, last frame;
        nbr_ptr [n+1]; nbr_idx)."""
        n = self.num_regions()
        total = lib().vso_graph_get_regions(self.h, None, None, None)
        regs = np.empty((n, 5), np.int32)
        ptr = np.empty(n + 1, np.int32)
        idx = np.empty(max(total, 1), np.int32)
        lib().vso_graph_get_regions(self.h, _ptr(regs), _ptr(ptr), _ptr(idx))
        return regs, ptr, idx[:total]

    def get_intervals(self, t):
        """Scan intervals of slice t: [m,4] = region index, y, left_x, right_x."""
        m = lib().vso_graph_get_intervals(self.h, t, None)
        out = np.empty((max(m, 1), 4), np.int32)
        lib().vso_graph_get_intervals(self.h, t, _ptr(out))
        return out[:m]

    def bucket_census(self):
        out = np.zeros((2048, 7), np.int64)
        lib().vso_graph_bucket_census(self.h, _ptr(out))
        return out
