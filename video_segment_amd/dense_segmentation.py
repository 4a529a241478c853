"""Python host mirror of the reference's operator interface for the dense over-segmentation path.

``DenseSegmentation`` mirrors segmentation::DenseSegmentation (segmentation/dense_segmentation.h:
112-186): ``process_frame(flush, features, flow)`` returns the number of results and the results
are serialized ``SegmentationDesc`` protobuf messages.  ``DenseSegGraph`` mirrors
segmentation::DenseSegGraphInterface (dense_seg_graph_interface.h:107-159).

Everything below is a thin ctypes layer over the C ABI (include/vsg.h); all computation happens in
the HIP library.  Inputs may be numpy arrays (host memory) or torch CUDA tensors (device memory,
used as plain device pointers).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VsgDiagnostics, VsgOptions, VsgTimings, check, lib


def default_options(**kw):
    o = VsgOptions()
    lib().vsg_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _is_torch(x):
    return type(x).__module__.startswith("torch")


_DTYPE_NAMES = {"uint8": ("uint8",), "float32": ("float32",), "int32": ("int32",)}


def _ptr_mem(x, dtype=None, strided_rows=False):
    """Returns (pointer, mem kind) of a numpy array or torch tensor (None -> (None, host)).

    dtype: required element type ("uint8", "float32", "int32").  The array has to be contiguous;
    strided_rows allows an H x W x 3 frame whose rows are `stride` bytes apart (a view of a wider
    buffer, like the reference's padded VideoFrames)."""
    if x is None:
        return None, _lib.VSG_MEM_HOST
    name = str(x.dtype).replace("torch.", "")
    if dtype is not None and name != dtype:
        raise TypeError("expected %s, got %s" % (dtype, name))
    if _is_torch(x):
        if not (x.is_contiguous() or (strided_rows and x.dim() == 3 and x.stride(2) == 1
                                       and x.stride(1) == x.shape[2])):
            raise ValueError("tensor has to be contiguous")
        mem = _lib.VSG_MEM_DEVICE if x.is_cuda else _lib.VSG_MEM_HOST
        return C.c_void_p(x.data_ptr()), mem
    if not (x.flags["C_CONTIGUOUS"] or (strided_rows and x.ndim == 3 and x.strides[2] == x.itemsize
                                         and x.strides[1] == x.shape[2] * x.itemsize)):
        raise ValueError("array has to be C-contiguous")
    return x.ctypes.data_as(C.c_void_p), _lib.VSG_MEM_HOST


def _same_mem(mem_a, mem_b, what):
    if mem_a != mem_b:
        raise ValueError("%s has to live in the same memory kind (host / device) as the frame" % what)


def _row_stride(bgr):
    if _is_torch(bgr):
        assert bgr.stride(2) == 1 and bgr.stride(1) == 3
        return bgr.stride(0) * bgr.element_size()
    assert bgr.strides[2] == 1 and bgr.strides[1] == 3
    return bgr.strides[0]


class DenseSegmentation:
    """Drop-in for segmentation::DenseSegmentation running on one MI355X."""

    def __init__(self, width, height, options=None, has_flow=False):
        self.W, self.H = width, height
        self.has_flow = has_flow
        self.opts = options if options is not None else default_options()
        h = C.c_void_p()
        check(lib().vsg_stream_create(C.byref(self.opts), width, height, C.byref(h)))
        self.h = h
        self._destroy = lib().vsg_stream_destroy

    def close(self):
        if getattr(self, "h", None):
            self._destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def chunk_size(self):
        return lib().vsg_stream_chunk_size(self.h)

    def process_frame(self, bgr, flow=None, flush=False):
        """bgr: HxWx3 uint8 (numpy or torch cuda), or None for a pure flush; flow: HxWx2 f32."""
        stride = 0
        if bgr is not None:
            assert tuple(bgr.shape) == (self.H, self.W, 3)
            stride = _row_stride(bgr)
        p_bgr, mem = _ptr_mem(bgr, "uint8", strided_rows=True)
        if flow is not None:
            assert tuple(flow.shape) == (self.H, self.W, 2)
            if not _is_torch(flow):
                flow = np.ascontiguousarray(flow, dtype=np.float32)
            p_flow, mem_f = _ptr_mem(flow, "float32")
            if bgr is not None:
                _same_mem(mem_f, mem, "the flow field")
            else:
                mem = mem_f
        else:
            p_flow = None
        n = C.c_int()
        check(lib().vsg_stream_process_frame(self.h, int(flush), p_bgr, stride, p_flow,
                                             int(self.has_flow), mem, C.byref(n)))
        self._n = n.value
        return n.value

    def result_bytes(self, i):
        p = C.c_void_p()
        n = C.c_size_t()
        check(lib().vsg_stream_result_bytes(self.h, i, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value)

    def result_id_image(self, i):
        out = np.empty((self.H, self.W), np.int32)
        check(lib().vsg_stream_result_id_image(self.h, i, out.ctypes.data_as(C.c_void_p)))
        return out

    def last_merge_stats(self):
        s = np.zeros(3, np.int64)
        check(lib().vsg_stream_last_merge_stats(self.h, s.ctypes.data_as(C.c_void_p)))
        return s

    def last_timings(self):
        t = VsgTimings()
        check(lib().vsg_stream_last_timings(self.h, C.byref(t)))
        return t

    def last_diagnostics(self):
        """vsg_stream_last_diagnostics of the last segmented chunk, as a dict."""
        d = VsgDiagnostics()
        check(lib().vsg_stream_last_diagnostics(self.h, C.byref(d)))
        return d.as_dict()

    def last_smoothed(self):
        out = np.empty((self.H, self.W, 3), np.float32)
        check(lib().vsg_stream_last_smoothed(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def export_halo(self):
        """Returns (dev_ptr_virtual, dev_ptr_constrained, scalars[4]) after a chunk boundary."""
        a, b = C.c_void_p(), C.c_void_p()
        s = np.zeros(4, np.int64)
        check(lib().vsg_stream_export_halo(self.h, C.byref(a), C.byref(b),
                                           s.ctypes.data_as(C.c_void_p)))
        return a.value, b.value, s

    def expect_halo(self):
        """Fresh stream in the middle of a video: frames may be fed before import_halo."""
        check(lib().vsg_stream_expect_halo(self.h))

    def restart(self):
        """Back to the state right after construction (device memory is kept)."""
        check(lib().vsg_stream_restart(self.h))

    def import_halo(self, labels_virtual, labels_constrained, scalars):
        n = self.W * self.H
        for a in (labels_virtual, labels_constrained):
            if int(np.prod(tuple(a.shape))) != n:
                raise ValueError("label planes have to hold W*H int32")
        pa, mem = _ptr_mem(labels_virtual, "int32")
        pb, mem_b = _ptr_mem(labels_constrained, "int32")
        _same_mem(mem_b, mem, "the second label plane")
        s = np.ascontiguousarray(scalars, dtype=np.int64)
        check(lib().vsg_stream_import_halo(self.h, pa, pb, mem, s.ctypes.data_as(C.c_void_p)))


class ChunkChain:
    """RCCL hand-off of the chunk halo between the GPUs of a node (vsg_chain_*, include/vsg.h).

    nonce: a value shared by the ranks of this run and not used by earlier runs (see vsg.h)."""

    def __init__(self, rank, world, id_file, nonce=0, device=-1):
        h = C.c_void_p()
        check(lib().vsg_chain_create(rank, world, id_file.encode(), C.c_uint64(nonce), device,
                                     C.byref(h)))
        self.h = h
        self._destroy = lib().vsg_chain_destroy

    def close(self):
        if getattr(self, "h", None):
            self._destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def info(self):
        """(rank, world) as the RCCL communicator reports them."""
        r, w = C.c_int(), C.c_int()
        check(lib().vsg_chain_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def send_halo(self, stream, dst):
        check(lib().vsg_chain_send_halo(self.h, stream.h, dst))

    def recv_halo(self, stream, src):
        check(lib().vsg_chain_recv_halo(self.h, stream.h, src))

    def exchange_halo(self, from_stream, dst, into_stream, src):
        check(lib().vsg_chain_exchange_halo(self.h, from_stream.h if from_stream else None, dst,
                                            into_stream.h if into_stream else None, src))


class DenseSegGraph:
    """Drop-in for segmentation::DenseSegGraphInterface (one chunk graph)."""

    def __init__(self, width, height, max_frames, l1=False, device=-1):
        self.W, self.H, self.max_frames = width, height, max_frames
        h = C.c_void_p()
        check(lib().vsg_graph_create(width, height, max_frames, int(l1), device, C.byref(h)))
        self.h = h
        self._destroy = lib().vsg_graph_destroy

    def close(self):
        if getattr(self, "h", None):
            self._destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add_frame_bgr(self, bgr, presmoothing=2, constraint_ids=None):
        assert tuple(bgr.shape) == (self.H, self.W, 3)
        p, mem = _ptr_mem(bgr, "uint8", strided_rows=True)
        if constraint_ids is not None and not _is_torch(constraint_ids):
            constraint_ids = np.ascontiguousarray(constraint_ids, np.int32)
        pc, mem_c = _ptr_mem(constraint_ids, "int32")
        if constraint_ids is not None:
            _same_mem(mem_c, mem, "constraint_ids")
        check(lib().vsg_graph_add_frame_bgr(self.h, p, _row_stride(bgr), presmoothing, pc, mem))

    def add_frame_features(self, feat, constraint_ids=None):
        if not _is_torch(feat):
            feat = np.ascontiguousarray(feat, np.float32)
        assert tuple(feat.shape) == (self.H, self.W, 3)
        p, mem = _ptr_mem(feat, "float32")
        if constraint_ids is not None and not _is_torch(constraint_ids):
            constraint_ids = np.ascontiguousarray(constraint_ids, np.int32)
        pc, mem_c = _ptr_mem(constraint_ids, "int32")
        if constraint_ids is not None:
            _same_mem(mem_c, mem, "constraint_ids")
        check(lib().vsg_graph_add_frame_features(self.h, p, pc, mem))

    def add_virtual_frame(self, constraint_ids):
        if not _is_torch(constraint_ids):
            constraint_ids = np.ascontiguousarray(constraint_ids, np.int32)
        p, mem = _ptr_mem(constraint_ids, "int32")
        check(lib().vsg_graph_add_virtual_frame(self.h, p, mem))

    def add_temporal(self, flow=None, is_virtual=False):
        if flow is not None and not _is_torch(flow):
            flow = np.ascontiguousarray(flow, np.float32)
        p, mem = _ptr_mem(flow, "float32")
        check(lib().vsg_graph_add_temporal(self.h, p, int(is_virtual), mem))

    def finish_building(self):
        check(lib().vsg_graph_finish_building(self.h))

    def segment_spatially(self):
        check(lib().vsg_graph_segment_spatially(self.h))

    def segment(self, min_region_size, force_constraints):
        check(lib().vsg_graph_segment(self.h, int(min_region_size), int(force_constraints)))

    def obtain_results(self, use_flows=False, enforce_n4=True, enforce_spatial_connectedness=True):
        check(lib().vsg_graph_obtain_results(self.h, int(use_flows), int(enforce_n4),
                                             int(enforce_spatial_connectedness)))

    def num_frames(self):
        return lib().vsg_graph_num_frames(self.h)

    def num_regions(self):
        return lib().vsg_graph_num_regions(self.h)

    def num_neighbor_links(self):
        return lib().vsg_graph_num_neighbor_links(self.h)

    def region_sizes(self):
        n = self.num_regions()
        s = np.empty(n, np.int32)
        c = np.empty(n, np.int32)
        check(lib().vsg_graph_region_sizes(self.h, s.ctypes.data_as(C.c_void_p),
                                           c.ctypes.data_as(C.c_void_p)))
        return s, c

    def index_image(self, t):
        out = np.empty((self.H, self.W), np.int32)
        check(lib().vsg_graph_index_image(self.h, t, out.ctypes.data_as(C.c_void_p)))
        return out

    def get_regions(self):
        """The RegionInfoList of ObtainResults + DetermineNeighborIds as arrays: (regions [n,5] =
        index, size, constrained_id, first frame, last frame; nbr_ptr [n+1]; nbr_idx)."""
        regs, n = C.c_void_p(), C.c_size_t()
        ptr, idx = C.c_void_p(), C.c_void_p()
        check(lib().vsg_graph_get_regions(self.h, C.byref(regs), C.byref(n), C.byref(ptr), C.byref(idx)))
        n = n.value
        r = np.ctypeslib.as_array(C.cast(regs, C.POINTER(C.c_int32)), shape=(max(n, 1) * 5,))[:n * 5]
        r = r.reshape(n, 5).copy()
        p = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(n + 1,)).copy()
        total = int(p[n])
        i = np.ctypeslib.as_array(C.cast(idx, C.POINTER(C.c_int32)), shape=(max(total, 1),))[:total].copy()
        return r, p, i

    def get_intervals(self, t):
        """Scan intervals of slice t: [m,4] = region index, y, left_x, right_x."""
        iv, m = C.c_void_p(), C.c_size_t()
        check(lib().vsg_graph_get_intervals(self.h, t, C.byref(iv), C.byref(m)))
        m = m.value
        a = np.ctypeslib.as_array(C.cast(iv, C.POINTER(C.c_int32)), shape=(max(m, 1) * 4,))[:m * 4]
        return a.reshape(m, 4).copy()

    def smoothed(self, t):
        out = np.empty((self.H, self.W, 3), np.float32)
        check(lib().vsg_graph_smoothed(self.h, t, out.ctypes.data_as(C.c_void_p)))
        return out

    def spatial_buckets(self, t):
        out = np.empty((4, self.H, self.W), np.uint16)
        check(lib().vsg_graph_spatial_buckets(self.h, t, out.ctypes.data_as(C.c_void_p)))
        return out

    def temporal_buckets(self, t):
        out = np.empty((9, self.H, self.W), np.uint16)
        pidx = np.empty((self.H, self.W), np.int32)
        check(lib().vsg_graph_temporal_buckets(self.h, t, out.ctypes.data_as(C.c_void_p),
                                               pidx.ctypes.data_as(C.c_void_p)))
        return out, pidx

    def node_roots(self):
        out = np.empty(self.W * self.H * self.num_frames(), np.int32)
        check(lib().vsg_graph_node_roots(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def merge_stats(self):
        s = np.zeros(3, np.int64)
        check(lib().vsg_graph_merge_stats(self.h, s.ctypes.data_as(C.c_void_p)))
        return s

    def timings(self):
        t = VsgTimings()
        check(lib().vsg_graph_timings(self.h, C.byref(t)))
        return t

    def diagnostics(self):
        """vsg_graph_diagnostics of the last segment() call, as a dict."""
        d = VsgDiagnostics()
        check(lib().vsg_graph_diagnostics(self.h, C.byref(d)))
        return d.as_dict()
