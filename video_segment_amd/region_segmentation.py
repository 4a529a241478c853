"""Python host mirror of segmentation::RegionSegmentation (segmentation/region_segmentation.h:
131-216): the hierarchical stage on top of the dense over-segmentation.  A thin ctypes layer over
vsg_region_* (include/vsg.h); frames, flow and messages live in host memory."""
import ctypes as C

import numpy as np

from ._lib import VsgRegionOptions, check, lib


def default_region_options(**kw):
    o = VsgRegionOptions()
    lib().vsg_regionseg_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def bgr_to_lab(bgr):
    """cv::cvtColor(BGR -> Lab, 8 bit) as the stage computes it (parity hook)."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty_like(bgr)
    check(lib().vsg_bgr_to_lab(bgr.ctypes.data_as(C.c_void_p), bgr.strides[0], bgr.shape[1], bgr.shape[0],
                               out.ctypes.data_as(C.c_void_p)))
    return out


class RegionSegmentation:
    """Drop-in for segmentation::RegionSegmentation."""

    def __init__(self, width, height, options=None):
        self.W, self.H = width, height
        self.opts = options if options is not None else default_region_options()
        h = C.c_void_p()
        check(lib().vsg_regionseg_create(C.byref(self.opts), width, height, C.byref(h)))
        self.h = h
        self._destroy = lib().vsg_regionseg_destroy

    def close(self):
        if getattr(self, "h", None):
            self._destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process_frame(self, seg_bytes, bgr, flow=None, flush=False):
        """seg_bytes: the frame's serialized over-segmentation (None together with bgr None for a
        pure flush); bgr: HxWx3 uint8; flow: HxWx2 f32 or None.  Returns the number of results."""
        n = C.c_int()
        if seg_bytes is None:
            check(lib().vsg_regionseg_process_frame(self.h, int(flush), None, 0, None, 0, None, C.byref(n)))
            return n.value
        assert tuple(bgr.shape) == (self.H, self.W, 3) and bgr.strides[2] == 1 and bgr.strides[1] == 3
        if flow is not None:
            flow = np.ascontiguousarray(flow, np.float32)
            assert tuple(flow.shape) == (self.H, self.W, 2)
        buf = C.create_string_buffer(seg_bytes, len(seg_bytes))
        check(lib().vsg_regionseg_process_frame(
            self.h, int(flush), C.cast(buf, C.c_void_p), len(seg_bytes), bgr.ctypes.data_as(C.c_void_p),
            bgr.strides[0], flow.ctypes.data_as(C.c_void_p) if flow is not None else None, C.byref(n)))
        return n.value

    def result_bytes(self, i):
        p, n = C.c_void_p(), C.c_size_t()
        check(lib().vsg_regionseg_result_bytes(self.h, i, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value)
