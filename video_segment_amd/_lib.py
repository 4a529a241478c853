"""ctypes loader of the product library libvsg_hip.so (C ABI: include/vsg.h).

The library is the HIP path; there is no Python or CPU fallback.  Loading fails loudly if the
shared object has not been built (run ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C video_segment_amd/csrc``).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "lib", "libvsg_hip.so")

VSG_OK = 0
VSG_ERR_INVALID, VSG_ERR_DEVICE, VSG_ERR_STATE, VSG_ERR_INTERNAL = -1, -2, -3, -4   # include/vsg.h
VSG_MEM_HOST = 0
VSG_MEM_DEVICE = 1


class VsgOptions(C.Structure):
    _fields_ = [
        ("presmoothing", C.c_int),
        ("frac_min_region_size", C.c_float),
        ("chunk_size", C.c_int),
        ("chunk_overlap_ratio", C.c_float),
        ("num_constraint_frames", C.c_int),
        ("enforce_n4_connectivity", C.c_int),
        ("enforce_spatial_connectedness", C.c_int),
        ("color_distance", C.c_int),
        ("device", C.c_int),
        ("two_stage_oversegment", C.c_int),
        ("compute_vectorization", C.c_int),
    ]


class VsgTimings(C.Structure):
    _fields_ = [
        ("preprocess_ms", C.c_float),
        ("edges_ms", C.c_float),
        ("sort_ms", C.c_float),
        ("merge_ms", C.c_float),
        ("readout_ms", C.c_float),
        ("host_post_ms", C.c_float),
        ("edges_total", C.c_int64),
        ("edges_active", C.c_int64),
        ("merges", C.c_int64),
        ("preprocess_launches", C.c_int64),
        ("edge_launches", C.c_int64),
        ("wave_kernel_ms", C.c_float),
        ("wave_kernel_launches", C.c_int64),
        ("wave_kernel_edges", C.c_int64),
        ("filter_kernel_ms", C.c_float),
        ("filter_kernel_launches", C.c_int64),
        ("spine_kernel_ms", C.c_float),
        ("spine_kernel_launches", C.c_int64),
        ("spine_kernel_edges", C.c_int64),
    ]


class VsgDiagnostics(C.Structure):
    _fields_ = [
        ("segment_wall_ms", C.c_double), ("prepare_ms", C.c_double), ("constrained_merge_ms", C.c_double),
        ("stages", C.c_int64), ("optimistic_stages", C.c_int64), ("rollbacks", C.c_int64),
        ("slab_growths", C.c_int64), ("slab_growth_ms", C.c_double),
        ("spine_pool_growths", C.c_int64), ("spine_pool_growth_ms", C.c_double),
        ("runtime_mallocs", C.c_int64), ("runtime_malloc_ms", C.c_double),
        ("runtime_frees", C.c_int64), ("runtime_free_ms", C.c_double),
        ("cache_hits", C.c_int64), ("device_syncs", C.c_int64), ("device_sync_ms", C.c_double),
        ("mail_waits", C.c_int64), ("mail_wait_ms", C.c_double), ("mail_wait_longest_ms", C.c_double),
        ("mail_mode", C.c_int),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class VsgMemoryStats(C.Structure):
    _fields_ = [
        ("bytes_in_use", C.c_int64), ("bytes_in_use_peak", C.c_int64), ("bytes_cached", C.c_int64),
        ("limit_bytes", C.c_int64),
        ("runtime_mallocs", C.c_int64), ("runtime_frees", C.c_int64), ("cache_hits", C.c_int64),
        ("device_syncs", C.c_int64),
        ("runtime_malloc_ms", C.c_double), ("runtime_free_ms", C.c_double), ("device_sync_ms", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class VsgRegionOptions(C.Structure):
    _fields_ = [
        ("min_region_num", C.c_int), ("max_region_num", C.c_int),
        ("level_cutoff_fraction", C.c_float), ("small_region_penalizer", C.c_float),
        ("luminance_bins", C.c_int), ("color_bins", C.c_int), ("flow_bins", C.c_int),
        ("chunk_set_size", C.c_int), ("chunk_set_overlap", C.c_int), ("constraint_chunks", C.c_int),
        ("use_appearance", C.c_int), ("use_flow", C.c_int), ("use_size_penalizer", C.c_int),
        ("compute_vectorization", C.c_int), ("save_descriptors", C.c_int),
    ]


# Every symbol include/vsg.h declares (checked by tests/test_capi_symbols.py).
EXPORTED_SYMBOLS = [
    "vsg_last_error", "vsg_version", "vsg_default_options", "vsg_device_count",
    "vsg_device_memory_stats", "vsg_device_memory_trim", "vsg_device_memory_limit",
    "vsg_vectorize_id_image",
    "vsg_stream_create", "vsg_stream_destroy", "vsg_stream_process_frame", "vsg_stream_chunk_size",
    "vsg_stream_result_bytes", "vsg_stream_result_id_image", "vsg_stream_last_merge_stats",
    "vsg_stream_last_timings", "vsg_stream_last_diagnostics", "vsg_stream_last_smoothed", "vsg_stream_export_halo",
    "vsg_stream_import_halo", "vsg_stream_expect_halo", "vsg_stream_restart",
    "vsg_chain_create", "vsg_chain_destroy", "vsg_chain_info", "vsg_chain_send_halo",
    "vsg_chain_recv_halo", "vsg_chain_exchange_halo",
    "vsg_regionseg_default_options", "vsg_regionseg_create", "vsg_regionseg_destroy", "vsg_regionseg_process_frame",
    "vsg_regionseg_result_bytes", "vsg_bgr_to_lab",
    "vsg_graph_create", "vsg_graph_destroy", "vsg_graph_add_frame_bgr",
    "vsg_graph_add_frame_features", "vsg_graph_add_virtual_frame", "vsg_graph_add_temporal",
    "vsg_graph_finish_building", "vsg_graph_segment_spatially", "vsg_graph_segment", "vsg_graph_obtain_results",
    "vsg_graph_num_frames", "vsg_graph_num_regions", "vsg_graph_num_neighbor_links",
    "vsg_graph_region_sizes", "vsg_graph_index_image", "vsg_graph_get_regions",
    "vsg_graph_get_intervals", "vsg_graph_smoothed",
    "vsg_graph_spatial_buckets", "vsg_graph_temporal_buckets", "vsg_graph_node_roots",
    "vsg_graph_merge_stats", "vsg_graph_timings", "vsg_graph_diagnostics",
    "vsg_debug_sort_pairs", "vsg_debug_sort_pairs_timed",
]


def build(force=False):
    """Compiles the HIP library in-tree (hipcc --offload-arch=gfx950)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", CSRC_DIR, "-j8", "-s"])
    else:
        # make decides what is stale
        subprocess.check_call(["make", "-C", CSRC_DIR, "-j8", "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvsg_hip.so is missing (%s): build the HIP extension first; there is no fallback "
            "path" % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own libamdhip64.  Two HIP runtimes in one process do not
    # work (whichever is loaded second sees no device), so when torch is installed it is imported
    # first and libvsg_hip.so then binds to the runtime torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.vsg_last_error.restype = C.c_char_p
    L.vsg_version.restype = C.c_int
    L.vsg_default_options.argtypes = [C.POINTER(VsgOptions)]
    L.vsg_device_count.restype = C.c_int
    L.vsg_device_memory_stats.argtypes = [C.c_int, C.POINTER(VsgMemoryStats)]
    L.vsg_device_memory_trim.argtypes = [C.c_int]
    L.vsg_device_memory_limit.argtypes = [C.c_int, C.c_int64]
    L.vsg_vectorize_id_image.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.vsg_stream_create.argtypes = [C.POINTER(VsgOptions), C.c_int, C.c_int, C.POINTER(vp)]
    L.vsg_stream_destroy.argtypes = [vp]
    L.vsg_stream_process_frame.argtypes = [vp, C.c_int, vp, C.c_size_t, vp, C.c_int, C.c_int,
                                           C.POINTER(C.c_int)]
    L.vsg_stream_chunk_size.argtypes = [vp]
    L.vsg_stream_result_bytes.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.vsg_stream_result_id_image.argtypes = [vp, C.c_int, vp]
    L.vsg_stream_last_merge_stats.argtypes = [vp, vp]
    L.vsg_stream_last_timings.argtypes = [vp, C.POINTER(VsgTimings)]
    L.vsg_stream_last_diagnostics.argtypes = [vp, C.POINTER(VsgDiagnostics)]
    L.vsg_stream_last_smoothed.argtypes = [vp, vp]
    L.vsg_stream_export_halo.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), vp]
    L.vsg_stream_import_halo.argtypes = [vp, vp, vp, C.c_int, vp]
    L.vsg_stream_expect_halo.argtypes = [vp]
    L.vsg_stream_restart.argtypes = [vp]
    L.vsg_chain_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.vsg_chain_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.vsg_chain_exchange_halo.argtypes = [vp, vp, C.c_int, vp, C.c_int]
    L.vsg_chain_destroy.argtypes = [vp]
    L.vsg_chain_send_halo.argtypes = [vp, vp, C.c_int]
    L.vsg_chain_recv_halo.argtypes = [vp, vp, C.c_int]
    L.vsg_regionseg_default_options.argtypes = [C.POINTER(VsgRegionOptions)]
    L.vsg_regionseg_create.argtypes = [C.POINTER(VsgRegionOptions), C.c_int, C.c_int, C.POINTER(vp)]
    L.vsg_regionseg_destroy.argtypes = [vp]
    L.vsg_regionseg_process_frame.argtypes = [vp, C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, C.POINTER(C.c_int)]
    L.vsg_regionseg_result_bytes.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.vsg_bgr_to_lab.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp]
    L.vsg_debug_sort_pairs.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int]
    L.vsg_debug_sort_pairs_timed.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.c_double)]
    L.vsg_graph_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.vsg_graph_destroy.argtypes = [vp]
    L.vsg_graph_add_frame_bgr.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_int]
    L.vsg_graph_add_frame_features.argtypes = [vp, vp, vp, C.c_int]
    L.vsg_graph_add_virtual_frame.argtypes = [vp, vp, C.c_int]
    L.vsg_graph_add_temporal.argtypes = [vp, vp, C.c_int, C.c_int]
    L.vsg_graph_finish_building.argtypes = [vp]
    L.vsg_graph_segment_spatially.argtypes = [vp]
    L.vsg_graph_segment.argtypes = [vp, C.c_int, C.c_int]
    L.vsg_graph_obtain_results.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.vsg_graph_num_frames.argtypes = [vp]
    L.vsg_graph_num_regions.argtypes = [vp]
    L.vsg_graph_num_neighbor_links.argtypes = [vp]
    L.vsg_graph_num_neighbor_links.restype = C.c_int64
    L.vsg_graph_region_sizes.argtypes = [vp, vp, vp]
    L.vsg_graph_index_image.argtypes = [vp, C.c_int, vp]
    L.vsg_graph_get_regions.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp),
                                        C.POINTER(vp)]
    L.vsg_graph_get_intervals.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.vsg_graph_smoothed.argtypes = [vp, C.c_int, vp]
    L.vsg_graph_spatial_buckets.argtypes = [vp, C.c_int, vp]
    L.vsg_graph_temporal_buckets.argtypes = [vp, C.c_int, vp, vp]
    L.vsg_graph_node_roots.argtypes = [vp, vp]
    L.vsg_graph_merge_stats.argtypes = [vp, vp]
    L.vsg_graph_timings.argtypes = [vp, C.POINTER(VsgTimings)]
    L.vsg_graph_diagnostics.argtypes = [vp, C.POINTER(VsgDiagnostics)]
    _lib = L
    return L


class VsgError(RuntimeError):
    """A negative status of the C ABI; `code` is the status (VSG_ERR_*)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


def memory_stats(device=-1):
    """vsg_device_memory_stats as a dict (bytes held by live handles, bytes cached for the next one,
    hipMalloc / hipFree calls the library issued so far)."""
    st = VsgMemoryStats()
    check(lib().vsg_device_memory_stats(device, C.byref(st)))
    return st.as_dict()


def memory_trim(device=-1):
    check(lib().vsg_device_memory_trim(device))


def memory_limit(nbytes, device=-1):
    check(lib().vsg_device_memory_limit(device, int(nbytes)))


def check(rc):
    if rc != VSG_OK:
        raise VsgError("vsg error %d: %s" % (rc, lib().vsg_last_error().decode()), rc)
