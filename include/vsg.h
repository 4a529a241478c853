/*
 * vsg.h -- C ABI of the MI355X-native dense over-segmentation hot path (libvsg_hip.so).
 *
 * This is the drop-in boundary: plain pointers, sizes and status ints, no C++ or torch types.
 * Two seams of the reference are mirrored (reference paths relative to the reference root):
 *
 *   vsg_stream_*  <->  segmentation::DenseSegmentation            (seam 2)
 *       segmentation/dense_segmentation.h:112-186  (ProcessFrame, ChunkSize)
 *       called by DenseSegmentationUnit::ProcessFrame/PostProcess, segmentation_unit.cpp:118-161
 *
 *   vsg_graph_*   <->  segmentation::DenseSegGraphInterface       (seam 3)
 *       segmentation/dense_seg_graph_interface.h:107-159 (13 pure virtuals, all mirrored; the
 *       RegionInfoList they fill is read back with vsg_graph_get_regions / _get_intervals)
 *       obtained through DenseSegmentation::CreateDenseSegGraph, dense_segmentation.cpp:253-266
 *
 * Results cross the boundary as serialized segmentation.proto `SegmentationDesc` messages
 * (segment_util/segmentation.proto:55-172), i.e. exactly what the reference's units exchange as
 * PointerFrame<SegmentationDesc>; INTEGRATION.md shows the reference-side binding.
 *
 * Conventions: every function returns VSG_OK (0) or a negative status; vsg_last_error() returns
 * a thread-local message for the last failure.  A handle is thread-compatible (one caller thread
 * per handle, like the reference's units) and owns one HIP stream.  Pointers returned by the
 * library stay valid until the next call on the same handle.  There is NO CPU fallback: if no
 * HIP device is usable, creation fails with VSG_ERR_DEVICE.
 */
#ifndef VSG_H_
#define VSG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSG_OK 0
#define VSG_ERR_INVALID -1   /* bad argument / contract violation (reference: glog CHECK)   */
#define VSG_ERR_DEVICE -2    /* HIP runtime error / no device                              */
#define VSG_ERR_STATE -3     /* call not valid in the handle's current state               */
#define VSG_ERR_INTERNAL -4  /* invariant violated inside the library                      */

/* Where the memory behind a frame / flow / label pointer lives. */
#define VSG_MEM_HOST 0
#define VSG_MEM_DEVICE 1

typedef struct vsg_stream vsg_stream;
typedef struct vsg_graph vsg_graph;

/* Mirrors segmentation::DenseSegmentationOptions (dense_segmentation.h:42-95). */
typedef struct vsg_options {
  int presmoothing;                   /* 0 PRESMOOTH_NONE, 1 PRESMOOTH_GAUSSIAN (3x3, sigma 1.5),
                                       * 2 PRESMOOTH_BILATERAL (default)  */
  float frac_min_region_size;         /* 0.01f                                              */
  int chunk_size;                     /* 20, must be >= 3                                   */
  float chunk_overlap_ratio;          /* 0.2f                                               */
  int num_constraint_frames;          /* 1                                                  */
  int enforce_n4_connectivity;        /* 1                                                  */
  int enforce_spatial_connectedness;  /* 1                                                  */
  int color_distance;                 /* 0 COLOR_DISTANCE_L1, 1 COLOR_DISTANCE_L2 (default) */
  int device;                         /* HIP device ordinal, -1 = current device            */
  int two_stage_oversegment;          /* 0 (default); 1 = SegmentGraphSpatially first         */
  int compute_vectorization;          /* 0 (default); 1 = region boundaries as polygons over a
                                         shared vector mesh in every SegmentationDesc
                                         (segmentation.cpp:527-532, seg_tree --over_segment)   */
} vsg_options;

/* Per-stage device time of the last segmented chunk, milliseconds (HIP events on the handle's
 * stream) and work counters; used by bench.py for the roofline line. */
typedef struct vsg_timings {
  float preprocess_ms;     /* min/max + bilateral, summed over the chunk's frames           */
  float edges_ms;          /* spatial + temporal edge weight / bucket kernels               */
  float sort_ms;           /* bucket counting sort                                          */
  float merge_ms;          /* ordered union-find merge (all buckets)                        */
  float readout_ms;        /* flatten, N4, run-length, neighbours (device part)             */
  float host_post_ms;      /* host: tube splitting, ids, SegmentationDesc assembly          */
  int64_t edges_total;     /* edges in the chunk graph                                      */
  int64_t edges_active;    /* edges that reached an exact merge worker                      */
  int64_t merges;          /* forced + regular + small merges                               */
  int64_t preprocess_launches;
  int64_t edge_launches;
  /* Dominant kernel (k_merge_wave: per-component ordered replay), measured with HIP events on
   * the handle's stream around every launch of the chunk. */
  float wave_kernel_ms;        /* summed launch durations                                    */
  int64_t wave_kernel_launches;
  int64_t wave_kernel_edges;   /* active edges replayed by wavefront workers                 */
  float filter_kernel_ms;      /* k_filter, summed                                           */
  int64_t filter_kernel_launches;
  /* k_spine (large components replayed along their Kruskal tree), same measurement. */
  float spine_kernel_ms;
  int64_t spine_kernel_launches;
  int64_t spine_kernel_edges;  /* side clusters absorbed along the spines                    */
} vsg_timings;

/* Where the last SegmentFullGraph call of a handle spent its host time (diagnostics: a window that
 * took ten times as long as its neighbours names its reason here).  Times are host wall-clock
 * milliseconds of the calling thread. */
typedef struct vsg_diagnostics {
  double segment_wall_ms;        /* the whole call                                               */
  double prepare_ms;             /* tables, scratch and pools before the first stage             */
  double constrained_merge_ms;   /* MergeConstrainedRegions (segmentation_graph.h:703-786)       */
  int64_t stages;                /* stages (filter -> components -> workers) the buckets took    */
  int64_t optimistic_stages, rollbacks;
  int64_t slab_growths;          /* the per-active-edge scratch had to grow inside a stage       */
  double slab_growth_ms;
  int64_t spine_pool_growths;    /* the scratch of the Kruskal-tree replay had to grow           */
  double spine_pool_growth_ms;
  int64_t runtime_mallocs;       /* hipMalloc / hipHostMalloc calls the call issued              */
  double runtime_malloc_ms;
  int64_t runtime_frees;         /* hipFree / hipHostFree calls                                  */
  double runtime_free_ms;
  int64_t cache_hits;            /* blocks adopted from the device cache instead                 */
  int64_t device_syncs;          /* device-wide synchronisations for blocks going back to it     */
  double device_sync_ms;
  int64_t mail_waits;            /* scalars the host waited for (mailbox, device_graph.h)        */
  double mail_wait_ms;           /* host time inside those waits (includes the kernels awaited)  */
  double mail_wait_longest_ms;
  int mail_mode;                 /* 0 spin, 1 yield, 2 sleep (VSG_MAIL_YIELD)                    */
} vsg_diagnostics;

/* Device memory of the library, per device.  A closed handle leaves its blocks in a process-wide
 * cache and the next handle adopts them (no hipMalloc / hipFree after the first window of a caller
 * that creates a graph per window, dense_seg_graph_interface.h:58-98); the cached, unused bytes are
 * bounded (default 40 % of the device's memory; VSG_DEVICE_CACHE_MB, 0 = no cache). */
typedef struct vsg_memory_stats {
  int64_t bytes_in_use;       /* device blocks held by live handles                               */
  int64_t bytes_in_use_peak;
  int64_t bytes_cached;       /* device blocks waiting for the next handle                        */
  int64_t limit_bytes;        /* bound of bytes_cached                                            */
  int64_t runtime_mallocs, runtime_frees, cache_hits, device_syncs;   /* since process start      */
  double runtime_malloc_ms, runtime_free_ms, device_sync_ms;
} vsg_memory_stats;
int vsg_device_memory_stats(int device, vsg_memory_stats* out);
/* Returns every cached block of the device to the HIP runtime (live handles keep theirs). */
int vsg_device_memory_trim(int device);
/* Sets the bound of the cached bytes (0: handles release straight to the runtime; < 0: default). */
int vsg_device_memory_limit(int device, int64_t bytes);

const char* vsg_last_error(void);
int vsg_version(void);
void vsg_default_options(vsg_options* o);
/* Number of visible HIP devices (0 => every create call fails loudly). */
int vsg_device_count(void);

/* Host-only parity hook (no HIP device needed): the SegmentationDesc -- Region2D list sorted by id,
 * boundaries vectorised as with compute_vectorization -- of a frame given as a W*H region-id image
 * (ids >= 0, N4-connected regions).  BoundaryComputation::ComputeBoundary + ComputeVectorization,
 * segmentation/boundary.cpp:121-244, 514-608.  *data stays valid until the thread's next call. */
int vsg_vectorize_id_image(const int32_t* ids, int width, int height, const uint8_t** data,
                           size_t* len);

/* ---- seam 2: DenseSegmentation ------------------------------------------------------------ */
/* DenseSegmentation::DenseSegmentation(options, frame_width, frame_height), cpp:50-106. */
int vsg_stream_create(const vsg_options* o, int width, int height, vsg_stream** out);
void vsg_stream_destroy(vsg_stream* s);
/* int DenseSegmentation::ProcessFrame(flush, features, flow, results), cpp:108-162.
 *   bgr    : H rows of W BGR24 pixels, `stride` bytes apart (VideoFrame::MatView); NULL with
 *            flush != 0 for a pure flush (DenseSegmentationUnit::PostProcess).
 *   flow   : backward flow, W*H interleaved (x,y) f32 (DenseFlowFrame::MatViewInterleaved), or
 *            NULL.  has_flow_stream says whether the unit has a flow stream at all; the flow of
 *            the first frame is ignored (segmentation_unit.cpp:124-130).
 *   mem    : VSG_MEM_HOST or VSG_MEM_DEVICE for both pointers.  Device buffers are read on the
 *            handle's own (non-blocking) HIP stream: whatever produced them has to be complete
 *            before the call (synchronise the producing stream or event first); they are only
 *            read during the call.
 * *num_results = number of SegmentationDesc now available (0: frame buffered). */
int vsg_stream_process_frame(vsg_stream* s, int flush, const uint8_t* bgr, size_t stride,
                             const float* flow, int has_flow_stream, int mem, int* num_results);
/* DenseSegmentation::ChunkSize(), h:130. */
int vsg_stream_chunk_size(const vsg_stream* s);
/* Result i of the last process_frame call as serialized SegmentationDesc. */
int vsg_stream_result_bytes(vsg_stream* s, int i, const uint8_t** data, size_t* len);
/* Convenience: SegmentationDescToIdImage(level 0) of result i (segmentation_util.cpp:741-770);
 * out is W*H int32 in host memory. */
int vsg_stream_result_id_image(vsg_stream* s, int i, int32_t* out);
int vsg_stream_last_merge_stats(const vsg_stream* s, int64_t* forced_regular_small);
int vsg_stream_last_timings(const vsg_stream* s, vsg_timings* t);
int vsg_stream_last_diagnostics(const vsg_stream* s, vsg_diagnostics* d);
/* Parity hook: smoothed feature planes of the most recently added frame, W*H*3 f32 BGR
 * interleaved, host memory (PreprocessFeatures output, cpp:164-198). */
int vsg_stream_last_smoothed(vsg_stream* s, float* out);

/* Multi-GPU hand-off: everything chunk c+1 needs from chunk c.
 * The two label planes are the region-id images of the two overlap frames
 * (overlap_segmentations_[0..1], dense_segmentation.cpp:300-308, 400-403) in DEVICE memory of the
 * exporting handle; scalars = {max_region_id_, chunk_id_, num_output_frames_, input_frames_}.
 * Valid right after a process_frame call that returned results without flush. */
int vsg_stream_export_halo(vsg_stream* s, const int32_t** dev_labels_virtual,
                           const int32_t** dev_labels_constrained, int64_t scalars[4]);
/* Starts a fresh stream in the middle of a video: the next chunk is constrained by the given
 * label planes (mem says where they live).  The caller then feeds the constrained overlap frame
 * first (with its flow), i.e. the frame `last_output_frame + 1` of the previous chunk. */
int vsg_stream_import_halo(vsg_stream* s, const int32_t* labels_virtual,
                           const int32_t* labels_constrained, int mem, const int64_t scalars[4]);

/* Overlapped order for the chain: a fresh stream that will start in the middle of a video may be
 * fed its frames (the constrained overlap frame first) BEFORE the halo exists -- features, edges
 * and the bucket sort of the chunk do not depend on the previous chunk -- and receives the halo
 * with vsg_stream_import_halo later, at the latest before the process_frame call that completes
 * the chunk (that call fails with VSG_ERR_STATE otherwise). */
int vsg_stream_expect_halo(vsg_stream* s);
/* Puts the handle back into the state right after vsg_stream_create, keeping its device memory
 * (one handle per GPU serves all the chunks the GPU owns). */
int vsg_stream_restart(vsg_stream* s);

/* Chunk hand-off between GPUs over RCCL (point-to-point ncclSend / ncclRecv over xGMI), for a
 * host that shards ONE video chunk-wise over the GPUs of a node: one process per
 * GPU, one vsg_chain per process.
 *   id_file : a path all ranks can read and rank 0 can write.  Rank 0 removes whatever is there,
 *             publishes {magic, nonce, ncclUniqueId} atomically and removes the file again once
 *             the communicator exists; the others poll (two minutes) for a record with THEIR nonce.
 *   nonce   : any value the ranks of this run share and earlier runs did not use (launch time,
 *             a hash of the launcher's run id ...): a record left behind by another run is
 *             ignored instead of joined.
 * send_halo ships what vsg_stream_export_halo describes (two W*H int32 label planes + 4 counters)
 * to rank dst; recv_halo receives it from rank src and imports it into the stream
 * (vsg_stream_import_halo semantics, including the overlapped order above).  exchange_halo does
 * both in one RCCL group (either stream may be NULL); with dst == src == the own rank it is the
 * hand-off between two handles of ONE GPU through the same RCCL calls.  All three block until
 * the transfer has completed; the streams have to live on the chain's device. */
typedef struct vsg_chain vsg_chain;
int vsg_chain_create(int rank, int world, const char* id_file, uint64_t nonce, int device,
                     vsg_chain** out);
void vsg_chain_destroy(vsg_chain* c);
/* Rank and size as the RCCL communicator reports them (ncclCommUserRank / ncclCommCount). */
int vsg_chain_info(const vsg_chain* c, int* rank, int* world);
int vsg_chain_send_halo(vsg_chain* c, vsg_stream* from, int dst);
int vsg_chain_recv_halo(vsg_chain* c, vsg_stream* into, int src);
int vsg_chain_exchange_halo(vsg_chain* c, vsg_stream* from, int dst, vsg_stream* into, int src);

/* ---- hierarchical stage: RegionSegmentation (BASELINE configs[4]) ------ */
/* segmentation::RegionSegmentation (segmentation/region_segmentation.h:131-216), called by
 * RegionSegmentationUnit::ProcessFrame / PostProcess (segmentation_unit.cpp:236-279): consumes the
 * dense unit's SegmentationDesc per frame together with the frame (and its backward flow) and
 * emits SegmentationDesc messages that carry the whole region hierarchy.  Host code, host memory:
 * the clustering works on a few thousand regions per chunk set (see region_segmentation.h); no HIP
 * device is needed for these entry points. */
typedef struct vsg_regionseg vsg_regionseg;
/* Mirrors segmentation::RegionSegmentationOptions (region_segmentation.h:41-83). */
typedef struct vsg_regionseg_options {
  int min_region_num;             /* 10    */
  int max_region_num;             /* 10000 */
  float level_cutoff_fraction;    /* 0.8f  */
  float small_region_penalizer;   /* 0.25f */
  int luminance_bins;             /* 10 */
  int color_bins;                 /* 20 */
  int flow_bins;                  /* 16 */
  int chunk_set_size;             /* 6: over-segmentation chunks per chunk set        */
  int chunk_set_overlap;          /* 2 */
  int constraint_chunks;          /* 1 */
  int use_appearance;             /* 1 */
  int use_flow;                   /* 1: RegionSegmentationUnit sets it to "a flow stream exists" */
  int use_size_penalizer;         /* 1 */
  int compute_vectorization;      /* 1 */
  int save_descriptors;           /* 0; 1 = SegmentationDesc.features on hierarchy frames: one RegionFeatures
                                   * { id } per region (segmentation.cpp:490-501; the descriptors of this
                                   * path add no extension to it, region_descriptor.cpp:137-138) */
} vsg_regionseg_options;
void vsg_regionseg_default_options(vsg_regionseg_options* o);
int vsg_regionseg_create(const vsg_regionseg_options* o, int width, int height, vsg_regionseg** out);
void vsg_regionseg_destroy(vsg_regionseg* r);
/* int RegionSegmentation::ProcessFrame(flush, segmentation, features, results), cpp:97-205.
 *   seg_desc : this frame's serialized over-segmentation (what vsg_stream_result_bytes returns),
 *              NULL together with bgr for a pure flush;
 *   bgr      : H rows of W BGR24 pixels `stride` bytes apart, host memory;
 *   flow     : W*H interleaved (x, y) f32 backward flow of the frame, or NULL (the first frame has
 *              none: segmentation_unit.cpp:314-323; always NULL with use_flow == 0).
 * VSG_ERR_INVALID where the reference aborts (glog CHECK), including the one plain input can reach:
 * two neighbouring regions at distance exactly 1.0 (region_segmentation_graph.cpp:165, see
 * region_segmentation.cpp). */
int vsg_regionseg_process_frame(vsg_regionseg* r, int flush, const uint8_t* seg_desc, size_t seg_len,
                             const uint8_t* bgr, size_t stride, const float* flow, int* num_results);
int vsg_regionseg_result_bytes(vsg_regionseg* r, int i, const uint8_t** data, size_t* len);
/* Parity hook: cv::cvtColor(BGR -> Lab, 8 bit) as the stage computes it; lab: W*H*3, rows packed. */
int vsg_bgr_to_lab(const uint8_t* bgr, size_t stride, int width, int height, uint8_t* lab);

/* ---- seam 3: DenseSegGraphInterface ------------------------------------------------------- */
/* CreateDenseSegGraph(frame_width, frame_height, max_frames) + InitializeGraph(),
 * dense_seg_graph_interface.h:46-48,112; l1 selects DistanceColorL1 (dense_segmentation.cpp:247-251). */
int vsg_graph_create(int width, int height, int max_frames, int l1, int device, vsg_graph** out);
void vsg_graph_destroy(vsg_graph* g);
/* PreprocessFeatures + AddNodesAndSpatialEdges[Constrained] on a BGR24 frame
 * (dense_segmentation.cpp:164-220; interface h:115-117).  constraint_ids: NULL or W*H int32. */
int vsg_graph_add_frame_bgr(vsg_graph* g, const uint8_t* bgr, size_t stride, int presmoothing,
                            const int32_t* constraint_ids, int mem);
/* AddNodesAndSpatialEdges[Constrained] on already smoothed features (W*H*3 f32 interleaved),
 * i.e. SpatialCvMatDistance3L2(feat). */
int vsg_graph_add_frame_features(vsg_graph* g, const float* feat, const int32_t* constraint_ids,
                                 int mem);
/* AddVirtualNodesConstrained(desc) with desc rendered to a W*H int32 id image, h:121. */
int vsg_graph_add_virtual_frame(vsg_graph* g, const int32_t* constraint_ids, int mem);
/* AddTemporalEdges / AddTemporalFlowEdges / AddTemporalVirtualEdges / AddTemporalFlowVirtualEdges
 * (h:124-132): connects the last two added slices.  flow NULL = no flow. */
int vsg_graph_add_temporal(vsg_graph* g, const float* flow, int is_virtual, int mem);
/* FinishBuildingGraph (h:135): waits for the asynchronous build kernels. */
int vsg_graph_finish_building(vsg_graph* g);
/* SegmentGraphSpatially(), h:138: merges along the spatial edges only (min_region_size 0, no
 * constraint merge).  Optional; the segment call that follows then only sees the spatial edges
 * this pass kept (two_stage_segmentation, segmentation/segmentation.cpp:280-283). */
int vsg_graph_segment_spatially(vsg_graph* g);
/* SegmentFullGraph(min_region_size, force_constraints), h:141. */
int vsg_graph_segment(vsg_graph* g, int min_region_size, int force_constraints);
/* ObtainResults(..., flows, false, enforce_n4, enforce_spatial_connectedness) followed by
 * DetermineNeighborIds (h:147-158).  use_flows: pass the flows given to add_temporal. */
int vsg_graph_obtain_results(vsg_graph* g, int use_flows, int enforce_n4,
                             int enforce_spatial_connectedness);
int vsg_graph_num_frames(const vsg_graph* g);
int vsg_graph_num_regions(const vsg_graph* g);
int64_t vsg_graph_num_neighbor_links(const vsg_graph* g);
/* Region table after obtain_results: size and constrained id per RegionInformation index. */
int vsg_graph_region_sizes(const vsg_graph* g, int32_t* sizes, int32_t* constrained_ids);
/* Per-pixel RegionInformation index of slice t from the rasterizations (host, W*H int32). */
int vsg_graph_index_image(const vsg_graph* g, int t, int32_t* out);
/* The RegionInfoList that ObtainResults / DetermineNeighborIds fill
 * (dense_seg_graph_interface.h:147-158; RegionInformation, segmentation_common.h:39-116), as
 * plain arrays an adapter rebuilds the reference's objects from.  Library-owned, valid until the
 * next call on the handle.
 *   regions[i]  : RegionInformation with index == i: size, constrained_id, first / last frame of
 *                 its Rasterization3D (-1, -1: raster == nullptr, a representative that only
 *                 appears as a neighbour).
 *   nbr_csr_ptr : num_regions + 1 offsets into nbr_csr_idx; nbr_csr_idx[ptr[i] .. ptr[i+1]) =
 *                 RegionInformation::neighbor_idx of region i (sorted, unique region indices).
 *   intervals   : the scan intervals of slice `frame`, region after region in index order, each
 *                 region's in the order of its Rasterization (scan order; merged tubes: sorted by
 *                 MergeRasterization) -- exactly what raster->find(frame)->second holds. */
typedef struct vsg_region {
  int32_t index, size, constrained_id, first_frame, last_frame;
} vsg_region;
typedef struct vsg_interval {
  int32_t region_index, y, left_x, right_x;
} vsg_interval;
int vsg_graph_get_regions(vsg_graph* g, const vsg_region** regions, size_t* num_regions,
                          const int32_t** nbr_csr_ptr, const int32_t** nbr_csr_idx);
int vsg_graph_get_intervals(vsg_graph* g, int frame, const vsg_interval** intervals, size_t* n);
/* Parity hooks (host memory outputs). */
int vsg_graph_smoothed(vsg_graph* g, int t, float* out /* W*H*3 interleaved */);
int vsg_graph_spatial_buckets(vsg_graph* g, int t, uint16_t* out /* 4*W*H, plane k */);
int vsg_graph_temporal_buckets(vsg_graph* g, int t, uint16_t* out /* 9*W*H */, int32_t* prev_idx);
/* Union-find representative (node id) of every node after segment(); n = W*H*frames. */
int vsg_graph_node_roots(vsg_graph* g, int32_t* out);
int vsg_graph_merge_stats(const vsg_graph* g, int64_t* forced_regular_small);
int vsg_graph_timings(const vsg_graph* g, vsg_timings* t);
int vsg_graph_diagnostics(const vsg_graph* g, vsg_diagnostics* d);

/* Test hook: the merge path's stable radix sort of (key, value) pairs by the low `end_bit` bits of
 * the keys (csrc/radix_sort.hip), on host arrays of n elements, run on `device`.  The reference sorts
 * nothing here -- its per-bucket vectors are in order by construction (segmentation_graph.h:336);
 * the sort is how the device path orders a stage's edges by component. */
int vsg_debug_sort_pairs(const uint32_t* keys, const uint32_t* values, int n, int end_bit,
                         uint32_t* keys_out, uint32_t* values_out, int device);
/* ... with the implementation chosen (0: what the merge would use for n, 1: hand-written, 2: rocPRIM),
 * repeated `reps` times; *avg_us = device time per sort (HIP events around the repetitions). */
int vsg_debug_sort_pairs_timed(const uint32_t* keys, const uint32_t* values, int n, int end_bit,
                               uint32_t* keys_out, uint32_t* values_out, int device, int impl, int reps,
                               double* avg_us);

#ifdef __cplusplus
}
#endif
#endif /* VSG_H_ */
