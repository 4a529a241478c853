/*
 * vs_oracle.h -- C interface of the CPU oracle for the dense over-segmentation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load this library,
 * and only as the checker / reported CPU baseline.
 *
 * The oracle is a from-scratch CPU restatement (plain C++17, no third-party code) of the
 * reference's DenseSegmentation / DenseSegmentationGraph / FastSegmentationGraph path; every
 * function in vs_oracle.cpp cites the reference file:line it follows.
 *
 * PINNING STATUS: the reference ships no tests, golden vectors or fixtures for this path, and
 * it cannot be compiled in this image without writing stand-ins for OpenCV/glog/gflags/
 * protobuf/Boost (forbidden for this build).  The oracle is pinned against the values the
 * survey session recorded from the reference's own code (label hashes,
 * region counts, first-region moments); see tests/test_oracle_pins.py.  The u8->f32 conversion
 * (OpenCV convertTo, un-vendored) is an assumption: float(u8) * float(1.0/255.0).
 */
#ifndef VS_ORACLE_H_
#define VS_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vso_stream vso_stream;   /* DenseSegmentation restatement (streaming).     */
typedef struct vso_graph vso_graph;     /* DenseSegGraphInterface restatement (seam 3).   */

/* Mirrors DenseSegmentationOptions (segmentation/dense_segmentation.h:42-95). */
typedef struct vso_options {
  int presmoothing;             /* 0 none, 2 bilateral (1 = gaussian: unsupported, OpenCV-only) */
  float frac_min_region_size;   /* 0.01f */
  int chunk_size;               /* 20 */
  float chunk_overlap_ratio;    /* 0.2f */
  int num_constraint_frames;    /* 1 */
  int enforce_n4_connectivity;  /* 1 */
  int enforce_spatial_connectedness; /* 1 */
  int color_distance;           /* 0 = L1, 1 = L2 */
  int two_stage_oversegment;    /* 0; 1 = SegmentGraphSpatially before SegmentFullGraph */
  int compute_vectorization;    /* 0; 1 = BoundaryComputation + vector mesh in every result
                                   (segmentation.cpp:527-532; seg_tree --over_segment) */
} vso_options;

void vso_default_options(vso_options* o);
/* Threads of the CPU baseline (results do not depend on it): 1 = everything on the calling thread;
 * n > 1 = the reference's defaults, i.e. one thread per Add*Edges call of the graph construction
 * (dense_segmentation_graph.cpp:31) and an n-way row-parallel bilateral filter. */
void vso_set_threads(int n);

/* ---- hierarchical stage: RegionSegmentation::ProcessFrame (segmentation/region_segmentation.h) -- */
typedef struct vso_region vso_region;
/* Mirrors RegionSegmentationOptions (region_segmentation.h:41-83). */
typedef struct vso_region_options {
  int min_region_num;            /* 10 */
  int max_region_num;            /* 10000 */
  float level_cutoff_fraction;   /* 0.8f */
  float small_region_penalizer;  /* 0.25f */
  int luminance_bins, color_bins, flow_bins;                      /* 10, 20, 16 */
  int chunk_set_size, chunk_set_overlap, constraint_chunks;       /* 6, 2, 1 */
  int use_appearance, use_flow, use_size_penalizer;               /* 1, 1, 1 */
  int compute_vectorization;                                      /* 1 */
  int save_descriptors;                                           /* 0: features { id } per region on hierarchy frames */
} vso_region_options;
void vso_region_default_options(vso_region_options* o);
vso_region* vso_region_create(const vso_region_options* o, int width, int height);
void vso_region_destroy(vso_region* r);
/* seg_desc: the serialized SegmentationDesc of this frame's over-segmentation (NULL together with
 * bgr for a pure flush); bgr: the frame (H rows, `stride` bytes apart); flow: W*H*2 f32 or NULL
 * (first frame / no flow stream; use_flow says whether a flow descriptor exists at all).
 * Returns the number of hierarchical SegmentationDesc now available, -1 on a malformed message,
 * -2 where the reference itself aborts on this input (a glog CHECK in
 * RegionAgglomerationGraph::SegmentGraph, region_segmentation_graph.cpp:165, reachable when two
 * adjacent regions have distance exactly 1.0 -- see vs_oracle_region.inc). */
int vso_region_process_frame(vso_region* r, int flush, const uint8_t* seg_desc, size_t seg_len,
                             const uint8_t* bgr, size_t stride, const float* flow);
int vso_region_result_bytes(const vso_region* r, int i, const uint8_t** data, size_t* len);
/* cv::cvtColor(BGR -> Lab) for 8-bit frames as restated by the oracle (parity unpinned). */
void vso_bgr_to_lab(const uint8_t* bgr, size_t stride, int width, int height, uint8_t* lab);

/* Mirror of vsg_vectorize_id_image: the SegmentationDesc (Region2D list sorted by id, boundaries
 * vectorised) of a frame given as a W*H region-id image.  *data is valid until the thread's next
 * call. */
int vso_vectorize_id_image(const int32_t* ids, int width, int height, const uint8_t** data, size_t* len);

/* ---- streaming level: DenseSegmentation::ProcessFrame ------------------------------- */
vso_stream* vso_stream_create(const vso_options* o, int width, int height);
void vso_stream_destroy(vso_stream* s);
/* bgr: H rows of W*3 bytes with byte stride `stride` (may be NULL together with flush=1).
 * flow: NULL (no flow stream) or W*H*2 floats (x,y interleaved); the flow of the very first
 * frame is ignored (reference: segmentation_unit.cpp:124-130).  has_flow_stream tells whether
 * a flow stream exists at all (then frame 0 stores an empty flow).
 * Returns the number of results now available (0 = buffered). */
int vso_stream_process_frame(vso_stream* s, int flush, const uint8_t* bgr, size_t stride,
                             const float* flow, int has_flow_stream);
int vso_stream_num_results(const vso_stream* s);
/* Serialized segmentation.proto SegmentationDesc (proto2 wire format) of result i. */
int vso_stream_result_bytes(const vso_stream* s, int i, const uint8_t** data, size_t* len);
/* Region-id image of result i (SegmentationDescToIdImage level 0), W*H int32, -1 if uncovered. */
int vso_stream_result_id_image(const vso_stream* s, int i, int32_t* out);
int vso_stream_result_num_regions(const vso_stream* s, int i);
int vso_stream_result_hierarchy_regions(const vso_stream* s, int i);
/* First Region2D of result i: id, then size, mean_x, mean_y, xx, xy, yy. */
int vso_stream_result_first_region(const vso_stream* s, int i, int* id, float* moments6);
/* merge statistics of the last segmented chunk: forced, regular, small. */
void vso_stream_last_merge_stats(const vso_stream* s, int64_t* stats3);
/* Smoothed feature frame of the most recently added frame (W*H*3 f32, BGR interleaved). */
int vso_stream_last_smoothed(const vso_stream* s, float* out);

/* Chunk hand-off for the multi-GPU chain (mirrors vsg_stream_export_halo / import_halo): id images
 * of the two overlap frames (W*H int32 each, host) + {max_region_id, chunk_id, num_output_frames,
 * input_frames}. */
int vso_stream_export_halo(const vso_stream* s, int32_t* virt, int32_t* cons, int64_t* scalars4);
void vso_stream_import_halo(vso_stream* s, const int32_t* virt, const int32_t* cons,
                            const int64_t* scalars4);

/* ---- stage level (kernel parity) ------------------------------------------------------- */
/* PreprocessFeatures: u8*(1/255) then bilateral(3.0, 0.25) if presmoothing==2. out: W*H*3. */
void vso_preprocess(const uint8_t* bgr, size_t stride, int width, int height, int presmoothing,
                    float* out);
/* Bilateral LUT + spatial weights as the reference builds them (image_filter.cpp:208-250).
 * lut: 12288 floats, space_w: 49 floats; returns scale. */
float vso_bilateral_tables(float min_val, float max_val, float* lut, float* space_w);
/* Bucket index of every spatial edge: out[k*W*H + y*W + x], k = 0 R, 1 B, 2 BL, 3 BR,
 * 0xFFFF where the edge does not exist.  l1 = use L1 distance. */
void vso_spatial_buckets(const float* feat, int width, int height, int l1, uint16_t* out);
/* Temporal edges of (cur -> prev): out[k*W*H + ...], k = 0..8 TL,T,TR,L,C,R,BL,B,BR around the
 * (flow displaced) location; prev_idx[y*W+x] = py*W+px of the centre. flow may be NULL. */
void vso_temporal_buckets(const float* cur, const float* prev, const float* flow, int width,
                          int height, int l1, uint16_t* out, int32_t* prev_idx);

/* ---- graph level: DenseSegGraphInterface (config 2 drives this seam directly) --------- */
vso_graph* vso_graph_create(int width, int height, int max_frames, int l1);
void vso_graph_destroy(vso_graph* g);
/* feat: smoothed W*H*3 f32 (kept by pointer until segment; caller keeps it alive).
 * constraint_ids: NULL or W*H int32 (AddNodesAndSpatialEdgesConstrained). */
void vso_graph_add_frame(vso_graph* g, const float* feat, const int32_t* constraint_ids);
void vso_graph_add_virtual_frame(vso_graph* g, const int32_t* constraint_ids);
/* Connects the last two added slices.  flow NULL = straight; is_virtual = weight 1e10. */
void vso_graph_add_temporal(vso_graph* g, const float* cur, const float* prev, const float* flow,
                            int is_virtual);
/* SegmentGraphSpatially (dense_seg_graph_interface.h:138): spatial lists only, before segment. */
void vso_graph_segment_spatially(vso_graph* g);
void vso_graph_segment(vso_graph* g, int min_region_size, int force_constraints);
/* ObtainResults + DetermineNeighborIds.  flows: NULL or array of num_frames pointers. */
void vso_graph_obtain_results(vso_graph* g, const float* const* flows, int enforce_n4,
                              int enforce_spatial_connectedness);
int vso_graph_num_regions(const vso_graph* g);
int64_t vso_graph_num_neighbor_links(const vso_graph* g);
/* Per node label after the merge (representative node id, before FlattenUnionFind). Must be
 * called between vso_graph_segment and vso_graph_obtain_results. n = W*H*frames. */
void vso_graph_node_roots(vso_graph* g, int32_t* out);
/* Per pixel region *index* (RegionInformation::index) of frame t after obtain_results, from the
 * rasterizations.  -1 where uncovered (virtual slices). */
void vso_graph_index_image(const vso_graph* g, int t, int32_t* out);
/* Region table: index -> size, constrained_id, #neighbors; neighbor ids concatenated. */
void vso_graph_region_sizes(const vso_graph* g, int32_t* sizes, int32_t* constrained);
/* The RegionInfoList itself (mirrors vsg_graph_get_regions / vsg_graph_get_intervals):
 * regions5: num_regions x {index, size, constrained_id, first_frame, last_frame} (-1, -1 without
 * a rasterization); nbr_ptr: num_regions + 1 offsets; nbr_idx: neighbor_idx lists concatenated
 * (call with NULL outputs first: returns the total number of neighbour entries). */
int64_t vso_graph_get_regions(const vso_graph* g, int32_t* regions5, int32_t* nbr_ptr, int32_t* nbr_idx);
/* Scan intervals of slice `frame`: {region index, y, left_x, right_x}, regions in index order,
 * each in the order of its Rasterization.  out NULL: returns the count only. */
int64_t vso_graph_get_intervals(const vso_graph* g, int frame, int32_t* out4);
void vso_graph_merge_stats(const vso_graph* g, int64_t* stats3);
/* Per bucket event census of the last vso_graph_segment: for each of 2048 buckets
 * {edges, internal, regular_merge, fail_finalize, small_merge, kept, forced_merge}. */
void vso_graph_bucket_census(const vso_graph* g, int64_t* out /* 2048*7 */);

#ifdef __cplusplus
}
#endif
#endif  /* VS_ORACLE_H_ */
