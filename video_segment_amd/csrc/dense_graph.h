// dense_graph.h -- DenseGraphHip: the MI355X implementation of the reference's
// DenseSegGraphInterface (segmentation/dense_seg_graph_interface.h:107-159) for
// DenseSegmentationGraph<DistanceColorL2|L1, ColorMeanDescriptorTraits>.
#ifndef VSG_DENSE_GRAPH_H_
#define VSG_DENSE_GRAPH_H_

#include <memory>
#include <unordered_map>
#include <vector>

#include "device_graph.h"
#include "host_model.h"

namespace vsg {

struct GraphTimings {
  float edges_ms = 0, sort_ms = 0, merge_ms = 0, readout_ms = 0, host_post_ms = 0;
  int64_t edges_total = 0, edges_active = 0;
  int64_t merges[3] = {0, 0, 0};   // forced, regular, small
  float wave_ms = 0, filter_ms = 0, spine_ms = 0;
  int64_t wave_launches = 0, filter_launches = 0, wave_edges = 0, spine_launches = 0, spine_edges = 0;
  int64_t optimistic_stages = 0, rollbacks = 0;
  // Diagnostics of the last Segment call (vsg_diagnostics, include/vsg.h): what a window that took
  // ten times as long as its neighbours spent its time on.
  int64_t stages = 0;
  int64_t slab_growths = 0, spine_growths = 0;
  double slab_growth_ms = 0, spine_growth_ms = 0;
  int64_t runtime_mallocs = 0, runtime_frees = 0, cache_hits = 0, device_syncs = 0;
  double runtime_malloc_ms = 0, runtime_free_ms = 0, device_sync_ms = 0;
  int64_t mail_waits = 0;
  double mail_wait_ms = 0, mail_wait_longest_ms = 0;
  int mail_mode = 0;
  double segment_wall_ms = 0, prepare_ms = 0, constrained_merge_ms = 0;
};

class DenseGraphHip {
 public:
  DenseGraphHip(int W, int H, int max_frames, bool l1, hipStream_t stream);
  ~DenseGraphHip();

  int W() const { return W_; }
  int H() const { return H_; }
  int max_frames() const { return max_frames_; }
  int num_frames() const { return num_frames_; }
  hipStream_t stream() const { return stream_; }

  // Starts a new (empty) graph reusing all device buffers.  max_frames may shrink/grow up to the
  // capacity given at construction.
  void Reset(int max_frames);
  // Forgets what the handle learned about its input (where the tree replay pays): a handle that
  // is restarted for another video starts like a fresh one.
  void ForgetLearned();

  // AddNodesAndSpatialEdges[Constrained].  feat: 3 planes of W*H f32 (B,G,R) in device memory,
  // must stay valid until the stream has executed the call.  cons: W*H int32 device or nullptr.
  void AddFrame(const float* feat_planar_dev, const int32_t* cons_dev);
  // AddVirtualNodesConstrained.  max_label: upper bound (exclusive) of the ids in the image.
  void AddVirtualFrame(const int32_t* ids_dev, int max_label);
  // The same with the label images still to come (multi-GPU chunk chain: the previous chunk is
  // being segmented on another GPU while this graph is built).  The virtual slice is reserved
  // and the next AddFrame is the constrained slice; nothing but the node initialisation of the
  // two slices depends on the labels, so features, edges and the bucket sort proceed, and
  // SetHaloLabels completes the two slices before Segment.
  void AddVirtualFrameDeferred();
  void SetHaloLabels(const int32_t* virtual_ids_dev, const int32_t* constrained_ids_dev,
                     int max_label);
  bool halo_pending() const { return halo_pending_; }
  // AddTemporal[Flow][Virtual]Edges: connects the last two slices.
  void AddTemporal(const float* cur_dev, const float* prev_dev, const float* flow_dev,
                   bool is_virtual);
  void FinishBuilding();
  void Segment(int min_region_size, bool force_constraints) {
    SegmentLists(min_region_size, force_constraints, spatial_pass_done_ ? 2 : 0);
  }
  // SegmentGraphSpatially (dense_segmentation_graph.h:406-416): the spatial bucket lists only,
  // min_region_size 0, no constraint merge; the Segment() call that follows only sees the
  // spatial edges this pass kept (two_stage_oversegment).
  void SegmentSpatially() {
    SegmentLists(0, false, 1);
    spatial_pass_done_ = true;
  }
  // ObtainResults + DetermineNeighborIds.  flows: per slice W*H*2 f32 *device* pointers (may be
  // null for slices without flow) or null: the tube analysis samples them at a few thousand
  // points (FindPreviousTube), so the fields never leave the device.
  void ObtainResults(const std::vector<const float*>* dev_flows, bool enforce_n4,
                     bool enforce_spatial_connectedness);

  std::vector<RegionInfo>& regions() { return regions_; }
  const std::vector<RegionInfo>& regions() const { return regions_; }
  const GraphTimings& timings() const { return timings_; }

  // parity hooks
  void CopySpatialBuckets(int t, uint16_t* out_host);                     // [4][H][W]
  void CopyTemporalBuckets(int t, uint16_t* out_host, int32_t* prev_idx_host);   // [9][H][W]
  void CopyNodeRoots(int32_t* out_host);

 private:
  struct ListBuf {
    DevBuf<uint32_t> slots;      // sorted slot ids
    DevBuf<int32_t> offsets;     // [kBucketSlots]
    DevBuf<int32_t> prev_idx;    // temporal only
    DevBuf<uint16_t> keys_dbg;   // unsorted keys kept for the parity hooks
    int type = 0;
    int n = 0;
    int base_a = 0, base_b = 0;
    bool used = false;
  };

  void EnsureScratch(size_t n_edges_max);
  void EnsureActiveScratch(size_t n_active_max);
  void DebugHash(const char* where);
  void SortList(ListBuf& lb, int per_px);
  void MergeConstrainedHostAssisted();
  NodeArrays nodes() {
    return NodeArrays{parent_.get(), desc_sz_.get(), cons_.get(), flags_.get(), hub8_.get()};
  }

  QuiesceGuard quiesce_;   // first member: spans the release of every buffer below (device_cache.h)
  int W_, H_, capacity_frames_, max_frames_;
  bool l1_;
  hipStream_t stream_;
  hipStream_t aux_stream_ = nullptr;   // second stream of the merge stages (see RunBucketStage)
  hipStream_t aux2_stream_ = nullptr;  // third: side clusters beside the next tree level (merge_spine.hip)
  hipEvent_t aux_fork_ = nullptr, aux_join_ = nullptr;
  int num_frames_ = 0;
  size_t wh_;
  bool has_constraints_ = false;
  std::vector<int> virtual_slices_;
  int min_region_size_ = 0;
  bool flattened_ = false;

  // node arrays
  DevBuf<int32_t> parent_;
  DevBuf<float4> desc_sz_;
  DevBuf<int32_t> cons_;
  DevBuf<uint8_t> flags_;
  DevBuf<uint8_t> hub8_;    // NodeArrays::hub8
  // N-sized scratch / result arrays
  DevBuf<int32_t> cc_, label_uf_, label_img_, adjust_;
  // lists
  std::vector<ListBuf> lists_;
  DevBuf<ListDesc> list_desc_dev_;
  DevBuf<uint32_t> list_slot_base_dev_;
  std::vector<uint32_t> list_slot_base_;
  DevBuf<uint8_t> kept_all_;
  DevBuf<uint8_t> kept_spatial_pass_;   // two-stage: what SegmentSpatially kept
  bool spatial_pass_done_ = false;
  bool halo_pending_ = false;
  int deferred_virtual_slice_ = -1, deferred_constrained_slice_ = -1;
  // pass: 0 all lists, 1 spatial lists only, 2 all lists after a spatial pass
  void SegmentLists(int min_region_size, bool force_constraints, int pass);
  void RunStageDebug(int b, int w, int windows, int j0, int n, const MergeParams& P, int inert_mode,
                     MergeScratch& S, bool debug_stages, StageInfo* info);
  DevBuf<int32_t> bucket_base_dev_;
  std::vector<int32_t> bucket_base_host_;
  // temporaries for edge generation / sorting
  DevBuf<uint16_t> keys_tmp_;
  DevBuf<int32_t> hist_tmp_, hist_sums_;   // edge_sort.hip scratch
  DevBuf<int32_t> first_label_scratch_;
  // merge scratch
  DevBuf<int32_t> e_ra_, e_rb_;
  DevBuf<uint32_t> e_gpos_;
  // The arrays sized by a stage's ACTIVE edges live in one slab (a single allocation: growing them
  // in the middle of a stage is one hipFree + hipMalloc, not nineteen), carved up by
  // EnsureActiveScratch.
  DevBuf<uint8_t> active_slab_;
  struct ActiveArrays {
    int32_t *e_active = nullptr, *e_apos = nullptr, *a_ra = nullptr, *a_rb = nullptr, *seg_cnt = nullptr,
            *seg_off = nullptr, *lead_pos = nullptr, *l_ra = nullptr, *l_rb = nullptr, *bk_cons = nullptr;
    uint32_t *a_gpos = nullptr, *a_comp = nullptr, *a_idx = nullptr, *s_comp = nullptr, *s_idx = nullptr,
             *seg_key = nullptr, *l_gpos = nullptr;
    float4* bk_ds = nullptr;
    uint8_t* bk_flags = nullptr;
  } act_;
  DevBuf<unsigned long long> filter_masks_;   // k_filter's verdict: 3 x one bit per edge
  DevBuf<int32_t> block_cnt_, block_off_;
  int64_t optimistic_stages_ = 0, rollbacks_ = 0;
  DevBuf<int32_t> scalars_;   // num_active, num_segs, misc
  DevBuf<int32_t> seg_table_dev_;   // k_filter: the segments of the current stage
  DevBuf<int32_t> hub_excl_;        // exclusion list of the hub regions (device_graph.h: kFlagHubExcluded)
  PinnedBuf<int32_t> iv_host_;      // read-out: the scan intervals on the host (label, frame|y, lx, rx)
  std::vector<int32_t> seg_table_host_, list_off_host_;
  Mailbox mail_;              // host-visible scalars (device_graph.h); mapped host memory
  void* mail_mem_ = nullptr;
  DevBuf<int32_t> zero_pool_mem_;
  ZeroPool zero_pool_;
  DevBuf<int32_t> bucket_prefix_dev_;   // edges in the buckets before b
  int spine_limit_bucket_ = 0x7fffffff;   // learned per stream: where the tree replay stops paying
  int spine_limit_age_ = 0;               // chunks since it was learned (forgotten after 8: one atypical
                                          // chunk must not switch the tree replay off for good)
  int spine_low_fails_[2] = {0, 0}, spine_low_cooldown_[2] = {0, 0};   // the same for buckets 0 and 1
  // ... and what the bucket cost (ms) the last time the replay failed in it / the last time it went
  // without: a bucket is skipped only while that is known to be the cheaper of the two (0: not known)
  double spine_low_cost_fail_[2] = {0, 0}, spine_low_cost_skip_[2] = {0, 0};
  int spine_low_cost_age_[2] = {0, 0};
  DevBuf<int32_t> spine_pool_;   // scratch of the Kruskal-tree replay (merge_spine.hip)
  static constexpr int64_t kNoWindowTarget = 1ll << 40;
  int64_t wave_target_active_ = kNoWindowTarget;   // active edges per stage (SegmentLists), learned
  std::vector<int64_t> window_target_;             // ... per bucket, from the last chunk (0: none yet)
  int window_target_age_ = 0;
  std::vector<int> window_last_seg_;      // per bucket: largest wave segment when the target was last halved
  std::vector<uint8_t> window_frozen_;    // per bucket: halving stopped paying
  std::vector<uint8_t> window_unpaid_;    // per bucket: halvings in a row that did not pay
  std::vector<int> window_peak_;          // per bucket: largest component of a stage since the targets were last reviewed
  // Hub regions per bucket (group): stages that used them and regions they absorbed in the chunk at
  // hand, and for how many more chunks the bucket goes without (its hubs absorbed next to nothing:
  // the pipeline around them -- marks, log, sort, one more host wait per stage -- is all cost there).
  std::vector<int> hub_bucket_stages_, hub_bucket_pause_;
  std::vector<long long> hub_bucket_absorbed_;
  double last_density_ = 1.0;                      // active / all edges of the last measured stage
  int spine_max_edges_grown_ = 0;   // what the pool was enlarged to for this video's largest stage
  DevBuf<unsigned long long> stats_;
  DevBuf<uint8_t> cub_temp_;      // temporary storage of the library radix sort
  DevBuf<int32_t> scan_sums_;     // tile sums of the hand-written scans (scan_device.h)
  size_t scratch_edges_ = 0;    // capacity of the per-edge stage scratch (EnsureScratch)
  size_t scratch_active_ = 0;   // capacity of the per-active-edge stage scratch (EnsureActiveScratch)
  // flow sampling for the tube analysis
  DevBuf<int32_t> flow_req_dev_;
  DevBuf<float2> flow_samples_dev_;
  DevBuf<const float*> flow_ptrs_dev_;
  void SampleFlows(const std::vector<FlowRequest>& req, const std::vector<const float*>& dev_flows,
                   std::vector<float>* samples);
  // readout scratch
  DevBuf<int32_t> row_counts_, row_offsets_;
  DevBuf<int32_t> iv_label_, iv_lx_, iv_rx_;
  DevBuf<uint32_t> iv_ty_;
  DevBuf<unsigned long long> pairs_, order_keys_, pairs_sorted_, pairs_unique_;
  DevBuf<int32_t> small_i32_a_, small_i32_b_, small_i32_c_;
  DevBuf<float4> small_f4_;
  // persistent small scratch (MergeConstrainedRegions write-back, tube relabelling, unseen keys)
  DevBuf<int32_t> mc_ids_, mc_par_, mc_cons_, mc_fl_, rl_lx_, rl_rx_, rl_nl_, un_keys_;
  DevBuf<float4> mc_ds_;
  DevBuf<uint32_t> rl_ty_;
  DevBuf<unsigned long long> un_min_;
  std::vector<hipEvent_t> ev_pool_;
  std::vector<std::pair<int, int>> ev_wave_, ev_filter_, ev_spine_;
  int ev_used_ = 0;

  std::vector<RegionInfo> regions_;
  std::unordered_map<int, int> key_to_region_;   // representative key -> index into regions_
  std::unordered_map<int, int> key_size_override_;   // regions erased by N4: size forced to 0
  int next_region_index_ = 0;
  GraphTimings timings_;
};

}  // namespace vsg

#endif  // VSG_DENSE_GRAPH_H_
