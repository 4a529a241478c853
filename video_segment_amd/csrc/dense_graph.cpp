// dense_graph.cpp -- host orchestration of the device graph (see dense_graph.h).
//
// Reference call sites mirrored: DenseSegmentationGraph::AddNodesAndSpatialEdges*,
// AddVirtualNodesConstrained, AddTemporal*Edges*, SegmentFullGraph, ObtainResults,
// DetermineNeighborIds (segmentation/dense_segmentation_graph.h:83-160, 327-579) and
// FastSegmentationGraph::SegmentGraph / MergeConstrainedRegions / DetermineNeighborIdsImpl
// (segmentation/segmentation_graph.h:339-496, 703-786).
#include "dense_graph.h"
#include "scan_device.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <atomic>
#include <functional>
#include <thread>
#include <unordered_set>

namespace vsg {

namespace {
// Persistent scratch that only ever grows (with slack): no hipMalloc / hipFree in the steady state.
template <class T>
DevBuf<T>& Grow(DevBuf<T>& b, size_t n) {
  if (b.size() < n) b.alloc(n + n / 2 + 1024);
  return b;
}
double NowMs() {
  using clk = std::chrono::steady_clock;
  return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}
template <class T>
void D2H(T* dst, const T* src, size_t n, hipStream_t s) {
  if (n == 0) return;
  VSG_HIP(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, s));
}
template <class T>
void H2D(T* dst, const T* src, size_t n, hipStream_t s) {
  if (n == 0) return;
  VSG_HIP(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, s));
}
}  // namespace

DenseGraphHip::DenseGraphHip(int W, int H, int max_frames, bool l1, hipStream_t stream)
    : W_(W), H_(H), capacity_frames_(max_frames), max_frames_(max_frames), l1_(l1),
      stream_(stream), wh_((size_t)W * H) {
  VSG_REQUIRE(W >= 2 && H >= 1 && W <= 65535 && H <= 65535, -1, "unsupported frame size");
  VSG_REQUIRE(max_frames >= 1 && max_frames < 4096, -1, "unsupported number of frames");
  const size_t N = wh_ * (size_t)max_frames;
  VSG_REQUIRE(N * 9 < (size_t)0xFFFFFFFFu, -1, "graph too large for 32-bit edge positions");
  parent_.alloc(N);
  desc_sz_.alloc(N);
  cons_.alloc(N);
  flags_.alloc(N);
  hub8_.alloc(N);
  cc_.alloc(N);
  label_uf_.alloc(N);
  label_img_.alloc(N);
  adjust_.alloc(N);
  lists_.resize(2 * max_frames - 1);
  list_desc_dev_.alloc(lists_.size());
  list_slot_base_dev_.alloc(lists_.size() + 1);
  const size_t total_slots = (size_t)(4 * max_frames + 9 * (max_frames - 1)) * wh_;
  kept_all_.alloc(total_slots);
  bucket_base_dev_.alloc((size_t)(kNumBuckets + 1) * (lists_.size() + 1));
  keys_tmp_.alloc(9 * wh_ + 8);
  hist_tmp_.alloc(EdgeSortHistInts(wh_));
  hist_sums_.alloc(EdgeSortSumInts(wh_) + 1);
  scalars_.alloc(32);
  stats_.alloc(96);
  hub_excl_.alloc(kHubListInts);
  VSG_HIP(hipMemsetAsync(hub_excl_.get(), 0, 4 * sizeof(int32_t), stream_));   // (the count; blocks of the cache are not zeroed)
  {
    // mailbox: 256 slots of four words + a list of 2 * 4095 ints, mapped and coherent (the device
    // writes it while kernels run, the host polls it)
    const int slots = 256;
    const size_t list_cap = 8192;
    const size_t bytes = (size_t)slots * kMailValues * sizeof(unsigned long long) + list_cap * sizeof(int32_t);
    mail_mem_ = CacheAlloc(bytes, kCacheMappedCoherent);
    std::memset(mail_mem_, 0, bytes);
    void* dev = nullptr;
    VSG_HIP(hipHostGetDevicePointer(&dev, mail_mem_, 0));
    mail_.host = static_cast<volatile unsigned long long*>(mail_mem_);
    mail_.dev = static_cast<unsigned long long*>(dev);
    mail_.slots = slots;
    mail_.list_host = reinterpret_cast<volatile int32_t*>(static_cast<char*>(mail_mem_) +
                                                          (size_t)slots * kMailValues * sizeof(unsigned long long));
    mail_.list_dev = reinterpret_cast<int32_t*>(static_cast<char*>(dev) +
                                                (size_t)slots * kMailValues * sizeof(unsigned long long));
    mail_.list_cap = (int)list_cap;
    zero_pool_mem_.alloc(1 << 20);
    zero_pool_.base = zero_pool_mem_.get();
    zero_pool_.cap = zero_pool_mem_.size();
    if (const char* e = getenv("VSG_ZERO_POOL")) {   // test hook: a small pool changes halves all the time
      zero_pool_.cap = std::min<size_t>(zero_pool_.cap, std::max<size_t>((size_t)atoll(e), 4096) + kZeroArenaInts);
    }
  }
  VSG_HIP(hipStreamCreateWithFlags(&aux_stream_, hipStreamNonBlocking));
  VSG_HIP(hipStreamCreateWithFlags(&aux2_stream_, hipStreamNonBlocking));
  VSG_HIP(hipEventCreateWithFlags(&aux_fork_, hipEventDisableTiming));
  VSG_HIP(hipEventCreateWithFlags(&aux_join_, hipEventDisableTiming));

  scan_sums_.alloc(kScanMaxTiles);
  Reset(max_frames);
  MailRegisterGraph(1);
}

DenseGraphHip::~DenseGraphHip() {
  // (the owner has synchronised the handle's stream; one device synchronisation here instead of one
  // per buffer that goes back to the cache)
  quiesce_.Begin();
  MailRegisterGraph(-1);
  for (hipEvent_t e : ev_pool_) (void)hipEventDestroy(e);
  if (aux_fork_) (void)hipEventDestroy(aux_fork_);
  if (aux_join_) (void)hipEventDestroy(aux_join_);
  if (aux_stream_) (void)hipStreamDestroy(aux_stream_);
  if (aux2_stream_) (void)hipStreamDestroy(aux2_stream_);
  if (mail_mem_) CacheFree(mail_mem_);
}

void DenseGraphHip::Reset(int max_frames) {
  VSG_REQUIRE(max_frames >= 1 && max_frames <= capacity_frames_, -1, "max_frames above capacity");
  max_frames_ = max_frames;
  num_frames_ = 0;
  has_constraints_ = false;
  virtual_slices_.clear();
  flattened_ = false;
  spatial_pass_done_ = false;
  halo_pending_ = false;
  deferred_virtual_slice_ = deferred_constrained_slice_ = -1;
  for (auto& lb : lists_) lb.used = false;
  regions_.clear();
  key_to_region_.clear();
  key_size_override_.clear();
  next_region_index_ = 0;
  timings_ = GraphTimings();
}

void DenseGraphHip::ForgetLearned() {
  spine_limit_bucket_ = 0x7fffffff;
  spine_limit_age_ = 0;
  for (int b = 0; b < 2; ++b) {
    spine_low_fails_[b] = spine_low_cooldown_[b] = spine_low_cost_age_[b] = 0;
    spine_low_cost_fail_[b] = spine_low_cost_skip_[b] = 0;
  }
  wave_target_active_ = kNoWindowTarget;
  window_target_.clear();
  hub_bucket_pause_.clear();
  window_target_age_ = 0;
  last_density_ = 1.0;
}

void DenseGraphHip::SortList(ListBuf& lb, int per_px) {
  const int n = (int)(per_px * wh_);
  lb.slots.ensure((size_t)n);
  lb.offsets.ensure(kBucketSlots);
  LaunchBucketSort(keys_tmp_.get(), wh_, per_px, hist_tmp_.get(), hist_sums_.get(),
                   lb.offsets.get(), lb.slots.get(), stream_);
  lb.n = n;
  lb.used = true;
}

void DenseGraphHip::AddFrame(const float* feat, const int32_t* cons_dev) {
  VSG_REQUIRE(num_frames_ < max_frames_, -1, "more frames than max_frames (CHECK_LE)");
  const int t = num_frames_;
  const int base = (int)(wh_ * t);
  LaunchInitNodes(feat, wh_, base, cons_dev, nodes(), stream_);
  ListBuf& lb = lists_[2 * t];
  lb.type = 0;
  lb.base_a = base;
  lb.base_b = base;
  LaunchSpatialKeys(feat, W_, H_, l1_ ? 1 : 0, keys_tmp_.get(), hist_tmp_.get(), stream_);
  SortList(lb, 4);
  if (cons_dev) has_constraints_ = true;
  ++num_frames_;
}

void DenseGraphHip::AddVirtualFrame(const int32_t* ids_dev, int max_label) {
  VSG_REQUIRE(num_frames_ < max_frames_, -1, "more frames than max_frames (CHECK_LE)");
  VSG_REQUIRE(max_label >= 1, -1, "max_label must be positive");
  const int t = num_frames_;
  first_label_scratch_.ensure((size_t)max_label);
  LaunchInitVirtualNodes(ids_dev, wh_, (int)(wh_ * t), max_label, first_label_scratch_.get(),
                         nodes(), stream_);
  virtual_slices_.push_back(t);
  has_constraints_ = true;
  ++num_frames_;
}

void DenseGraphHip::AddVirtualFrameDeferred() {
  VSG_REQUIRE(num_frames_ < max_frames_, -1, "more frames than max_frames (CHECK_LE)");
  deferred_virtual_slice_ = num_frames_;
  deferred_constrained_slice_ = num_frames_ + 1;
  virtual_slices_.push_back(num_frames_);
  has_constraints_ = true;
  halo_pending_ = true;
  ++num_frames_;
}

void DenseGraphHip::SetHaloLabels(const int32_t* virtual_ids_dev, const int32_t* constrained_ids_dev,
                                  int max_label) {
  VSG_REQUIRE(halo_pending_, -3, "no deferred virtual slice");
  VSG_REQUIRE(num_frames_ > deferred_constrained_slice_, -3,
              "the constrained slice has to be added before its labels");
  VSG_REQUIRE(max_label >= 1, -1, "max_label must be positive");
  first_label_scratch_.ensure((size_t)max_label);
  LaunchInitVirtualNodes(virtual_ids_dev, wh_, (int)(wh_ * deferred_virtual_slice_), max_label,
                         first_label_scratch_.get(), nodes(), stream_);
  // the constrained slice keeps the features it was initialised with; only its labels arrive
  VSG_HIP(hipMemcpyAsync(cons_.get() + wh_ * (size_t)deferred_constrained_slice_, constrained_ids_dev,
                         wh_ * sizeof(int32_t), hipMemcpyDeviceToDevice, stream_));
  halo_pending_ = false;
}

void DenseGraphHip::AddTemporal(const float* cur, const float* prev, const float* flow,
                                bool is_virtual) {
  VSG_REQUIRE(num_frames_ >= 2, -3, "temporal edges need two slices");
  const int t = num_frames_ - 1;
  ListBuf& lb = lists_[2 * t - 1];
  lb.type = 1;
  lb.base_a = (int)(wh_ * t);
  lb.base_b = (int)(wh_ * (t - 1));
  lb.prev_idx.ensure(wh_);
  LaunchTemporalKeys(cur, prev, flow, W_, H_, l1_ ? 1 : 0, is_virtual ? 1 : 0, keys_tmp_.get(),
                     lb.prev_idx.get(), hist_tmp_.get(), stream_);
  SortList(lb, 9);
}

void DenseGraphHip::FinishBuilding() { VSG_HIP(hipStreamSynchronize(stream_)); }

// Stage scratch comes in two sizes.  Per EDGE of a stage (the largest bucket: 139 M edges at 1080p):
// only what k_filter leaves for every edge -- the packed records of the active / settled edges, three
// mask bits, the per-block counts.  Per ACTIVE edge (a few per cent of a stage, 30 % in the bucket of
// the giant components): everything the compaction, the run leaders, the sort by component, the
// workers' segments and the undo buffers need -- 106 bytes per edge, which sized for the largest
// bucket was 18 of the 29 GB a 1080p stream held.  The active arrays grow on demand (RunBucketStage
// learns the number of active edges before it uses them) and keep their size between chunks.
void DenseGraphHip::EnsureScratch(size_t n) {
  if (n <= scratch_edges_) return;
  n = n + n / 8 + 1024;
  e_ra_.alloc(n);
  e_rb_.alloc(n);
  e_gpos_.alloc(n);
  // three mask words per 64 edges (rounded up to whole workgroups of 256 edges)
  filter_masks_.alloc(3 * (4 * ((n + 255) / 256) + 4));
  block_cnt_.alloc((n + 255) / 256 + 2);
  block_off_.alloc((n + 255) / 256 + 2);
  scratch_edges_ = n;
}

void DenseGraphHip::EnsureActiveScratch(size_t n) {
  if (n <= scratch_active_) return;
  // (half as much again: a stage that needs more than the last one is usually followed by one that
  // needs more still, and every growth drains the streams)
  n = std::max(n + n / 8 + 1024, scratch_active_ + scratch_active_ / 2);
  size_t bytes = 0;
  auto carve = [&bytes](size_t elems, size_t elem_size) {
    const size_t at = bytes;
    bytes += (elems * elem_size + 255) & ~(size_t)255;
    return at;
  };
  const size_t o_e_active = carve(n, 4), o_e_apos = carve(n, 4), o_a_ra = carve(n, 4), o_a_rb = carve(n, 4),
               o_a_gpos = carve(n, 4), o_a_comp = carve(n, 4), o_a_idx = carve(n, 4), o_s_comp = carve(n, 4),
               o_s_idx = carve(n, 4), o_seg_key = carve(n, 4), o_seg_cnt = carve(n, 4), o_seg_off = carve(n, 4),
               o_lead_pos = carve(n, 4), o_l_ra = carve(n, 4), o_l_rb = carve(n, 4), o_l_gpos = carve(n, 4),
               o_bk_ds = carve(2 * n, sizeof(float4)), o_bk_cons = carve(2 * n, 4), o_bk_flags = carve(2 * n, 1);
  // (an allocation that throws must not leave the old pointers behind: the slab they point into is
  // released first, and a later call would return early on the old capacity and bind them)
  scratch_active_ = 0;
  act_ = ActiveArrays();
  active_slab_.alloc(bytes);
  uint8_t* b = active_slab_.get();
  act_.e_active = reinterpret_cast<int32_t*>(b + o_e_active);
  act_.e_apos = reinterpret_cast<int32_t*>(b + o_e_apos);
  act_.a_ra = reinterpret_cast<int32_t*>(b + o_a_ra);
  act_.a_rb = reinterpret_cast<int32_t*>(b + o_a_rb);
  act_.a_gpos = reinterpret_cast<uint32_t*>(b + o_a_gpos);
  act_.a_comp = reinterpret_cast<uint32_t*>(b + o_a_comp);
  act_.a_idx = reinterpret_cast<uint32_t*>(b + o_a_idx);
  act_.s_comp = reinterpret_cast<uint32_t*>(b + o_s_comp);
  act_.s_idx = reinterpret_cast<uint32_t*>(b + o_s_idx);
  act_.seg_key = reinterpret_cast<uint32_t*>(b + o_seg_key);
  act_.seg_cnt = reinterpret_cast<int32_t*>(b + o_seg_cnt);
  act_.seg_off = reinterpret_cast<int32_t*>(b + o_seg_off);
  act_.lead_pos = reinterpret_cast<int32_t*>(b + o_lead_pos);
  act_.l_ra = reinterpret_cast<int32_t*>(b + o_l_ra);
  act_.l_rb = reinterpret_cast<int32_t*>(b + o_l_rb);
  act_.l_gpos = reinterpret_cast<uint32_t*>(b + o_l_gpos);
  act_.bk_ds = reinterpret_cast<float4*>(b + o_bk_ds);
  act_.bk_cons = reinterpret_cast<int32_t*>(b + o_bk_cons);
  act_.bk_flags = b + o_bk_flags;
  // (the largest sort of a stage: the arcs of the spanning trees, two per tree edge -- merge_spine.hip)
  const size_t temp = SortPairsU32TempBytes((int)std::min<size_t>(2 * n, 0x7fffffff));
  if (temp > cub_temp_.size()) cub_temp_.alloc(temp);
  scratch_active_ = n;
}

// ---------------------------------------------------------------------------------------------
// SegmentFullGraph
// ---------------------------------------------------------------------------------------------
// One stage, with the per-stage counters of VSG_DEBUG_STAGES printed around it.
void DenseGraphHip::RunStageDebug(int b, int w, int windows, int j0, int n, const MergeParams& P,
                                  int inert_mode, MergeScratch& S, bool debug_stages, StageInfo* info) {
  unsigned long long s0[96] = {0}, s1[96] = {0};
  double ts = 0;
  size_t ev0 = 0, evf0 = 0;
  if (debug_stages) {
    VSG_HIP(hipMemsetAsync(stats_.get() + 16, 0, 2 * sizeof(unsigned long long), stream_));
    VSG_HIP(hipMemsetAsync(stats_.get() + 48, 0, 16 * sizeof(unsigned long long), stream_));
    D2H(s0, stats_.get(), 96, stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
    ts = NowMs();
    ev0 = ev_wave_.size();
    evf0 = ev_filter_.size();
  }
  RunBucketStage(b, j0, n, list_desc_dev_.get(), bucket_base_dev_.get(), list_slot_base_dev_.get(),
                 kept_all_.get(), nodes(), P, inert_mode, S, stream_, info);
  if (!debug_stages) return;
  D2H(s1, stats_.get(), 96, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
  const double te = NowMs();
  float wave_ms = 0;
  for (size_t k = ev0; k < ev_wave_.size(); ++k) {
    float ms = 0;
    VSG_HIP(hipEventElapsedTime(&ms, ev_pool_[ev_wave_[k].first], ev_pool_[ev_wave_[k].second]));
    wave_ms += ms;
  }
  float filter_ms = 0;
  for (size_t k = evf0; k < ev_filter_.size(); ++k) {
    float ms = 0;
    VSG_HIP(hipEventElapsedTime(&ms, ev_pool_[ev_filter_[k].first], ev_pool_[ev_filter_[k].second]));
    filter_ms += ms;
  }
  if (te - ts <= 1.0) return;
  std::fprintf(stderr, "[vsg] stage b=%d w=%d/%d n=%d wall %.2f ms filter %.2f ms workers %.2f ms | replayed %d components %d | "
               "wave edges %llu batches %llu rounds %llu nwin %llu chain %llu | max_seg %llu slowest %.2f Mcyc | "
               "cyc load %.0f M loop %.0f M wait %.0f M\n",
               b, w, windows, n, te - ts, filter_ms, wave_ms, info ? info->replayed : -1, info ? info->components : -1,
               s1[3] - s0[3], s1[7] - s0[7], s1[5] - s0[5], s1[4] - s0[4], s1[20] - s0[20],
               s1[17], s1[16] / 1e6, (s1[18] - s0[18]) / 1e6, (s1[19] - s0[19]) / 1e6,
               (s1[26] - s0[26]) / 1e6);
  std::fprintf(stderr, "[vsg]   largest: %.2f Mcyc batches %llu rounds %llu live %llu chain %llu nwin %llu | kcyc/batch: "
               "wait %.1f stage+table %.1f loop %.1f (reserve %.1f closure %.1f masks+generic %.1f chain %.1f) | cuts %llu kept %llu\n",
               s1[48] / 1e6, s1[49], s1[50], s1[51], s1[55], s1[56],
               s1[52] / 1e3 / std::max(1.0, (double)s1[49]), s1[53] / 1e3 / std::max(1.0, (double)s1[49]),
               s1[54] / 1e3 / std::max(1.0, (double)s1[49]), s1[57] / 1e3 / std::max(1.0, (double)s1[49]),
               s1[58] / 1e3 / std::max(1.0, (double)s1[49]), s1[59] / 1e3 / std::max(1.0, (double)s1[49]),
               s1[60] / 1e3 / std::max(1.0, (double)s1[49]), s1[61], s1[62]);
  std::fprintf(stderr, "[vsg]   chain ends (%llu): not owner %llu, flags %llu, constraint %llu, larger %llu, failed/mode %llu, "
               "hot small %llu, none %llu\n", s1[71] - s0[71], s1[64] - s0[64], s1[65] - s0[65], s1[66] - s0[66],
               s1[67] - s0[67], s1[68] - s0[68], s1[69] - s0[69], s1[70] - s0[70]);
}

void DenseGraphHip::SegmentLists(int min_region_size, bool force_constraints, int pass) {
  VSG_REQUIRE(num_frames_ >= 1, -3, "no frames");
  VSG_REQUIRE(!halo_pending_, -3, "the labels of the previous chunk have not been imported");
  min_region_size_ = min_region_size;
  const int L = (int)lists_.size();
  const size_t N = wh_ * (size_t)num_frames_;
  const double t_enter = NowMs();
  const ThreadAllocCounters alloc0 = ThreadAllocSnapshot();
  MailWaitResetLongest();
  const MailWaitCounters mail0 = MailWaitSnapshot();
  int64_t diag_stages = 0, diag_slab_growths = 0, diag_spine_growths = 0, diag_hub_stages = 0, diag_hub_absorbed = 0;
  double diag_slab_ms = 0, diag_spine_ms = 0;

  // List table.
  std::vector<ListDesc> desc(L);
  list_slot_base_.assign(L + 1, 0);
  uint32_t acc = 0;
  int64_t edges_total = 0;
  for (int l = 0; l < L; ++l) {
    ListBuf& lb = lists_[l];
    list_slot_base_[l] = acc;
    ListDesc d = {};
    if (lb.used) {
      d.slots = lb.slots.get();
      d.kept = kept_all_.get() + acc;
      d.prev_idx = lb.type == 1 ? lb.prev_idx.get() : nullptr;
      // a spatial-only pass does not see the temporal lists (their positions in kept_all stay)
      d.offsets = (pass == 1 && lb.type == 1) ? nullptr : lb.offsets.get();
      d.type = lb.type;
      d.base_a = lb.base_a;
      d.base_b = lb.base_b;
      d.n = lb.n;
      acc += (uint32_t)lb.n;
    }
    desc[l] = d;
  }
  list_slot_base_[L] = acc;
  H2D(list_desc_dev_.get(), desc.data(), (size_t)L, stream_);
  H2D(list_slot_base_dev_.get(), list_slot_base_.data(), (size_t)L + 1, stream_);
  if (pass == 2) {
    // Edges of the spatial lists that the spatial pass did not keep are gone
    // (segmentation_graph.h:442: the kept edges replace the bucket's contents).
    kept_spatial_pass_.ensure(kept_all_.size());
    VSG_HIP(hipMemcpyAsync(kept_spatial_pass_.get(), kept_all_.get(), acc, hipMemcpyDeviceToDevice,
                           stream_));
  }
  VSG_HIP(hipMemsetAsync(kept_all_.get(), 0, acc, stream_));
  LaunchBuildBucketTable(list_desc_dev_.get(), L, bucket_base_dev_.get(), stream_);
  bucket_base_host_.resize((size_t)(kNumBuckets + 1) * (L + 1));
  D2H(bucket_base_host_.data(), bucket_base_dev_.get(), bucket_base_host_.size(), stream_);
  VSG_HIP(hipMemsetAsync(stats_.get(), 0, 96 * sizeof(unsigned long long), stream_));
  // (the counters of the last chunk's stages: nothing is in flight on the other streams here)
  VSG_HIP(hipMemsetAsync(zero_pool_.base, 0, zero_pool_.cap * sizeof(int32_t), stream_));
  zero_pool_.used = 0;
  zero_pool_.second_half = false;
  zero_pool_.arena_used = 0;
  LaunchInitIdentity(cc_.get(), N, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));

  // Consecutive buckets above the force-merge weight share their thresholds, and a stage is exact
  // for any consecutive range of the edge sequence (RunBucketStage): the long tail of nearly empty
  // buckets is replayed in groups, hundreds of stages of a few edges each would cost more in
  // launches than in work.  A group must stay small, though: its filter sees the regions as they
  // are when the group starts, so what the group's own earlier buckets settle (regions finalized
  // by a failed test make their later edges inert) is only known to the workers.  The width
  // follows the number of edges the last group had to replay.
  const bool group_buckets = !getenv("VSG_GROUP_BUCKETS") || atoi(getenv("VSG_GROUP_BUCKETS")) != 0;
  std::vector<int32_t> bucket_prefix(kNumBuckets + 2, 0);
  int first_plain = kNumBuckets;
  {
    const float scale = 2048.0f / (1.0f + 1e-6f);
    const float inv_scale = (float)(1.0 / (double)scale);
    const float force_w = l1_ ? 0.002f : 0.001f;
    for (int b = 0; b < kNumBuckets; ++b) {
      if (!((float)b * inv_scale < force_w)) {
        first_plain = b;
        break;
      }
    }
  }
  int64_t n_max = 0;
  for (int b = 0; b <= kNumBuckets; ++b) {
    const int tot = bucket_base_host_[(size_t)b * (L + 1) + L];
    edges_total += tot;
    bucket_prefix[b + 1] = bucket_prefix[b] + tot;
  }
  for (int b = 0; b < kNumBuckets; ++b) n_max = std::max<int64_t>(n_max, bucket_prefix[b + 1] - bucket_prefix[b]);
  VSG_REQUIRE(bucket_prefix[kNumBuckets] >= 0, -1, "too many edges");
  EnsureScratch((size_t)std::max<int64_t>(n_max, 1));
  // (a first guess for a handle that has not seen a chunk yet: a sixth of the largest bucket (a quarter
  // until round 6: the rest of a bucket is now replayed in pieces of at most 1.4 N edges, below).  The
  // stage of the giant components of a first, unconstrained 1080p chunk has 61 M active edges of
  // 139 M and the window graph of configs[1] 9.5 M of 17 M -- one growth of the slab each, a
  // millisecond --, a 3840x2160 chunk stays below the quarter, and half the bucket would be 33 GB
  // there.  Later chunks start with what the earlier ones needed.)
  if (const char* e = getenv("VSG_ACTIVE_CAP")) {   // test hook: start small, grow inside the stages
    EnsureActiveScratch((size_t)std::max(1, atoi(e)));
  } else {
    EnsureActiveScratch(std::min<size_t>((size_t)std::max<int64_t>(n_max, 1),
                                         std::max<size_t>((size_t)n_max / 6, (size_t)1 << 20)));
  }
  bucket_prefix_dev_.ensure(bucket_prefix.size());
  H2D(bucket_prefix_dev_.get(), bucket_prefix.data(), bucket_prefix.size(), stream_);

  MergeScratch S = {};
  auto bind_scratch = [this, &S]() {
    S.e_ra = e_ra_.get();
    S.e_rb = e_rb_.get();
    S.e_gpos = e_gpos_.get();
    S.e_active = act_.e_active;
    S.e_apos = act_.e_apos;
    S.a_ra = act_.a_ra;
    S.a_rb = act_.a_rb;
    S.a_gpos = act_.a_gpos;
    S.a_comp = act_.a_comp;
    S.a_idx = act_.a_idx;
    S.s_comp = act_.s_comp;
    S.s_idx = act_.s_idx;
    S.seg_key = act_.seg_key;
    S.seg_cnt = act_.seg_cnt;
    S.seg_off = act_.seg_off;
    S.lead_pos = act_.lead_pos;
    S.l_ra = act_.l_ra;
    S.l_rb = act_.l_rb;
    S.l_gpos = act_.l_gpos;
    S.bk_ds = act_.bk_ds;
    S.bk_cons = act_.bk_cons;
    S.bk_flags = act_.bk_flags;
    S.cub_temp = cub_temp_.get();
    S.cub_temp_bytes = cub_temp_.size();
    S.scan = ScanScratch{scan_sums_.get(), stream_};
    S.active_cap = (int)std::min<size_t>(scratch_active_, 0x7fffffff);
  };
  bind_scratch();
  S.grow_active = [this, bind_scratch, &diag_slab_growths, &diag_slab_ms](long long need) {
    // (nothing of the stage is in the active arrays yet: RunBucketStage asks before it uses them)
    const double tg0 = NowMs();
    VSG_HIP(hipStreamSynchronize(stream_));
    VSG_HIP(hipStreamSynchronize(aux_stream_));
    VSG_HIP(hipStreamSynchronize(aux2_stream_));
    const size_t before = scratch_active_;
    EnsureActiveScratch((size_t)need);
    bind_scratch();
    ++diag_slab_growths;
    diag_slab_ms += NowMs() - tg0;
    if (getenv("VSG_DEBUG_STATS")) {
      std::fprintf(stderr, "[vsg] stage scratch: %zu -> %zu active edges (needed %lld), %.2f ms\n", before,
                   scratch_active_, need, NowMs() - tg0);
    }
    return true;
  };
  S.num_active = scalars_.get();
  S.num_segs = scalars_.get() + 1;
  S.node_key_bits = 1;
  while (S.node_key_bits < 32 && ((size_t)1 << S.node_key_bits) < N) ++S.node_key_bits;
  {
    const size_t words = 4 * ((scratch_edges_ + 255) / 256) + 4;
    S.masks.active = filter_masks_.get();
    S.masks.settled = filter_masks_.get() + words;
    S.masks.tentative = filter_masks_.get() + 2 * words;
    S.masks.block_cnt = block_cnt_.get();
    S.block_off = block_off_.get();
  }
  S.force_rollback = getenv("VSG_FORCE_ROLLBACK") ? 1 : 0;
  S.small_seg = getenv("VSG_SMALL_SEG") ? std::max(1, atoi(getenv("VSG_SMALL_SEG"))) : 24;
  // The wide worker (merge_wide.hip: several wavefronts replay one component in lock-step rounds) is
  // OFF by default: measured in round 6, it is exact (every parity test with VSG_WIDE_MIN=25) and no
  // faster -- half the edges of a percolating component hang on ONE region, whose chain is cut at
  // the first blocked lane wherever the batch ends, so a batch of 256 lanes takes four times the
  // rounds of a batch of 64.  VSG_WIDE_MIN=n hands it the components of at least n
  // replayed edges (the largest size class of the work list then starts there).
  S.wide_min = getenv("VSG_WIDE_MIN") ? atoi(getenv("VSG_WIDE_MIN")) : 0;
  S.wide_waves = getenv("VSG_WIDE_WAVES") ? atoi(getenv("VSG_WIDE_WAVES")) : 4;
  S.chain_relax = getenv("VSG_CHAIN_RELAX") ? atoi(getenv("VSG_CHAIN_RELAX")) : 1;
  const int hubs_default = getenv("VSG_HUBS") ? atoi(getenv("VSG_HUBS")) : 1;
  S.hubs = hubs_default;
  S.hub_excl = hub_excl_.get();
  S.hub_max_splits = getenv("VSG_HUB_SPLITS") ? std::max(0, atoi(getenv("VSG_HUB_SPLITS"))) : kHubMaxSplits;
  S.hub_cut_min_work = getenv("VSG_CUT_MIN_WORK") ? atoi(getenv("VSG_CUT_MIN_WORK")) : 0;
  // (a stage takes its marks off again; an exception in the middle of one must not leave any behind)
  VSG_HIP(hipMemsetAsync(hub8_.get(), 0, N, stream_));
  // Sizes that follow the graph rather than the 1080p bench: the tree replay's scratch pool holds
  // the large components of one stage, which together are at most about one bucket of the chunk
  // graph (1.15 N edges: 48 M at 1080p x 21 slices; the stamped ranks of the spanning forest allow
  // 2^27); a component is worth the tree replay from a fixed number of edges on (what one
  // wavefront replays in about a millisecond), whatever the frame size.
  // (re-tuned in round 4, once the tree machinery had got cheaper: 4096 -> 2048, nested levels from
  // the same size on)
  S.spine_min = getenv("VSG_SPINE_MIN") ? atoi(getenv("VSG_SPINE_MIN")) : 2048;
  {
    // (the handle's capacity, not this chunk's N.  Round 6 measured 0.55 N: 1.7 GB less at 1080p and
    // the same time on the bench input, but a stage that then wants more makes the pool grow by the
    // dearer formula below -- 8 GB instead of 3.2 -- and at 3840x2160 the large components no longer
    // fit the 2^27 edges the stamps address: kept at 1.15 N)
    const double want = 1.15 * (double)wh_ * (double)capacity_frames_;
    const int by_graph = (int)std::min<double>(std::max<double>(want, 1 << 20), 120 << 20);
    S.spine_max_edges = getenv("VSG_SPINE_MAX_EDGES") ? atoi(getenv("VSG_SPINE_MAX_EDGES"))
                                                      : std::max(by_graph, spine_max_edges_grown_);
  }
  S.spine_off = 0;
  if (spine_limit_bucket_ != 0x7fffffff && ++spine_limit_age_ > 8) {   // probe again
    spine_limit_bucket_ = 0x7fffffff;
    spine_limit_age_ = 0;
  }
  S.spine_limit_bucket = &spine_limit_bucket_;
  // The first two buckets hold the large components the tree replay is for; on an input where its
  // assumption keeps failing there (two chunks in a row) it is skipped for eight chunks, then
  // tried again.
  int spine_low_failed[2] = {0, 0}, spine_low_skip[2] = {0, 0};
  for (int b = 0; b < 2; ++b) {
    if (spine_low_cooldown_[b] > 0) {
      --spine_low_cooldown_[b];
      spine_low_skip[b] = 1;
    }
  }
  S.spine_low_failed = spine_low_failed;
  S.spine_low_skip = spine_low_skip;
  S.spine_nested_factor = getenv("VSG_SPINE_NESTED") ? std::max(1, atoi(getenv("VSG_SPINE_NESTED"))) : 1;
  S.spine_debug = getenv("VSG_SPINE_DEBUG") ? 1 : 0;
  S.spine_check = getenv("VSG_SPINE_CHECK") ? 1 : 0;
  S.rank_split_min = getenv("VSG_RANK_SPLIT_MIN") ? atoi(getenv("VSG_RANK_SPLIT_MIN")) : (1 << 20);
  S.spine_fast = getenv("VSG_SPINE_FAST") ? atoi(getenv("VSG_SPINE_FAST")) : 2;   // streamed passes
  S.spine_fast_min = getenv("VSG_SPINE_FAST_MIN") ? atoi(getenv("VSG_SPINE_FAST_MIN")) : 32768;
  if (S.spine_min > 0) {
    const size_t ints = SpinePoolInts((size_t)S.spine_max_edges);
    if (spine_pool_.size() < ints) spine_pool_.alloc(ints);
  }
  S.spine_pool = spine_pool_.get();
  S.spine_pool_ints = spine_pool_.size();
  S.grow_spine_pool = [this, &S, &diag_spine_growths, &diag_spine_ms](long long edges) {
    // up to what the rank stamps of the spanning forest address (2^27 edges, merge_spine.hip)
    if (getenv("VSG_SPINE_MAX_EDGES") || edges >= (120ll << 20)) return false;
    const long long want = std::min<long long>(edges + edges / 8, 120ll << 20);
    // SpinePoolInts prices a stage whose components are dense (a fifth to two thirds of the edges
    // are tree edges, the rest of the 16 ints per edge is room for nested levels); a stage this
    // large is the opposite -- most of the volume in a few components, nearly every node a tree
    // edge -- and rooting the trees takes 29 ints per tree edge on top of the 7 per edge.
    const long long nodes = (long long)wh_ * (long long)std::max(capacity_frames_, 1);
    const size_t ints = SpinePoolInts((size_t)want) + (size_t)(29 * std::min(want, nodes));
    if (ints <= spine_pool_.size()) return false;
    const double tg0 = NowMs();
    size_t free_b = 0, total_b = 0;
    VSG_HIP(hipMemGetInfo(&free_b, &total_b));
    if (free_b + spine_pool_.size() * sizeof(int32_t) < 2 * ints * sizeof(int32_t)) {
      // (what closed handles left in the cache counts as free: give it back and look again)
      int dev = 0;
      VSG_HIP(hipGetDevice(&dev));
      CacheTrim(dev);
      VSG_HIP(hipMemGetInfo(&free_b, &total_b));
      if (free_b + spine_pool_.size() * sizeof(int32_t) < 2 * ints * sizeof(int32_t)) return false;
    }
    VSG_HIP(hipStreamSynchronize(stream_));
    VSG_HIP(hipStreamSynchronize(aux_stream_));
    VSG_HIP(hipStreamSynchronize(aux2_stream_));
    spine_pool_.alloc(ints);
    ++diag_spine_growths;
    diag_spine_ms += NowMs() - tg0;
    S.spine_pool = spine_pool_.get();
    S.spine_pool_ints = spine_pool_.size();
    S.spine_max_edges = (int)want;
    spine_max_edges_grown_ = S.spine_max_edges;
    return true;
  };
  S.nmap[0] = label_uf_.get();
  S.nmap[1] = label_img_.get();
  S.nmap[2] = adjust_.get();
  S.aux_stream = aux_stream_;
  S.aux2_stream = aux2_stream_;
  S.aux_fork = aux_fork_;
  S.aux_join = aux_join_;
  S.use_rle = getenv("VSG_RLE") ? atoi(getenv("VSG_RLE")) : 1;
  S.wave_dbg = getenv("VSG_WAVE_DBG") ? atoi(getenv("VSG_WAVE_DBG")) : 0;
  S.wave_debug = (S.wave_dbg != 0 || getenv("VSG_DEBUG_STAGES") || getenv("VSG_DEBUG_STATS")) ? 1 : 0;
  optimistic_stages_ = 0;
  rollbacks_ = 0;
  S.optimistic_stages = &optimistic_stages_;
  S.rollbacks = &rollbacks_;
  S.cc = cc_.get();
  S.stats = stats_.get();
  ev_used_ = 0;
  ev_wave_.clear();
  ev_filter_.clear();
  ev_spine_.clear();
  S.ev_pool = &ev_pool_;
  S.ev_wave = &ev_wave_;
  S.ev_filter = &ev_filter_;
  S.ev_spine = &ev_spine_;
  S.ev_used = &ev_used_;
  S.mail = &mail_;
  S.zeros = &zero_pool_;
  S.main_stream = stream_;

  MergeParams P;
  P.spatial_survivors = pass == 2 ? kept_spatial_pass_.get() : nullptr;
  P.W = W_;
  P.H = H_;
  P.num_lists = L;
  P.min_region_size = min_region_size;
  P.force_merge_weight = l1_ ? 0.002f : 0.001f;
  const float scale = 2048.0f / (1.0f + 1e-6f);
  P.inv_scale = (float)(1.0 / (double)scale);
  // Largest float s whose correctly rounded sqrtf still satisfies the reference's comparison.
  auto largest_s = [](double t, bool inclusive) {
    auto ok = [&](float s) {
      const double r = (double)std::sqrt(s);   // float sqrt, correctly rounded
      return inclusive ? r <= t : r < t;
    };
    float s = (float)(t * t);
    while (!ok(s)) s = std::nextafter(s, 0.0f);
    while (ok(std::nextafter(s, 10.0f))) s = std::nextafter(s, 10.0f);
    return s;
  };
  P.s_lt_005 = largest_s((double)0.05f, false);   // d < MergeDistanceThreshold (0.05f)
  P.s_lt_02 = largest_s(0.2, false);              // dist < 0.2 (double literal)
  P.s_le_015 = largest_s((double)0.15f, true);    // !(d > SplitDistanceThreshold (0.15f))

  const double t0 = NowMs();
  int inert_mode = has_constraints_ ? 2 : 1;
  if (const char* e = getenv("VSG_INERT_MODE")) inert_mode = std::min(inert_mode, atoi(e));
  const bool debug_stages = getenv("VSG_DEBUG_STAGES") != nullptr;
  // Rank windows of bucket 0: six up to 1080p, more on larger frames (the components of a window
  // grow with the frame, the per-window chain of small launches does not), in proportion to the
  // frame area up to 12.  Merge per chunk at 3840x2160: 342-371 ms with six, 311-320 with 10-12,
  // 343 with 32; at 2560x1440: 149-151 with six, 143-147 with 10-12.
  const double frame_ratio = (double)W_ * (double)H_ / (1920.0 * 1080.0);
  const int default_windows = std::max(6, std::min(12, (int)std::lround(6.0 * frame_ratio)));
  const int num_windows = getenv("VSG_WINDOWS") ? std::max(1, atoi(getenv("VSG_WINDOWS"))) : default_windows;
  const int window_bushy = getenv("VSG_WINDOW_BUSHY") ? atoi(getenv("VSG_WINDOW_BUSHY")) : 64;
  const bool adapt_windows = !getenv("VSG_ADAPT_WINDOWS") || atoi(getenv("VSG_ADAPT_WINDOWS")) != 0;
  // (in units of the frame: a bucket is split into rank windows from two frames' worth of edges
  // on -- 4 M at 1080p --, a probe window that replays less than half a frame's worth -- 1 M --
  // ends the splitting)
  const int window_min_edges =
      getenv("VSG_WINDOW_MIN") ? atoi(getenv("VSG_WINDOW_MIN")) : (int)std::min<size_t>(2 * wh_, 1u << 30);
  const int window_min_replayed =
      getenv("VSG_WINDOW_MIN_REPLAYED") ? atoi(getenv("VSG_WINDOW_MIN_REPLAYED"))
                                        : (getenv("VSG_WINDOW_MIN") ? 0 : (int)std::min<size_t>(wh_ / 2, 1u << 30));
  // A bucket is replayed as consecutive *rank windows* (each a full stage: filter -> components ->
  // workers; exact for any split, see RunBucketStage).  While the regions of a bucket are still
  // many small clusters growing side by side, the edges of one window fall into many small
  // components that the whole GPU replays at once, where one stage over the whole bucket would
  // chain them into a few huge components replayed by one wavefront each.  Once one region
  // dominates (its component is a chain whatever the split) windows only add overhead.  The first
  // window tells which case it is: average component size below `bushy` -> keep splitting.
  S.bucket_prefix = bucket_prefix_dev_.get();
  S.bucket_prefix_host = bucket_prefix.data();
  {
    // start of every bucket inside every list's sorted slots (= the list's offsets array), from the
    // host copy of the bucket table
    list_off_host_.assign((size_t)L * (kNumBuckets + 2), 0);
    for (int l = 0; l < L; ++l) {
      int32_t* off = list_off_host_.data() + (size_t)l * (kNumBuckets + 2);
      int acc_l = 0;
      for (int b = 0; b <= kNumBuckets; ++b) {
        off[b] = acc_l;
        const int32_t* row = bucket_base_host_.data() + (size_t)b * (L + 1);
        acc_l += row[l + 1] - row[l];
      }
      off[kNumBuckets + 1] = acc_l;
    }
    const size_t cap = (size_t)3 * 65536;
    if (seg_table_dev_.size() < cap) seg_table_dev_.alloc(cap);
    S.bucket_base_host = getenv("VSG_FILTER_SEGS") && atoi(getenv("VSG_FILTER_SEGS")) == 0 ? nullptr
                                                                                          : bucket_base_host_.data();
    S.list_off_host = list_off_host_.data();
    S.list_slot_base_host = list_slot_base_.data();
    S.seg_dev = seg_table_dev_.get();
    S.seg_cap = seg_table_dev_.size();
    S.seg_host = &seg_table_host_;
  }
  if (++window_target_age_ > 8) {
    window_target_age_ = 0;
    // A target grows again only on evidence of slack: the largest component of the bucket's stages over
    // the chunks since the last look was a quarter of the limit or less.  (Unconditionally, as up to
    // round 5, the targets of a stationary noisy input drifted upwards -- doubled here, halved twice
    // without the largest component falling by 60 %, which is how a percolating bucket looks between
    // one stage and the next, restored and frozen one doubling higher -- and a stream of 20 chunks was
    // back at one-second merges: round 6, 400 frames of the +-40 noise input.)
    for (size_t b = 0; b < window_target_.size(); ++b) {
      int64_t& t = window_target_[b];
      if (t != 0 && t != kNoWindowTarget && window_peak_[b] <= 4096) t = t * 2 > (1ll << 28) ? 0 : t * 2;
    }
    std::fill(window_peak_.begin(), window_peak_.end(), 0);
    std::fill(window_last_seg_.begin(), window_last_seg_.end(), 0);
    std::fill(window_frozen_.begin(), window_frozen_.end(), 0);
    std::fill(window_unpaid_.begin(), window_unpaid_.end(), 0);
  }
  int group_width = 1;
  double low_bucket_ms[2] = {0, 0};
  for (int b = 0, hi = 0; b < kNumBuckets; b = hi) {
    const double t_bucket0 = b < 2 ? NowMs() : 0;
    struct LowBucketTimer {   // (what buckets 0 and 1 cost: see the cooldown of the tree replay below)
      double* out;
      double t0;
      ~LowBucketTimer() { if (out) *out += NowMs() - t0; }
    } low_timer{b < 2 ? &low_bucket_ms[b] : nullptr, t_bucket0};
    hi = b + 1;
    if (group_buckets && b >= first_plain) {   // as many buckets as fit the scratch arrays
      while (hi < kNumBuckets && hi - b < group_width &&
             (int64_t)bucket_prefix[hi + 1] - bucket_prefix[b] <= n_max) {
        ++hi;
      }
    }
    S.group_hi = hi;
    const int n_b = bucket_prefix[hi] - bucket_prefix[b];
    if (n_b == 0) continue;
    int64_t group_active = 0;
    const int windows = n_b >= window_min_edges ? num_windows : 1;
    // the window target of this bucket: what the last chunk learned for it, else what the
    // bucket before it ended with (the force-merge buckets, where the tree replay takes the large
    // components, hand nothing on)
    if (window_target_.empty()) {
      window_target_.assign(kNumBuckets + 1, 0);
      window_last_seg_.assign(kNumBuckets + 1, 0);
      window_frozen_.assign(kNumBuckets + 1, 0);
      window_unpaid_.assign(kNumBuckets + 1, 0);
      window_peak_.assign(kNumBuckets + 1, 0);
    }
    if (hub_bucket_pause_.empty()) {
      hub_bucket_pause_.assign(kNumBuckets + 1, 0);
      hub_bucket_stages_.assign(kNumBuckets + 1, 0);
      hub_bucket_absorbed_.assign(kNumBuckets + 1, 0);
    }
    // (hubs are a choice per stage: a stage is exact with or without them)
    S.hubs = hubs_default && hub_bucket_pause_[b] == 0 ? 1 : 0;
    if (hub_bucket_pause_[b] > 0) --hub_bucket_pause_[b];
    hub_bucket_stages_[b] = 0;
    hub_bucket_absorbed_[b] = 0;
    if (window_target_[b] != 0) {
      wave_target_active_ = window_target_[b];
    } else if (b <= first_plain) {
      wave_target_active_ = kNoWindowTarget;
    }
    // One stage; afterwards the window target follows the largest component a single wavefront
    // had to replay.  (On a noisy input the active edges of the middle buckets percolate: one
    // component of a million edges between a million small regions, 0.3 s on one wavefront and
    // three seconds per chunk.  Below the percolation threshold -- fewer active edges per stage --
    // the same edges fall into thousands of components.  The target is counted in active edges,
    // kept per bucket between chunks, halved while a stage still produces a component above
    // 16 K edges; it only grows between chunks -- doubled every eighth chunk, to follow the video
    // -- because the threshold is sharp: twice the edges per stage that give components of 5 K
    // give one of 60 K.)
    auto run = [&](int w, int j0, int n, int measure, bool limited) {
      StageInfo info;
      info.want_components = debug_stages ? 2 : measure;
      ++diag_stages;
      RunStageDebug(b, w, windows, j0, n, P, inert_mode, S, debug_stages, &info);
      diag_hub_stages += info.hub_stages;
      diag_hub_absorbed += info.hub_absorbed;
      hub_bucket_stages_[b] += info.hub_stages;
      hub_bucket_absorbed_[b] += info.hub_absorbed;
      group_active += info.replayed;
      if (debug_stages && info.want_components && wave_target_active_ != kNoWindowTarget) {
        std::fprintf(stderr, "[vsg]   window: max wave segment %d, target %lld active (density %.4f), limited %d\n",
                     info.max_wave_segment, (long long)wave_target_active_, last_density_, (int)limited);
      }
      // (a stage that was run again with broken hubs as ordinary regions says nothing about the
      // window: its large components are the neighbourhoods of those regions)
      if (measure && adapt_windows && info.hub_retries == 0 && (measure > 1 || info.replayed > 16384)) {
        last_density_ = std::max((double)info.replayed / (double)std::max(n, 1), 1e-6);
        window_peak_[b] = std::max(window_peak_[b], info.max_wave_segment);
        if (info.max_wave_segment > 16384 && !window_frozen_[b]) {
          // Halving has to pay: below the percolation threshold the largest component collapses
          // (184 K -> 57 K -> 10 K edges); the edges of ONE region against its neighbours just
          // split in two with the window, the same serial work in twice the stages -- then the
          // last step is undone and the bucket keeps its target.
          // (two halvings in a row that did not pay: the windows differ, one is not evidence)
          const bool paid = window_last_seg_[b] == 0 || info.max_wave_segment <= 0.4 * window_last_seg_[b];
          if (!paid && window_frozen_[b] == 0 && ++window_unpaid_[b] >= 2) {
            if (wave_target_active_ != kNoWindowTarget) wave_target_active_ *= 4;
            window_frozen_[b] = 1;
          } else {
            if (paid) window_unpaid_[b] = 0;
            window_last_seg_[b] = info.max_wave_segment;
            wave_target_active_ =
                std::max<int64_t>(std::min<int64_t>(wave_target_active_, info.replayed) / 2, 8192);
          }
        }
        (void)limited;
      }
      return info;
    };
    // edges the next stage may take (from `left`), by the window target
    auto limit = [&](int64_t want, bool* limited) {
      *limited = false;
      if (wave_target_active_ == kNoWindowTarget) return want;
      // (at most 64 stages per bucket, whatever the target)
      const int64_t by_target = std::max<int64_t>(
          std::max<int64_t>((int64_t)((double)wave_target_active_ / last_density_), 16384), ((int64_t)n_b + 63) / 64);
      if (by_target < want) {
        *limited = true;
        return by_target;
      }
      return want;
    };
    const bool big_enough = n_b >= (1 << 18);   // (the extra synchronisation of a measured stage)
    int pos = 0;
    for (int w = 0; w < windows && pos < n_b; ++w) {
      const int j1 = (int)((int64_t)n_b * (w + 1) / windows);
      if (j1 <= pos) continue;
      const bool probe = windows > 1 && w == 0 && window_bushy > 0;
      bool rest_mode = false;
      while (pos < j1) {
        bool limited = false;
        const int take = (int)limit(j1 - pos, &limited);
        const StageInfo info = run(w, pos, take,
                                   probe ? 2 : ((big_enough && (limited || windows == 1)) ||
                                                wave_target_active_ != kNoWindowTarget) ? 1 : 0, limited);
        pos += take;
        // (and a window that replays few edges is all overhead)
        if (probe && ((int64_t)info.replayed >= (int64_t)window_bushy * std::max(info.components, 1) ||
                      info.replayed < (int64_t)window_min_replayed * take / std::max(j1, 1))) {
          rest_mode = true;
        }
      }
      if (rest_mode) {
        // few, large components already: the rest of the bucket in one stage -- in as few as the
        // tree replay can take (its rank stamps address 2^27 edges; the giant components of a
        // low-contrast 4K bucket have 260 M, and left to the wave worker they cost seconds)
        // (and in pieces of at most 1.4 N edges -- 61 M at 1080p --: the active edges of a stage size the
        // 106-byte-per-edge scratch, and the rest of a first, unconstrained chunk's bucket in ONE stage
        // was 61 M active edges = 6.5 GB for nothing: two pieces take the same time)
        const int64_t rest_default = std::min<int64_t>(
            119ll << 20, std::max<int64_t>(16ll << 20, (int64_t)(1.4 * (double)wh_ * (double)capacity_frames_)));
        const int64_t rest_cap = getenv("VSG_REST_CAP") ? atoll(getenv("VSG_REST_CAP")) : rest_default;
        const int64_t rest_n = (int64_t)n_b - pos;
        const int pieces = (int)std::max<int64_t>(1, (rest_n + rest_cap - 1) / rest_cap);
        const int64_t piece = (rest_n + pieces - 1) / pieces;
        for (int q = 0; pos < n_b; ++q) {
          bool limited = false;
          const int take = (int)limit(std::min<int64_t>(piece, n_b - pos), &limited);
          run(-1 - q, pos, take, big_enough ? 1 : 0, limited);
          pos += take;
        }
      }
    }
    window_target_[b] = wave_target_active_;
    // Hubs that absorbed less than a region per stage: the bucket does without them for eight chunks.
    if (hub_bucket_stages_[b] > 0 && hub_bucket_absorbed_[b] < hub_bucket_stages_[b]) hub_bucket_pause_[b] = 8;
    if (debug_stages && wave_target_active_ != kNoWindowTarget) {
      std::fprintf(stderr, "[vsg]   bucket %d leaves target %lld (frozen %d, unpaid %d, last %d, peak %d)\n", b,
                   (long long)wave_target_active_, (int)window_frozen_[b], (int)window_unpaid_[b], window_last_seg_[b],
                   window_peak_[b]);
    }
    if (b >= first_plain) {
      if (group_active < 2048) {
        group_width = std::min(group_width * 4, 512);
      } else if (group_active > 16384) {
        group_width = std::max(group_width / 4, 1);
      }
    }
  }
  // The tree replay in the first two buckets: where its assumption keeps failing (two chunks in a row)
  // the bucket goes without it for a while -- but only while that is the cheaper of the two.  On the
  // headline input, once constraints have accumulated for 35 chunks, an edge between regions of
  // different constraints sits inside the giant component of bucket 1: a failed replay costs the chunk
  // 30-40 ms (the stage again with the ordinary workers), a bucket without the replay 180 ms (the giant
  // component on single wavefronts, 64 stages by the window target) -- eight chunks in a row, as up to
  // round 5, was the wrong way round.  Both costs are measured (the bucket's wall time); a skip that is
  // not known to be cheaper lasts one chunk, which measures it.
  for (int b = 0; b < 2; ++b) {
    if (++spine_low_cost_age_[b] > 64) {   // (what was measured ages)
      spine_low_cost_age_[b] = 0;
      spine_low_cost_fail_[b] = spine_low_cost_skip_[b] = 0;
    }
    if (spine_low_skip[b]) {
      spine_low_cost_skip_[b] = low_bucket_ms[b];
      if (spine_low_cost_fail_[b] > 0 && spine_low_cost_skip_[b] >= spine_low_cost_fail_[b]) spine_low_cooldown_[b] = 0;
      continue;
    }
    if (spine_low_failed[b]) spine_low_cost_fail_[b] = low_bucket_ms[b];
    spine_low_fails_[b] = spine_low_failed[b] ? spine_low_fails_[b] + 1 : 0;
    if (spine_low_fails_[b] >= 2) {
      spine_low_fails_[b] = 0;
      if (spine_low_cost_skip_[b] <= 0) {
        spine_low_cooldown_[b] = 1;                                   // not known: one chunk measures it
      } else if (spine_low_cost_skip_[b] < spine_low_cost_fail_[b]) {
        spine_low_cooldown_[b] = 8;
      }
    }
  }
  timings_.optimistic_stages = optimistic_stages_;
  timings_.rollbacks = rollbacks_;
  LaunchKeepVirtualBucket(list_desc_dev_.get(), L, stream_);
  ResetHubExclusions(S, nodes(), stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
  const double t_buckets = NowMs();
  if (getenv("VSG_DEBUG_HASH")) DebugHash("after buckets");
  if (force_constraints && has_constraints_) MergeConstrainedHostAssisted();
  if (getenv("VSG_DEBUG_HASH")) DebugHash("after constrained merge");
  const double t_mc = NowMs();
  if (getenv("VSG_DEBUG_STATS")) {
    std::fprintf(stderr, "[vsg] segment: buckets %.1f ms, merge-constrained %.1f ms\n",
                 t_buckets - t0, t_mc - t_buckets);
  }
  unsigned long long st[80] = {0};
  D2H(st, stats_.get(), 80, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
  if (st[23] != 0) {
    std::fprintf(stderr, "[vsg] chain self check: %llu mismatches\n", st[23]);
    throw Error(-4 /* VSG_ERR_INTERNAL */, "merge worker: the chain self check found mismatches");
  }
  if (st[22] != 0) throw Error(-4 /* VSG_ERR_INTERNAL */, "merge worker: a batch did not converge");
  if (getenv("VSG_DEBUG_STATS")) {
    std::fprintf(stderr, "[vsg] wave: edges %llu batches %llu rounds %llu generic %llu (solo %llu) chain %llu "
                 "cuts %llu; optimistic stages %lld rollbacks %lld\n",
                 st[3], st[7], st[5], st[4], st[6], st[20], st[21], (long long)optimistic_stages_,
                 (long long)rollbacks_);
    std::fprintf(stderr, "[vsg] hubs: %lld stages used hub regions, %lld regions absorbed through them, %lld stages cut at an edge, %lld redone "
                 "(broken %lld, inherit %lld, shape %lld, marked %lld, pair %lld, split %lld)\n",
                 (long long)diag_hub_stages, (long long)diag_hub_absorbed, S.hub_splits, S.hub_retries, S.hub_reasons[0],
                 S.hub_reasons[1], S.hub_reasons[2], S.hub_reasons[3], S.hub_reasons[4], S.hub_reasons[5]);
    std::fprintf(stderr, "[vsg] wide: edges %llu batches %llu (%.1f lanes each) rounds %llu (%.1f per batch), chain lanes %llu, "
                 "kept-lane iterations %llu; kcyc per batch: staging %.1f rounds %.1f (%.2f per round)\n",
                 st[31], st[45], (double)st[72] / std::max(1.0, (double)st[45]), st[44],
                 (double)st[44] / std::max(1.0, (double)st[45]), st[73], st[74],
                 st[46] / 1e3 / std::max(1.0, (double)st[45]), st[47] / 1e3 / std::max(1.0, (double)st[45]),
                 st[47] / 1e3 / std::max(1.0, (double)st[44]));
  }
  timings_.merge_ms = (float)(NowMs() - t0);
  for (auto& pr : ev_wave_) {
    float ms = 0;
    VSG_HIP(hipEventElapsedTime(&ms, ev_pool_[pr.first], ev_pool_[pr.second]));
    timings_.wave_ms += ms;
  }
  for (auto& pr : ev_filter_) {
    float ms = 0;
    VSG_HIP(hipEventElapsedTime(&ms, ev_pool_[pr.first], ev_pool_[pr.second]));
    timings_.filter_ms += ms;
  }
  for (auto& pr : ev_spine_) {
    float ms = 0;
    VSG_HIP(hipEventElapsedTime(&ms, ev_pool_[pr.first], ev_pool_[pr.second]));
    timings_.spine_ms += ms;
  }
  timings_.spine_launches += (int64_t)ev_spine_.size();
  timings_.spine_edges += (int64_t)st[24];
  timings_.wave_launches += (int64_t)ev_wave_.size();
  timings_.filter_launches += (int64_t)ev_filter_.size();
  timings_.wave_edges += (int64_t)st[3];
  timings_.edges_active += (int64_t)st[3];
  timings_.edges_total = edges_total;
  timings_.merges[0] += (int64_t)st[0];
  timings_.merges[1] += (int64_t)st[1];
  timings_.merges[2] += (int64_t)st[2];
  {
    const ThreadAllocCounters a1 = ThreadAllocSnapshot();
    const MailWaitCounters m1 = MailWaitSnapshot();
    timings_.stages = diag_stages;
    timings_.slab_growths = diag_slab_growths;
    timings_.slab_growth_ms = diag_slab_ms;
    timings_.spine_growths = diag_spine_growths;
    timings_.spine_growth_ms = diag_spine_ms;
    timings_.runtime_mallocs = a1.runtime_mallocs - alloc0.runtime_mallocs;
    timings_.runtime_malloc_ms = a1.runtime_malloc_ms - alloc0.runtime_malloc_ms;
    timings_.runtime_frees = a1.runtime_frees - alloc0.runtime_frees;
    timings_.runtime_free_ms = a1.runtime_free_ms - alloc0.runtime_free_ms;
    timings_.cache_hits = a1.cache_hits - alloc0.cache_hits;
    timings_.device_syncs = a1.device_syncs - alloc0.device_syncs;
    timings_.device_sync_ms = a1.device_sync_ms - alloc0.device_sync_ms;
    timings_.mail_waits = m1.waits - mail0.waits;
    timings_.mail_wait_ms = m1.wait_ms - mail0.wait_ms;
    timings_.mail_wait_longest_ms = m1.longest_ms;
    timings_.mail_mode = MailYieldMode();
    timings_.prepare_ms = t0 - t_enter;
    timings_.constrained_merge_ms = t_mc - t_buckets;
    timings_.segment_wall_ms = NowMs() - t_enter;
  }
}

// ---------------------------------------------------------------------------------------------
// MergeConstrainedRegions (segmentation_graph.h:703-786), host assisted.
//
// The reference walks all non-virtual nodes in id order, then all virtual nodes.  A node only
// matters through (its own constraint field >= 0, its current representative), and consecutive
// nodes with the same representative repeat the same step until it becomes a no-op, so the device
// reduces the node walk to runs of equal representatives and the host replays the runs on the few
// thousand region states involved.
// ---------------------------------------------------------------------------------------------
namespace {
struct SimRegion {
  int parent;
  float d[3];
  int sz, cons, flags;
  bool dirty;
};
}  // namespace

// Debug aid (VSG_DEBUG_HASH): FNV hashes of the partition (labels renumbered by first occurrence),
// of the representative identities and of the representatives' states, to compare two runs.
void DenseGraphHip::DebugHash(const char* where) {
  const size_t N = wh_ * (size_t)num_frames_;
  std::vector<int32_t> parent(N), cons(N);
  std::vector<float4> ds(N);
  std::vector<uint8_t> flags(N);
  D2H(parent.data(), parent_.get(), N, stream_);
  D2H(cons.data(), cons_.get(), N, stream_);
  D2H(ds.data(), desc_sz_.get(), N, stream_);
  D2H(flags.data(), flags_.get(), N, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
  auto fnv = [](uint64_t h, const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
  };
  std::vector<int32_t> root(N), canon(N, -1);
  uint64_t h_part = 1469598103934665603ull, h_root = h_part, h_state = h_part, h_cons = h_part;
  int next = 0, nroots = 0;
  for (size_t i = 0; i < N; ++i) {
    int r = (int)i;
    while (parent[(size_t)r] != r) r = parent[(size_t)r];
    root[i] = r;
    if (canon[(size_t)r] < 0) canon[(size_t)r] = next++;
    const int32_t c = canon[(size_t)r];
    h_part = fnv(h_part, &c, 4);
    h_root = fnv(h_root, &r, 4);
  }
  for (size_t i = 0; i < N; ++i) {
    if (parent[i] != (int)i) continue;
    ++nroots;
    h_state = fnv(h_state, &ds[i], 16);
    h_state = fnv(h_state, &flags[i], 1);
    h_cons = fnv(h_cons, &cons[i], 4);
  }
  std::fprintf(stderr, "[vsg] hash %s: roots %d partition %016llx root-ids %016llx states %016llx cons %016llx\n",
               where, nroots, (unsigned long long)h_part, (unsigned long long)h_root,
               (unsigned long long)h_state, (unsigned long long)h_cons);
}

void DenseGraphHip::MergeConstrainedHostAssisted() {
  const int N = (int)(wh_ * (size_t)num_frames_);
  // Ranges: non-virtual nodes first (in id order), then virtual nodes (in id order).
  std::vector<std::pair<int, int>> vranges;
  for (int t : virtual_slices_) vranges.emplace_back((int)(wh_ * t), (int)(wh_ * (t + 1)));
  std::sort(vranges.begin(), vranges.end());
  std::vector<std::pair<int, int>> nvranges;
  int cursor = 0;
  for (auto& vr : vranges) {
    if (vr.first > cursor) nvranges.emplace_back(cursor, vr.first);
    cursor = vr.second;
  }
  if (cursor < N) nvranges.emplace_back(cursor, N);

  // scratch: label_uf_ = flags, label_img_ = values, adjust_ = offsets (all N ints)
  int32_t* d_flags = label_uf_.get();
  int32_t* d_vals = label_img_.get();
  int32_t* d_offs = adjust_.get();
  // Run heads (readout_kernels.hip: LaunchConstrainedRuns) of a node range, in node order:
  // (first node, value) with value = representative / -2 unconstrained representative / -1
  // terminator; a run ends where the next entry starts.
  struct Runs {
    std::vector<int32_t> node, value;
  };
  auto collect = [&](int begin, int end, Runs* out) {
    const int n = end - begin;
    if (n <= 0) return;
    LaunchConstrainedRuns(nodes(), begin, end, d_vals, d_flags, stream_);
    ExclusiveSum(ScanScratch{scan_sums_.get()}, d_flags, d_offs, n, stream_);
    int last_off = 0, last_flag = 0;
    D2H(&last_off, d_offs + (n - 1), 1, stream_);
    D2H(&last_flag, d_flags + (n - 1), 1, stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
    const int k = last_off + last_flag;
    const size_t old = out->node.size();
    if (k > 0) {
      EnsureActiveScratch((size_t)k);
      LaunchCompactIndexValue(d_flags, d_offs, d_vals, n, act_.a_ra, act_.a_rb, stream_);
      out->node.resize(old + k);
      out->value.resize(old + k);
      D2H(out->node.data() + old, act_.a_ra, (size_t)k, stream_);
      D2H(out->value.data() + old, act_.a_rb, (size_t)k, stream_);
      VSG_HIP(hipStreamSynchronize(stream_));
      for (size_t i = old; i < old + (size_t)k; ++i) out->node[i] += begin;
    }
    out->node.push_back(end);     // terminator of the range
    out->value.push_back(-1);
  };

  const double tm0 = NowMs();
  Runs nv, vr;
  for (auto& r : nvranges) collect(r.first, r.second, &nv);
  for (auto& r : vranges) collect(r.first, r.second, &vr);
  const double tm1 = NowMs();

  // Distinct representatives of the nodes whose own constraint is >= 0, and their states.
  std::vector<int32_t> ids;
  {
    std::unordered_set<int32_t> seen;
    for (const Runs* v : {&nv, &vr}) {
      for (int32_t r : v->value) {
        if (r >= 0 && seen.insert(r).second) ids.push_back(r);
      }
    }
    std::sort(ids.begin(), ids.end());
  }
  const int m = (int)ids.size();
  if (m == 0) return;
  small_i32_a_.ensure((size_t)m);
  small_i32_b_.ensure((size_t)m);
  small_i32_c_.ensure((size_t)m);
  small_f4_.ensure((size_t)m);
  H2D(small_i32_a_.get(), ids.data(), (size_t)m, stream_);
  LaunchGatherStates(nodes(), small_i32_a_.get(), m, small_f4_.get(), small_i32_b_.get(),
                     small_i32_c_.get(), stream_);
  std::vector<float4> h_ds(m);
  std::vector<int32_t> h_cons(m), h_flags(m);
  D2H(h_ds.data(), small_f4_.get(), (size_t)m, stream_);
  D2H(h_cons.data(), small_i32_b_.get(), (size_t)m, stream_);
  D2H(h_flags.data(), small_i32_c_.get(), (size_t)m, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));

  const double tm2 = NowMs();
  std::unordered_map<int, SimRegion> sim;
  sim.reserve((size_t)m * 2);
  for (int i = 0; i < m; ++i) {
    SimRegion r;
    r.parent = ids[i];
    r.d[0] = h_ds[i].x;
    r.d[1] = h_ds[i].y;
    r.d[2] = h_ds[i].z;
    std::memcpy(&r.sz, &h_ds[i].w, 4);
    r.cons = h_cons[i];
    r.flags = h_flags[i];
    r.dirty = false;
    sim.emplace(ids[i], r);
  }
  auto find = [&sim](int r) {
    for (;;) {
      const SimRegion& s = sim.at(r);
      if (s.parent == r) return r;
      r = s.parent;
    }
  };
  const float force_w = l1_ ? 0.002f : 0.001f;
  auto distance = [&](const SimRegion& a, const SimRegion& b, float w) {
    const float x = a.d[0] - b.d[0], y = a.d[1] - b.d[1], z = a.d[2] - b.d[2];
    const float dist = (float)std::sqrt((double)((x * x + y * y + z * z) * (1.0f / 3.0f)));
    if (w < force_w && (double)dist < 0.2) return 0.0f;
    return dist;
  };
  // MergeRegions(rep_1, rep_2); returns nothing, updates sim.
  auto merge = [&](int r1, int r2) {
    SimRegion& a = sim.at(r1);
    SimRegion& b = sim.at(r2);
    const bool first = a.sz > b.sz;
    SimRegion& mrg = first ? a : b;
    SimRegion& oth = first ? b : a;
    const int mid = first ? r1 : r2;
    if (!((mrg.flags | oth.flags) & kFlagNoDesc)) {
      const float denom = 1.0f / (float)(oth.sz + mrg.sz);
      const float ca = (float)oth.sz * denom;
      const float cb = (float)mrg.sz * denom;
      for (int c = 0; c < 3; ++c) mrg.d[c] = ca * oth.d[c] + cb * mrg.d[c];
    }
    mrg.sz += oth.sz;
    mrg.cons = std::max(a.cons, b.cons);
    oth.parent = mid;
    mrg.dirty = true;
    oth.dirty = true;
  };

  std::unordered_map<int, int> c2r;
  // One visit of the reference's loop body for a node whose representative (before the pass)
  // is r0.  Returns true if anything changed.
  auto visit_nonvirtual = [&](int r0) -> bool {
    const int my = find(r0);
    SimRegion& me = sim.at(my);
    auto pos = c2r.find(me.cons);
    if (pos == c2r.end()) {
      c2r.emplace(me.cons, my);
      return true;
    }
    const int crep = find(pos->second);
    if (crep == my) return false;
    SimRegion& cr = sim.at(crep);
    const float dist = distance(me, cr, 1.0f);
    if (dist > 0.15f) {
      if ((double)me.sz < (double)cr.sz * 0.3) {
        const bool ch = me.cons != -1;
        me.cons = -1;
        me.dirty = true;
        return ch;
      } else if ((double)cr.sz < (double)me.sz * 0.3) {
        const bool ch = (cr.cons != -1) || (pos->second != my);
        cr.cons = -1;
        cr.dirty = true;
        pos->second = my;
        return ch;
      }
      me.cons = -1;
      cr.cons = -1;
      me.dirty = cr.dirty = true;
      c2r.erase(pos);
      return true;
    }
    merge(my, crep);
    return true;
  };
  auto visit_virtual = [&](int r0) -> bool {   // never reset, always merge
    const int my = find(r0);
    SimRegion& me = sim.at(my);
    auto pos = c2r.find(me.cons);
    if (pos == c2r.end()) {
      c2r.emplace(me.cons, my);
      return true;
    }
    const int crep = find(pos->second);
    if (crep == my) return false;
    merge(my, crep);
    return true;
  };

  // Non-virtual pass, visit order = node id order.  A node that is (or was) a representative is
  // tested with its *current* own constraint field (it changes when the region is unconstrained
  // or inherits a constraint through a merge); other nodes keep the value they had before the
  // pass (>= 0 for every node of a plain run).  A run is visited node by node until a visit
  // changes nothing: the nodes after that repeat the same no-op.
  for (size_t k = 0; k + 1 < nv.node.size(); ++k) {
    const int node = nv.node[k], v = nv.value[k];
    if (v == -1) continue;
    const bool is_rep = v == -2 || v == node;
    if (is_rep) {
      auto it = sim.find(node);
      if (it == sim.end() || it->second.cons < 0) continue;   // region->constraint_id < 0
      visit_nonvirtual(node);
      continue;
    }
    const int len = nv.node[k + 1] - node;
    for (int i = 0; i < len; ++i) {
      if (!visit_nonvirtual(v)) break;
    }
  }
  // Virtual pass.
  for (size_t k = 0; k + 1 < vr.node.size(); ++k) {
    const int node = vr.node[k], v = vr.value[k];
    if (v < 0) continue;   // terminator / a representative whose own field is < 0 is not listed
    const int len = (v == node) ? 1 : vr.node[k + 1] - node;
    for (int i = 0; i < len; ++i) {
      if (!visit_virtual(v)) break;
    }
  }

  // Write back.
  const double tm3 = NowMs();
  if (getenv("VSG_DEBUG_STATS")) {
    std::fprintf(stderr, "[vsg] merge-constrained: collect %.1f ms (%zu + %zu runs), states %.1f ms "
                 "(%d representatives), replay %.1f ms\n", tm1 - tm0, nv.node.size(), vr.node.size(),
                 tm2 - tm1, m, tm3 - tm2);
  }
  std::vector<int32_t> u_ids, u_parent, u_cons, u_flags;
  std::vector<float4> u_ds;
  for (auto& kv : sim) {
    if (!kv.second.dirty) continue;
    u_ids.push_back(kv.first);
    u_parent.push_back(kv.second.parent);
    float w;
    std::memcpy(&w, &kv.second.sz, 4);
    u_ds.push_back(make_float4(kv.second.d[0], kv.second.d[1], kv.second.d[2], w));
    u_cons.push_back(kv.second.cons);
    u_flags.push_back(kv.second.flags);
  }
  const int u = (int)u_ids.size();
  if (u == 0) return;
  // (persistent scratch: a hipFree per call would synchronise the whole device, i.e. every other
  // stream of the GPU as well)
  DevBuf<int32_t>&d_ids = Grow(mc_ids_, u), &d_par = Grow(mc_par_, u), &d_cons = Grow(mc_cons_, u),
                 &d_fl = Grow(mc_fl_, u);
  DevBuf<float4>& d_ds = Grow(mc_ds_, u);
  H2D(d_ids.get(), u_ids.data(), (size_t)u, stream_);
  H2D(d_par.get(), u_parent.data(), (size_t)u, stream_);
  H2D(d_cons.get(), u_cons.data(), (size_t)u, stream_);
  H2D(d_fl.get(), u_flags.data(), (size_t)u, stream_);
  H2D(d_ds.get(), u_ds.data(), (size_t)u, stream_);
  LaunchScatterStates(nodes(), d_ids.get(), u, d_par.get(), d_ds.get(), d_cons.get(), d_fl.get(),
                      stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
}

// ---------------------------------------------------------------------------------------------
// ObtainResults + DetermineNeighborIds
// ---------------------------------------------------------------------------------------------
void DenseGraphHip::SampleFlows(const std::vector<FlowRequest>& req,
                                const std::vector<const float*>& dev_flows,
                                std::vector<float>* samples) {
  const int n = (int)req.size();
  samples->assign(2 * (size_t)n, 0.f);
  if (n == 0) return;
  static_assert(sizeof(FlowRequest) == 3 * sizeof(int32_t), "FlowRequest is three ints");
  flow_req_dev_.ensure(3 * (size_t)n);
  flow_samples_dev_.ensure((size_t)n);
  flow_ptrs_dev_.ensure(dev_flows.size());
  H2D(flow_req_dev_.get(), reinterpret_cast<const int32_t*>(req.data()), 3 * (size_t)n, stream_);
  H2D(flow_ptrs_dev_.get(), dev_flows.data(), dev_flows.size(), stream_);
  LaunchGatherFlow(flow_req_dev_.get(), n, flow_ptrs_dev_.get(), W_, flow_samples_dev_.get(), stream_);
  D2H(reinterpret_cast<float2*>(samples->data()), flow_samples_dev_.get(), (size_t)n, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
}

void DenseGraphHip::ObtainResults(const std::vector<const float*>* dev_flows, bool enforce_n4,
                                  bool enforce_spatial_connectedness) {
  const double t_start = NowMs();
  const size_t N = wh_ * (size_t)num_frames_;
  if (dev_flows) VSG_REQUIRE((int)dev_flows->size() == num_frames_, -1, "one flow per frame");

  // 1. representative key per node (FlattenUnionFind).
  LaunchFlatten(nodes(), N, label_uf_.get(), stream_);
  VSG_HIP(hipMemcpyAsync(label_img_.get(), label_uf_.get(), N * sizeof(int32_t),
                         hipMemcpyDeviceToDevice, stream_));
  VSG_HIP(hipMemsetAsync(adjust_.get(), 0, N * sizeof(int32_t), stream_));

  // 2. N4 connectivity on every rasterised slice (constrained_slices_ is never filled in the
  //    reference's live path, SURVEY A.7-1, so constrained slices are swept as well).
  std::vector<int32_t> frames;
  for (int t = 0; t < num_frames_; ++t) {
    if (!std::binary_search(virtual_slices_.begin(), virtual_slices_.end(), t)) frames.push_back(t);
  }
  const int nf = (int)frames.size();
  small_i32_a_.ensure((size_t)std::max(nf, 1));
  H2D(small_i32_a_.get(), frames.data(), (size_t)nf, stream_);
  const int rows = nf * H_;
  row_counts_.ensure((size_t)rows + 1);
  row_offsets_.ensure((size_t)rows + 1);
  if (enforce_n4 && nf > 0) {
    // (the row flags of the sweep live in the array the run counts of step 3 are written to next)
    VSG_HIP(hipMemsetAsync(row_counts_.get(), 0, (size_t)rows * sizeof(int32_t), stream_));
    LaunchEnforceN4(label_img_.get(), W_, H_, small_i32_a_.get(), nf, row_counts_.get(), adjust_.get(), stream_);
  }

  // 3. run-length intervals in (slice, y, x) order.
  for (int i = 0; i < nf; ++i) {
    LaunchRowRunCounts(label_img_.get(), W_, H_, frames[i], row_counts_.get() + (size_t)i * H_,
                       stream_);
  }
  VSG_HIP(hipMemsetAsync(row_counts_.get() + rows, 0, sizeof(int32_t), stream_));
  ExclusiveSum(ScanScratch{scan_sums_.get()}, row_counts_.get(), row_offsets_.get(), rows + 1, stream_);
  int num_iv = 0;
  D2H(&num_iv, row_offsets_.get() + rows, 1, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
  iv_label_.ensure((size_t)num_iv);
  iv_ty_.ensure((size_t)num_iv);
  iv_lx_.ensure((size_t)num_iv);
  iv_rx_.ensure((size_t)num_iv);
  IntervalArrays iva{iv_label_.get(), iv_ty_.get(), iv_lx_.get(), iv_rx_.get()};
  for (int i = 0; i < nf; ++i) {
    LaunchWriteIntervals(label_img_.get(), W_, H_, frames[i],
                         row_offsets_.get() + (size_t)i * H_, iva, stream_);
  }
  // (pinned host memory, kept between chunks: the 14 MB of a many-region chunk took 1.5 ms into
  // freshly allocated pageable vectors)
  iv_host_.ensure(4 * (size_t)std::max(num_iv, 1));
  int32_t* h_label = iv_host_.get();
  uint32_t* h_ty = reinterpret_cast<uint32_t*>(iv_host_.get() + (size_t)num_iv);
  int32_t* h_lx = iv_host_.get() + 2 * (size_t)num_iv;
  int32_t* h_rx = iv_host_.get() + 3 * (size_t)num_iv;
  D2H(h_label, iv_label_.get(), (size_t)num_iv, stream_);
  D2H(h_ty, iv_ty_.get(), (size_t)num_iv, stream_);
  D2H(h_lx, iv_lx_.get(), (size_t)num_iv, stream_);
  D2H(h_rx, iv_rx_.get(), (size_t)num_iv, stream_);

  // 4. size adjustments of the N4 pass (sparse).
  std::vector<int32_t> adj_keys, adj_vals;
  if (enforce_n4) {
    // cc_ is free after the merge and label_img_ after the interval kernels (stream order).
    int32_t* d_flags = cc_.get();
    int32_t* d_offs = label_img_.get();
    LaunchNonzeroFlags(adjust_.get(), (int)N, d_flags, stream_);
    ExclusiveSum(ScanScratch{scan_sums_.get()}, d_flags, d_offs, (int)N, stream_);
    int lo = 0, lf = 0;
    D2H(&lo, d_offs + (N - 1), 1, stream_);
    D2H(&lf, d_flags + (N - 1), 1, stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
    const int cnt = lo + lf;
    if (cnt > 0) {
      small_i32_b_.ensure((size_t)cnt);
      small_i32_c_.ensure((size_t)cnt);
      LaunchCompactIndexValue(d_flags, d_offs, adjust_.get(), (int)N, small_i32_b_.get(),
                              small_i32_c_.get(), stream_);
      adj_keys.resize(cnt);
      adj_vals.resize(cnt);
      D2H(adj_keys.data(), small_i32_b_.get(), (size_t)cnt, stream_);
      D2H(adj_vals.data(), small_i32_c_.get(), (size_t)cnt, stream_);
      VSG_HIP(hipStreamSynchronize(stream_));
    }
  }
  VSG_HIP(hipStreamSynchronize(stream_));
  const double t_dev1 = NowMs();

  double th[6] = {t_dev1, t_dev1, t_dev1, t_dev1, t_dev1, t_dev1};   // debug: phases of the host part
  long long sum_tubes = 0;
  int most_tubes = 0;
  // 5. regions in first-appearance order of their intervals (GetCreateRegionInformation via
  //    AddIntervalToRasterization, dense_segmentation_graph.h:432-466).
  regions_.clear();
  key_to_region_.clear();
  next_region_index_ = 0;
  std::vector<int32_t> region_keys;
  {
    // The intervals arrive in (slice, y, x) order.  One host thread per slice collects the slice's
    // regions in order of first appearance with their rasters (a small open-addressing map: a
    // std::unordered_map lookup per interval was 15 ms for 0.9 M intervals); the slices are then
    // merged in order, which gives the same region numbering as one pass over all intervals.
    struct SliceTable {
      int frame = 0;
      std::vector<int32_t> keys;       // first-appearance order within the slice
      std::vector<Raster> rasters;     // parallel to keys
    };
    struct FlatMap {   // key -> value, keys >= 0
      std::vector<int32_t> k, v;
      uint32_t mask = 0;
      size_t used = 0;
      explicit FlatMap(size_t cap_pow2) : k(cap_pow2, -1), v(cap_pow2, 0), mask((uint32_t)cap_pow2 - 1) {}
      int32_t* find_or_insert(int32_t key, int32_t value_if_new, bool* inserted) {
        if (2 * (used + 1) > k.size()) grow();
        uint32_t h = ((uint32_t)key * 2654435761u) & mask;
        for (;; h = (h + 1) & mask) {
          if (k[h] == key) {
            *inserted = false;
            return &v[h];
          }
          if (k[h] < 0) {
            k[h] = key;
            v[h] = value_if_new;
            ++used;
            *inserted = true;
            return &v[h];
          }
        }
      }
      void grow() {
        FlatMap bigger(k.size() * 2);
        bool ins;
        for (size_t i = 0; i < k.size(); ++i) {
          if (k[i] >= 0) bigger.find_or_insert(k[i], v[i], &ins);
        }
        *this = std::move(bigger);
      }
    };
    std::vector<int> slice_begin;   // interval index where every slice starts
    for (int i = 0; i < num_iv; ++i) {
      if (i == 0 || (h_ty[i] >> 16) != (h_ty[i - 1] >> 16)) slice_begin.push_back(i);
    }
    const int ns = (int)slice_begin.size();
    slice_begin.push_back(num_iv);
    std::vector<SliceTable> tables((size_t)ns);
    {
      std::atomic<int> next(0);
      auto work = [&]() {
        for (int sidx = next.fetch_add(1); sidx < ns; sidx = next.fetch_add(1)) {
          SliceTable& st = tables[(size_t)sidx];
          const int b = slice_begin[(size_t)sidx], e = slice_begin[(size_t)sidx + 1];
          st.frame = (int)(h_ty[b] >> 16);
          FlatMap local(1024);
          int last_key = -1, last_idx = -1;
          for (int i = b; i < e; ++i) {
            const int key = h_label[i];
            int idx;
            if (key == last_key) {
              idx = last_idx;
            } else {
              bool ins;
              idx = *local.find_or_insert(key, (int32_t)st.keys.size(), &ins);
              if (ins) {
                st.keys.push_back(key);
                st.rasters.emplace_back();
              }
              last_key = key;
              last_idx = idx;
            }
            st.rasters[(size_t)idx].push_back(Interval{(int)(h_ty[i] & 0xFFFFu), h_lx[i], h_rx[i]});
          }
        }
      };
      const int hw = (int)std::thread::hardware_concurrency();
      const int nt = std::max(1, std::min({ns, hw > 0 ? hw : 1, HostThreadCap()}));
      std::vector<std::thread> pool;
      for (int t = 1; t < nt; ++t) pool.emplace_back(work);
      work();
      for (std::thread& t : pool) t.join();
    }
    FlatMap global(4096);
    for (SliceTable& st : tables) {
      for (size_t j = 0; j < st.keys.size(); ++j) {
        bool ins;
        const int idx = *global.find_or_insert(st.keys[j], next_region_index_, &ins);
        if (ins) {
          ++next_region_index_;
          regions_.emplace_back();
          regions_.back().index = idx;
          regions_.back().has_raster = true;
          region_keys.push_back(st.keys[j]);
        }
        regions_[(size_t)idx].raster.push_back(RasterSlice{st.frame, std::move(st.rasters[j])});
      }
    }
    key_to_region_.reserve(region_keys.size() * 2);
    for (size_t i = 0; i < region_keys.size(); ++i) key_to_region_.emplace(region_keys[i], (int)i);
  }
  // sizes / constraints of the representatives
  auto fetch_states = [&](const std::vector<int32_t>& keys, std::vector<int32_t>* sz,
                          std::vector<int32_t>* cons) {
    const int m = (int)keys.size();
    sz->resize(m);
    cons->resize(m);
    if (m == 0) return;
    small_i32_a_.ensure((size_t)m);
    small_i32_b_.ensure((size_t)m);
    small_i32_c_.ensure((size_t)m);
    small_f4_.ensure((size_t)m);
    H2D(small_i32_a_.get(), keys.data(), (size_t)m, stream_);
    LaunchGatherStates(nodes(), small_i32_a_.get(), m, small_f4_.get(), small_i32_b_.get(),
                       small_i32_c_.get(), stream_);
    std::vector<float4> ds(m);
    D2H(ds.data(), small_f4_.get(), (size_t)m, stream_);
    D2H(cons->data(), small_i32_b_.get(), (size_t)m, stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
    for (int i = 0; i < m; ++i) std::memcpy(&(*sz)[i], &ds[i].w, 4);
  };
  {
    std::vector<int32_t> sz, cons;
    fetch_states(region_keys, &sz, &cons);
    for (size_t i = 0; i < region_keys.size(); ++i) {
      regions_[i].size = sz[i];
      regions_[i].constrained_id = cons[i];
    }
  }

  std::unordered_map<int, int> size_adjust;
  for (size_t i = 0; i < adj_keys.size(); ++i) size_adjust[adj_keys[i]] = adj_vals[i];
  th[0] = NowMs();

  // 6. EnforceSpatialConnectedness (dense_segmentation_graph.h:666-904).
  std::vector<uint32_t> rl_ty;
  std::vector<int32_t> rl_lx, rl_rx, rl_new;
  int next_new_key = (int)N;
  if (enforce_spatial_connectedness) {
    const bool have_flows = dev_flows != nullptr;
    const int num_regions = (int)regions_.size();
    // Phase A (host, one task per region): components and shapes of every slice, the list of
    // flow samples the matching will read; one gather on the device; then the matching.  Only
    // regions that really split need the representative of their tubes' first pixel, so those
    // few node labels are gathered from the device afterwards instead of copying the whole
    // label volume.
    std::vector<std::pair<int, TubeResult>> split;
    {
      std::vector<TubeSplitter> splitters((size_t)num_regions);
      std::vector<std::vector<FlowRequest>> reqs((size_t)num_regions);
      // One task per region, the regions with the most scan intervals first (a region's analysis
      // grows faster than its size); the number of threads follows the number of intervals.
      std::vector<int> by_work((size_t)num_regions);
      std::vector<long long> work_of((size_t)num_regions, 0);
      long long total_work = 0;
      for (int r = 0; r < num_regions; ++r) {
        by_work[r] = r;
        if (!regions_[r].has_raster) continue;
        for (const RasterSlice& sl : regions_[r].raster) work_of[r] += (long long)sl.raster.size();
        total_work += work_of[r];
      }
      std::stable_sort(by_work.begin(), by_work.end(), [&](int a, int b) { return work_of[a] > work_of[b]; });
      auto parallel_regions = [&](const std::function<void(int)>& fn) {
        std::atomic<int> next(0);
        auto work = [&]() {
          for (int i = next.fetch_add(1); i < num_regions; i = next.fetch_add(1)) fn(by_work[i]);
        };
        const int hw = (int)std::thread::hardware_concurrency();
        const int nt = std::max(1, std::min({(int)(total_work / 16384), num_regions, hw > 0 ? hw : 1, HostThreadCap()}));
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (std::thread& t : pool) t.join();
      };
      parallel_regions([&](int r) {
        if (regions_[r].has_raster) splitters[r].Prepare(regions_[r].raster, have_flows ? &reqs[r] : nullptr);
      });
      th[1] = NowMs();
      std::vector<size_t> req_off((size_t)num_regions + 1, 0);
      for (int r = 0; r < num_regions; ++r) req_off[r + 1] = req_off[r] + reqs[r].size();
      std::vector<float> samples;
      if (have_flows) {
        std::vector<FlowRequest> all;
        all.reserve(req_off[num_regions]);
        for (int r = 0; r < num_regions; ++r) all.insert(all.end(), reqs[r].begin(), reqs[r].end());
        SampleFlows(all, *dev_flows, &samples);
      }
      th[2] = NowMs();
      if (const char* dump = getenv("VSG_DUMP_TUBES")) {
        // debug: the input of the tube analysis of this chunk (tools/tube_harness.cpp reads it)
        if (FILE* f = std::fopen(dump, "wb")) {
          auto put = [&](int v) { std::fwrite(&v, 4, 1, f); };
          put(W_); put(H_); put(num_regions); put(have_flows ? 1 : 0);
          for (int r = 0; r < num_regions; ++r) {
            put(regions_[r].has_raster ? (int)regions_[r].raster.size() : -1);
            if (!regions_[r].has_raster) continue;
            for (const RasterSlice& sl : regions_[r].raster) {
              put(sl.frame); put((int)sl.raster.size());
              std::fwrite(sl.raster.data(), sizeof(Interval), sl.raster.size(), f);
            }
            put((int)reqs[r].size());
            if (have_flows) std::fwrite(samples.data() + 2 * req_off[r], 8, reqs[r].size(), f);
          }
          std::fclose(f);
        }
      }
      std::vector<TubeResult> results((size_t)num_regions);
      parallel_regions([&](int r) {
        if (!regions_[r].has_raster || !splitters[r].MaySplit()) return;
        splitters[r].Finish(W_, H_, have_flows ? samples.data() + 2 * req_off[r] : nullptr, &results[r]);
      });
      th[3] = NowMs();
      for (int r = 0; r < num_regions; ++r) {
        most_tubes = std::max(most_tubes, results[r].tubes_matched);
        sum_tubes += results[r].tubes_matched;
      }
      for (int r = 0; r < num_regions; ++r) {
        if (results[r].tubes.size() <= 1) continue;
        split.emplace_back(r, TubeResult());
        split.back().second.tubes.swap(results[r].tubes);
        split.back().second.areas.swap(results[r].areas);
        split.back().second.tube_to_keep = results[r].tube_to_keep;
      }
    }
    // A region whose rasterization was replaced by an earlier one's tube (see phase B) is split
    // again at its turn, with its own small gather.
    auto split_again = [&](const Raster3D& raster, TubeResult* out) {
      TubeSplitter ts;
      std::vector<FlowRequest> req;
      ts.Prepare(raster, have_flows ? &req : nullptr);
      std::vector<float> samples;
      if (have_flows) SampleFlows(req, *dev_flows, &samples);
      ts.Finish(W_, H_, have_flows ? samples.data() : nullptr, out);
    };
    auto first_node_of = [&](const Raster3D& tube) {
      const RasterSlice& s0 = tube[0];
      return (int32_t)((size_t)s0.frame * wh_ + (size_t)s0.raster[0].y * W_ + s0.raster[0].lx);
    };
    auto fetch_labels = [&](const std::vector<int32_t>& node_ids, std::vector<int32_t>* keys) {
      const int m = (int)node_ids.size();
      keys->resize(m);
      if (m == 0) return;
      small_i32_a_.ensure((size_t)m);
      small_i32_b_.ensure((size_t)m);
      H2D(small_i32_a_.get(), node_ids.data(), (size_t)m, stream_);
      LaunchGatherI32(label_uf_.get(), small_i32_a_.get(), m, small_i32_b_.get(), stream_);
      D2H(keys->data(), small_i32_b_.get(), (size_t)m, stream_);
      VSG_HIP(hipStreamSynchronize(stream_));
    };
    std::vector<int32_t> first_nodes, first_keys;
    std::vector<size_t> key_offset(split.size());
    for (size_t si = 0; si < split.size(); ++si) {
      key_offset[si] = first_nodes.size();
      for (auto& tube : split[si].second.tubes) first_nodes.push_back(first_node_of(tube));
    }
    fetch_labels(first_nodes, &first_keys);
    // Phase B: bookkeeping in the reference's order (region, then tube).  The reference looks the
    // tube's representative up through the union-find of its first pixel, which after an N4 swap
    // can belong to ANOTHER region whose rasterization is then replaced (reference behaviour,
    // kept); if that region is still to be visited its tube split is recomputed at its turn.
    std::vector<char> dirty((size_t)num_regions, 0);
    size_t si = 0;
    for (int r = 0; r < num_regions; ++r) {
      TubeResult local;
      TubeResult* trp = nullptr;
      std::vector<int32_t> keys;
      const bool cached = si < split.size() && split[si].first == r;
      if (dirty[r]) {
        if (cached) ++si;
        if (!regions_[r].has_raster) continue;
        split_again(regions_[r].raster, &local);
        if (local.tubes.size() <= 1) continue;
        std::vector<int32_t> nodes_needed;
        for (auto& tube : local.tubes) nodes_needed.push_back(first_node_of(tube));
        fetch_labels(nodes_needed, &keys);
        trp = &local;
      } else if (cached) {
        trp = &split[si].second;
        keys.assign(first_keys.begin() + key_offset[si],
                    first_keys.begin() + key_offset[si] + trp->tubes.size());
        ++si;
      } else {
        continue;
      }
      TubeResult& tr = *trp;
      for (int k = 0; k < (int)tr.tubes.size(); ++k) {
        int rep_key = keys[k];
        if (k != tr.tube_to_keep) {
          int& adj = size_adjust[rep_key];
          adj = (int)((float)adj - tr.areas[k]);   // int -= float
          rep_key = next_new_key++;
          const int idx = next_region_index_++;
          key_to_region_.emplace(rep_key, idx);
          regions_.emplace_back();
          regions_.back().index = idx;
          regions_.back().size = (int)tr.areas[k];
          regions_.back().constrained_id = -1;
          for (const RasterSlice& sl : tr.tubes[k]) {
            for (const Interval& iv : sl.raster) {
              rl_ty.push_back(((uint32_t)sl.frame << 16) | (uint32_t)iv.y);
              rl_lx.push_back(iv.lx);
              rl_rx.push_back(iv.rx);
              rl_new.push_back(rep_key);
            }
          }
        }
        auto it = key_to_region_.find(rep_key);
        int idx;
        if (it == key_to_region_.end()) {
          // representative without rasterization (possible only after N4 swaps)
          std::vector<int32_t> sz, cons;
          fetch_states(std::vector<int32_t>{rep_key}, &sz, &cons);
          idx = next_region_index_++;
          key_to_region_.emplace(rep_key, idx);
          regions_.emplace_back();
          regions_.back().index = idx;
          regions_.back().size = sz[0];
          regions_.back().constrained_id = cons[0];
        } else {
          idx = it->second;
        }
        if (idx != r && idx > r && idx < num_regions) dirty[idx] = 1;
        regions_[idx].has_raster = true;
        regions_[idx].raster.swap(tr.tubes[k]);
      }
    }
  }

  // 7. size adjustments (dense_segmentation_graph.h:566-578).
  for (const auto& kv : size_adjust) {
    auto it = key_to_region_.find(kv.first);
    if (it == key_to_region_.end()) {
      key_size_override_[kv.first] = 0;
      continue;
    }
    regions_[it->second].size += kv.second;
  }
  const double t_host1 = NowMs();

  // 8. DetermineNeighborIds: relabel split tubes on the device, then collect region pairs.
  const int n_rl = (int)rl_ty.size();
  if (n_rl > 0) {
    DevBuf<uint32_t>& d_ty = Grow(rl_ty_, n_rl);
    DevBuf<int32_t>&d_lx = Grow(rl_lx_, n_rl), &d_rx = Grow(rl_rx_, n_rl), &d_nl = Grow(rl_nl_, n_rl);
    H2D(d_ty.get(), rl_ty.data(), (size_t)n_rl, stream_);
    H2D(d_lx.get(), rl_lx.data(), (size_t)n_rl, stream_);
    H2D(d_rx.get(), rl_rx.data(), (size_t)n_rl, stream_);
    H2D(d_nl.get(), rl_new.data(), (size_t)n_rl, stream_);
    LaunchRelabelIntervals(d_ty.get(), d_lx.get(), d_rx.get(), d_nl.get(), n_rl, W_, H_,
                           label_uf_.get(), stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
  }
  const int L = (int)lists_.size();
  int capacity = (int)std::max<size_t>(pairs_.size(), (size_t)1 << 20);
  int count = 0;
  // The pairs through a hash table first (what is left to sort is the distinct pairs); a chunk with
  // more distinct pairs than half the table lists all of them, as before.
  bool hashed = false;
  if (!getenv("VSG_PAIR_TABLE") || atoi(getenv("VSG_PAIR_TABLE")) != 0) {
    const size_t cap = (size_t)1 << 21;
    pairs_sorted_.ensure(cap);
    pairs_unique_.ensure(cap);
    pairs_.ensure((size_t)capacity);
    order_keys_.ensure((size_t)capacity);
    PairTable table{pairs_sorted_.get(), pairs_unique_.get(), (unsigned)(cap - 1)};
    LaunchNeighborPairsHashed(list_desc_dev_.get(), L, label_uf_.get(), W_, table, scalars_.get() + 3, stream_);
    int distinct = 0;
    D2H(&distinct, scalars_.get() + 3, 1, stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
    if (distinct <= (int)(cap / 2) && distinct <= capacity) {
      LaunchPairTableCompact(table, pairs_.get(), order_keys_.get(), scalars_.get() + 3, stream_);
      D2H(&count, scalars_.get() + 3, 1, stream_);
      VSG_HIP(hipStreamSynchronize(stream_));
      VSG_REQUIRE(count == distinct, -4, "pair table: entries lost");
      hashed = true;
    }
  }
  for (int attempt = 0; attempt < 2 && !hashed; ++attempt) {
    pairs_.ensure((size_t)capacity);
    order_keys_.ensure((size_t)capacity);
    LaunchNeighborPairs(list_desc_dev_.get(), L, label_uf_.get(), W_, pairs_.get(),
                        order_keys_.get(), scalars_.get() + 3, capacity, stream_);
    D2H(&count, scalars_.get() + 3, 1, stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
    if (count <= capacity) break;
    VSG_REQUIRE(attempt == 0, -4, "neighbour pair buffer overflow");
    capacity = count + 1024;
  }
  std::vector<unsigned long long> uniq;
  if (count > 0) {
    pairs_sorted_.ensure((size_t)count);
    pairs_unique_.ensure((size_t)count);
    size_t temp = std::max(SortKeysU64TempBytes(count), UniqueU64TempBytes(count));
    if (temp > cub_temp_.size()) cub_temp_.alloc(temp);
    SortKeysU64(cub_temp_.get(), cub_temp_.size(), pairs_.get(), pairs_sorted_.get(), count, stream_);
    UniqueU64(cub_temp_.get(), cub_temp_.size(), pairs_sorted_.get(), pairs_unique_.get(),
              scalars_.get() + 4, count, stream_);
    int nu = 0;
    D2H(&nu, scalars_.get() + 4, 1, stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
    uniq.resize(nu);
    D2H(uniq.data(), pairs_unique_.get(), (size_t)nu, stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
  }
  const double t_dev2 = NowMs();

  // Representatives that own no interval get a RegionInformation in order of first appearance
  // (bucket, list, position, first-then-second end point).
  std::vector<int32_t> unseen;
  for (unsigned long long pr : uniq) {
    const int ka = (int)(uint32_t)(pr >> 32), kb = (int)(uint32_t)(pr & 0xFFFFFFFFull);
    if (!key_to_region_.count(ka)) unseen.push_back(ka);
    if (!key_to_region_.count(kb)) unseen.push_back(kb);
  }
  std::sort(unseen.begin(), unseen.end());
  unseen.erase(std::unique(unseen.begin(), unseen.end()), unseen.end());
  if (!unseen.empty()) {
    const int U = (int)unseen.size();
    DevBuf<int32_t>& d_keys = Grow(un_keys_, U);
    DevBuf<unsigned long long>& d_min = Grow(un_min_, U);
    H2D(d_keys.get(), unseen.data(), (size_t)U, stream_);
    VSG_HIP(hipMemsetAsync(d_min.get(), 0xFF, (size_t)U * sizeof(unsigned long long), stream_));
    LaunchFirstOrderOfKeys(pairs_.get(), order_keys_.get(), count, d_keys.get(), U, d_min.get(),
                           stream_);
    std::vector<unsigned long long> h_min(U);
    D2H(h_min.data(), d_min.get(), (size_t)U, stream_);
    VSG_HIP(hipStreamSynchronize(stream_));
    std::vector<int> order(U);
    for (int i = 0; i < U; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return h_min[a] < h_min[b]; });
    std::vector<int32_t> sz, cons;
    fetch_states(unseen, &sz, &cons);
    for (int oi : order) {
      const int key = unseen[oi];
      const int idx = next_region_index_++;
      key_to_region_.emplace(key, idx);
      regions_.emplace_back();
      RegionInfo& ri = regions_.back();
      ri.index = idx;
      auto ov = key_size_override_.find(key);
      ri.size = (ov != key_size_override_.end()) ? ov->second : sz[oi];
      ri.constrained_id = cons[oi];
      ri.has_raster = false;
    }
  }
  for (unsigned long long pr : uniq) {
    const int ka = (int)(uint32_t)(pr >> 32), kb = (int)(uint32_t)(pr & 0xFFFFFFFFull);
    const int ia = key_to_region_.at(ka), ib = key_to_region_.at(kb);
    regions_[ia].neighbors.push_back(ib);
    regions_[ib].neighbors.push_back(ia);
  }
  for (RegionInfo& ri : regions_) {
    std::sort(ri.neighbors.begin(), ri.neighbors.end());
    ri.neighbors.erase(std::unique(ri.neighbors.begin(), ri.neighbors.end()), ri.neighbors.end());
  }
  const double t_end = NowMs();
  if (getenv("VSG_DEBUG_STATS")) {
    std::fprintf(stderr, "[vsg] readout: device1 %.1f host1 %.1f device2 %.1f host2 %.1f ms (intervals %d, pairs %d, unique %zu)\n",
                 t_dev1 - t_start, t_host1 - t_dev1, t_dev2 - t_host1, t_end - t_dev2, num_iv, count,
                 uniq.size());
    std::fprintf(stderr, "[vsg]   host1: region table %.1f, tube prepare %.1f, flow samples %.1f, tube finish %.1f, "
                 "bookkeeping %.1f ms (%zu regions, %lld matched tubes, at most %d in a region)\n", th[0] - t_dev1,
                 th[1] - th[0], th[2] - th[1], th[3] - th[2], t_host1 - th[3], regions_.size(), sum_tubes, most_tubes);
  }
  timings_.readout_ms = (float)((t_dev1 - t_start) + (t_dev2 - t_host1));
  timings_.host_post_ms = (float)((t_host1 - t_dev1) + (t_end - t_dev2));
}

// ---------------------------------------------------------------------------------------------
// Parity hooks.
// ---------------------------------------------------------------------------------------------
void DenseGraphHip::CopySpatialBuckets(int t, uint16_t* out) {
  VSG_REQUIRE(t >= 0 && t < num_frames_ && lists_[2 * t].used, -1, "no spatial list for slice");
  ListBuf& lb = lists_[2 * t];
  std::vector<uint32_t> slots(lb.n);
  std::vector<int32_t> offs(kBucketSlots);
  D2H(slots.data(), lb.slots.get(), (size_t)lb.n, stream_);
  D2H(offs.data(), lb.offsets.get(), (size_t)kBucketSlots, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
  for (size_t i = 0; i < 4 * wh_; ++i) out[i] = 0xFFFF;
  for (int b = 0; b <= kNumBuckets; ++b) {
    for (int p = offs[b]; p < offs[b + 1]; ++p) {
      const uint32_t s = slots[p];
      out[(size_t)(s & 3u) * wh_ + (s >> 2)] = (uint16_t)b;
    }
  }
}

void DenseGraphHip::CopyTemporalBuckets(int t, uint16_t* out, int32_t* prev_idx) {
  VSG_REQUIRE(t >= 1 && t < num_frames_ && lists_[2 * t - 1].used, -1, "no temporal list");
  ListBuf& lb = lists_[2 * t - 1];
  std::vector<uint32_t> slots(lb.n);
  std::vector<int32_t> offs(kBucketSlots);
  D2H(slots.data(), lb.slots.get(), (size_t)lb.n, stream_);
  D2H(offs.data(), lb.offsets.get(), (size_t)kBucketSlots, stream_);
  D2H(prev_idx, lb.prev_idx.get(), wh_, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
  for (size_t i = 0; i < 9 * wh_; ++i) out[i] = 0xFFFF;
  for (int b = 0; b <= kNumBuckets; ++b) {
    for (int p = offs[b]; p < offs[b + 1]; ++p) {
      const uint32_t s = slots[p];
      const uint32_t pix = s / 9u, k = s - pix * 9u;
      out[(size_t)k * wh_ + pix] = (uint16_t)b;
    }
  }
}

void DenseGraphHip::CopyNodeRoots(int32_t* out) {
  const size_t N = wh_ * (size_t)num_frames_;
  LaunchFlatten(nodes(), N, label_uf_.get(), stream_);
  D2H(out, label_uf_.get(), N, stream_);
  VSG_HIP(hipStreamSynchronize(stream_));
}

}  // namespace vsg
