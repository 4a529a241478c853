// device_graph.h -- device-side data layout and kernel launch prototypes.
//
// HBM layout of one chunk graph (N = W*H*max_frames nodes, node id = t*W*H + y*W + x as in
// dense_segmentation_graph.h:1180-1228):
//   parent   int32[N]    union-find parent (FastSegmentationGraph::Region::my_id)
//   desc_sz  float4[N]   xyz = mean colour descriptor (B,G,R), w = region size as int bits
//   cons     int32[N]    constraint id (-1 unconstrained)
//   flags    uint8[N]    bit0 region_finalized, bit1 "no descriptor" (virtual / fresh reps)
// Edges are never stored as (a,b) pairs.  Every bucket list l (spatial slice t -> l = 2t,
// temporal slice t -> l = 2t-1, same numbering as the reference) owns
//   slots    uint32[n_l] edge slot ids, stably sorted by bucket => (bucket, scan order, k)
//   offsets  int32[2050] start of every bucket inside slots
//   kept     uint8[n_l]  1 = edge survived the merge (the reference's `remaining_edges`)
// with slot = pix*4+k (spatial) or pix*9+k (temporal); a temporal list also owns
// prev_idx int32[W*H], the (flow displaced) centre pixel in the previous slice.
#ifndef VSG_DEVICE_GRAPH_H_
#define VSG_DEVICE_GRAPH_H_

#include <functional>

#include "common.h"

namespace vsg {

constexpr uint8_t kFlagFinalized = 1;
constexpr uint8_t kFlagNoDesc = 2;
constexpr uint8_t kFlagTentative = 4;   // region has tentatively settled edges in this stage
// Hub regions of a stage (merge_stage.hip, "hubs"): a finalized region of at least the minimum size
// absorbs a small unconstrained neighbour whatever its own mean and size are, so its edges to such
// neighbours do not tie the neighbours' components together; the absorbed regions are logged and the
// hub's size and mean are brought up to date, in sequence order, after the workers (k_hub_apply).
// The hub mark itself lives in NodeArrays::hub8 (one byte per region, plain idempotent stores by
// the filter: thousands of edges mark the same region, and the flags byte is written with plain
// read-modify-write stores by the same kernel); the workers fold it into RState::flags as kFlagHub.
constexpr uint8_t kFlagHub = 8;         // (RState::flags only) the stage uses the region as a hub
constexpr uint8_t kFlagHubBroken = 16;  // the region is on the exclusion list (keeps the list free of duplicates:
                                        // set with atomics, possibly lost to a plain store -- a hint)
// A region that broke a hub rule is put on the stage's exclusion list (MergeScratch::hub_excl: count,
// then the regions) and carries kFlagHubExcluded through the retries of that stage: their filter
// does not make it a hub again (a hint only -- a stage is exact with any set of hubs).
constexpr uint8_t kFlagHubExcluded = 32;
constexpr int kHubExclCap = 16384;      // entries of the exclusion list
constexpr int kHubMaxAttempts = 4;      // retries of a stage with a longer list before it runs without hubs
// A stage is exact in front of the earliest edge that broke a hub rule, and that edge on its own is an
// ordinary edge: the stage is cut there -- [start, edge) with hubs, the edge alone, (edge, end) with hubs
// again -- up to kHubMaxSplits times per stage, before the exclusion list is used.
constexpr int kHubMaxSplits = 64;
// One buffer (MergeScratch::hub_excl) holds the exclusion list and the edges at which hub rules were
// broken: [0] regions on the exclusion list, [1] / [2] edges recorded by the filter (position inside the
// stage) / by the workers and k_hub_apply (number of the work edge), [4 ..) the regions, then the two
// lists of edges (kHubCutCap each; more violations than that are found by the parts of the cut stage).
constexpr int kHubCutCap = 64;
constexpr int kHubListInts = 4 + kHubExclCap + 3 * kHubCutCap;   // ([3] / third list: kept positions of edges, kind 2)

struct NodeArrays {
  int32_t* parent;
  float4* desc_sz;
  int32_t* cons;
  uint8_t* flags;
  uint8_t* hub8;    // 1: hub of the current stage (kFlagHub above); zero between stages
};

// One bucket list as seen by the merge / neighbour kernels.
struct ListDesc {
  const uint32_t* slots;     // sorted slot ids
  uint8_t* kept;             // per sorted position
  const int32_t* prev_idx;   // temporal lists only
  const int32_t* offsets;    // [kBucketSlots]
  int32_t type;              // 0 spatial, 1 temporal
  int32_t base_a;            // node id of pixel 0 of the slice owning the edges
  int32_t base_b;            // temporal: node id of pixel 0 of the previous slice
  int32_t n;                 // number of slots (valid + invalid)
};

struct MergeParams {
  int W, H;
  int num_lists;
  int min_region_size;
  float force_merge_weight;   // 0.001 (L2) / 0.002 (L1), dense_segmentation.cpp:259-264
  float inv_scale;            // (float)(1.0 / scale_), segmentation_graph.h:348
  // Squared-distance thresholds (see merge_common.h): largest float s with
  // sqrtf(s) < 0.05f, (double)sqrtf(s) < 0.2 and sqrtf(s) <= 0.15f respectively.
  float s_lt_005, s_lt_02, s_le_015;
  // Two-stage over-segmentation, full pass: per edge position, 1 = the spatial pass kept the
  // edge.  Spatial edges it did not keep no longer exist.  Null otherwise.
  const uint8_t* spatial_survivors;
};

// ---- build_kernels.hip ----------------------------------------------------------------
void UploadSpaceWeights(const float* w49, hipStream_t stream);
void LaunchMinMax(const uint8_t* bgr, size_t stride, int W, int H, int* mm, hipStream_t s);
void LaunchBilateral(const uint8_t* bgr, size_t stride, int W, int H, const float* lut,
                     float scale, float* planes, hipStream_t s);
// cv::GaussianBlur 3x3 with the symmetric kernel {k1, k0, k1} (PRESMOOTH_GAUSSIAN).
void LaunchGaussian3(const uint8_t* bgr, size_t stride, int W, int H, float k0, float k1, float* planes,
                     hipStream_t s);
void LaunchConvertPlanar(const uint8_t* bgr, size_t stride, int W, int H, float* planes,
                         hipStream_t s);
void LaunchInterleavedToPlanar(const float* in, size_t n, float* planes, hipStream_t s);
void LaunchPlanarToInterleaved(const float* planes, size_t n, float* out, hipStream_t s);
// ---- edge_sort.hip: edge keys + stable bucket (counting) sort of the edge slots -----------------
// keys: u16 per slot (slot = pix*4+k / pix*9+k); hist: bin-major tile histograms (scratch of
// EdgeSortHistInts ints, sums: EdgeSortSumInts ints).
size_t EdgeSortHistInts(size_t n_px);
size_t EdgeSortSumInts(size_t n_px);
void LaunchSpatialKeys(const float* feat, int W, int H, int l1, uint16_t* keys, int32_t* hist,
                       hipStream_t s);
void LaunchTemporalKeys(const float* cur, const float* prev, const float* flow, int W, int H,
                        int l1, int is_virtual, uint16_t* keys, int32_t* prev_idx, int32_t* hist,
                        hipStream_t s);
// offsets[kBucketSlots]: start of every bucket; slots_out: slot ids of the existing edges, stably
// sorted by bucket.
void LaunchBucketSort(const uint16_t* keys, size_t n_px, int per_px, int32_t* hist, int32_t* sums,
                      int32_t* offsets, uint32_t* slots_out, hipStream_t s);
void LaunchInitNodes(const float* feat, size_t n, int base, const int32_t* cons_in,
                     NodeArrays nodes, hipStream_t s);
void LaunchInitVirtualNodes(const int32_t* labels, size_t n, int base, int num_labels,
                            int32_t* first_scratch, NodeArrays nodes, hipStream_t s);

// ---- scan_device.h: hand-written device-wide scans (the merge path and the read-out) -------------------
// Scratch of the scans of one stream (all scans of a graph run on its main stream): the tile sums.
constexpr int kScanMaxTiles = 4096;
struct ScanScratch {
  int32_t* sums;   // [kScanMaxTiles]
  hipStream_t owner = nullptr;   // the one stream whose scans may use `sums` (null: not checked)
};

// ---- radix_sort.hip: the stable (key, value) sort of the merge path, hand-written ----------------------
// (sort_scan.hip: the rocPRIM 64-bit sort / unique of the read-out)
// SortPairsU32 = the hand-written sort inside its window of sizes, the library outside (measured, radix_sort.hip).
size_t SortPairsU32HandTempBytes(int n);
void SortPairsU32Hand(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                      const uint32_t* vals_in, uint32_t* vals_out, int n, int end_bit, hipStream_t s);
size_t SortPairsU32LibTempBytes(int n);
void SortPairsU32Lib(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                     const uint32_t* vals_in, uint32_t* vals_out, int n, int end_bit, hipStream_t s);
size_t SortPairsU32TempBytes(int n);
void SortPairsU32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                  const uint32_t* vals_in, uint32_t* vals_out, int n, int end_bit, hipStream_t s);
size_t SortKeysU64TempBytes(int n);
void SortKeysU64(void* temp, size_t temp_bytes, const unsigned long long* in,
                 unsigned long long* out, int n,
                 hipStream_t s);
size_t UniqueU64TempBytes(int n);
void UniqueU64(void* temp, size_t temp_bytes, const unsigned long long* in,
               unsigned long long* out,
               int32_t* num_out, int n, hipStream_t s);

// Scalars the host has to wait for (sizes, flags) reach it through a mailbox in mapped, coherent
// host memory: the kernel that produces the value stores (sequence number << 32 | value) with system
// scope and the host spins on the word -- no copy kernel, no stream synchronisation (a read-back
// through hipMemcpyAsync + hipStreamSynchronize costs 30-40 us of idle GPU, 250 times per chunk).
// A slot holds up to four values; slots rotate, sequence numbers never repeat.
constexpr int kMailValues = 4;
struct MailSlot {
  unsigned long long* dev;              // device address of the slot's words
  volatile unsigned long long* host;    // the same words as the host sees them
  unsigned seq;
};
struct Mailbox {
  unsigned long long* dev = nullptr;
  volatile unsigned long long* host = nullptr;
  int slots = 0;
  int next = 0;
  unsigned seq = 0;
  // a small list in mapped memory (the large components of a stage, merge_spine.hip)
  int32_t* list_dev = nullptr;
  volatile int32_t* list_host = nullptr;
  int list_cap = 0;
};
inline MailSlot NextMail(Mailbox& m) {
  MailSlot s{m.dev + (size_t)kMailValues * m.next, m.host + (size_t)kMailValues * m.next, ++m.seq};
  m.next = (m.next + 1) % m.slots;
  return s;
}
// Waits until the first `count` values of the slot have arrived (spins; looks at the stream for
// errors now and then) and returns them.
void MailWait(const MailSlot& slot, int count, int* values, hipStream_t s);
// The waiting policy (merge_stage.hip: VSG_MAIL_YIELD): graphs register themselves so that a process
// with more streams than cores stops spinning by itself.
void MailRegisterGraph(int delta);
int MailYieldMode();
// What the calling thread spent in MailWait so far (diagnostics: the difference around a call).
struct MailWaitCounters {
  long long waits = 0;
  double wait_ms = 0, longest_ms = 0;
};
MailWaitCounters MailWaitSnapshot();
void MailWaitResetLongest();
// Posts up to four device scalars from a one-thread kernel (where no kernel of the algorithm is at
// hand to do it): values[i] = *ptrs[i].
void LaunchMailPost(const MailSlot& slot, const int32_t* p0, const int32_t* p1, const int32_t* p2,
                    const int32_t* p3, hipStream_t s);

// Zeroed device counters, handed out in order and cleared in one go per chunk (a hipMemsetAsync per
// counter per stage was 450 launches per chunk).
constexpr size_t kZeroArenaInts = 1024;   // head of the pool: scalars that live as long as a stage
struct ZeroPool {
  int32_t* base = nullptr;
  size_t cap = 0, used = 0;     // `used` counts inside the current half (TakeZeroed)
  bool second_half = false;
  size_t arena_used = 0;        // TakeStageScalars
};

// ---- merge_stage.hip (worker: merge_wave.hip) ----------------------------------------------------------------
// What k_filter found, three bits per edge of the stage: one 64-bit word per wavefront (64
// consecutive edges) and class, and the number of active edges per workgroup of 256 edges.
struct FilterMasks {
  unsigned long long* active;      // the edge goes to a worker
  unsigned long long* settled;     // kept by the filter (inert)
  unsigned long long* tentative;   // settled under an assumption (inert_mode 2)
  int32_t* block_cnt;
};
// The non-empty (bucket, list) segments of a stage's edge range (k_filter): start inside the stage,
// list index, position of the segment's first edge in the list's sorted slots.
struct FilterSegs {
  const int32_t* start;
  const int32_t* list;
  const int32_t* pos0;
  int n;
};
struct MergeScratch {
  // e_ra / e_rb / e_gpos, the masks and the block counts are sized for the largest bucket; everything
  // else holds active edges only: active_cap of them (grow_active enlarges those arrays -- all
  // streams are drained, nothing of the current stage may be in them yet -- and rebinds the pointers)
  int active_cap;
  std::function<bool(long long need)> grow_active;
  int32_t* e_ra;         // root of node a at filter time (per bucket edge)
  int32_t* e_rb;
  uint32_t* e_gpos;      // global kept position (list_slot_base[l] + pos)
  int32_t* e_active;     // scratch (run-leader flags)
  int32_t* e_apos;       // scratch (kept positions in component order)
  FilterMasks masks;     // the filter's verdict per edge
  int32_t* block_off;    // exclusive scan of masks.block_cnt
  int32_t* a_ra;         // compacted active edges (bucket order)
  int32_t* a_rb;
  uint32_t* a_gpos;
  uint32_t* a_comp;      // component id (cc root) per active edge
  uint32_t* a_idx;       // 0..n_active-1
  uint32_t* s_comp;      // sorted by component (stable)
  uint32_t* s_idx;
  uint32_t* seg_key;     // run-length encoding of s_comp
  int32_t* seg_cnt;
  int32_t* seg_off;
  int32_t* num_active;   // device scalars: [0] num_active [1] num_segs [2] num_ti [3] violation [5] num_leaders
  int32_t* num_segs;     // = num_active + 1
  float4* bk_ds;         // undo buffers of an optimistic stage (2 entries per active edge)
  int32_t* bk_cons;
  uint8_t* bk_flags;
  int force_rollback;    // test hook: treat every optimistic stage as violated
  int32_t* lead_pos;     // exclusive scan of the run-leader flags (per active edge)
  int32_t* l_ra;         // run leaders: what the workers replay when use_rle is set
  int32_t* l_rb;
  uint32_t* l_gpos;
  int use_rle;           // replay one edge per run of equal root pairs (default on)
  // A stage may cover several consecutive buckets with the same thresholds: [bucket, group_hi)
  // (group_hi <= bucket: one bucket); bucket_prefix[b] = edges in the buckets before b (device).
  int group_hi;
  const int32_t* bucket_prefix;
  const int32_t* bucket_prefix_host;   // the same table on the host
  const int32_t* bucket_base_host;     // host copy of the bucket table [(kNumBuckets + 1) x (L + 1)]
  const int32_t* list_off_host;        // [L][kNumBuckets + 2]: start of every bucket in every list's slots
  int32_t* seg_dev;                    // device buffer of the stage's segment table (FilterSegs)
  size_t seg_cap;                      // ... its capacity in ints
  std::vector<int32_t>* seg_host;      // staging
  // Kruskal-tree replay of the large components (merge_spine.hip)
  int spine_min;         // components of at least this many replayed edges; 0: never
  int spine_off;             // the current stage is a replay without the tree replay
  int* spine_low_failed;     // [2] out: the assumption failed in bucket 0 / 1 during this chunk
  const int* spine_low_skip; // [2] in: bucket 0 / 1 skip the tree replay in this chunk
  int* spine_limit_bucket;   // buckets from this one on skip the tree replay (its assumption failed there)
  int spine_nested_factor;   // side clusters go one level down from spine_min * this many edges
  int spine_max_edges;   // at most this many edges per stage (scratch pool)
  int spine_debug, spine_check;
  int rank_split_min;    // Euler tours of at least this many arcs are ranked by sampling (k_rank_walk)
  int spine_fast;        // the plain steps of a spine through the streamed chain (k_spine_chain)
  int spine_fast_min;    // ... from this many tree edges on (seven more launches)
  int32_t* spine_pool;   // scratch, SpinePoolInts(spine_max_edges) ints
  size_t spine_pool_ints;
  // Enlarges the pool so that it holds `edges` edges (all streams of the graph are idle when it is
  // called); updates spine_pool / spine_pool_ints / spine_max_edges and returns true, or false if
  // the request is beyond what the tree replay can address.
  std::function<bool(long long edges)> grow_spine_pool;
  int32_t* nmap[3];      // three more [N] scratch maps (free during the bucket stages)
  hipStream_t aux_stream;   // the ordinary workers run here while the trees are built on the main stream
  hipStream_t aux2_stream;  // the ordinary side clusters of a tree level, beside the level below
  hipEvent_t aux_fork, aux_join;
  int small_seg;         // components of up to this many replayed edges: one lane each (k_merge_small)
  int wide_min;          // components of at least this many replayed edges: wide worker (merge_wide.hip), 0: off
  int wide_waves;        // wavefronts per component of the wide worker (2 or 4)
  int chain_relax;       // StageThr::relax (VSG_CHAIN_RELAX, default 1)
  int hubs;              // hub regions (VSG_HUBS, default 1): see kFlagHub
  int hubs_off;          // (> 0 while a stage that violated a hub rule is replayed without hubs)
  int hub_attempt;       // retries of the current stage with hubs (kHubMaxAttempts)
  int hub_splits_left;   // times the current top-level stage may still be cut at a violating edge (below)
  int hub_split_depth;   // (> 0 inside the parts of a cut stage)
  int hub_list_dirty;    // the head of the list has to be cleared before the next stage uses it
  int hub_max_splits;    // cuts per top-level stage (VSG_HUB_SPLITS, kHubMaxSplits; 0: a violated stage is rerun as a whole)
  int hub_cut_min_work;  // stages of fewer replayed edges are not cut (VSG_CUT_MIN_WORK, 0: measured, 32768 is worse)
  long long hub_splits;  // cuts in this Segment call
  const uint32_t* list_slot_base_host;   // first kept position of every list (host copy), for StagePosition
  int32_t* hub_excl;     // exclusion list of the stage and the edges that broke hub rules (kHubListInts)
  long long hub_retries; // such replays in this Segment call
  long long hub_reasons[6];   // ... by reason (kHubVioBroken .. kHubVioSplit)
  int wave_debug;        // use the instrumented build of the wave worker (counters, self checks)
  int wave_dbg;          // debug hook (bit mask): 1 no chain, 4 no hot region, 8 one generic lane per round,
                         // 16 chain self check, 32 no jumping over pending lanes, 64 one chain lane per round
  int64_t* optimistic_stages;
  int64_t* rollbacks;
  int32_t* cc;           // [N] component scratch (identity outside a stage)
  unsigned long long* stats;   // [16]: forced, regular, small, wave edges, debug x4; [8..15] undo copy
  void* cub_temp;        // temporary storage of the radix sort
  size_t cub_temp_bytes;
  ScanScratch scan;
  int node_key_bits;     // bits of the largest node id: what a radix sort by vertex / component has to look at
  // HIP event pairs recorded around k_filter / k_merge_wave launches (resolved by the caller
  // after the stream has been synchronised).
  std::vector<hipEvent_t>* ev_pool;
  std::vector<std::pair<int, int>>* ev_wave;     // indices into ev_pool
  std::vector<std::pair<int, int>>* ev_filter;
  std::vector<std::pair<int, int>>* ev_spine;    // k_spine launches
  int* ev_used;
  Mailbox* mail;
  ZeroPool* zeros;
  hipStream_t main_stream;
};
// n zeroed ints (see ZeroPool) for kernels that are launched right away; when the current half of
// the pool is used up all three streams are drained and the other half is cleared and taken over.
int32_t* TakeZeroed(MergeScratch& S, size_t n);

// n zeroed ints that stay intact until the stage that took them (at its entry) has ended.
int32_t* TakeStageScalars(MergeScratch& S, size_t n);

// bucket_base[b * (L+1) + l] = number of bucket-b edges in lists < l; [.. + L] = total.
void LaunchBuildBucketTable(const ListDesc* lists, int num_lists, int32_t* bucket_base,
                            hipStream_t s);
void LaunchInitIdentity(int32_t* a, size_t n, hipStream_t s);
// Runs one stage (filter -> components -> exact workers) on the edges [j0, j0 + n_b) of the
// bucket's edge sequence (list order, then position).  Any split of a bucket into consecutive
// windows run one after the other is equivalent to one stage over the whole bucket.
struct StageInfo {
  int want_components = 0;   // in: 1: report `components` / `max_wave_segment` of a stage that replays
                             // more than 16 K edges (one more synchronisation), 2: of any stage
  int replayed = 0;     // edges handed to the workers (run leaders)
  int components = 0;   // independent components they fell into
  int max_wave_segment = 0;   // (with want_components) edges of the largest component that one
                              // wavefront replayed -- those of the tree replay do not count
  int hub_stages = 0;         // the stage used hub regions (kFlagHub)
  long long hub_absorbed = 0; // regions its hubs absorbed
  int hub_retries = 0;        // times the stage was run again with broken hubs as ordinary regions: the
                              // neighbourhood of such a region is one large component of the retry --
                              // not the percolation the window target reacts to
};
// Takes the marks of the hub exclusion list off the regions and empties the list.
void ResetHubExclusions(MergeScratch& S, NodeArrays nodes, hipStream_t s);
void RunBucketStage(int bucket, int j0, int n_b, const ListDesc* lists, const int32_t* bucket_base,
                    const uint32_t* list_slot_base, uint8_t* kept_all, NodeArrays nodes,
                    const MergeParams& P, int inert_mode, MergeScratch& S, hipStream_t s,
                    StageInfo* info = nullptr);
// Scratch ints the Kruskal-tree replay (merge_spine.hip) needs for max_edges edges per stage.
size_t SpinePoolInts(size_t max_edges);
// Marks every edge of bucket 2048 (virtual edges) as kept.
void LaunchKeepVirtualBucket(const ListDesc* lists, int num_lists, hipStream_t s);

// ---- readout_kernels.hip --------------------------------------------------------------
void LaunchFlatten(NodeArrays nodes, size_t n, int32_t* label_uf, hipStream_t s);
// N4 sweep on the listed slices of label_img (one workgroup per slice); adjust[key] receives the
// per-region size change.
void LaunchEnforceN4(int32_t* label_img, int W, int H, const int32_t* frames_dev, int num_frames,
                     int32_t* row_flags /* [num_frames * H] scratch, zeroed */, int32_t* adjust /* [N] */,
                     hipStream_t s);
// Run-length encoding of one slice: counts per row, then intervals.
void LaunchRowRunCounts(const int32_t* label_img, int W, int H, int frame, int32_t* row_counts,
                        hipStream_t s);
struct IntervalArrays {
  int32_t* label;     // region representative key
  uint32_t* ty;       // frame << 16 | y
  int32_t* lx;        // left_x
  int32_t* rx;        // right_x
};
void LaunchWriteIntervals(const int32_t* label_img, int W, int H, int frame,
                          const int32_t* row_offsets /* exclusive, global */, IntervalArrays out,
                          hipStream_t s);
// Relabels nodes covered by the given intervals (tube splitting).
void LaunchRelabelIntervals(const uint32_t* ty, const int32_t* lx, const int32_t* rx,
                            const int32_t* new_label, int n, int W, int H, int32_t* label_uf,
                            hipStream_t s);
// Hash table of the region pairs of a chunk: pair -> smallest order key (readout_kernels.hip).
struct PairTable {
  unsigned long long* key;     // [mask + 1], all ones = empty
  unsigned long long* order;   // [mask + 1]
  unsigned mask;               // capacity - 1 (a power of two)
};
// The pairs go into the table instead of a list (*distinct = pairs entered); LaunchPairTableCompact
// then lists the table's entries as (pairs, order_keys), *out_count of them.
void LaunchNeighborPairsHashed(const ListDesc* lists, int num_lists, const int32_t* label_uf, int W,
                               PairTable table, int32_t* distinct, hipStream_t s);
void LaunchPairTableCompact(PairTable table, unsigned long long* pairs, unsigned long long* order_keys,
                            int32_t* out_count, hipStream_t s);
// Emits (ka << 32 | kb) for every kept edge whose end labels differ.  count is a device scalar.
void LaunchNeighborPairs(const ListDesc* lists, int num_lists, const int32_t* label_uf, int W,
                         unsigned long long* pairs, unsigned long long* order_keys,
                         int32_t* count, int capacity, hipStream_t s);
void LaunchGatherStates(NodeArrays nodes, const int32_t* ids, int n, float4* desc_sz_out,
                        int32_t* cons_out, int32_t* flags_out, hipStream_t s);
void LaunchScatterStates(NodeArrays nodes, const int32_t* ids, int n, const int32_t* parent_in,
                         const float4* desc_sz_in, const int32_t* cons_in, const int32_t* flags_in,
                         hipStream_t s);
// Nodes with own constraint >= 0 in [begin, end): flag array for compaction + their roots.
void LaunchConstrainedRoots(NodeArrays nodes, int begin, int end, int32_t* flag_out,
                            int32_t* root_out, hipStream_t s);
// The same walk reduced to runs of equal representatives (see readout_kernels.hip): value_out[j] =
// representative of node begin + j, -2 for an unconstrained representative, -1 for a node that
// does not matter; flag_out[j] = the node starts a run (or terminates the one before it).
void LaunchConstrainedRuns(NodeArrays nodes, int begin, int end, int32_t* value_out,
                           int32_t* flag_out, hipStream_t s);
void LaunchCompactI32(const int32_t* flags, const int32_t* offsets, const int32_t* values, int n,
                      int32_t* out, hipStream_t s);
void LaunchGatherI32(const int32_t* src, const int32_t* idx, int n, int32_t* out, hipStream_t s);
// out[i] = flows[req[3i]][req[3i+1] * W + req[3i+2]] (interleaved x, y); flows: device array of
// per-slice device pointers (null: no flow for the slice).
void LaunchGatherFlow(const int32_t* req, int n, const float* const* flows, int W, float2* out,
                      hipStream_t s);

void LaunchNonzeroFlags(const int32_t* a, int n, int32_t* flags, hipStream_t s);
void LaunchCompactIndexValue(const int32_t* flags, const int32_t* offsets, const int32_t* values,
                             int n, int32_t* out_idx, int32_t* out_val, hipStream_t s);
void LaunchFirstOrderOfKeys(const unsigned long long* pairs, const unsigned long long* order_keys,
                            int m, const int32_t* keys_sorted, int num_keys,
                            unsigned long long* out_min, hipStream_t s);

}  // namespace vsg

#endif  // VSG_DEVICE_GRAPH_H_
