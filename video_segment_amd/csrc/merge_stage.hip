// merge_stage.hip -- the ordered union-find merge (K6), gfx950: one stage of the edge sequence.
//
// Reference semantics restated: FastSegmentationGraph::SegmentGraph / GetRegion / MergeRegions
// (segmentation/segmentation_graph.h:339-463, 651-701) with ColorMeanDescriptorTraits
// (segmentation/pixel_distance.h:469-521).  The reference walks every edge sequentially in
// (bucket, bucket list, insertion) order and its merge predicate depends on evolving float state,
// so the result is order dependent.  This file keeps that order *exactly* and extracts the
// parallelism that is provably free:
//
//   stage = a consecutive range of the edge sequence: a bucket, a rank window of a bucket, or a
//   group of buckets with the same thresholds (the argument below never uses that the range is a
//   bucket).  k_filter (all CUs): find both roots with path compression, drop edges that are
//   already internal, settle edges between two finalized, large, unconstrained-graph regions as
//   "kept" (they can never change state again), and hook the roots of the remaining *active*
//   edges into a scratch union-find (ECL-CC style atomicCAS hooking).
//   Two active edges can only influence each other if they are connected through active edges of
//   the same stage, so each connected component of that scratch graph is an independent
//   sequential sub-problem.  Active edges are stably sorted by component and every component is
//   replayed in the reference's order by its own worker: one lane for a small component
//   (k_merge_small), a reader + a consumer wavefront for an ordinary one (merge_wave.hip), the
//   Kruskal-tree replay for a large one (merge_spine.hip).  A stage that relies on an assumption
//   (tentatively settled edges, run leaders, the tree structure) is undoable: region states are
//   backed up, a worker that sees the assumption fail raises the violation flag, and the stage is
//   restored and replayed without it.
//
// This is memory-latency / dependency bound integer + scalar float work: no MFMA, no LDS tiling;
// what matters is coalesced streaming in the filter and keeping the serial chains in registers.
// Shared device helpers in merge_common.h.
#include <sched.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "merge_common.h"
#include "scan_device.h"

namespace vsg {

__global__ __launch_bounds__(256) void k_build_bucket_table(const ListDesc* __restrict__ lists,
                                                             int num_lists,
                                                             int32_t* __restrict__ bucket_base) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b > kNumBuckets) return;
  int acc = 0;
  int32_t* row = bucket_base + (size_t)b * (num_lists + 1);
  for (int l = 0; l < num_lists; ++l) {
    row[l] = acc;
    const int32_t* off = lists[l].offsets;
    if (off) acc += off[b + 1] - off[b];
  }
  row[num_lists] = acc;
}

void LaunchBuildBucketTable(const ListDesc* lists, int num_lists, int32_t* bucket_base,
                            hipStream_t s) {
  hipLaunchKernelGGL(k_build_bucket_table, dim3((kNumBuckets + 1 + 255) / 256), dim3(256), 0, s,
                     lists, num_lists, bucket_base);
  VSG_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_init_identity(int32_t* a, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = (int32_t)i;
}

void LaunchInitIdentity(int32_t* a, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_init_identity, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, n);
  VSG_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// Mailbox and zeroed counters (device_graph.h).
// ------------------------------------------------------------------------------------------
__global__ void k_mail_post(unsigned long long* slot, unsigned seq, const int32_t* p0, const int32_t* p1,
                            const int32_t* p2, const int32_t* p3) {
  if (p0) MailPost(slot, seq, 0, *p0);
  if (p1) MailPost(slot, seq, 1, *p1);
  if (p2) MailPost(slot, seq, 2, *p2);
  if (p3) MailPost(slot, seq, 3, *p3);
}

void LaunchMailPost(const MailSlot& slot, const int32_t* p0, const int32_t* p1, const int32_t* p2,
                    const int32_t* p3, hipStream_t s) {
  hipLaunchKernelGGL(k_mail_post, dim3(1), dim3(1), 0, s, slot.dev, slot.seq, p0, p1, p2, p3);
  VSG_HIP(hipGetLastError());
}

// How a waiting host thread treats its core.  Every stream has one thread in MailWait for most of a
// merge (about 150 waits per chunk); on a node with fewer free cores than streams (8 GPUs x S
// streams) threads that only ever `pause` starve each other and the units feeding them.
//   VSG_MAIL_YIELD=0  spin (lowest latency: about 14 us from the producing kernel to the next launch)
//   VSG_MAIL_YIELD=1  spin briefly, then sched_yield between polls
//   VSG_MAIL_YIELD=2  spin briefly, then sleep between polls (20 us, doubling up to 200 us)
//   unset             0 while the process holds at most half as many graphs as it may use cores, 1 beyond
static std::atomic<int> g_live_graphs{0};
void MailRegisterGraph(int delta) { g_live_graphs.fetch_add(delta, std::memory_order_relaxed); }

static int UsableCores() {   // (of the calling thread, now: a syscall per wait is nothing beside the wait)
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) return (int)CPU_COUNT(&set);
  const unsigned h = std::thread::hardware_concurrency();
  return h ? (int)h : 1;
}

int MailYieldMode() {
  if (const char* e = getenv("VSG_MAIL_YIELD")) return std::max(0, std::min(2, atoi(e)));
  return 2 * g_live_graphs.load(std::memory_order_relaxed) > UsableCores() ? 1 : 0;
}

static thread_local MailWaitCounters t_mail_counters;
MailWaitCounters MailWaitSnapshot() { return t_mail_counters; }
void MailWaitResetLongest() { t_mail_counters.longest_ms = 0; }

namespace {
struct MailWaitTimer {   // host time of one MailWait call, whichever way it ends
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~MailWaitTimer() {
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    t_mail_counters.waits += 1;
    t_mail_counters.wait_ms += ms;
    if (ms > t_mail_counters.longest_ms) t_mail_counters.longest_ms = ms;
  }
};
}  // namespace

void MailWait(const MailSlot& slot, int count, int* values, hipStream_t s) {
  using clk = std::chrono::steady_clock;
  MailWaitTimer wait_timer;
  clk::time_point t0;
  bool timed = false;
  const int mode = MailYieldMode();
  constexpr unsigned long long kBriefSpins = 512;   // a few microseconds: most values are there by then
  long sleep_ns = 20000;
  for (int i = 0; i < count; ++i) {
    for (unsigned long long spins = 0;; ++spins) {
      const unsigned long long w = slot.host[i];
      if ((unsigned)(w >> 32) == slot.seq) {
        values[i] = (int)(unsigned)(w & 0xffffffffull);
        break;
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      if (mode != 0 && spins >= kBriefSpins) {
        if (mode == 1) {
          std::this_thread::yield();
        } else {
          const timespec ts = {0, sleep_ns};
          nanosleep(&ts, nullptr);
          sleep_ns = std::min(sleep_ns * 2, 200000l);
        }
      }
      // nothing for a while (65536 spins, or about 10 ms of polite polls): a kernel that died
      // would never post
      if ((mode == 0 && (spins & 0xffffull) == 0xffffull) || (mode != 0 && (spins & 0x3ffull) == 0x3ffull)) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipSuccess && e != hipErrorNotReady) VSG_HIP(e);
        if (!timed) {
          t0 = clk::now();
          timed = true;
        } else if (e == hipSuccess && clk::now() - t0 > std::chrono::seconds(2)) {
          // the stream is idle and the value is not there: whoever should post it did not
          const unsigned long long w2 = slot.host[i];
          if ((unsigned)(w2 >> 32) != slot.seq) throw Error(-4 /* VSG_ERR_INTERNAL */, "mailbox: a value never arrived");
        }
        if (mode == 0) std::this_thread::yield();
      }
    }
  }
}

static void DrainStreams(MergeScratch& S) {
  VSG_HIP(hipStreamSynchronize(S.main_stream));
  if (S.aux_stream) VSG_HIP(hipStreamSynchronize(S.aux_stream));
  if (S.aux2_stream) VSG_HIP(hipStreamSynchronize(S.aux2_stream));
}

int32_t* TakeZeroed(MergeScratch& S, size_t n) {
  // Behind the stage arena (below) the pool is used in two halves, for counters that are consumed
  // by kernels launched right after they are taken: when one half is used up all streams are drained
  // -- so every kernel that was given a counter of either half is complete --, the OTHER half is
  // cleared and taken over.  A counter that a kernel launched AFTER a later TakeZeroed still has to
  // find intact (the stage's violation word) must not come from here: TakeStageScalars.
  ZeroPool& z = *S.zeros;
  n = (n + 3) & ~(size_t)3;
  VSG_REQUIRE(z.cap > 2 * kZeroArenaInts, -4, "zero pool: too small");
  const size_t half = (z.cap - kZeroArenaInts) / 2;
  VSG_REQUIRE(n <= half, -4, "zero pool: request too large");
  if (z.used + n > half) {
    DrainStreams(S);
    z.second_half = !z.second_half;
    z.used = 0;
    VSG_HIP(hipMemsetAsync(z.base + kZeroArenaInts + (z.second_half ? half : 0), 0, half * sizeof(int32_t),
                           S.main_stream));
  }
  int32_t* p = z.base + kZeroArenaInts + (z.second_half ? half : 0) + z.used;
  z.used += n;
  return p;
}

int32_t* TakeStageScalars(MergeScratch& S, size_t n) {
  // The scalars a stage keeps from its first kernel to its last (tentative-edge flag, violation
  // word) live in their own arena at the head of the pool, which the half swaps above never touch.
  // It is only ever recycled here, at the entry of a stage, where no earlier stage's scalars are
  // live any more (a stage reads its violation word back before it returns or replays itself).
  ZeroPool& z = *S.zeros;
  n = (n + 3) & ~(size_t)3;
  VSG_REQUIRE(n <= kZeroArenaInts, -4, "zero pool: stage scalars too large");
  if (z.arena_used + n > kZeroArenaInts) {
    DrainStreams(S);
    VSG_HIP(hipMemsetAsync(z.base, 0, kZeroArenaInts * sizeof(int32_t), S.main_stream));
    z.arena_used = 0;
  }
  int32_t* p = z.base + z.arena_used;
  z.arena_used += n;
  return p;
}

// ------------------------------------------------------------------------------------------
// Stage step 1: filter.
// ------------------------------------------------------------------------------------------
// inert_mode 0: every non-internal edge is active.
// inert_mode 1 (graph without constraints): an edge between two finalized regions of at least
//   min size is kept and changes no state whenever it is visited -- exact, because such a region
//   stays finalized and large and there is no constraint that could force a merge.
// inert_mode 2 (graph with constraints): the same edges, and edges between regions with different
//   constraints, are *tentatively* settled as kept.  That is only valid while the constraint of
//   the regions involved does not change during this stage, so both regions are marked
//   (kFlagTentative in the region flags, which travel with the state into the workers); a worker
//   that changes the constraint of a marked region raises the stage's violation flag and the host
//   rolls the stage back and replays it with inert_mode 0 (see RunBucketStage).
// The stage covers the edges [j0, j0 + n_b) of the buckets [bucket, bucket_hi) (one after the
// other; bucket_prefix[b] = edges in the buckets before b).
// Every thread filters kFilterPer edges, 256 apart: the kernel is a chain of dependent loads per edge
// (slot -> flow-displaced pixel -> parent -> parent of parent ..., 91 % of its wave cycles in
// s_waitcnt at full occupancy), so the only parallelism left to add is inside the thread -- the
// chains of a thread's edges advance together, every step one batch of independent loads.  The
// bookkeeping downstream (masks, packed records, per-block counts) stays in units of 256 edges:
// a workgroup simply covers kFilterPer such blocks.
constexpr int kFilterPer = 2;
constexpr int kFilterEdges = 256 * kFilterPer;   // edges per workgroup

__global__ __launch_bounds__(256) void k_filter(int bucket, int bucket_hi, int j0, int n_b,
                                                 const ListDesc* __restrict__ lists,
                                                 const int32_t* __restrict__ bucket_base,
                                                 const int32_t* __restrict__ bucket_prefix,
                                                 const uint32_t* __restrict__ list_slot_base,
                                                 uint8_t* __restrict__ kept_all, NodeArrays nodes,
                                                 MergeParams P, int inert_mode,
                                                 int32_t* __restrict__ cc,
                                                 int32_t* __restrict__ e_ra,
                                                 int32_t* __restrict__ e_rb,
                                                 uint32_t* __restrict__ e_gpos,
                                                 FilterMasks M,
                                                 int32_t* __restrict__ num_ti, FilterSegs segs, int hubs,
                                                 int32_t* __restrict__ num_hub, int32_t* __restrict__ hub_excl,
                                                 float split_s) {
  __shared__ int wave_cnt[kFilterPer][4], wave_kept[kFilterPer][4];
  __shared__ int s_start[kFilterEdges], s_pos0[kFilterEdges];
  // ... with what the edges of a segment need from its list (a 48-byte descriptor per thread
  // otherwise: six loads of the 22 a wavefront issued)
  __shared__ const uint32_t* s_slots[kFilterEdges];
  __shared__ const int32_t* s_prev[kFilterEdges];
  __shared__ int s_type[kFilterEdges], s_base_a[kFilterEdges], s_base_b[kFilterEdges];
  __shared__ uint32_t s_slot_base[kFilterEdges];
  const int jf = blockIdx.x * kFilterEdges;   // first edge of the workgroup inside the stage's window
  // Where an edge lives: the host lists the stage's non-empty (bucket, list) segments (start inside
  // the stage, list, position of the segment's first edge in the list's sorted slots).  A
  // workgroup's edges touch at most kFilterEdges of them, found by the workgroup together and kept
  // in LDS -- the two binary searches per thread over tables in global memory (up to fifteen
  // dependent loads before the first byte of the edge itself) made the kernel latency bound.
  int m_segs = 0;
  if (segs.start) {
    int first = 0;
    if (segs.n > kFilterEdges) {   // the last segment that starts at or before the workgroup's first edge
      const int stride = (segs.n + 255) / 256;
      const int t1 = threadIdx.x * stride;
      const int c1 = __syncthreads_count(t1 < segs.n && segs.start[t1] <= jf);
      const int base = (c1 - 1) * stride;
      const int t2 = base + threadIdx.x;
      const int c2 = __syncthreads_count(threadIdx.x < stride && t2 < segs.n && segs.start[t2] <= jf);
      first = base + c2 - 1;
    }
    m_segs = min(kFilterEdges, segs.n - first);
    for (int i = threadIdx.x; i < m_segs; i += 256) {
      s_start[i] = segs.start[first + i];
      const int l = segs.list[first + i];
      s_pos0[i] = segs.pos0[first + i];
      const ListDesc L = lists[l];
      s_slots[i] = L.slots;
      // (a spatial list has no displaced pixels: the unconditional load below reads its slots instead)
      s_prev[i] = L.type != 0 ? L.prev_idx : reinterpret_cast<const int32_t*>(L.slots);
      s_type[i] = L.type;
      s_base_a[i] = L.base_a;
      s_base_b[i] = L.base_b;
      s_slot_base[i] = list_slot_base[l];
    }
    __syncthreads();
  }
  bool valid[kFilterPer];
  int a[kFilterPer], b[kFilterPer], l_type[kFilterPer];
  uint32_t gpos[kFilterPer];
  if (segs.start) {
    // ---- decode: every step is one batch of independent loads over the thread's edges ------------------------
    const uint32_t* slots_p[kFilterPer];
    const int32_t* prev_p[kFilterPer];
    int pos[kFilterPer], base_a[kFilterPer], base_b[kFilterPer];
#pragma unroll
    for (int e = 0; e < kFilterPer; ++e) {
      const int j = jf + e * 256 + threadIdx.x;
      valid[e] = j < n_b;
      const int jc = valid[e] ? j : n_b - 1;   // (an edge past the end repeats the last one, without effects)
      int lo = 0, hi = m_segs;   // the last loaded segment with start <= jc
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_start[mid] <= jc) lo = mid; else hi = mid;
      }
      pos[e] = s_pos0[lo] + (jc - s_start[lo]);
      slots_p[e] = s_slots[lo];
      prev_p[e] = s_prev[lo];
      l_type[e] = s_type[lo];
      base_a[e] = s_base_a[lo];
      base_b[e] = s_base_b[lo];
      gpos[e] = s_slot_base[lo] + (uint32_t)pos[e];
    }
    uint32_t slot[kFilterPer];
#pragma unroll
    for (int e = 0; e < kFilterPer; ++e) slot[e] = slots_p[e][pos[e]];
    uint32_t pix[kFilterPer];
    int pv[kFilterPer];
#pragma unroll
    for (int e = 0; e < kFilterPer; ++e) {
      // (selected with a mask: as a conditional the compiler branches, and waits for edge 0's
      // displaced pixel before it issues the load of edge 1's)
      const uint32_t tm = (uint32_t)-(int)(l_type[e] != 0);
      pix[e] = ((slot[e] / 9u) & tm) | ((slot[e] >> 2) & ~tm);
      pv[e] = prev_p[e][pix[e]];
    }
#pragma unroll
    for (int e = 0; e < kFilterPer; ++e) {   // DecodeEdge, without branches (a branch per edge would
      a[e] = base_a[e] + (int)pix[e];        // put a wait for its loads between the edges)
      const int kt = (int)(slot[e] - pix[e] * 9u);
      const int dy = kt / 3 - 1, dx = kt - (kt / 3) * 3 - 1;
      const int bt = base_b[e] + pv[e] + dy * P.W + dx;
      const int ks = (int)(slot[e] & 3u);   // 0 right, 1 bottom, 2 bottom-left, 3 bottom-right
      const int bs = a[e] + (ks != 0 ? P.W : 0) + (ks == 0 || ks == 3 ? 1 : 0) - (ks == 2 ? 1 : 0);
      const int tm = -(int)(l_type[e] != 0);
      b[e] = (bt & tm) | (bs & ~tm);
    }
  } else {
#pragma unroll
    for (int e = 0; e < kFilterPer; ++e) {
      const int j = jf + e * 256 + threadIdx.x;
      valid[e] = j < n_b;
      const int jc = valid[e] ? j : n_b - 1;
      int bk = bucket;
      int jb = j0 + jc;                              // index inside the bucket
      if (bucket_hi > bucket + 1) {                  // several buckets: the last b with prefix[b] <= position
        const int jg = bucket_prefix[bucket] + jb;
        int lo = bucket, hi = bucket_hi;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (bucket_prefix[mid] <= jg) lo = mid; else hi = mid;
        }
        bk = lo;
        jb = jg - bucket_prefix[bk];
      }
      const int32_t* base_row = bucket_base + (size_t)bk * (P.num_lists + 1);
      const int l = LocateList(base_row, P.num_lists, jb);
      const int pos = lists[l].offsets[bk] + (jb - base_row[l]);
      const ListDesc L = lists[l];
      l_type[e] = L.type;
      DecodeEdge(L, L.slots[pos], P.W, a[e], b[e]);
      gpos[e] = list_slot_base[l] + (uint32_t)pos;
    }
  }
  // ---- both roots of every edge (GetRegion with path compression), all chains of the thread together -------
  constexpr int kChains = 2 * kFilterPer;
  int x0[kChains], root[kChains];
#pragma unroll
  for (int e = 0; e < kFilterPer; ++e) {
    x0[2 * e] = a[e];
    x0[2 * e + 1] = b[e];
  }
#pragma unroll
  for (int c = 0; c < kChains; ++c) root[c] = x0[c];
  for (bool any = true; any;) {
    int p[kChains];
#pragma unroll
    for (int c = 0; c < kChains; ++c) p[c] = nodes.parent[root[c]];
    any = false;
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      if (p[c] != root[c]) {
        root[c] = p[c];
        any = true;
      }
    }
  }
  {
    // compression: every node on the path that does not point at the root yet (concurrent callers
    // may race on parent[] writes; every value ever written is an ancestor of the node, so any
    // interleaving leaves a valid forest with the same roots -- FindCompress, merge_common.h)
    int cur[kChains];
#pragma unroll
    for (int c = 0; c < kChains; ++c) cur[c] = x0[c];
    for (bool any = true; any;) {
      int nx[kChains];
#pragma unroll
      for (int c = 0; c < kChains; ++c) nx[c] = cur[c] != root[c] ? nodes.parent[cur[c]] : root[c];
      any = false;
#pragma unroll
      for (int c = 0; c < kChains; ++c) {
        if (cur[c] != root[c]) {
          if (nx[c] != root[c]) nodes.parent[cur[c]] = root[c];
          cur[c] = nx[c];
          any = any || (cur[c] != root[c]);
        }
      }
    }
  }
  int ti[kFilterPer], active[kFilterPer], settled[kFilterPer];
#pragma unroll
  for (int e = 0; e < kFilterPer; ++e) {
    ti[e] = active[e] = settled[e] = 0;
    const int ra = root[2 * e], rb = root[2 * e + 1];
    if (!valid[e] || ra == rb) continue;
    if (P.spatial_survivors && l_type[e] == 0 && !P.spatial_survivors[gpos[e]]) continue;   // the edge is gone
    bool inert = false;
    int f1 = 0, f2 = 0, s1 = 0, s2 = 0, c1 = -1, c2 = -1;
    if (inert_mode != 0) {
      f1 = nodes.flags[ra];
      f2 = nodes.flags[rb];
      if (f1 & kFlagFinalized) s1 = __float_as_int(nodes.desc_sz[ra].w);
      if (f2 & kFlagFinalized) s2 = __float_as_int(nodes.desc_sz[rb].w);
      const bool both_final_large = (f1 & kFlagFinalized) && (f2 & kFlagFinalized) &&
                                    (s1 >= P.min_region_size) && (s2 >= P.min_region_size);
      if (inert_mode == 1) {
        inert = both_final_large;
      } else {
        c1 = nodes.cons[ra];
        c2 = nodes.cons[rb];
        if (c1 >= 0 && c2 >= 0) {
          inert = (c1 != c2);            // different constraints: never merged
        } else {
          inert = both_final_large;      // at least one unconstrained
        }
        if (inert) {
          ti[e] = 1;
          // (plain stores: every writer of the kernel that matters writes this bit into what it read;
          // the one other writer -- kFlagHubBroken, with atomics -- only keeps a list free of duplicates)
          if (!(f1 & kFlagTentative)) nodes.flags[ra] = (uint8_t)(f1 | kFlagTentative);
          if (!(f2 & kFlagTentative)) nodes.flags[rb] = (uint8_t)(f2 | kFlagTentative);
        }
      }
    }
    if (inert) {
      kept_all[gpos[e]] = 1;
      settled[e] = 1;
    } else {
      active[e] = 1;
      // Hubs (device_graph.h: kFlagHub): an edge between a finalized region of at least the minimum
      // size and a region that has a mean and is unconstrained (or shares the hub's constraint: merged
      // subject to a test that is verified later) is decided by the latter alone -- it does not
      // tie the two into one component.  The record lists the other region first (the component key is
      // taken from the first root), the hub second.  An active edge that needs a hub's exact state
      // marks it broken and puts it on the exclusion list: a stage that uses such a region as a hub
      // elsewhere is redone with the region as an ordinary one.
      bool hub_edge = false;
      if (hubs) {
        constexpr int kNoHub = kFlagNoDesc | kFlagHubExcluded;
        const bool h1 = (f1 & kFlagFinalized) && !(f1 & kNoHub) && s1 >= P.min_region_size;
        const bool h2 = (f2 & kFlagFinalized) && !(f2 & kNoHub) && s2 >= P.min_region_size;
        if (h1 != h2) {
          const int h = h1 ? ra : rb, x = h1 ? rb : ra;
          const int fx = h1 ? f2 : f1, cx = h1 ? c2 : c1, fh = h1 ? f1 : f2;
          const int ch = h1 ? c1 : c2;
          if ((cx < 0 || cx == ch) && !(fx & kFlagNoDesc)) {
            hub_edge = true;
            root[2 * e] = x;
            root[2 * e + 1] = h;
            if (!nodes.hub8[h]) nodes.hub8[h] = 1;
            if (!(*num_hub & 1)) atomicOr(num_hub, 1);
            if (cx >= 0) {
              // Equal constraints: absorbed unless the means are further apart than the split threshold
              // (HubEdge, verified by k_hub_apply against the hub's mean of that moment).  With the
              // hub's mean of NOW the answer is almost always the same, and a test that fails is an
              // edge at which the stage is cut (bit 4 of the word the host waits for) before a worker
              // has run.
              const float4 dx = nodes.desc_sz[x], dh = nodes.desc_sz[h];
              const float u = dh.x - dx.x, v = dh.y - dx.y, w = dh.z - dx.z;
              if ((u * u + v * v + w * w) * (1.0f / 3.0f) > split_s) {
                HubViolationAt(hub_excl, 0, jf + e * 256 + (int)threadIdx.x);
                if (!(*num_hub & 16)) atomicOr(num_hub, 16);
              }
            }
          } else {
            HubViolationAt(hub_excl, 0, jf + e * 256 + (int)threadIdx.x);
            if (!(fh & kFlagHubBroken)) HubExclude(hub_excl, nodes.flags, h);
          }
        } else if (h1 && h2) {
          // (not inert: equal constraints -- the split test reads both means.)  With the smaller of
          // the two an ordinary region the edge is a hub edge of the larger one -- absorbed subject to
          // the test (HubEdge) -- and only the smaller one's neighbours are tied into a component.
          HubViolationAt(hub_excl, 0, jf + e * 256 + (int)threadIdx.x);
          if (s1 <= s2 && !(f1 & kFlagHubBroken)) HubExclude(hub_excl, nodes.flags, ra);
          if (s2 <= s1 && !(f2 & kFlagHubBroken)) HubExclude(hub_excl, nodes.flags, rb);
        }
      }
      if (!hub_edge) CcUnion(cc, ra, rb);
    }
  }
  // What the edge turned out to be is three bits per edge, one 64-bit word per wavefront and
  // class (a dense flag + code array per edge was 5 of the 17 bytes this kernel wrote per edge,
  // with 2 % of the edges active), plus the number of active edges per block of 256 edges for the
  // ordered compaction (k_compact_active).
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_blocks = (n_b + 255) >> 8;
  unsigned long long ma[kFilterPer], ms[kFilterPer];
#pragma unroll
  for (int e = 0; e < kFilterPer; ++e) {
    ma[e] = __ballot(active[e] != 0);
    ms[e] = __ballot(settled[e] != 0);
    const unsigned long long mt = __ballot(ti[e] != 0);
    const int vb = blockIdx.x * kFilterPer + e;   // the block of 256 edges
    if (lane == 0 && vb < n_blocks) {
      const size_t gw = (size_t)vb * 4 + w;
      M.active[gw] = ma[e];
      M.settled[gw] = ms[e];
      M.tentative[gw] = mt;
      wave_cnt[e][w] = (int)__popcll(ma[e]);
      wave_kept[e][w] = (int)__popcll(ma[e] | ms[e]);
      // (only "any" is read back; an atomic per wavefront on this one word would serialise the
      // kernel wherever constrained regions meet)
      if (mt != 0 && *num_ti == 0) *num_ti = 1;
    }
  }
  __syncthreads();
  // Roots and position only where something reads them back (the compaction: active; the
  // clearing of the marks: tentative; the rollback: every kept mark the filter set) -- most
  // edges of a stage are internal, so the records of a block's active / settled edges are
  // packed at the head of the block's 256 slots (FilterSlot: the readers recompute the rank
  // from the masks); scattered 4-byte stores cost a 64-byte sector each.
#pragma unroll
  for (int e = 0; e < kFilterPer; ++e) {
    const int vb = blockIdx.x * kFilterPer + e;
    if (active[e] | settled[e]) {
      int rank = (int)__popcll((ma[e] | ms[e]) & ((1ull << lane) - 1ull));
      for (int k = 0; k < w; ++k) rank += wave_kept[e][k];
      const int slot = vb * 256 + rank;
      e_ra[slot] = root[2 * e];
      e_rb[slot] = root[2 * e + 1];
      e_gpos[slot] = gpos[e];
    }
    if (threadIdx.x == 0 && vb < n_blocks) {
      M.block_cnt[vb] = wave_cnt[e][0] + wave_cnt[e][1] + wave_cnt[e][2] + wave_cnt[e][3];
    }
  }
}

// Where k_filter left the record (roots, kept position) of edge j of the stage: the rank of the
// edge among the active / settled edges of its workgroup.  Only valid for such an edge.
__device__ __forceinline__ int FilterSlot(const FilterMasks& M, int j) {
  const int block = j >> 8, w = (j >> 6) & 3, lane = j & 63;
  const size_t gw0 = (size_t)block * 4;
  int rank = (int)__popcll((M.active[gw0 + w] | M.settled[gw0 + w]) & ((1ull << lane) - 1ull));
  for (int k = 0; k < w; ++k) rank += (int)__popcll(M.active[gw0 + k] | M.settled[gw0 + k]);
  return block * 256 + rank;
}

// Clears the tentative marks of a stage: on the regions marked by the filter and on whatever
// region they have been merged into since.
__global__ __launch_bounds__(256) void k_clear_tentative(int n_b, FilterMasks M,
                                                          const int32_t* __restrict__ e_ra,
                                                          const int32_t* __restrict__ e_rb,
                                                          NodeArrays nodes) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_b || !((M.tentative[j >> 6] >> (j & 63)) & 1ull)) return;
  const int slot = FilterSlot(M, j);
  int r[2] = {e_ra[slot], e_rb[slot]};
  for (int k = 0; k < 2; ++k) {
    int x = r[k];
    for (;;) {
      const int f = nodes.flags[x];
      if (f & kFlagTentative) nodes.flags[x] = (uint8_t)(f & ~kFlagTentative);
      const int p = nodes.parent[x];
      if (p == x) break;
      x = p;
    }
  }
}

// Undo support for an optimistic stage: region states of every active edge's two regions.
__global__ __launch_bounds__(256) void k_backup_roots(int n, const int32_t* __restrict__ a_ra,
                                                       const int32_t* __restrict__ a_rb,
                                                       NodeArrays nodes, float4* __restrict__ bk_ds,
                                                       int32_t* __restrict__ bk_cons,
                                                       uint8_t* __restrict__ bk_flags,
                                                       unsigned long long* __restrict__ stats) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 8) stats[8 + i] = stats[i];   // the merge statistics are undone with the stage
  if (i >= n) return;
  const int ra = a_ra[i], rb = a_rb[i];
  bk_ds[2 * i] = nodes.desc_sz[ra];
  bk_cons[2 * i] = nodes.cons[ra];
  bk_flags[2 * i] = nodes.flags[ra];
  bk_ds[2 * i + 1] = nodes.desc_sz[rb];
  bk_cons[2 * i + 1] = nodes.cons[rb];
  bk_flags[2 * i + 1] = nodes.flags[rb];
}

__global__ __launch_bounds__(256) void k_restore_roots(int n, const int32_t* __restrict__ a_ra,
                                                        const int32_t* __restrict__ a_rb,
                                                        NodeArrays nodes,
                                                        const float4* __restrict__ bk_ds,
                                                        const int32_t* __restrict__ bk_cons,
                                                        const uint8_t* __restrict__ bk_flags,
                                                        unsigned long long* __restrict__ stats) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 8) stats[i] = stats[8 + i];
  if (i >= n) return;
  const int ra = a_ra[i], rb = a_rb[i];
  nodes.parent[ra] = ra;
  nodes.desc_sz[ra] = bk_ds[2 * i];
  nodes.cons[ra] = bk_cons[2 * i];
  nodes.flags[ra] = bk_flags[2 * i];
  nodes.parent[rb] = rb;
  nodes.desc_sz[rb] = bk_ds[2 * i + 1];
  nodes.cons[rb] = bk_cons[2 * i + 1];
  nodes.flags[rb] = bk_flags[2 * i + 1];
}

__global__ __launch_bounds__(256) void k_clear_kept(int n_b, FilterMasks M,
                                                     const uint32_t* __restrict__ e_gpos,
                                                     uint8_t* __restrict__ kept_all) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_b) return;
  // every edge whose kept mark the stage may have set
  if (((M.active[j >> 6] | M.settled[j >> 6]) >> (j & 63)) & 1ull) kept_all[e_gpos[FilterSlot(M, j)]] = 0;
}

// Ordered compaction of the active edges: block_off = exclusive scan of the per-workgroup counts
// the filter left; inside a workgroup the position follows from the wavefronts' masks.
__global__ __launch_bounds__(256) void k_compact_active(int n_b, FilterMasks M,
                                                         const int32_t* __restrict__ block_off,
                                                         const int32_t* __restrict__ e_ra,
                                                         const int32_t* __restrict__ e_rb,
                                                         const uint32_t* __restrict__ e_gpos,
                                                         int32_t* __restrict__ a_ra,
                                                         int32_t* __restrict__ a_rb,
                                                         uint32_t* __restrict__ a_gpos,
                                                         int32_t* __restrict__ num_active,
                                                         const int32_t* __restrict__ num_ti,
                                                         unsigned long long* __restrict__ mail, unsigned mail_seq,
                                                         int a_cap, const int32_t* __restrict__ num_hub) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t gw0 = (size_t)blockIdx.x * 4;
  const unsigned long long m = M.active[gw0 + w];
  int base = block_off[blockIdx.x];
  for (int k = 0; k < w; ++k) base += (int)__popcll(M.active[gw0 + k]);
  if ((m >> lane) & 1ull) {
    const int p = base + (int)__popcll(m & ((1ull << lane) - 1ull));
    // (more active edges than the arrays hold: the host sees the total, enlarges them and compacts again)
    if (p < a_cap) {
      const int slot = FilterSlot(M, j);
      a_ra[p] = e_ra[slot];
      a_rb[p] = e_rb[slot];
      a_gpos[p] = e_gpos[slot];
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
    const int total = base + (int)__popcll(m);
    *num_active = total;
    MailPost(mail, mail_seq, 0, total);
    MailPost(mail, mail_seq, 1, *num_ti);   // (k_filter's sum: complete, it is the kernel before)
    MailPost(mail, mail_seq, 2, *num_hub);
  }
}

__global__ __launch_bounds__(256) void k_component_ids(int n, const int32_t* __restrict__ a_ra,
                                                        int32_t* __restrict__ cc,
                                                        uint32_t* __restrict__ a_comp,
                                                        uint32_t* __restrict__ a_idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  a_comp[i] = (uint32_t)CcFind(cc, a_ra[i]);
  a_idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_reset_cc(int n, const int32_t* __restrict__ a_ra,
                                                   const int32_t* __restrict__ a_rb,
                                                   int32_t* __restrict__ cc,
                                                   const int32_t* __restrict__ violation,
                                                   unsigned long long* __restrict__ mail, unsigned mail_seq) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  // (the workers of the stage are done: this kernel follows the join of their streams)
  if (i == 0 && mail) MailPost(mail, mail_seq, 0, *violation);
  if (i >= n) return;
  const int ra = a_ra[i], rb = a_rb[i];
  cc[ra] = ra;
  cc[rb] = rb;
}

// Active edges in component order (contiguous input of the workers).
__global__ __launch_bounds__(256) void k_gather_sorted(int n, const uint32_t* __restrict__ s_idx,
                                                        const int32_t* __restrict__ a_ra,
                                                        const int32_t* __restrict__ a_rb,
                                                        const uint32_t* __restrict__ a_gpos,
                                                        int32_t* __restrict__ s_ra,
                                                        int32_t* __restrict__ s_rb,
                                                        uint32_t* __restrict__ s_gpos) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const uint32_t i = s_idx[p];
  s_ra[p] = a_ra[i];
  s_rb[p] = a_rb[i];
  s_gpos[p] = a_gpos[i];
}

// ------------------------------------------------------------------------------------------
// Worker A: one lane replays one small component.
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_merge_small(const int32_t* __restrict__ num_segs,
                                                      const int32_t* __restrict__ seg_off,
                                                      const int32_t* __restrict__ seg_cnt,
                                                      const int32_t* __restrict__ s_ra,
                                                      const int32_t* __restrict__ s_rb,
                                                      const uint32_t* __restrict__ s_gpos,
                                                      NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                      StageThr T, int optimistic,
                                                      int32_t* __restrict__ violation,
                                                      unsigned long long* __restrict__ stats,
                                                      int small_seg, int wave_max,
                                                      uint32_t* __restrict__ work_list,
                                                      int work_cap, int32_t* __restrict__ work_ctl,
                                                      int cls_min0, int32_t* __restrict__ hub_mark,
                                                      const uint32_t* __restrict__ s_seq,
                                                      int32_t* __restrict__ hub_excl) {
  const int seg = blockIdx.x * 256 + threadIdx.x;
  unsigned n_forced = 0, n_regular = 0, n_small = 0;
  const int cnt = seg < *num_segs ? seg_cnt[seg] : 0;
  if (work_list) {
    // the wave worker's: filed under its size class (the order inside a class is arbitrary --
    // components are independent); one reservation per wavefront and class
    const int cls = (cnt > small_seg && cnt < wave_max) ? (cnt >= cls_min0 ? 0 : (cnt >= kWaveClassMin1 ? 1 : 2)) : -1;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < kWaveClasses; ++c) {
      const unsigned long long m = __ballot(cls == c);
      if (!m) continue;
      int base = 0;
      if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&work_ctl[c], (int)__popcll(m));
      base = __shfl(base, (int)__builtin_ctzll(m));
      if (cls == c) work_list[(size_t)c * work_cap + base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)seg;
    }
  }
  if (seg < *num_segs) {
    if (cnt <= small_seg) {
      const int beg = seg_off[seg];
      for (int p = beg; p < beg + cnt; ++p) {
        // An optimistic stage must stay undoable from the backed-up region states alone, so it
        // does not compress paths.
        const int r1 = optimistic ? FindReadOnly(nodes.parent, s_ra[p])
                                  : FindCompress(nodes.parent, s_ra[p]);
        const int r2 = optimistic ? FindReadOnly(nodes.parent, s_rb[p])
                                  : FindCompress(nodes.parent, s_rb[p]);
        if (r1 == r2) continue;
        RState s1 = LoadState(nodes, r1);
        RState s2 = LoadState(nodes, r2);
        if (T.hubs) {
          if (nodes.hub8[r1]) s1.flags |= kFlagHub;
          if (nodes.hub8[r2]) s2.flags |= kFlagHub;
        }
        if (T.hubs && ((s1.flags | s2.flags) & kFlagHub)) {
          // an edge on a hub of the stage (merge_common.h: HubEdge): decided by the other region alone
          const bool hub1 = (s1.flags & kFlagHub) != 0, hub2 = (s2.flags & kFlagHub) != 0;
          const RState& x = hub1 ? s2 : s1;
          const RState& h = hub1 ? s1 : s2;
          const int act = (hub1 && hub2) ? HubHubEdge(s1.cons, s1.flags, s2.cons, s2.flags)
                                         : HubEdge(x, h.cons, h.flags, h.sz, T);
          if (act >= kHubViolation) {
            atomicOr(violation, act);
            HubViolationAt(hub_excl, 1, (int)s_seq[p]);
            if (hub1) HubExclude(hub_excl, nodes.flags, r1);
            if (hub2) HubExclude(hub_excl, nodes.flags, r2);
          } else if (act == kHubKeep) {
            kept_all[s_gpos[p]] = 1;
          } else {
            const int xr = hub1 ? r2 : r1, hr = hub1 ? r1 : r2;
            nodes.parent[xr] = hr;
            if (x.flags & kFlagTentative) AtomicOrFlags(nodes.flags, hr, kFlagTentative);
            hub_mark[s_seq[p]] = xr | (act == kHubAbsorbTest ? kHubTestBit : 0);
            if (act == kHubAbsorbTest) ++n_forced; else ++n_small;
          }
          continue;
        }
        const RState o1 = s1, o2 = s2;
        int stat;
        const int out = DecideEdge(s1, s2, T, stat);
        if (optimistic) {
          const bool v = (out == kOutKeep)     ? TentativeViolated(o1, o2, s1, s2)
                         : (out == kOutMerge1) ? TentativeViolated(o1, o2, s1, s1)
                                               : TentativeViolated(o1, o2, s2, s2);
          if (v) {
            atomicOr(violation, kVioCut);
            HubViolationAt(hub_excl, 1, (int)s_seq[p]);
          }
        }
        if (stat == 4 && T.rle) {
          atomicOr(violation, kVioCut);
          HubViolationAt(hub_excl, 1, (int)s_seq[p]);
        }
        n_forced += (stat == 1);
        n_regular += (stat == 2);
        n_small += (stat == 3);
        if (out == kOutKeep) {
          if (T.side) {   // (a side cluster of a tree replay does not end up as one region: the stage is cut here)
            atomicOr(violation, 2);
            HubViolationAt(hub_excl, 2, (int)s_gpos[p]);
          }
          kept_all[s_gpos[p]] = 1;
          StoreState(nodes, r1, s1);
          StoreState(nodes, r2, s2);
        } else if (out == kOutMerge1) {
          StoreState(nodes, r1, s1);
          nodes.parent[r2] = r1;
        } else {
          StoreState(nodes, r2, s2);
          nodes.parent[r1] = r2;
        }
      }
    }
  }
  // wave-level reduction of the statistics
  for (int off = 32; off > 0; off >>= 1) {
    n_forced += __shfl_down(n_forced, off);
    n_regular += __shfl_down(n_regular, off);
    n_small += __shfl_down(n_small, off);
  }
  if ((threadIdx.x & 63) == 0) {
    if (n_forced) atomicAdd(&stats[0], (unsigned long long)n_forced);
    if (n_regular) atomicAdd(&stats[1], (unsigned long long)n_regular);
    if (n_small) atomicAdd(&stats[2], (unsigned long long)n_small);
  }
}

// ------------------------------------------------------------------------------------------
// Runs of equal root pairs.
// ------------------------------------------------------------------------------------------
// Consecutive active edges (bucket order) with the same pair of roots are a *run*: the nine
// temporal edges of a pixel that hang on the same two regions, the edges along the common
// boundary of two regions.  They are consecutive in their component's sequence as well, so once
// the first edge of the run (the leader) has been replayed, the followers see exactly the state it
// left behind and repeat its outcome without changing anything:
//   leader internal or merged  -> followers internal;
//   leader kept (Case U: a finalized end and both ends large; Case D: different constraints)
//                              -> followers kept, same state.
// The one exception is a leader that is kept by a *constrained split* (Case S with d > split
// threshold): it drops a constraint, and its followers would be decided by Case U.  Such a stage
// is reported through the violation flag and replayed edge by edge (rollback, like a violated
// optimistic stage).  Workers therefore replay leaders only; k_resolve_followers copies the kept
// marks afterwards.
// (one fused scan: the leader flag of position p, its exclusive prefix = the leader's index, the
// leaders moved together, their number posted to the host -- scan_device.h)
struct LeaderValue {
  const int32_t* a_ra;
  const int32_t* a_rb;
  __device__ int operator()(int p) const {
    if (p == 0) return 1;
    const int ra = a_ra[p], rb = a_rb[p], qa = a_ra[p - 1], qb = a_rb[p - 1];
    return !((ra == qa && rb == qb) || (ra == qb && rb == qa));
  }
};
struct LeaderEmit {
  const int32_t* a_ra;
  const int32_t* a_rb;
  const uint32_t* a_gpos;
  int32_t* lead;
  int32_t* lpos;
  int32_t* l_ra;
  int32_t* l_rb;
  uint32_t* l_gpos;
  __device__ void operator()(int p, int is_lead, int q) const {
    lead[p] = is_lead;
    lpos[p] = q;
    if (is_lead) {
      l_ra[q] = a_ra[p];
      l_rb[q] = a_rb[p];
      l_gpos[q] = a_gpos[p];
    }
  }
};
struct LeaderFinish {
  int32_t* num_leaders;
  unsigned long long* mail;
  unsigned mail_seq;
  __device__ void operator()(int total) const {
    *num_leaders = total;
    MailPost(mail, mail_seq, 0, total);
  }
};

__global__ __launch_bounds__(256) void k_resolve_followers(int n, const int32_t* __restrict__ lead,
                                                            const int32_t* __restrict__ lpos,
                                                            const uint32_t* __restrict__ a_gpos,
                                                            const uint32_t* __restrict__ l_gpos,
                                                            uint8_t* __restrict__ kept_all) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n || lead[p]) return;
  // exclusive scan of the leader flags: lpos[p] - 1 is the run's leader
  if (kept_all[l_gpos[lpos[p] - 1]]) kept_all[a_gpos[p]] = 1;
}

// ------------------------------------------------------------------------------------------
// Hubs: what the workers logged is applied in sequence order (device_graph.h: kFlagHub).
// ------------------------------------------------------------------------------------------
// hub_mark[seq] = region absorbed by a hub at the work edge with sequence number seq (or -1): the
// absorbed regions in sequence order (one fused scan), with the hub of each -- its parent.
struct HubMarkValue {
  const int32_t* mark;
  __device__ int operator()(int i) const { return mark[i] >= 0; }
};
struct HubMarkEmit {
  const int32_t* mark;
  const int32_t* parent;
  uint32_t* seq_list;
  uint32_t* key_list;
  __device__ void operator()(int i, int is_set, int q) const {
    if (is_set) {
      seq_list[q] = (uint32_t)i;   // (the absorbed region is mark[i], with kHubTestBit)
      key_list[q] = (uint32_t)parent[mark[i] & ~kHubTestBit];
    }
  }
};
struct HubMarkFinish {
  int32_t* count;
  unsigned long long* mail;
  unsigned mail_seq;
  __device__ void operator()(int total) const {
    *count = total;
    MailPost(mail, mail_seq, 0, total);
  }
};

// A hub that absorbed hundreds of regions is a chain of dependent f32 operations -- one thread walks it at
// ~19 ns per region, every load behind the one before; on a noisy input a few such hubs per stage were
// 80 ms per chunk (and a thread that prefetches four regions at a time still waits three dependent loads
// per four).  From kHubWaveMin regions on a hub gets a wavefront: 64 regions at a time, all lanes
// load (sequence number -> region -> state) and compute what does not depend on the mean -- the hub's
// size before every absorption is a prefix sum, so are the weights ca, cb and the products ca * x --,
// then lane ch replays channel ch through the 64 steps  h = ca * x + cb * h  (coefficients from LDS,
// the mean before every step left behind there), and all lanes check the split tests of their
// regions against those means.  Same operations, same order, same roundings as the thread form.
constexpr int kHubWaveMin = 32;
__global__ __launch_bounds__(64) void k_hub_apply_wave(const int32_t* __restrict__ num_runs,
                                                        const int32_t* __restrict__ run_off,
                                                        const int32_t* __restrict__ run_cnt,
                                                        const uint32_t* __restrict__ hub_sorted,
                                                        const uint32_t* __restrict__ seq_sorted,
                                                        const int32_t* __restrict__ mark, NodeArrays nodes,
                                                        float split_s, int32_t* __restrict__ violation,
                                                        int32_t* __restrict__ hub_excl) {
  __shared__ float coef[64][4];     // ca * x (three channels), cb
  __shared__ float before[64][4];   // the hub's mean before step l
  const int lane = threadIdx.x;
  const int n_runs = *num_runs;
  for (int r = blockIdx.x; r < n_runs; r += gridDim.x) {
    const int cnt = run_cnt[r];
    if (cnt < kHubWaveMin) continue;
    const int off = run_off[r];
    const int hub = (int)hub_sorted[off];
    const float4 hs = nodes.desc_sz[hub];
    float h = lane == 0 ? hs.x : lane == 1 ? hs.y : hs.z;   // (lanes 0..2: one channel each)
    int S = __float_as_int(hs.w);
    int bad_seq = -1;
    for (int base = 0; base < cnt; base += 64) {
      const int i = base + lane;
      const bool valid = i < cnt;
      uint32_t sq = 0, xv = 0;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        sq = seq_sorted[off + i];
        xv = (uint32_t)mark[sq];
        o = nodes.desc_sz[xv & ~(uint32_t)kHubTestBit];
      }
      const int osz = valid ? __float_as_int(o.w) : 0;
      const int incl = WaveInclusiveSum(osz);
      const int Sb = S + incl - osz;   // the hub's size before this absorption
      float4 q = make_float4(0.f, 0.f, 0.f, 1.f);   // (no region: the mean stays)
      if (valid) {
        const float denom = 1.0f / (float)(osz + Sb);
        const float ca = (float)osz * denom;
        const float cb = (float)Sb * denom;
        q = make_float4(ca * o.x, ca * o.y, ca * o.z, cb);
      }
      coef[lane][0] = q.x;
      coef[lane][1] = q.y;
      coef[lane][2] = q.z;
      coef[lane][3] = q.w;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < 3) {
#pragma unroll 8
        for (int l = 0; l < 64; ++l) {
          before[l][lane] = h;
          h = coef[l][lane] + coef[l][3] * h;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      bool fails = false;
      if (valid && (xv & (uint32_t)kHubTestBit)) {
        const float x = before[lane][0] - o.x, y = before[lane][1] - o.y, z = before[lane][2] - o.z;
        fails = (x * x + y * y + z * z) * (1.0f / 3.0f) > split_s;
      }
      const unsigned long long fm = __ballot(fails);
      if (fm && bad_seq < 0) bad_seq = __builtin_amdgcn_readlane((int)sq, (int)__builtin_ctzll(fm));
      S += __builtin_amdgcn_readlane(incl, 63);
      __builtin_amdgcn_wave_barrier();
    }
    const float r0 = ReadLaneF(h, 0), r1 = ReadLaneF(h, 1), r2 = ReadLaneF(h, 2);
    if (lane == 0) {
      if (bad_seq >= 0) {
        atomicOr(violation, kHubVioSplit);
        HubViolationAt(hub_excl, 1, bad_seq);
        HubExclude(hub_excl, nodes.flags, hub);
      }
      nodes.desc_sz[hub] = make_float4(r0, r1, r2, __int_as_float(S));
    }
  }
}

// One thread per hub: MergeStates (merge_common.h) with the hub as the survivor, once per absorbed
// region, in sequence order (the list is sorted by hub, stably).  The loads do not depend on the
// recurrence: four regions are fetched ahead of the arithmetic.
__global__ __launch_bounds__(64) void k_hub_apply(const int32_t* __restrict__ num_runs,
                                                   const int32_t* __restrict__ run_off,
                                                   const int32_t* __restrict__ run_cnt,
                                                   const uint32_t* __restrict__ hub_sorted,
                                                   const uint32_t* __restrict__ seq_sorted,
                                                   const int32_t* __restrict__ mark,
                                                   NodeArrays nodes, float split_s,
                                                   int32_t* __restrict__ violation,
                                                   int32_t* __restrict__ hub_excl) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= *num_runs) return;
  const int off = run_off[r], cnt = run_cnt[r];
  if (cnt >= kHubWaveMin) return;   // (k_hub_apply_wave)
  const int hub = (int)hub_sorted[off];
  const float4 hs = nodes.desc_sz[hub];
  float h0 = hs.x, h1 = hs.y, h2 = hs.z;
  int S = __float_as_int(hs.w);
  int bad_seq = -1;   // the first absorption (in sequence order) that fails its test
  auto absorb = [&](const float4& o, bool test, uint32_t seq) {
    if (test) {   // equal constraints: DecideEdge keeps the edge (and drops constraints) beyond the split threshold
      const float x = h0 - o.x, y = h1 - o.y, z = h2 - o.z;
      if ((x * x + y * y + z * z) * (1.0f / 3.0f) > split_s && bad_seq < 0) bad_seq = (int)seq;
    }
    const int osz = __float_as_int(o.w);
    const float denom = 1.0f / (float)(osz + S);
    const float ca = (float)osz * denom;
    const float cb = (float)S * denom;
    h0 = ca * o.x + cb * h0;
    h1 = ca * o.y + cb * h1;
    h2 = ca * o.z + cb * h2;
    S += osz;
  };
  int k = 0;
  for (; k + 4 <= cnt; k += 4) {
    float4 o[4];
    uint32_t xv[4], sq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sq[q] = seq_sorted[off + k + q];
#pragma unroll
    for (int q = 0; q < 4; ++q) xv[q] = (uint32_t)mark[sq[q]];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = nodes.desc_sz[xv[q] & ~(uint32_t)kHubTestBit];
#pragma unroll
    for (int q = 0; q < 4; ++q) absorb(o[q], (xv[q] & (uint32_t)kHubTestBit) != 0, sq[q]);
  }
  for (; k < cnt; ++k) {
    const uint32_t sq = seq_sorted[off + k];
    const uint32_t xv = (uint32_t)mark[sq];
    absorb(nodes.desc_sz[xv & ~(uint32_t)kHubTestBit], (xv & (uint32_t)kHubTestBit) != 0, sq);
  }
  if (bad_seq >= 0) {
    atomicOr(violation, kHubVioSplit);
    HubViolationAt(hub_excl, 1, bad_seq);
    HubExclude(hub_excl, nodes.flags, hub);
  }
  nodes.desc_sz[hub] = make_float4(h0, h1, h2, __int_as_float(S));
}

// The hub marks of a stage go with the stage: off both roots of every active edge.
__global__ __launch_bounds__(256) void k_hub_clear(int n, const int32_t* __restrict__ a_ra,
                                                    const int32_t* __restrict__ a_rb, NodeArrays nodes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ra = a_ra[i], rb = a_rb[i];
  if (nodes.hub8[ra]) nodes.hub8[ra] = 0;
  if (nodes.hub8[rb]) nodes.hub8[rb] = 0;
  // (several edges clear the same region: each writes what the others write)
  const int fa = nodes.flags[ra], fb = nodes.flags[rb];
  if (fa & kFlagHubBroken) nodes.flags[ra] = (uint8_t)(fa & ~(int)kFlagHubBroken);
  if (fb & kFlagHubBroken) nodes.flags[rb] = (uint8_t)(fb & ~(int)kFlagHubBroken);
}

// The regions on the exclusion list are no hubs for the rest of the chunk (device_graph.h:
// kFlagHubExcluded); launched again after every undone stage -- the restored flags are older.
__global__ __launch_bounds__(256) void k_hub_exclude(const int32_t* __restrict__ excl, NodeArrays nodes) {
  const int n = min(excl[0], kHubExclCap);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = excl[4 + i];
    if (!(nodes.flags[r] & kFlagHubExcluded)) AtomicOrFlags(nodes.flags, r, kFlagHubExcluded);
  }
}

// A region that the filter put on the exclusion list (an edge needs its exact state) while other
// edges made it a hub: known before any worker runs -- bit 1 of the word the host waits for; bit 2: the
// list overflowed, which regions are broken is not known; bit 3: the list is not empty (regions carry
// kFlagHubBroken, to be cleared with the stage's other marks).
__global__ __launch_bounds__(256) void k_hub_check(const int32_t* __restrict__ excl, NodeArrays nodes,
                                                    int32_t* __restrict__ num_hub) {
  const int cnt = excl[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // (bits 8 ..: how many edges the filter recorded -- what a cut of the stage would cost)
    const int bits = (cnt > 0 ? (cnt > kHubExclCap ? 12 : 8) : 0) | (min(excl[1], 0xfffff) << 8);
    if (bits) atomicOr(num_hub, bits);
  }
  const int n = min(cnt, kHubExclCap);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    if (nodes.hub8[excl[4 + i]]) {
      atomicOr(num_hub, 2);
      break;
    }
  }
}

// ... and the marks are taken off again when the stage is through (the list starts empty).
__global__ __launch_bounds__(256) void k_hub_unexclude(int32_t* __restrict__ excl, NodeArrays nodes) {
  const int n = min(excl[0], kHubExclCap);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    AtomicAndFlags(nodes.flags, excl[4 + i], 0xff & ~(kFlagHubExcluded | kFlagHubBroken));
  }
}
__global__ void k_hub_excl_reset(int32_t* __restrict__ excl) { excl[0] = excl[1] = excl[2] = excl[3] = 0; }
// The work edges that broke a rule, as kept positions (what the host can locate).
__global__ void k_hub_cut_gpos(int32_t* __restrict__ list, const uint32_t* __restrict__ work_gpos) {
  const int n = min(list[2], kHubCutCap);
  int32_t* at = list + 4 + kHubExclCap + kHubCutCap;
  for (int i = threadIdx.x; i < n; i += blockDim.x) at[i] = (int32_t)work_gpos[at[i]];
}
void ResetHubExclusions(MergeScratch& S, NodeArrays nodes, hipStream_t s) {
  if (!S.hub_excl) return;
  hipLaunchKernelGGL(k_hub_unexclude, dim3(16), dim3(256), 0, s, S.hub_excl, nodes);
  hipLaunchKernelGGL(k_hub_excl_reset, dim3(1), dim3(1), 0, s, S.hub_excl);
  VSG_HIP(hipGetLastError());
}

// One edge of a bucket on its own -- the edge at which a stage is cut: both roots, DecideEdge, the states
// and the kept mark, exactly what the lane worker does for an edge without any of the stage's assumptions.
__global__ void k_plain_edge(int bucket, int jb, const ListDesc* __restrict__ lists,
                             const int32_t* __restrict__ bucket_base, const uint32_t* __restrict__ list_slot_base,
                             uint8_t* __restrict__ kept_all, NodeArrays nodes, MergeParams P, StageThr T,
                             unsigned long long* __restrict__ stats) {
  const int32_t* base_row = bucket_base + (size_t)bucket * (P.num_lists + 1);
  const int l = LocateList(base_row, P.num_lists, jb);
  const int pos = lists[l].offsets[bucket] + (jb - base_row[l]);
  const ListDesc L = lists[l];
  int a, b;
  DecodeEdge(L, L.slots[pos], P.W, a, b);
  const uint32_t gpos = list_slot_base[l] + (uint32_t)pos;
  if (P.spatial_survivors && L.type == 0 && !P.spatial_survivors[gpos]) return;   // the edge is gone
  int r1 = a, r2 = b;
  for (int p; (p = nodes.parent[r1]) != r1;) r1 = p;
  for (int p; (p = nodes.parent[r2]) != r2;) r2 = p;
  if (r1 == r2) return;
  RState s1 = LoadState(nodes, r1), s2 = LoadState(nodes, r2);
  int stat;
  const int out = DecideEdge(s1, s2, T, stat);
  if (stat >= 1 && stat <= 3) atomicAdd(&stats[stat - 1], 1ull);
  if (out == kOutKeep) {
    kept_all[gpos] = 1;
    StoreState(nodes, r1, s1);
    StoreState(nodes, r2, s2);
  } else if (out == kOutMerge1) {
    StoreState(nodes, r1, s1);
    nodes.parent[r2] = r1;
  } else {
    StoreState(nodes, r2, s2);
    nodes.parent[r1] = r2;
  }
}

// Position, in the edge sequence (bucket by bucket, inside a bucket list by list), of the edge with kept
// position `gpos`, which lies in one of the buckets [b_lo, b_hi): from the host copies of the bucket tables.
static long long SequencePosition(const MergeScratch& S, const MergeParams& P, int b_lo, int b_hi, uint32_t gpos) {
  const int L = P.num_lists;
  int lo = 0, hi = L;   // the last list that starts at or before gpos
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (S.list_slot_base_host[mid] <= gpos) lo = mid; else hi = mid;
  }
  const int l = lo;
  const long long p = (long long)gpos - (long long)S.list_slot_base_host[l];
  const int32_t* off = S.list_off_host + (size_t)l * (kNumBuckets + 2);   // where the list's edges of bucket b start
  int b0 = b_lo, b1 = b_hi;   // the last bucket of the range whose part of the list starts at or before p
  while (b1 - b0 > 1) {
    const int mid = (b0 + b1) >> 1;
    if (off[mid] <= p) b0 = mid; else b1 = mid;
  }
  const int32_t* row = S.bucket_base_host + (size_t)b0 * (L + 1);
  return (long long)S.bucket_prefix_host[b0] + row[l] + (p - (long long)off[b0]);
}
// The bucket of [b_lo, b_hi) that holds sequence position g (the last one that starts at or before it).
static int BucketOfPosition(const MergeScratch& S, int b_lo, int b_hi, long long g) {
  int b0 = b_lo, b1 = b_hi;
  while (b1 - b0 > 1) {
    const int mid = (b0 + b1) >> 1;
    if ((long long)S.bucket_prefix_host[mid] <= g) b0 = mid; else b1 = mid;
  }
  return b0;
}

// ------------------------------------------------------------------------------------------
// Host driver of one stage.
// ------------------------------------------------------------------------------------------
static inline unsigned Blocks(int n) { return (unsigned)((n + 255) / 256); }

__global__ void k_max_segment(int max_segs, const int32_t* __restrict__ num_segs,
                              const int32_t* __restrict__ seg_cnt, int below, int32_t* __restrict__ out);

void RunBucketStage(int bucket, int j0, int n_b, const ListDesc* lists, const int32_t* bucket_base,
                    const uint32_t* list_slot_base, uint8_t* kept_all, NodeArrays nodes,
                    const MergeParams& P, int inert_mode, MergeScratch& S, hipStream_t s,
                    StageInfo* info) {
  if (info) {
    info->replayed = 0;
    info->components = 0;
  }
  if (n_b <= 0) return;
  const int bucket_hi = S.group_hi > bucket ? S.group_hi : bucket + 1;
  int32_t* d_num_ti = TakeStageScalars(S, 8);   // fresh counters per stage: nothing to clear
  int32_t* d_violation = d_num_ti + 1;
  int32_t* d_num_hub = d_num_ti + 2;
  int32_t* d_hub_count = d_num_ti + 3;
  int32_t* d_hub_runs = d_num_ti + 4;
  // Hubs (device_graph.h: kFlagHub) only in stages that the tree replay cannot take (it works on the
  // region states directly) and never in the conservative replay of a violated stage.
  const bool spine_possible = S.spine_min > 0 && !S.spine_off && bucket < *S.spine_limit_bucket &&
                              !(bucket < 2 && S.spine_low_skip[bucket]);
  const bool try_hubs = S.hubs && !S.hubs_off && inert_mode != 0 && !spine_possible && S.wide_min <= 0;
  // (the list of a stage starts empty: what the filter of an earlier stage put there without
  // consequences -- a broken region that no edge used as a hub -- is not excluded from this one)
  // (... and the edges that broke a rule are those of this run of the stage -- recorded with or without
  // hubs.  Cleared only when a stage has left something there: the host knows -- the filter's entries
  // show in the word it waits for, a worker's come with a violation.)
  if (S.hub_excl && S.hub_list_dirty) {
    if (S.hub_attempt == 0) {
      VSG_HIP(hipMemsetAsync(S.hub_excl, 0, 4 * sizeof(int32_t), s));
      S.hub_list_dirty = 0;
    } else {
      VSG_HIP(hipMemsetAsync(S.hub_excl + 1, 0, 2 * sizeof(int32_t), s));   // (the exclusion list stays)
    }
  }
  if (S.hub_attempt == 0 && S.hub_split_depth == 0) S.hub_splits_left = S.hub_max_splits;
  int32_t* d_num_leaders = S.num_active + 5;
  // The stage's non-empty (bucket, list) segments, from the host copy of the bucket table.
  FilterSegs segs = {nullptr, nullptr, nullptr, 0};
  if (S.bucket_base_host && S.list_off_host && S.seg_dev) {
    const int L = P.num_lists;
    std::vector<int32_t>& h = *S.seg_host;
    h.clear();
    const int64_t g0 = (int64_t)S.bucket_prefix_host[bucket] + j0, g1 = g0 + n_b;
    std::vector<int32_t> st, li, po;
    for (int b = bucket; b < bucket_hi; ++b) {
      const int32_t* row = S.bucket_base_host + (size_t)b * (L + 1);
      const int64_t bstart = S.bucket_prefix_host[b];
      if (bstart + row[L] <= g0) continue;
      if (bstart >= g1) break;
      for (int l = 0; l < L; ++l) {
        const int cnt = row[l + 1] - row[l];
        if (cnt == 0) continue;
        const int64_t s0 = bstart + row[l], s1 = s0 + cnt;
        if (s1 <= g0 || s0 >= g1) continue;
        const int64_t first = std::max(s0, g0);
        st.push_back((int32_t)(first - g0));
        li.push_back(l);
        po.push_back(S.list_off_host[(size_t)l * (kNumBuckets + 2) + b] + (int32_t)(first - s0));
      }
    }
    const int nseg = (int)st.size();
    if (nseg > 0 && nseg <= 65536 && (size_t)3 * nseg <= S.seg_cap) {
      h.reserve((size_t)3 * nseg);
      h.insert(h.end(), st.begin(), st.end());
      h.insert(h.end(), li.begin(), li.end());
      h.insert(h.end(), po.begin(), po.end());
      VSG_HIP(hipMemcpyAsync(S.seg_dev, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
      segs.start = S.seg_dev;
      segs.list = S.seg_dev + nseg;
      segs.pos0 = S.seg_dev + 2 * (size_t)nseg;
      segs.n = nseg;
    }
  }
  const float split_s = ((float)bucket * P.inv_scale < P.force_merge_weight) ? P.s_lt_02 : P.s_le_015;   // (StageThr::split_s)
  const int ef0 = NextEvent(S);
  if (ef0 >= 0) VSG_HIP(hipEventRecord((*S.ev_pool)[ef0], s));
  hipLaunchKernelGGL(k_filter, dim3((unsigned)((n_b + kFilterEdges - 1) / kFilterEdges)), dim3(256), 0, s, bucket, bucket_hi, j0, n_b, lists,
                     bucket_base, S.bucket_prefix, list_slot_base, kept_all, nodes, P, inert_mode, S.cc, S.e_ra, S.e_rb, S.e_gpos,
                     S.masks, d_num_ti, segs, try_hubs ? 1 : 0, d_num_hub, S.hub_excl, split_s);
  const int ef1 = NextEvent(S);
  if (ef1 >= 0) {
    VSG_HIP(hipEventRecord((*S.ev_pool)[ef1], s));
    S.ev_filter->emplace_back(ef0, ef1);
  }
  if (try_hubs) hipLaunchKernelGGL(k_hub_check, dim3(16), dim3(256), 0, s, S.hub_excl, nodes, d_num_hub);
  ExclusiveSum(S.scan, S.masks.block_cnt, S.block_off, (int)Blocks(n_b), s);
  int h[3] = {0, 0, 0};   // num_active, num_ti, hubs (k_hub_check)
  for (int attempt = 0;; ++attempt) {
    const MailSlot m_active = NextMail(*S.mail);
    hipLaunchKernelGGL(k_compact_active, dim3(Blocks(n_b)), dim3(256), 0, s, n_b, S.masks, S.block_off,
                       S.e_ra, S.e_rb, S.e_gpos, S.a_ra, S.a_rb, S.a_gpos, S.num_active, d_num_ti, m_active.dev,
                       m_active.seq, S.active_cap, d_num_hub);
    VSG_HIP(hipGetLastError());
    MailWait(m_active, 3, h, s);
    if (h[0] <= S.active_cap) break;
    // the arrays that hold active edges are too small for this stage: enlarge them, compact again
    VSG_REQUIRE(attempt == 0 && S.grow_active && S.grow_active(h[0]) && h[0] <= S.active_cap, -4,
                "stage scratch: cannot hold the active edges");
  }
  const int n_active = h[0];
  const int n_ti = h[1];
  const bool hubs_used = try_hubs && (h[2] & 1) != 0;
  if (h[2] & (8 | 16)) S.hub_list_dirty = 1;   // (the filter put regions / edges on the list)
  auto clear_hub_marks = [&]() {
    if (try_hubs && (h[2] & 9) != 0 && n_active > 0) {
      hipLaunchKernelGGL(k_hub_clear, dim3(Blocks(n_active)), dim3(256), 0, s, n_active, S.a_ra, S.a_rb, nodes);
    }
  };
  auto clear_marks = [&]() {
    if (n_ti > 0) {
      hipLaunchKernelGGL(k_clear_tentative, dim3(Blocks(n_b)), dim3(256), 0, s, n_b, S.masks, S.e_ra,
                         S.e_rb, nodes);
    }
  };
  if (n_active == 0) {
    clear_marks();
    return;
  }
  // A stage whose hubs broke a rule.  (Called with the stage undone.)  The stage is exact in front of
  // the earliest edge at which that happened, and that edge on its own is an ordinary edge: the stage
  // is cut there -- [start, edge) with hubs, the edge alone, (edge, end) with hubs again; whatever
  // made the hub unusable (a constraint it was about to inherit, a region of its own constraint
  // that fails the split test, its like) is behind it then.  After kHubMaxSplits cuts, in a stage over
  // several buckets, or without the host tables that locate an edge, the stage is run again with
  // the regions that broke a rule as ordinary regions (they are on the exclusion list), and after
  // kHubMaxAttempts of those -- or when the list has overflowed -- without hubs.
  // (a stage whose edges the host tables can locate -- one bucket or a group of buckets --, with cuts left)
  // ... and a stage for which the cuts cost less than its edge-by-edge replay could: a cut is two more
  // stages (0.2-0.3 ms of launches and waits each), the replay of a stage is one -- whose largest component,
  // on one wavefront at ~0.2 us per edge, is at worst all of its n_b edges (a replay settles nothing in
  // advance).  At 1920x1080 a stage has 0.3-8 M edges and tens of violations: cut.  A 320x240 stream of
  // uniformly random frames has stages of a few thousand edges with as many violations each: cutting
  // them made its merge 1.5 s per chunk instead of 0.4.  (VSG_CUT_MIN_WORK: a floor on the replayed edges
  // as well -- measured at 2 K / 8 K / 32 K on the long 1080p noise stream: all worse than 0.)
  auto can_cut = [&](int work, long long violations) {
    return work >= S.hub_cut_min_work && (long long)n_b > 1500ll * (violations + 1) && S.hub_splits_left > 0 &&
           S.bucket_base_host && S.bucket_prefix_host && S.list_off_host && S.list_slot_base_host;
  };
  auto retry_without_broken_hubs = [&](int violated, int work, bool list_complete, const uint32_t* work_gpos) -> bool {
    for (int q = 0; q < 6; ++q) S.hub_reasons[q] += (violated >> (2 + q)) & 1;
    ++S.hub_retries;
    if (info) ++info->hub_retries;
    // The edges at which rules were broken, as positions inside the stage, in order.
    std::vector<int> cuts;
    const long long g_stage = (long long)S.bucket_prefix_host[bucket] + j0;   // the stage's first edge in the sequence
    int head[4] = {0, 0, 0, 0};
    if (can_cut(work, 0)) {
      VSG_HIP(hipMemcpyAsync(head, S.hub_excl, sizeof(head), hipMemcpyDeviceToHost, s));
      VSG_HIP(hipStreamSynchronize(s));
    }
    if (can_cut(work, (long long)head[1] + (work_gpos ? head[2] + head[3] : 0))) {
      int at[3 * kHubCutCap];
      if (work_gpos) hipLaunchKernelGGL(k_hub_cut_gpos, dim3(1), dim3(64), 0, s, S.hub_excl, work_gpos);
      VSG_HIP(hipMemcpyAsync(at, S.hub_excl + 4 + kHubExclCap, sizeof(at), hipMemcpyDeviceToHost, s));
      VSG_HIP(hipStreamSynchronize(s));
      for (int i = 0; i < std::min(head[1], kHubCutCap); ++i) cuts.push_back(at[i]);
      if (work_gpos) {
        for (int i = 0; i < std::min(head[2], kHubCutCap); ++i) {
          cuts.push_back((int)(SequencePosition(S, P, bucket, bucket_hi, (uint32_t)at[kHubCutCap + i]) - g_stage));
        }
        for (int i = 0; i < std::min(head[3], kHubCutCap); ++i) {   // (kept positions as they are)
          cuts.push_back((int)(SequencePosition(S, P, bucket, bucket_hi, (uint32_t)at[2 * kHubCutCap + i]) - g_stage));
        }
      }
      std::sort(cuts.begin(), cuts.end());
      cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
      while (!cuts.empty() && cuts.back() >= n_b) cuts.pop_back();
      while (!cuts.empty() && cuts.front() < 0) cuts.erase(cuts.begin());
      // (What the filter found: only the earliest.  A hub about to inherit a constraint has as many such
      // edges as constrained neighbours, and all but the first are ordinary hub edges once it has; the
      // rest of the stage finds what is left at the cost of a filter pass.)
      if (!work_gpos && cuts.size() > 1) cuts.resize(1);
      if ((int)cuts.size() > S.hub_splits_left) cuts.resize((size_t)S.hub_splits_left);
    }
    if (getenv("VSG_DEBUG_STAGES")) {
      int cnt = 0;
      VSG_HIP(hipMemcpyAsync(&cnt, S.hub_excl, sizeof(int), hipMemcpyDeviceToHost, s));
      VSG_HIP(hipStreamSynchronize(s));
      std::fprintf(stderr, "[vsg]   hub rule violated (mask %x) in b=%d j0=%d n=%d work %d, attempt %d, %d regions on the "
                   "list, %zu cuts (first at %d, %d left)\n", violated, bucket, j0, n_b, work, S.hub_attempt, cnt,
                   cuts.size(), cuts.empty() ? -1 : cuts[0], S.hub_splits_left);
    }
    if (!cuts.empty()) {
      // (the positions after the first are hints: what the stage does behind a cut may differ from this
      // run -- every part checks itself and is cut again if need be)
      S.hub_splits_left -= (int)cuts.size();
      S.hub_splits += (long long)cuts.size();
      int replayed = 0;
      auto part = [&](int off, int n, bool plain) {
        if (n <= 0) return;
        StageInfo sub;
        if (info) sub.want_components = info->want_components;
        ++S.hub_split_depth;
        if (plain) ++S.hubs_off;
        const int b_at = BucketOfPosition(S, bucket, bucket_hi, g_stage + off);   // (a group: the part starts in a later bucket)
        RunBucketStage(b_at, (int)(g_stage + off - S.bucket_prefix_host[b_at]), n, lists, bucket_base, list_slot_base,
                       kept_all, nodes, P, inert_mode, S, s, info ? &sub : nullptr);
        if (plain) --S.hubs_off;
        --S.hub_split_depth;
        replayed += sub.replayed;
        if (info) {
          info->hub_stages += sub.hub_stages;
          info->hub_absorbed += sub.hub_absorbed;
          info->hub_retries += sub.hub_retries;
          info->components += sub.components;
          info->max_wave_segment = std::max(info->max_wave_segment, sub.max_wave_segment);
        }
      };
      StageThr plain;   // (the thresholds of the bucket, none of a stage's assumptions)
      plain.pass_s = ((float)bucket * P.inv_scale < P.force_merge_weight) ? P.s_lt_02 : P.s_lt_005;
      plain.split_s = split_s;
      plain.min_size = P.min_region_size;
      plain.rle = 0;
      plain.side = 0;
      plain.hubs = 0;
      int prev = 0;
      for (const int c : cuts) {
        part(prev, c - prev, false);
        const int b_c = BucketOfPosition(S, bucket, bucket_hi, g_stage + c);
        hipLaunchKernelGGL(k_plain_edge, dim3(1), dim3(1), 0, s, b_c, (int)(g_stage + c - S.bucket_prefix_host[b_c]), lists,
                           bucket_base, list_slot_base, kept_all, nodes, P, plain, S.stats);
        ++replayed;
        prev = c + 1;
      }
      part(prev, n_b - prev, false);
      if (info) info->replayed = replayed;
      return true;
    }
    if (violated & (kVioCut | 2)) return false;   // (not a matter of hubs: the caller replays the stage its own way)
    int& depth = (list_complete && S.hub_attempt + 1 < kHubMaxAttempts) ? S.hub_attempt : S.hubs_off;
    if (&depth == &S.hub_attempt) {
      hipLaunchKernelGGL(k_hub_exclude, dim3(16), dim3(256), 0, s, S.hub_excl, nodes);
    }
    ++depth;
    RunBucketStage(bucket, j0, n_b, lists, bucket_base, list_slot_base, kept_all, nodes, P, inert_mode, S, s, info);
    --depth;
    // (the exclusions go with the stage: a hub that was about to inherit a constraint, or to meet
    // its like, is an ordinary hub again once that edge is behind it)
    if (S.hub_attempt == 0 && S.hubs_off == 0) ResetHubExclusions(S, nodes, s);
    return true;
  };
  if (hubs_used && ((h[2] & 6) != 0 || ((h[2] & 16) != 0 && can_cut(n_active, (long long)(h[2] >> 8))))) {
    // The filter itself found a hub whose exact state an edge needs: nothing has been replayed yet.
    hipLaunchKernelGGL(k_reset_cc, dim3(Blocks(n_active)), dim3(256), 0, s, n_active, S.a_ra, S.a_rb, S.cc,
                       d_violation, nullptr, 0u);
    hipLaunchKernelGGL(k_clear_kept, dim3(Blocks(n_b)), dim3(256), 0, s, n_b, S.masks, S.e_gpos, kept_all);
    clear_marks();
    clear_hub_marks();
    VSG_HIP(hipGetLastError());
    if (retry_without_broken_hubs(kHubVioBroken, n_active, (h[2] & 4) == 0, nullptr)) return;
    VSG_REQUIRE(false, -4, "hub stage: no way to rerun it");
  }
  if (hubs_used && info) ++info->hub_stages;

  // Run leaders (inert_mode 0 is the conservative replay of a violated stage: every edge).
  // In a graph with constraints a run-compressed stage has to stay undoable (constrained split).
  const bool rle = S.use_rle && inert_mode != 0 && n_active >= 64;
  const int32_t* w_ra = S.a_ra;     // what the workers replay
  const int32_t* w_rb = S.a_rb;
  const uint32_t* w_gpos = S.a_gpos;
  int n_work = n_active;
  int32_t* lead = S.e_active;       // free after the compaction
  if (rle) {
    const MailSlot m_lead = NextMail(*S.mail);
    FusedScan(S.scan, LeaderValue{S.a_ra, S.a_rb},
              LeaderEmit{S.a_ra, S.a_rb, S.a_gpos, lead, S.lead_pos, S.l_ra, S.l_rb, S.l_gpos},
              LeaderFinish{d_num_leaders, m_lead.dev, m_lead.seq}, n_active, s);
    MailWait(m_lead, 1, &n_work, s);
    w_ra = S.l_ra;
    w_rb = S.l_rb;
    w_gpos = S.l_gpos;
  }

  hipLaunchKernelGGL(k_component_ids, dim3(Blocks(n_work)), dim3(256), 0, s, n_work, w_ra, S.cc,
                     S.a_comp, S.a_idx);
  SortPairsU32(S.cub_temp, S.cub_temp_bytes, S.a_comp, S.s_comp, S.a_idx, S.s_idx, n_work, S.node_key_bits, s);
  // the components = the runs of equal keys: first position and length of every run
  RunsOfSortedKeys(S.scan, S.s_comp, n_work, S.seg_off, S.seg_cnt, S.num_segs, s);

  // The large components are replayed along their Kruskal tree (merge_spine.hip): the choice is
  // made on the host from the list of components above the threshold.
  SpineInput spine_in;
  int spine_thr = 0x7fffffff;
  if (S.spine_min > 0 && !S.spine_off && bucket < *S.spine_limit_bucket &&
      !(bucket < 2 && S.spine_low_skip[bucket]) && inert_mode != 0 && n_work >= S.spine_min) {
    long long wanted = 0;
    spine_thr = SelectLargeSegments(n_work, S.num_segs, S.seg_off, S.seg_cnt, S.spine_min,
                                    S.spine_max_edges, S, s, &spine_in, &wanted);
    if (wanted > S.spine_max_edges && S.grow_spine_pool && S.grow_spine_pool(wanted)) {
      // the pool was too small for this input (it is sized for a typical bucket, not for the
      // largest one a video can produce): it has grown, choose again
      spine_thr = SelectLargeSegments(n_work, S.num_segs, S.seg_off, S.seg_cnt, S.spine_min,
                                      S.spine_max_edges, S, s, &spine_in, nullptr);
    }
  }
  const bool spine = !spine_in.segs.empty();
  // A stage that settles edges tentatively, replays run leaders only or relies on the spine
  // structure has to stay undoable.
  const bool optimistic = ((inert_mode == 2) && (n_ti > 0 || rle)) || spine || hubs_used;
  if (optimistic) {
    hipLaunchKernelGGL(k_backup_roots, dim3(Blocks(n_work)), dim3(256), 0, s, n_work, w_ra, w_rb,
                       nodes, S.bk_ds, S.bk_cons, S.bk_flags, S.stats);
  }

  const float weight = (float)bucket * P.inv_scale;
  const bool force = weight < P.force_merge_weight;
  StageThr T;
  T.pass_s = force ? P.s_lt_02 : P.s_lt_005;
  T.split_s = force ? P.s_lt_02 : P.s_le_015;
  T.min_size = P.min_region_size;
  T.rle = rle ? 1 : 0;
  T.side = 0;
  T.relax = S.chain_relax;
  T.hubs = hubs_used ? 1 : 0;
  // Replayed edges in component order; the scratch arrays of the earlier steps are free by now.
  int32_t* s_ra = reinterpret_cast<int32_t*>(S.a_comp);
  int32_t* s_rb = reinterpret_cast<int32_t*>(S.a_idx);
  uint32_t* s_gpos = reinterpret_cast<uint32_t*>(S.e_apos);
  hipLaunchKernelGGL(k_gather_sorted, dim3(Blocks(n_work)), dim3(256), 0, s, n_work, S.s_idx, w_ra,
                     w_rb, w_gpos, s_ra, s_rb, s_gpos);
  const int small_seg = S.small_seg;
  auto WaveGrid = [small_seg](int n) {
    const int g = n / (small_seg + 1);
    return g < 1 ? 1 : (g > 8192 ? 8192 : g);
  };
  const int wave_grid = WaveGrid(n_work);
  WorkerArgs wa;
  wa.num_segs = S.num_segs;
  wa.seg_off = S.seg_off;
  wa.seg_cnt = S.seg_cnt;
  wa.s_ra = s_ra;
  wa.s_rb = s_rb;
  wa.s_gpos = s_gpos;
  wa.nodes = nodes;
  wa.kept_all = kept_all;
  wa.T = T;
  wa.optimistic = optimistic ? 1 : 0;
  wa.violation = d_violation;
  wa.stats = S.stats;
  wa.small_seg = small_seg;
  wa.wave_min = small_seg;
  wa.wave_max = spine_thr;
  // work list of the wave worker (seg_key: scratch of n_work entries), the control words follow the
  // stage's scalars
  wa.work_cap = n_work / (small_seg + 1) + 1;
  wa.work_list = (size_t)kWaveClasses * wa.work_cap <= (size_t)n_work ? S.seg_key : nullptr;
  wa.work_ctl = wa.work_list ? TakeZeroed(S, 2 * kWaveClasses) : nullptr;
  wa.wide_min = S.wide_min;
  wa.wide_waves = S.wide_waves;
  wa.s_seq = S.s_idx;          // (number of a work edge in the stage's sequence: where a rule was broken)
  wa.hub_excl = S.hub_excl;
  if (hubs_used) {
    // (the sorted component keys are free once the runs are known: one mark per work edge)
    wa.hub_mark = reinterpret_cast<int32_t*>(S.s_comp);
    VSG_HIP(hipMemsetAsync(wa.hub_mark, 0xFF, (size_t)n_work * sizeof(int32_t), s));
  }
  auto general_workers = [&](const WorkerArgs& w, int small_threads, int grid, hipStream_t s) {
    if (w.wave_min == w.small_seg) {
      hipLaunchKernelGGL(k_merge_small, dim3(Blocks(small_threads)), dim3(256), 0, s, w.num_segs,
                         w.seg_off, w.seg_cnt, w.s_ra, w.s_rb, w.s_gpos, w.nodes, w.kept_all, w.T,
                         w.optimistic, w.violation, w.stats, w.small_seg, w.wave_max, w.work_list, w.work_cap,
                         w.work_ctl,
                         // (the largest size class starts where the wide worker does, if that is lower)
                         (w.wide_min > 0 && w.wide_min < kWaveClassMin0) ? std::max(w.wide_min, w.small_seg + 1)
                                                                        : kWaveClassMin0,
                         w.hub_mark, w.s_seq, w.hub_excl);
    }
    // the wave worker is timed on the stream it runs on
    const int ew0 = NextEvent(S);
    if (ew0 >= 0) VSG_HIP(hipEventRecord((*S.ev_pool)[ew0], s));
    // The large components of the work list go to the wide worker (several wavefronts per component,
    // merge_wide.hip); the wave worker leaves them alone.  (small_threads = the edges of the launch)
    const bool wide = w.work_list != nullptr && w.wide_min > 0 && w.wide_min < w.wave_max &&
                      small_threads >= w.wide_min && w.wave_min == w.small_seg;
    if (wide) {
      WorkerArgs narrow = w;
      narrow.wave_max = w.wide_min;
      LaunchMergeWave(grid, narrow, S.wave_debug != 0, S.wave_dbg, s);
      const int wide_grid = std::min(small_threads / w.wide_min + 1, 1024);
      LaunchMergeWide(wide_grid, w, w.wide_waves, s);
    } else {
      LaunchMergeWave(grid, w, S.wave_debug != 0, S.wave_dbg, s);
    }
    const int ew1 = NextEvent(S);
    if (ew1 >= 0) {
      VSG_HIP(hipEventRecord((*S.ev_pool)[ew1], s));
      S.ev_wave->emplace_back(ew0, ew1);
    }
  };
  if (!spine) {
    general_workers(wa, n_work, wave_grid, s);
  } else {
    // The ordinary components on the second stream, beside the tree replay of the large ones
    // (disjoint regions; the statistics and the violation flag are atomics / idempotent stores).
    VSG_HIP(hipEventRecord(S.aux_fork, s));
    VSG_HIP(hipStreamWaitEvent(S.aux_stream, S.aux_fork, 0));
    general_workers(wa, n_work, wave_grid, S.aux_stream);
    VSG_HIP(hipEventRecord(S.aux_join, S.aux_stream));
    const bool done = RunSpineComponents(spine_in, wa, S, s, [&](const WorkerArgs& w, int n, hipStream_t st) {
      general_workers(w, n, WaveGrid(n), st);
    }, 0, 0);
    if (!done) {   // no room in the scratch pool: the wave worker replays them
      WorkerArgs w3 = wa;
      w3.wave_min = spine_thr - 1;
      w3.wave_max = 0x7fffffff;
      w3.work_list = nullptr;   // (the list belongs to the launch on the second stream)
      general_workers(w3, n_work, wave_grid, s);
    }
    VSG_HIP(hipStreamWaitEvent(s, S.aux_join, 0));
  }
  if (hubs_used) {
    // What the hubs absorbed, in sequence order, grouped by hub (stable sort), one thread per hub --
    // before the violation word is read: an absorption that was subject to the split test may fail it.
    uint32_t* seq_list = S.a_comp;                                 // (the workers' inputs are free now)
    uint32_t* key_list = S.a_idx;
    uint32_t* key_sorted = reinterpret_cast<uint32_t*>(S.e_apos);
    uint32_t* seq_sorted = S.s_idx;                                // (the marks themselves stay where they are: s_comp)
    int32_t* run_off = reinterpret_cast<int32_t*>(S.seg_key);      // (the work list)
    int32_t* run_cnt = S.seg_off;                                  // (the segments are through)
    const MailSlot m_hub = NextMail(*S.mail);
    FusedScan(S.scan, HubMarkValue{wa.hub_mark}, HubMarkEmit{wa.hub_mark, nodes.parent, seq_list, key_list},
              HubMarkFinish{d_hub_count, m_hub.dev, m_hub.seq}, n_work, s);
    int n_abs = 0;
    MailWait(m_hub, 1, &n_abs, s);
    if (n_abs > 0) {
      SortPairsU32(S.cub_temp, S.cub_temp_bytes, key_list, key_sorted, seq_list, seq_sorted, n_abs, S.node_key_bits, s);
      RunsOfSortedKeys(S.scan, key_sorted, n_abs, run_off, run_cnt, d_hub_runs, s);
      hipLaunchKernelGGL(k_hub_apply, dim3((unsigned)((n_abs + 63) / 64)), dim3(64), 0, s, d_hub_runs, run_off,
                         run_cnt, key_sorted, seq_sorted, wa.hub_mark, nodes, T.split_s, d_violation, S.hub_excl);
      if (n_abs >= kHubWaveMin) {   // (the hubs that absorbed many: a wavefront each)
        hipLaunchKernelGGL(k_hub_apply_wave, dim3((unsigned)std::min(n_abs / kHubWaveMin, 2048)), dim3(64), 0, s,
                           d_hub_runs, run_off, run_cnt, key_sorted, seq_sorted, wa.hub_mark, nodes, T.split_s,
                           d_violation, S.hub_excl);
      }
    }
    if (info) info->hub_absorbed += n_abs;
  }
  MailSlot m_vio = {nullptr, nullptr, 0};
  if (optimistic) m_vio = NextMail(*S.mail);
  hipLaunchKernelGGL(k_reset_cc, dim3(Blocks(n_work)), dim3(256), 0, s, n_work, w_ra, w_rb, S.cc, d_violation,
                     m_vio.dev, m_vio.seq);
  VSG_HIP(hipGetLastError());
  if (optimistic) {
    int violated = 0;
    MailWait(m_vio, 1, &violated, s);
    ++*S.optimistic_stages;
    if (violated || S.force_rollback) {
      // Undo the stage and replay it without any tentatively settled edge, every edge on its own.
      ++*S.rollbacks;
      S.hub_list_dirty = 1;   // (whatever the workers recorded)
      hipLaunchKernelGGL(k_restore_roots, dim3(Blocks(n_work)), dim3(256), 0, s, n_work, w_ra, w_rb,
                         nodes, S.bk_ds, S.bk_cons, S.bk_flags, S.stats);
      hipLaunchKernelGGL(k_clear_kept, dim3(Blocks(n_b)), dim3(256), 0, s, n_b, S.masks, S.e_gpos, kept_all);
      clear_marks();
      clear_hub_marks();   // (the restored flags carry the marks the filter had set)
      VSG_HIP(hipGetLastError());
      // (A kept edge in a side cluster of the tree replay, at a known position: the stage is cut there and
      // the tree replay runs again on the parts -- not the whole stage on the ordinary workers.)
      // (Only in the first two buckets, where the replay is the rule: from the third on a failure moves the
      // bucket limit below, and cutting instead kept the replay failing stage after stage -- +-40 noise 74 -> 65
      // frames/s.)
      if (spine && violated == 2 && bucket < 2 && !S.force_rollback && retry_without_broken_hubs(2, n_work, true, w_gpos)) {
        return;
      }
      if (spine && violated == 2 && !S.force_rollback) {
        // Only the tree replay's assumption failed (an edge of a large component was kept): the
        // same stage again with the ordinary workers.  From the third bucket on that is the rule
        // rather than the exception, so the later buckets (of this and the next chunks) skip it.
        if (bucket >= 2 && bucket < *S.spine_limit_bucket) *S.spine_limit_bucket = bucket;
        if (bucket < 2) S.spine_low_failed[bucket] = 1;   // the first two buckets: see SegmentLists
        const int off = S.spine_off;
        S.spine_off = 1;
        RunBucketStage(bucket, j0, n_b, lists, bucket_base, list_slot_base, kept_all, nodes, P, inert_mode, S,
                       s, info);
        S.spine_off = off;
        return;
      }
      if ((violated & (kHubViolationMask | kVioCut)) != 0 && (violated & ~(kHubViolationMask | kVioCut)) == 0 &&
          !spine && !S.force_rollback) {
        // Only a hub rule failed (an edge met a hub that it would have changed, a split test on a hub
        // did not pass), or edges changed what the filter had assumed for the edges behind them.
        if (retry_without_broken_hubs(violated, n_work, true, w_gpos)) return;
      }
      if (bucket_hi > bucket + 1) {
        // A group of buckets: its two halves (each optimistic again, split further if it fails
        // again) -- the conservative replay of everything the group holds would chain it into one
        // huge component, and bucket by bucket would cost a stage per bucket.
        const int group_hi = S.group_hi;
        const int64_t g0 = (int64_t)S.bucket_prefix_host[bucket] + j0, g1 = g0 + n_b;
        const int mid = bucket + (bucket_hi - bucket) / 2;
        const int halves[2][2] = {{bucket, mid}, {mid, bucket_hi}};
        int replayed = 0;
        for (const auto& h : halves) {
          const int64_t lo = std::max<int64_t>(g0, S.bucket_prefix_host[h[0]]);
          const int64_t hi = std::min<int64_t>(g1, S.bucket_prefix_host[h[1]]);
          if (hi <= lo) continue;
          S.group_hi = h[1];
          StageInfo sub;
          RunBucketStage(h[0], (int)(lo - S.bucket_prefix_host[h[0]]), (int)(hi - lo), lists, bucket_base,
                         list_slot_base, kept_all, nodes, P, inert_mode, S, s, info ? &sub : nullptr);
          replayed += sub.replayed;
        }
        S.group_hi = group_hi;
        if (info) info->replayed = replayed;
        return;
      }
      RunBucketStage(bucket, j0, n_b, lists, bucket_base, list_slot_base, kept_all, nodes, P, 0, S,
                     s, info);
      return;
    }
  }
  if (info) {   // how the stage decomposed (the caller sizes the next windows / groups with it)
    info->replayed = n_work;
    // (a component above the window threshold needs that many replayed edges: small stages are
    // not worth the synchronisation)
    if (info->want_components && (info->want_components > 1 || n_work > 16384)) {
      int32_t* d_max = TakeZeroed(S, 1);
      hipLaunchKernelGGL(k_max_segment, dim3(Blocks(n_work)), dim3(256), 0, s, n_work, S.num_segs, S.seg_cnt,
                         spine ? spine_thr : 0x7fffffff, d_max);
      const MailSlot m_info = NextMail(*S.mail);
      LaunchMailPost(m_info, S.num_segs, d_max, nullptr, nullptr, s);
      int v[2] = {0, 0};
      MailWait(m_info, 2, v, s);
      info->components = v[0];
      info->max_wave_segment = v[1];
    }
  }
  if (rle && n_work < n_active) {
    hipLaunchKernelGGL(k_resolve_followers, dim3(Blocks(n_active)), dim3(256), 0, s, n_active, lead,
                       S.lead_pos, S.a_gpos, S.l_gpos, kept_all);
  }
  clear_marks();
  clear_hub_marks();
  VSG_HIP(hipGetLastError());
}

// seg_cnt[r] = seg_off[r + 1] - seg_off[r], the last run ends at n.
__global__ __launch_bounds__(256) void k_run_counts(int n, const int32_t* __restrict__ num_runs,
                                                     const int32_t* __restrict__ seg_off,
                                                     int32_t* __restrict__ seg_cnt) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int m = *num_runs;
  if (r >= m) return;
  seg_cnt[r] = (r + 1 < m ? seg_off[r + 1] : n) - seg_off[r];
}

void RunsOfSortedKeys(const ScanScratch& sc, const uint32_t* keys, int n, int32_t* seg_off, int32_t* seg_cnt,
                      int32_t* num_runs, hipStream_t s) {
  if (n <= 0) return;
  FusedScan(sc, RunHeadValue{keys}, RunHeadEmit{seg_off}, RunHeadFinish{num_runs}, n, s);
  // (the number of runs is only known on the device: one thread per position, most return at once)
  hipLaunchKernelGGL(k_run_counts, dim3(Blocks(n)), dim3(256), 0, s, n, num_runs, seg_off, seg_cnt);
  VSG_HIP(hipGetLastError());
}

// Largest component (in replayed edges) below `below`.
__global__ __launch_bounds__(256) void k_max_segment(int max_segs, const int32_t* __restrict__ num_segs,
                                                      const int32_t* __restrict__ seg_cnt, int below,
                                                      int32_t* __restrict__ out) {
  const int seg = blockIdx.x * 256 + threadIdx.x;
  int v = 0;
  if (seg < max_segs && seg < *num_segs) {
    v = seg_cnt[seg];
    if (v >= below) v = 0;
  }
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
  if ((threadIdx.x & 63) == 0 && v > *out) atomicMax(out, v);   // (a stale read only costs an atomic)
}

__global__ __launch_bounds__(256) void k_keep_virtual_bucket(const ListDesc* __restrict__ lists,
                                                              int num_lists) {
  // grid.y = list
  const int l = blockIdx.y;
  if (l >= num_lists) return;
  const ListDesc L = lists[l];
  if (!L.offsets) return;
  const int beg = L.offsets[kNumBuckets], end = L.offsets[kNumBuckets + 1];
  for (int p = beg + blockIdx.x * 256 + threadIdx.x; p < end; p += gridDim.x * 256) L.kept[p] = 1;
}

void LaunchKeepVirtualBucket(const ListDesc* lists, int num_lists, hipStream_t s) {
  hipLaunchKernelGGL(k_keep_virtual_bucket, dim3(64, num_lists), dim3(256), 0, s, lists,
                     num_lists);
  VSG_HIP(hipGetLastError());
}

}  // namespace vsg
