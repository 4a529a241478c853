// merge_common.h -- device helpers shared by the merge kernels (merge_stage.hip and the worker
// merge_wave.hip): union-find, the exact edge semantics on
// plain values, wave-level primitives, and the launchers of the workers.
#ifndef VSG_MERGE_COMMON_H_
#define VSG_MERGE_COMMON_H_

#include <functional>
#include <vector>

#include "device_graph.h"

namespace vsg {

constexpr int kSmallSegment = 24;   // default of MergeScratch::small_seg: components with more
                                    // replayed edges go to a wavefront
constexpr int kTabDirty = 0x100;    // region table entry changed since it was loaded

// ------------------------------------------------------------------------------------------
// Union-find helpers.
// ------------------------------------------------------------------------------------------
// Find with full path compression (GetRegion, segmentation_graph.h:651-669).  Concurrent callers
// may race on parent[] writes; every value ever written is an ancestor of the node, so any
// interleaving leaves a valid forest with the same roots.
__device__ __forceinline__ int FindCompress(int32_t* __restrict__ parent, int x) {
  int root = x;
  int p = parent[root];
  while (p != root) {
    root = p;
    p = parent[root];
  }
  int cur = x;
  while (cur != root) {
    const int next = parent[cur];
    if (next != root) parent[cur] = root;
    cur = next;
  }
  return root;
}

__device__ __forceinline__ int FindReadOnly(const int32_t* __restrict__ parent, int x) {
  int p = parent[x];
  while (p != x) {
    x = p;
    p = parent[x];
  }
  return x;
}

// Scratch component structure (min-id hooking with atomicCAS, as in ECL-CC).
__device__ __forceinline__ int CcFind(int32_t* cc, int x) {
  int p = cc[x];
  while (p != x) {
    const int gp = cc[p];
    if (gp != p) cc[x] = gp;   // path halving
    x = p;
    p = cc[x];
  }
  return x;
}

__device__ __forceinline__ void CcUnion(int32_t* cc, int a, int b) {
  a = CcFind(cc, a);
  b = CcFind(cc, b);
  while (a != b) {
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }
    // a > b: hook a under b if a is still a root.
    const int old = atomicCAS(&cc[a], a, b);
    if (old == a) return;
    a = CcFind(cc, old);
    b = CcFind(cc, b);
  }
}

// Value i of a mailbox slot (device_graph.h).
__device__ __forceinline__ void MailPost(unsigned long long* slot, unsigned seq, int i, int value) {
  __hip_atomic_store(slot + i, ((unsigned long long)seq << 32) | (unsigned)value, __ATOMIC_RELEASE,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------
// Edge decoding.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void DecodeEdge(const ListDesc& L, uint32_t slot, int W, int& a, int& b) {
  if (L.type == 0) {
    const uint32_t pix = slot >> 2;
    const int k = (int)(slot & 3u);
    a = L.base_a + (int)pix;
    const int off = (k == 0) ? 1 : (k == 1) ? W : (k == 2) ? (W - 1) : (W + 1);
    b = a + off;
  } else {
    const uint32_t pix = slot / 9u;
    const int k = (int)(slot - pix * 9u);
    const int dy = k / 3 - 1, dx = k - (k / 3) * 3 - 1;
    a = L.base_a + (int)pix;
    b = L.base_b + L.prev_idx[pix] + dy * W + dx;
  }
}

// bucket_base row for one bucket: base[l] = #edges of this bucket in lists < l, base[L] = total.
__device__ __forceinline__ int LocateList(const int32_t* __restrict__ base, int num_lists, int j) {
  int lo = 0, hi = num_lists;   // largest l with base[l] <= j
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (base[mid] <= j) lo = mid; else hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------
// Exact edge semantics on plain values (shared by the lane and the wave worker).
// ------------------------------------------------------------------------------------------
struct RState {
  float d0, d1, d2;
  int sz;
  int cons;
  int flags;
};

enum : int { kOutSkip = 0, kOutKeep = 1, kOutMerge1 = 2, kOutMerge2 = 3 };
// kOutMerge1: region 1 survives (s1 holds the merged state); kOutMerge2: region 2 survives.

// ColorMeanDescriptorTraits::DescriptorDistance (pixel_distance.h:479-493) is
//   dist = sqrt((dx^2+dy^2+dz^2) * (1/3));  return (w < force_w && dist < 0.2) ? 0 : dist
// and is only ever compared with a threshold.  sqrtf is correctly rounded and monotone, so every
// comparison is rewritten on s = (dx^2+dy^2+dz^2) * (1/3) against a float threshold that the host
// derives with the same correctly rounded sqrtf (dense_graph.cpp: SquaredThresholds):
//   regular merge test  d < 0.05f   <=>  s <= pass_s
//   constrained split   d > 0.15f   <=>  s >  split_s
// (with the force-merge rule of the stage's edge weight folded in).
__device__ __forceinline__ float SquaredDistance(const RState& a, const RState& b) {
  const float x = a.d0 - b.d0, y = a.d1 - b.d1, z = a.d2 - b.d2;
  return (x * x + y * y + z * z) * (1.0f / 3.0f);
}

// MergeRegions (segmentation_graph.h:671-701) + MergeDescriptor (pixel_distance.h:495-505).
// Returns kOutMerge1 / kOutMerge2; the survivor's RState receives the merged values.
__device__ __forceinline__ int MergeStates(RState& s1, RState& s2) {
  const bool first_wins = s1.sz > s2.sz;       // ties keep rep_2
  RState& m = first_wins ? s1 : s2;
  const RState& o = first_wins ? s2 : s1;
  if (!((m.flags | o.flags) & kFlagNoDesc)) {
    const float denom = 1.0f / (float)(o.sz + m.sz);
    const float a = (float)o.sz * denom;
    const float b = (float)m.sz * denom;
    m.d0 = a * o.d0 + b * m.d0;
    m.d1 = a * o.d1 + b * m.d1;
    m.d2 = a * o.d2 + b * m.d2;
  }
  m.sz += o.sz;
  m.cons = max(s1.cons, s2.cons);
  m.flags |= (o.flags & kFlagTentative);   // tentatively settled edges of `o` now hang on `m`
  return first_wins ? kOutMerge1 : kOutMerge2;
}

// One edge of SegmentGraph (segmentation_graph.h:374-440).  s1/s2 are updated in place (flags,
// constraints, merged state).  stat: 0 none, 1 forced, 2 regular, 3 small, 4 constrained split
// (the edge is kept and at least one constraint is dropped).
struct StageThr {
  float pass_s;    // regular test passes  <=> s <= pass_s
  float split_s;   // constrained split    <=> s >  split_s
  int min_size;
  int rle;         // the stage replays run leaders only (see k_mark_leaders): a constrained split
                   // invalidates the run's followers and has to be reported as a violation
  int side;        // the segments are side clusters of a spine (merge_spine.hip): a kept edge means
                   // the cluster does not end up as one region and is reported as a violation
  int relax = 1;   // a certainly kept lane may join the chain whatever marks its partner carries
  int hubs = 0;    // regions with kFlagHub are hubs of this stage (HubEdge below)
};

__device__ __forceinline__ int DecideEdge(RState& s1, RState& s2, const StageThr& T, int& stat) {
  stat = 0;
  if (s1.cons < 0 || s2.cons < 0) {
    if (!((s1.flags | s2.flags) & kFlagFinalized)) {
      if (SquaredDistance(s1, s2) <= T.pass_s) {   // d < MergeDistanceThreshold
        stat = 2;
        return MergeStates(s1, s2);
      }
      s1.flags |= kFlagFinalized;
      s2.flags |= kFlagFinalized;
    }
    // at least one finalized here
    if (s1.sz < T.min_size || s2.sz < T.min_size) {
      stat = 3;
      return MergeStates(s1, s2);
    }
    return kOutKeep;
  } else if (s1.cons == s2.cons) {
    if (SquaredDistance(s1, s2) > T.split_s) {     // d > SplitDistanceThreshold
      if ((double)s1.sz < (double)s2.sz * 0.3) {
        s1.cons = -1;
      } else if ((double)s2.sz < (double)s1.sz * 0.3) {
        s2.cons = -1;
      } else {
        s1.cons = -1;
        s2.cons = -1;
      }
      stat = 4;
      return kOutKeep;
    }
    stat = 1;
    return MergeStates(s1, s2);
  }
  return kOutKeep;
}

// Partner of a chain edge (merge_wave.hip, merge_spine.hip): a region the larger (hot) region can
// absorb without DecideEdge's bookkeeping -- no mark to hand on, no missing descriptor.  It may be
// finalized: then (unconstrained case) the edge is never tested, whatever the hot region's flag --
// the small one of the two is absorbed, two large ones are kept and nothing changes.  That is the
// rule on noisy and on low-contrast inputs: a giant region whose neighbours have all failed a
// test before.
// (A hub of the stage is no partner: its edges are decided by their other end.  The other hub marks
// -- broken, excluded -- say nothing about the region's state.)
__device__ __forceinline__ bool PlainPartner(int flags) {
  return (flags & (kFlagNoDesc | kFlagTentative | kFlagHub)) == 0;
}

// DecideEdge keeps the edge and changes neither state: different constraints, or (unconstrained
// rule) one of the two finalized -- no test -- and both at least of minimum size.
__device__ __forceinline__ bool NoopPair(const RState& s1, const RState& s2, const StageThr& T) {
  if (s1.cons >= 0 && s2.cons >= 0) return s1.cons != s2.cons;
  return ((s1.flags | s2.flags) & kFlagFinalized) && s1.sz >= T.min_size && s2.sz >= T.min_size;
}

// ------------------------------------------------------------------------------------------
// Hub edges (device_graph.h: kFlagHub).  x: the region at the other end of an edge to the hub h.
// What DecideEdge would do follows from x alone while h is a finalized region of at least the
// minimum size whose constraint does not change during the stage:
//   different constraints (both >= 0)             -> kept, nothing changes (kHubKeep)
//   x unconstrained, x.sz >= min size              -> kept, nothing changes (no test: h is finalized)
//   x unconstrained, x.sz < min size, x has a mean -> h absorbs x ("small" merge; kHubAbsorb): the
//        hub's constraint stays (max(h.cons, -1)), the absorbed state is logged, the hub's own
//        state is not touched here
//   equal constraints >= 0, x smaller than h       -> h absorbs x ("forced" merge) provided the split
//        test passes when the log is applied (kHubAbsorbTest)
//   anything else (x constrained and h not: the hub would inherit the constraint; x without a mean;
//        a broken hub; a marked x whose constraint would change) -> the stage cannot use hubs
//        (kHubViolation: bit 2 of the stage's violation word)
// ------------------------------------------------------------------------------------------
enum : int { kHubKeep = 0, kHubAbsorb = 1, kHubAbsorbTest = 3, kHubViolation = 4 };
// (a violation is kHubViolation or one of the bits above it: the reason, all of them in
// kHubViolationMask of the stage's violation word -- bit 0: optimistic stage, bit 1: tree replay)
constexpr int kHubViolationMask = 0xfc;
constexpr int kHubVioBroken = 4, kHubVioInherit = 8, kHubVioShape = 16, kHubVioMarked = 32, kHubVioPair = 64,
              kHubVioSplit = 128;
// Bit 8 of the violation word: an edge changed what the filter had assumed for the edges BEHIND it -- the
// constraint of a region with tentatively settled edges, the outcome of a run's leader (a constrained
// split) -- so the stage is exact in front of that edge (recorded with HubViolationAt) and can be cut
// there like a stage whose hub broke a rule, instead of being replayed edge by edge as a whole.
constexpr int kVioCut = 256;
constexpr int kHubTestBit = 1 << 30;   // in a hub mark: the absorption is subject to the split test (k_hub_apply)
// h_sz: the hub's size when the stage started (it only grows).  Returns kHubKeep / kHubAbsorb /
// kHubAbsorbTest or a violation bit.
__device__ __forceinline__ int HubEdge(const RState& x, int h_cons, int h_flags, int h_sz, const StageThr& T) {
  if (h_flags & kFlagHubBroken) return kHubVioBroken;
  if (x.cons >= 0) {
    if (h_cons < 0) return kHubVioInherit;       // the hub would inherit the constraint
    if (h_cons != x.cons) return kHubKeep;       // different constraints: never merged
    // Equal constraints: merged unless the means are further apart than the split threshold -- which
    // only the hub's mean at that moment can tell.  Nearly every such edge merges (two million
    // forced merges per constrained 1080p chunk, a handful of splits), so the absorption is logged
    // as one to be VERIFIED when the log is applied in order; a failed test undoes the stage.
    if ((x.flags & kFlagNoDesc) || x.sz >= h_sz) return kHubVioShape;
    return kHubAbsorbTest;
  }
  if (x.sz >= T.min_size) return kHubKeep;
  if (x.flags & kFlagNoDesc) return kHubVioShape;
  if ((x.flags & kFlagTentative) && h_cons != x.cons) return kHubVioMarked;   // TentativeViolated
  return kHubAbsorb;
}
// Two hubs: both finalized and large -- kept unless their (equal) constraints ask for the split test.
__device__ __forceinline__ int HubHubEdge(int c1, int f1, int c2, int f2) {
  if ((f1 | f2) & kFlagHubBroken) return kHubVioBroken;
  return (c1 >= 0 && c1 == c2) ? kHubVioPair : kHubKeep;
}
// Sets flag bits of a region with an atomic on the word that holds its flags byte (several kinds of
// marks are set concurrently: a byte read-modify-write would lose one).
// Returns the region's flags before.
__device__ __forceinline__ int AtomicOrFlags(uint8_t* flags, int r, int bits) {
  unsigned* w = reinterpret_cast<unsigned*>(flags + ((size_t)r & ~(size_t)3));
  return (int)((atomicOr(w, (unsigned)bits << (8 * (r & 3))) >> (8 * (r & 3))) & 0xffu);
}
// A region that cannot be a hub of its stage (the filter found an edge that needs its exact state, a
// worker an edge that breaks a hub rule): marked broken -- the stage is redone -- and, once, put on
// the exclusion list that the retry turns into kFlagHubExcluded marks.
// An edge of the stage that broke a hub rule (device_graph.h: kHubCutCap): the stage is exact in front of
// the earliest one.  kind 0: position inside the stage (the filter), 1: number of the work edge, 2: kept
// position of the edge (the workers of a tree replay's side clusters, whose edge numbers are their own).
__device__ __forceinline__ void HubViolationAt(int32_t* list, int kind, int position) {
  const int q = atomicAdd(&list[1 + kind], 1);
  if (q < kHubCutCap) list[4 + kHubExclCap + kind * kHubCutCap + q] = position;
}
__device__ __forceinline__ void HubExclude(int32_t* excl, uint8_t* flags, int r) {
  if (AtomicOrFlags(flags, r, kFlagHubBroken) & kFlagHubBroken) return;
  const int q = atomicAdd(&excl[0], 1);
  if (q < kHubExclCap) excl[4 + q] = r;
}
__device__ __forceinline__ void AtomicAndFlags(uint8_t* flags, int r, int keep_bits) {
  unsigned* w = reinterpret_cast<unsigned*>(flags + ((size_t)r & ~(size_t)3));
  atomicAnd(w, ~((unsigned)(~keep_bits & 0xff) << (8 * (r & 3))));
}

// A tentatively settled edge stays settled only while the constraints of its two regions do not
// change.  o1/o2: states before the edge, n1/n2: states that replace them (for a merge both are
// the survivor's state).
__device__ __forceinline__ bool TentativeViolated(const RState& o1, const RState& o2,
                                                  const RState& n1, const RState& n2) {
  return ((o1.flags & kFlagTentative) && n1.cons != o1.cons) ||
         ((o2.flags & kFlagTentative) && n2.cons != o2.cons);
}

__device__ __forceinline__ RState LoadState(const NodeArrays& nodes, int r) {
  const float4 ds = nodes.desc_sz[r];
  RState s;
  s.d0 = ds.x;
  s.d1 = ds.y;
  s.d2 = ds.z;
  s.sz = __float_as_int(ds.w);
  s.cons = nodes.cons[r];
  s.flags = nodes.flags[r];
  return s;
}

__device__ __forceinline__ void StoreState(const NodeArrays& nodes, int r, const RState& s) {
  nodes.desc_sz[r] = make_float4(s.d0, s.d1, s.d2, __int_as_float(s.sz));
  nodes.cons[r] = s.cons;
  nodes.flags[r] = (uint8_t)(s.flags & ~(int)kFlagHub);
}
// ... in a stage with hubs: with the hub mark (device_graph.h: NodeArrays::hub8) as kFlagHub.
__device__ __forceinline__ RState LoadStateHub(const NodeArrays& nodes, int r, int hubs) {
  RState s = LoadState(nodes, r);
  if (hubs && nodes.hub8[r]) s.flags |= kFlagHub;
  return s;
}

// ------------------------------------------------------------------------------------------
// Wave-level primitives.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int ReadLaneI(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float ReadLaneF(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ RState ReadLaneState(const RState& s, int lane) {
  RState r;
  r.d0 = ReadLaneF(s.d0, lane);
  r.d1 = ReadLaneF(s.d1, lane);
  r.d2 = ReadLaneF(s.d2, lane);
  r.sz = ReadLaneI(s.sz, lane);
  r.cons = ReadLaneI(s.cons, lane);
  r.flags = ReadLaneI(s.flags, lane);
  return r;
}

// c ? a : b, field by field (a conditional on the structs themselves goes through the stack).
__device__ __forceinline__ RState SelectState(bool c, const RState& a, const RState& b) {
  RState r;
  r.d0 = c ? a.d0 : b.d0;
  r.d1 = c ? a.d1 : b.d1;
  r.d2 = c ? a.d2 : b.d2;
  r.sz = c ? a.sz : b.sz;
  r.cons = c ? a.cons : b.cons;
  r.flags = c ? a.flags : b.flags;
  return r;
}

__device__ __forceinline__ bool SameState(const RState& a, const RState& b) {
  return a.sz == b.sz && a.cons == b.cons && a.flags == b.flags;   // descriptor only changes with sz
}

// Whole-wave shift by one lane (DPP wave_shr:1, gfx9): lane i receives lane i-1; lane 0 receives
// zero / `first`.
__device__ __forceinline__ float DppWaveShr1Zero(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float DppWaveShr1Old(float v, float first) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

// Wave-wide inclusive prefix sum / maximum with DPP row shifts and row broadcasts (no LDS round
// trips).  Lanes without a source keep the identity 0 (`old` operand, bound_ctrl off).
template <int kCtrl, int kRowMask, int kBankMask>
__device__ __forceinline__ int Dpp0(int v) {
  return __builtin_amdgcn_update_dpp(0, v, kCtrl, kRowMask, kBankMask, false);
}

__device__ __forceinline__ int WaveInclusiveSum(int v) {
  int t = v + Dpp0<0x111, 0xf, 0xf>(v);     // row_shr:1
  t += Dpp0<0x112, 0xf, 0xf>(v);            // row_shr:2
  int o = t + Dpp0<0x113, 0xf, 0xf>(v);     // row_shr:3
  o += Dpp0<0x114, 0xf, 0xe>(o);            // row_shr:4, banks 1-3
  o += Dpp0<0x118, 0xf, 0xc>(o);            // row_shr:8, banks 2-3
  o += Dpp0<0x142, 0xa, 0xf>(o);            // row_bcast:15 into rows 1 and 3
  o += Dpp0<0x143, 0xc, 0xf>(o);            // row_bcast:31 into rows 2 and 3
  return o;
}

// Maximum of non-negative values, returned to every lane.
__device__ __forceinline__ int WaveMax(int v) {
  int t = max(v, Dpp0<0x111, 0xf, 0xf>(v));
  t = max(t, Dpp0<0x112, 0xf, 0xf>(v));
  int o = max(t, Dpp0<0x113, 0xf, 0xf>(v));
  o = max(o, Dpp0<0x114, 0xf, 0xe>(o));
  o = max(o, Dpp0<0x118, 0xf, 0xc>(o));
  o = max(o, Dpp0<0x142, 0xa, 0xf>(o));
  o = max(o, Dpp0<0x143, 0xc, 0xf>(o));
  return __builtin_amdgcn_readlane(o, 63);
}

// ------------------------------------------------------------------------------------------
// Workers of the components with more than kSmallSegment active edges (one launch per stage).
// ------------------------------------------------------------------------------------------
struct WorkerArgs {
  const int32_t* num_segs;
  const int32_t* seg_off;
  const int32_t* seg_cnt;
  const int32_t* s_ra;      // active edges in component order: roots at filter time, kept position
  const int32_t* s_rb;
  const uint32_t* s_gpos;
  NodeArrays nodes;
  uint8_t* kept_all;
  StageThr T;
  int optimistic;
  int32_t* violation;
  unsigned long long* stats;
  int small_seg;            // components of up to small_seg edges are replayed by one lane (k_merge_small)
  int wave_min;             // the wave worker takes components of more than wave_min and less than
  int wave_max;             // wave_max edges (k_merge_small: up to kSmallSegment, when wave_min is that)
  // Work list of the wave worker (null: every workgroup strides over all segments).  k_merge_small
  // visits every segment anyway and files the ones the wave worker will replay into three size
  // classes (kWaveClasses: largest first); the wave worker's workgroups then draw tickets, so the
  // long components start first and nobody strides over a hundred thousand small segments.
  uint32_t* work_list;      // [kWaveClasses][work_cap] segment indices
  int work_cap;
  int32_t* work_ctl;        // [kWaveClasses] counts, [kWaveClasses] the wave worker's ticket, [kWaveClasses + 1]
                            // the wide worker's -- zeroed before k_merge_small
  // Components of the largest class with at least wide_min edges (and less than wave_max) are
  // replayed by the wide worker (merge_wide.hip: wide_waves wavefronts per component); 0: none.
  int wide_min = 0;
  int wide_waves = 4;
  // Hubs (T.hubs): hub_mark[seq] receives the absorbed region of the work edge with sequence number seq
  // (-1 otherwise; cleared by the stage driver), seq = s_seq[position in the component-sorted arrays].
  int32_t* hub_mark = nullptr;
  const uint32_t* s_seq = nullptr;
  int32_t* hub_excl = nullptr;   // MergeScratch::hub_excl
};
constexpr int kWaveClasses = 3;
constexpr int kWaveClassMin1 = 192;    // class 1: at least this many edges
constexpr int kWaveClassMin0 = 1536;   // class 0: at least this many edges
// Round-based replay by one consumer wavefront + one reader wavefront (the default).
void LaunchMergeWave(int grid, const WorkerArgs& a, bool instrumented, int dbg_flags, hipStream_t s);
// Lock-step replay by `waves` wavefronts per component (merge_wide.hip); draws its components from
// class 0 of the work list, so a.work_list must not be null.
void LaunchMergeWide(int grid, const WorkerArgs& a, int waves, hipStream_t s);

// Next event of the stage's pool (HIP events recorded around the dominant kernels; resolved by the
// caller once the stream has been synchronised); -1 without a pool.
inline int NextEvent(MergeScratch& S) {
  if (!S.ev_pool) return -1;
  if (*S.ev_used >= (int)S.ev_pool->size()) {
    hipEvent_t e;
    VSG_HIP(hipEventCreate(&e));
    S.ev_pool->push_back(e);
  }
  return (*S.ev_used)++;
}

// Kruskal-tree replay of the large components (merge_spine.hip).
struct SpineSeg {
  int off;   // first edge in the component-sorted arrays
  int cnt;
};
struct SpineInput {
  std::vector<SpineSeg> segs;
};
// Launches the ordinary workers (k_merge_small + wave worker) on the given segments.
using SpineWorkers = std::function<void(const WorkerArgs&, int n_edges, hipStream_t)>;
// The list SelectLargeSegments reads (the mailbox's mapped list, device_graph.h).
constexpr int kSpineListCap = 4095;
// Chooses the components (segments) of at least min_cnt edges for the Kruskal-tree replay, raising
// the threshold until all of them fit max_edges; returns the threshold (the ordinary workers leave
// segments of at least that many edges alone), 0x7fffffff when there is none.  Waits for the list.
// wanted_edges (optional): edges of all components of at least min_cnt edges (what the pool would
// have to hold for none of them to be left to the wave worker).
int SelectLargeSegments(int max_segs, const int32_t* num_segs, const int32_t* seg_off, const int32_t* seg_cnt,
                        int min_cnt, long long max_edges, MergeScratch& S, hipStream_t s, SpineInput* out,
                        long long* wanted_edges);
// pool_used: ints of S.spine_pool already taken (by the caller's lists and outer levels).
bool RunSpineComponents(const SpineInput& in, const WorkerArgs& wa, MergeScratch& S, hipStream_t s,
                        const SpineWorkers& run_workers, size_t pool_used, int depth);

}  // namespace vsg

#endif  // VSG_MERGE_COMMON_H_
