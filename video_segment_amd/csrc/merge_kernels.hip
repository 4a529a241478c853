// merge_kernels.hip -- the ordered union-find merge (K6), gfx950.
//
// Reference semantics restated: FastSegmentationGraph::SegmentGraph / GetRegion / MergeRegions
// (segmentation/segmentation_graph.h:339-463, 651-701) with ColorMeanDescriptorTraits
// (segmentation/pixel_distance.h:469-521).  The reference walks every edge sequentially in
// (bucket, bucket list, insertion) order and its merge predicate depends on evolving float state,
// so the result is order dependent.  This file keeps that order *exactly* and extracts the
// parallelism that is provably free:
//
//   stage = one bucket.  k_filter (all CUs): find both roots with path compression, drop edges
//   that are already internal, settle edges between two finalized, large, unconstrained-graph
//   regions as "kept" (they can never change state again), and hook the roots of the remaining
//   *active* edges into a scratch union-find (ECL-CC style atomicCAS hooking).
//   Two active edges can only influence each other if they are connected through active edges of
//   the same bucket, so each connected component of that scratch graph is an independent
//   sequential sub-problem.  Active edges are stably sorted by component and every component is
//   replayed in the reference's order by its own worker: one lane for a small component, one
//   64-lane wavefront for a large one (lanes prefetch roots + region state for 64 edges, then the
//   wave resolves them in order with readlane broadcasts, keeping region state in registers).
//
// This is memory-latency / dependency bound integer + scalar float work: no MFMA, no LDS tiling;
// what matters is coalesced streaming in the filter and keeping the serial chains in registers.
#include "device_graph.h"

namespace vsg {

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
__global__ __launch_bounds__(256) void k_filter(int bucket, int n_b,
                                                 const ListDesc* __restrict__ lists,
                                                 const int32_t* __restrict__ base_row,
                                                 const uint32_t* __restrict__ list_slot_base,
                                                 uint8_t* __restrict__ kept_all, NodeArrays nodes,
                                                 MergeParams P, int inert_mode,
                                                 int32_t* __restrict__ cc,
                                                 int32_t* __restrict__ e_ra,
                                                 int32_t* __restrict__ e_rb,
                                                 uint32_t* __restrict__ e_gpos,
                                                 int32_t* __restrict__ e_active,
                                                 uint8_t* __restrict__ e_ti,
                                                 int32_t* __restrict__ num_ti) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  int ti = 0;
  if (j < n_b) {
    const int l = LocateList(base_row, P.num_lists, j);
    const ListDesc L = lists[l];
    const int pos = L.offsets[bucket] + (j - base_row[l]);
    int a, b;
    DecodeEdge(L, L.slots[pos], P.W, a, b);
    const int ra = FindCompress(nodes.parent, a);
    const int rb = FindCompress(nodes.parent, b);
    const uint32_t gpos = list_slot_base[l] + (uint32_t)pos;
    int active = 0;
    if (ra != rb) {
      bool inert = false;
      if (inert_mode != 0) {
        const int f1 = nodes.flags[ra], f2 = nodes.flags[rb];
        bool both_final_large = false;
        if ((f1 & kFlagFinalized) && (f2 & kFlagFinalized)) {
          const int s1 = __float_as_int(nodes.desc_sz[ra].w);
          const int s2 = __float_as_int(nodes.desc_sz[rb].w);
          both_final_large = (s1 >= P.min_region_size) && (s2 >= P.min_region_size);
        }
        if (inert_mode == 1) {
          inert = both_final_large;
        } else {
          const int c1 = nodes.cons[ra], c2 = nodes.cons[rb];
          if (c1 >= 0 && c2 >= 0) {
            inert = (c1 != c2);            // different constraints: never merged
          } else {
            inert = both_final_large;      // at least one unconstrained
          }
          if (inert) {
            ti = 1;
            if (!(f1 & kFlagTentative)) nodes.flags[ra] = (uint8_t)(f1 | kFlagTentative);
            if (!(f2 & kFlagTentative)) nodes.flags[rb] = (uint8_t)(f2 | kFlagTentative);
          }
        }
      }
      if (inert) {
        kept_all[gpos] = 1;
      } else {
        active = 1;
        CcUnion(cc, ra, rb);
      }
    }
    e_ra[j] = ra;
    e_rb[j] = rb;
    e_gpos[j] = gpos;
    e_active[j] = active;
    e_ti[j] = (uint8_t)ti;
  }
  const unsigned long long m = __ballot(ti != 0);
  if (m != 0 && (threadIdx.x & 63) == 0) atomicAdd(num_ti, (int)__popcll(m));
}

// Clears the tentative marks of a stage: on the regions marked by the filter and on whatever
// region they have been merged into since.
__global__ __launch_bounds__(256) void k_clear_tentative(int n_b, const uint8_t* __restrict__ e_ti,
                                                          const int32_t* __restrict__ e_ra,
                                                          const int32_t* __restrict__ e_rb,
                                                          NodeArrays nodes) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_b || !e_ti[j]) return;
  int r[2] = {e_ra[j], e_rb[j]};
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
                                                       uint8_t* __restrict__ bk_flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
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
                                                        const uint8_t* __restrict__ bk_flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
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

__global__ __launch_bounds__(256) void k_clear_kept(int n_b, const uint32_t* __restrict__ e_gpos,
                                                     uint8_t* __restrict__ kept_all) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n_b) kept_all[e_gpos[j]] = 0;
}

__global__ __launch_bounds__(256) void k_compact_active(int n_b, const int32_t* __restrict__ e_active,
                                                         const int32_t* __restrict__ e_apos,
                                                         const int32_t* __restrict__ e_ra,
                                                         const int32_t* __restrict__ e_rb,
                                                         const uint32_t* __restrict__ e_gpos,
                                                         int32_t* __restrict__ a_ra,
                                                         int32_t* __restrict__ a_rb,
                                                         uint32_t* __restrict__ a_gpos,
                                                         int32_t* __restrict__ num_active) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_b) return;
  if (e_active[j]) {
    const int p = e_apos[j];
    a_ra[p] = e_ra[j];
    a_rb[p] = e_rb[j];
    a_gpos[p] = e_gpos[j];
  }
  if (j == n_b - 1) *num_active = e_apos[j] + e_active[j];
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
                                                   int32_t* __restrict__ cc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ra = a_ra[i], rb = a_rb[i];
  cc[ra] = ra;
  cc[rb] = rb;
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
// constraints, merged state).  stat: 0 none, 1 forced, 2 regular, 3 small.
struct StageThr {
  float pass_s;    // regular test passes  <=> s <= pass_s
  float split_s;   // constrained split    <=> s >  split_s
  int min_size;
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
      return kOutKeep;
    }
    stat = 1;
    return MergeStates(s1, s2);
  }
  return kOutKeep;
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
  nodes.flags[r] = (uint8_t)s.flags;
}

// ------------------------------------------------------------------------------------------
// Worker A: one lane replays one small component.
// ------------------------------------------------------------------------------------------
constexpr int kSmallSegment = 24;   // components with more active edges go to a wavefront

__global__ __launch_bounds__(256) void k_merge_small(const int32_t* __restrict__ num_segs,
                                                      const int32_t* __restrict__ seg_off,
                                                      const int32_t* __restrict__ seg_cnt,
                                                      const uint32_t* __restrict__ s_idx,
                                                      const int32_t* __restrict__ a_ra,
                                                      const int32_t* __restrict__ a_rb,
                                                      const uint32_t* __restrict__ a_gpos,
                                                      NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                      StageThr T, int optimistic,
                                                      int32_t* __restrict__ violation,
                                                      unsigned long long* __restrict__ stats) {
  const int seg = blockIdx.x * 256 + threadIdx.x;
  unsigned n_forced = 0, n_regular = 0, n_small = 0;
  if (seg < *num_segs) {
    const int cnt = seg_cnt[seg];
    if (cnt <= kSmallSegment) {
      const int beg = seg_off[seg];
      for (int p = beg; p < beg + cnt; ++p) {
        const uint32_t i = s_idx[p];
        // An optimistic stage must stay undoable from the backed-up region states alone, so it
        // does not compress paths.
        const int r1 = optimistic ? FindReadOnly(nodes.parent, a_ra[i])
                                  : FindCompress(nodes.parent, a_ra[i]);
        const int r2 = optimistic ? FindReadOnly(nodes.parent, a_rb[i])
                                  : FindCompress(nodes.parent, a_rb[i]);
        if (r1 == r2) continue;
        RState s1 = LoadState(nodes, r1);
        RState s2 = LoadState(nodes, r2);
        const RState o1 = s1, o2 = s2;
        int stat;
        const int out = DecideEdge(s1, s2, T, stat);
        if (optimistic) {
          const bool v = (out == kOutKeep)     ? TentativeViolated(o1, o2, s1, s2)
                         : (out == kOutMerge1) ? TentativeViolated(o1, o2, s1, s1)
                                               : TentativeViolated(o1, o2, s2, s2);
          if (v) *violation = 1;
        }
        n_forced += (stat == 1);
        n_regular += (stat == 2);
        n_small += (stat == 3);
        if (out == kOutKeep) {
          kept_all[a_gpos[i]] = 1;
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
// Worker B: one wavefront replays one large component, 64 edges per batch.
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

__device__ __forceinline__ bool SameState(const RState& a, const RState& b) {
  return a.sz == b.sz && a.cons == b.cons && a.flags == b.flags;   // descriptor only changes with sz
}

// The common pattern inside a large component is a chain: one big region absorbs neighbour after
// neighbour.  The winner of the last merge is therefore kept as the "hot" region: its state lives
// in (wave-uniform) registers, is used instead of any lane's cached copy, and is written back
// only when another region becomes hot or the component is finished.  Lane copies of a region are
// refreshed (12 v_cndmask) only when it stops being hot or on the rare flag/constraint change, so
// a chain step costs two id readlanes, six state readlanes, the decision and one parent store.
__global__ __launch_bounds__(64) void k_merge_wave(const int32_t* __restrict__ num_segs,
                                                    const int32_t* __restrict__ seg_off,
                                                    const int32_t* __restrict__ seg_cnt,
                                                    const uint32_t* __restrict__ s_idx,
                                                    const int32_t* __restrict__ a_ra,
                                                    const int32_t* __restrict__ a_rb,
                                                    const uint32_t* __restrict__ a_gpos,
                                                    NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                    StageThr T, int optimistic,
                                                    int32_t* __restrict__ violation,
                                                    unsigned long long* __restrict__ stats) {
  const int lane = threadIdx.x;
  const int nseg = *num_segs;
  unsigned n_forced = 0, n_regular = 0, n_small = 0;   // counted on lane 0 (uniform decisions)
  unsigned dbg_iters = 0, dbg_hot = 0, dbg_internal = 0, dbg_batches = 0;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int cnt = seg_cnt[seg];
    if (cnt <= kSmallSegment) continue;
    const int beg = seg_off[seg];
    const int end = beg + cnt;
    if (lane == 0) atomicAdd(&stats[3], (unsigned long long)cnt);
    int hot = -1;          // wave-uniform
    RState H = {};         // wave-uniform state of region `hot` (authoritative while hot >= 0)
    for (int base = beg; base < end; base += 64) {
      const int p = base + lane;
      const bool valid = p < end;
      int ra = -1, rb = -2;
      uint32_t gpos = 0;
      RState A = {}, B = {};
      if (valid) {
        const uint32_t i = s_idx[p];
        ra = optimistic ? FindReadOnly(nodes.parent, a_ra[i]) : FindCompress(nodes.parent, a_ra[i]);
        rb = optimistic ? FindReadOnly(nodes.parent, a_rb[i]) : FindCompress(nodes.parent, a_rb[i]);
        gpos = a_gpos[i];
        if (ra != rb) {
          A = LoadState(nodes, ra);
          B = LoadState(nodes, rb);
        }
      }
      bool my_kept = false;
      unsigned long long pending = __ballot(valid && ra != rb);
      // Per-lane flags: the cached region is a plain partner (unconstrained, not finalized, has a
      // descriptor, not tentatively marked).
      bool ok_a = (A.cons < 0 && A.flags == 0);
      bool ok_b = (B.cons < 0 && B.flags == 0);
      if (lane == 0) ++dbg_batches;
      while (pending) {
        // ---- tight chain loop: the hot region absorbs plain partners smaller than itself ------
        // Every lane keeps a "partner view" relative to the hot region (which of its two regions
        // is the partner, the partner's state), re-derived with a dozen VALU ops after each
        // merge, so that one chain step is: 5 readlanes, the test, the mean update, one parent
        // store and the id rename -- the same arithmetic as DecideEdge/MergeStates for this case
        // (Case U; the hot region keeps its flags and constraint; the partner carries no mark).
        if (hot >= 0 && !(H.flags & kFlagNoDesc)) {
          for (;;) {
            const bool e1 = (ra == hot), e2 = (rb == hot);
            const unsigned long long internal = __ballot(ra == rb);
            const unsigned long long chain = __ballot((e1 != e2) && (e1 ? ok_b : ok_a));
            // drop leading pending lanes that became internal
            while (pending && ((internal >> __builtin_ctzll(pending)) & 1ull)) {
              pending &= pending - 1;
              if (lane == 0) ++dbg_internal;
            }
            if (!pending) break;
            const int j = __builtin_ctzll(pending);
            if (!((chain >> j) & 1ull)) break;
            const float px0 = e1 ? B.d0 : A.d0;
            const float px1 = e1 ? B.d1 : A.d1;
            const float px2 = e1 ? B.d2 : A.d2;
            const int pszv = e1 ? B.sz : A.sz;
            const int pidv = e1 ? rb : ra;
            const int psz = ReadLaneI(pszv, j);
            if (!(H.sz > psz)) break;
            const float p0 = ReadLaneF(px0, j), p1 = ReadLaneF(px1, j), p2 = ReadLaneF(px2, j);
            bool merge;
            if (!(H.flags & kFlagFinalized)) {
              const float x = H.d0 - p0, y = H.d1 - p1, z = H.d2 - p2;
              if (!((x * x + y * y + z * z) * (1.0f / 3.0f) <= T.pass_s)) break;   // -> generic
              merge = true;
              ++n_regular;
            } else {
              merge = (psz < T.min_size || H.sz < T.min_size);
              n_small += merge;
            }
            pending &= pending - 1;
            if (lane == 0) { ++dbg_iters; ++dbg_hot; }
            if (merge) {
              const float denom = 1.0f / (float)(psz + H.sz);
              const float ca = (float)psz * denom;
              const float cb = (float)H.sz * denom;
              H.d0 = ca * p0 + cb * H.d0;
              H.d1 = ca * p1 + cb * H.d1;
              H.d2 = ca * p2 + cb * H.d2;
              H.sz += psz;
              const int pid = ReadLaneI(pidv, j);
              if (lane == j) nodes.parent[pid] = hot;
              if (ra == pid) ra = hot;
              if (rb == pid) rb = hot;
            } else if (lane == j) {
              my_kept = true;   // both regions finalized / large: kept, nothing changes
            }
          }
          if (!pending) break;
        }
        const int j = __builtin_ctzll(pending);
        pending &= pending - 1;
        const int r1 = ReadLaneI(ra, j);
        const int r2 = ReadLaneI(rb, j);
        if (r1 == r2) { if (lane == 0) ++dbg_internal; continue; }   // became internal
        if (lane == 0) { ++dbg_iters; dbg_hot += (r1 == hot || r2 == hot); }
        RState s1, s2;
        if (r1 == hot) s1 = H; else s1 = ReadLaneState(A, j);
        if (r2 == hot) s2 = H; else s2 = ReadLaneState(B, j);
        const RState o1 = s1, o2 = s2;
        int stat;
        const int out = DecideEdge(s1, s2, T, stat);
        if (optimistic) {
          const bool v = (out == kOutKeep)     ? TentativeViolated(o1, o2, s1, s2)
                         : (out == kOutMerge1) ? TentativeViolated(o1, o2, s1, s1)
                                               : TentativeViolated(o1, o2, s2, s2);
          if (v && lane == 0) *violation = 1;
        }
        n_forced += (stat == 1);
        n_regular += (stat == 2);
        n_small += (stat == 3);
        if (out == kOutKeep) {
          if (lane == j) my_kept = true;
          // rare: finalisation or constraint reset changed one or both regions
          if (!SameState(o1, s1)) {
            if (r1 == hot) H = s1;
            if (ra == r1) A = s1;
            if (rb == r1) B = s1;
            if (lane == j) StoreState(nodes, r1, s1);
          }
          if (!SameState(o2, s2)) {
            if (r2 == hot) H = s2;
            if (ra == r2) A = s2;
            if (rb == r2) B = s2;
            if (lane == j) StoreState(nodes, r2, s2);
          }
        } else {
          const int win = (out == kOutMerge1) ? r1 : r2;
          const int lose = (out == kOutMerge1) ? r2 : r1;
          const RState sw = (out == kOutMerge1) ? s1 : s2;
          if (win != hot) {
            if (hot >= 0 && hot != lose) {
              // the previous hot region leaves the registers: refresh lane copies + memory
              if (ra == hot) A = H;
              if (rb == hot) B = H;
              if (lane == 0) StoreState(nodes, hot, H);
            }
            hot = win;
          }
          H = sw;
          if (lane == j) nodes.parent[lose] = win;
          if (ra == lose) ra = win;
          if (rb == lose) rb = win;
        }
        // the generic path may have refreshed lane copies: re-derive the partner flags
        ok_a = (A.cons < 0 && A.flags == 0);
        ok_b = (B.cons < 0 && B.flags == 0);
      }
      if (valid && my_kept) kept_all[gpos] = 1;
      // Make this batch's stores visible to the next batch's loads (same CU: L1 is shared).
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      // Next batch reloads region states from memory; the hot region's memory copy is stale, so
      // write it back here once per batch (one 21-byte store) instead of once per merge.
      if (hot >= 0 && lane == 0) StoreState(nodes, hot, H);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  }
  if (lane == 0) {
    if (n_forced) atomicAdd(&stats[0], (unsigned long long)n_forced);
    if (n_regular) atomicAdd(&stats[1], (unsigned long long)n_regular);
    if (n_small) atomicAdd(&stats[2], (unsigned long long)n_small);
    atomicAdd(&stats[4], (unsigned long long)dbg_iters);
    atomicAdd(&stats[5], (unsigned long long)dbg_hot);
    atomicAdd(&stats[6], (unsigned long long)dbg_internal);
    atomicAdd(&stats[7], (unsigned long long)dbg_batches);
  }
}

// ------------------------------------------------------------------------------------------
// Host driver of one bucket stage.
// ------------------------------------------------------------------------------------------
static inline unsigned Blocks(int n) { return (unsigned)((n + 255) / 256); }

static int NextEvent(MergeScratch& S) {
  if (!S.ev_pool) return -1;
  if (*S.ev_used >= (int)S.ev_pool->size()) {
    hipEvent_t e;
    VSG_HIP(hipEventCreate(&e));
    S.ev_pool->push_back(e);
  }
  return (*S.ev_used)++;
}

void RunBucketStage(int bucket, int n_b, const ListDesc* lists, const int32_t* bucket_base,
                    const uint32_t* list_slot_base, uint8_t* kept_all, NodeArrays nodes,
                    const MergeParams& P, int inert_mode, MergeScratch& S, hipStream_t s) {
  if (n_b <= 0) return;
  const int32_t* base_row = bucket_base + (size_t)bucket * (P.num_lists + 1);
  int32_t* d_num_ti = S.num_active + 2;
  int32_t* d_violation = S.num_active + 3;
  VSG_HIP(hipMemsetAsync(d_num_ti, 0, 2 * sizeof(int32_t), s));
  const int ef0 = NextEvent(S);
  if (ef0 >= 0) VSG_HIP(hipEventRecord((*S.ev_pool)[ef0], s));
  hipLaunchKernelGGL(k_filter, dim3(Blocks(n_b)), dim3(256), 0, s, bucket, n_b, lists, base_row,
                     list_slot_base, kept_all, nodes, P, inert_mode, S.cc, S.e_ra, S.e_rb, S.e_gpos,
                     S.e_active, S.e_ti, d_num_ti);
  const int ef1 = NextEvent(S);
  if (ef1 >= 0) {
    VSG_HIP(hipEventRecord((*S.ev_pool)[ef1], s));
    S.ev_filter->emplace_back(ef0, ef1);
  }
  ExclusiveSumI32(S.cub_temp, S.cub_temp_bytes, S.e_active, S.e_apos, n_b, s);
  hipLaunchKernelGGL(k_compact_active, dim3(Blocks(n_b)), dim3(256), 0, s, n_b, S.e_active,
                     S.e_apos, S.e_ra, S.e_rb, S.e_gpos, S.a_ra, S.a_rb, S.a_gpos, S.num_active);
  VSG_HIP(hipGetLastError());
  int h[4] = {0, 0, 0, 0};   // num_active, num_segs (unused), num_ti, violation
  VSG_HIP(hipMemcpyAsync(h, S.num_active, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  VSG_HIP(hipStreamSynchronize(s));
  const int n_active = h[0];
  const int n_ti = h[2];
  auto clear_marks = [&]() {
    if (n_ti > 0) {
      hipLaunchKernelGGL(k_clear_tentative, dim3(Blocks(n_b)), dim3(256), 0, s, n_b, S.e_ti, S.e_ra,
                         S.e_rb, nodes);
    }
  };
  if (n_active == 0) {
    clear_marks();
    return;
  }

  hipLaunchKernelGGL(k_component_ids, dim3(Blocks(n_active)), dim3(256), 0, s, n_active, S.a_ra,
                     S.cc, S.a_comp, S.a_idx);
  SortPairsU32(S.cub_temp, S.cub_temp_bytes, S.a_comp, S.s_comp, S.a_idx, S.s_idx, n_active, 32,
               s);
  RunLengthEncodeU32(S.cub_temp, S.cub_temp_bytes, S.s_comp, S.seg_key, S.seg_cnt, S.num_segs,
                     n_active, s);
  // Segment offsets: exclusive scan over n_active counts (only the first num_segs are defined;
  // the prefix of an exclusive scan never depends on later elements).
  ExclusiveSumI32(S.cub_temp, S.cub_temp_bytes, S.seg_cnt, S.seg_off, n_active, s);

  const bool optimistic = (inert_mode == 2) && n_ti > 0;
  if (optimistic) {
    hipLaunchKernelGGL(k_backup_roots, dim3(Blocks(n_active)), dim3(256), 0, s, n_active, S.a_ra,
                       S.a_rb, nodes, S.bk_ds, S.bk_cons, S.bk_flags);
    VSG_HIP(hipMemcpyAsync(S.stats + 8, S.stats, 8 * sizeof(unsigned long long),
                           hipMemcpyDeviceToDevice, s));
  }

  const float weight = (float)bucket * P.inv_scale;
  const bool force = weight < P.force_merge_weight;
  StageThr T;
  T.pass_s = force ? P.s_lt_02 : P.s_lt_005;
  T.split_s = force ? P.s_lt_02 : P.s_le_015;
  T.min_size = P.min_region_size;
  hipLaunchKernelGGL(k_merge_small, dim3(Blocks(n_active)), dim3(256), 0, s, S.num_segs, S.seg_off,
                     S.seg_cnt, S.s_idx, S.a_ra, S.a_rb, S.a_gpos, nodes, kept_all, T,
                     optimistic ? 1 : 0, d_violation, S.stats);
  const int wave_grid = n_active / (kSmallSegment + 1) < 1 ? 1
                        : (n_active / (kSmallSegment + 1) > 8192 ? 8192
                                                                 : n_active / (kSmallSegment + 1));
  const int ew0 = NextEvent(S);
  if (ew0 >= 0) VSG_HIP(hipEventRecord((*S.ev_pool)[ew0], s));
  hipLaunchKernelGGL(k_merge_wave, dim3(wave_grid), dim3(64), 0, s, S.num_segs, S.seg_off,
                     S.seg_cnt, S.s_idx, S.a_ra, S.a_rb, S.a_gpos, nodes, kept_all, T,
                     optimistic ? 1 : 0, d_violation, S.stats);
  const int ew1 = NextEvent(S);
  if (ew1 >= 0) {
    VSG_HIP(hipEventRecord((*S.ev_pool)[ew1], s));
    S.ev_wave->emplace_back(ew0, ew1);
  }
  hipLaunchKernelGGL(k_reset_cc, dim3(Blocks(n_active)), dim3(256), 0, s, n_active, S.a_ra, S.a_rb,
                     S.cc);
  VSG_HIP(hipGetLastError());
  if (optimistic) {
    int violated = 0;
    VSG_HIP(hipMemcpyAsync(&violated, d_violation, sizeof(int), hipMemcpyDeviceToHost, s));
    VSG_HIP(hipStreamSynchronize(s));
    ++*S.optimistic_stages;
    if (violated || S.force_rollback) {
      // Undo the stage and replay it without any tentatively settled edge.
      ++*S.rollbacks;
      hipLaunchKernelGGL(k_restore_roots, dim3(Blocks(n_active)), dim3(256), 0, s, n_active, S.a_ra,
                         S.a_rb, nodes, S.bk_ds, S.bk_cons, S.bk_flags);
      VSG_HIP(hipMemcpyAsync(S.stats, S.stats + 8, 8 * sizeof(unsigned long long),
                             hipMemcpyDeviceToDevice, s));
      hipLaunchKernelGGL(k_clear_kept, dim3(Blocks(n_b)), dim3(256), 0, s, n_b, S.e_gpos, kept_all);
      clear_marks();
      VSG_HIP(hipGetLastError());
      RunBucketStage(bucket, n_b, lists, bucket_base, list_slot_base, kept_all, nodes, P, 0, S, s);
      return;
    }
  }
  clear_marks();
  VSG_HIP(hipGetLastError());
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
