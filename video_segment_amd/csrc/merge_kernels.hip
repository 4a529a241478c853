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
                                                      const int32_t* __restrict__ s_ra,
                                                      const int32_t* __restrict__ s_rb,
                                                      const uint32_t* __restrict__ s_gpos,
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
        // An optimistic stage must stay undoable from the backed-up region states alone, so it
        // does not compress paths.
        const int r1 = optimistic ? FindReadOnly(nodes.parent, s_ra[p])
                                  : FindCompress(nodes.parent, s_ra[p]);
        const int r2 = optimistic ? FindReadOnly(nodes.parent, s_rb[p])
                                  : FindCompress(nodes.parent, s_rb[p]);
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
__global__ __launch_bounds__(64) void k_merge_wave_v1(const int32_t* __restrict__ num_segs,
                                                       const int32_t* __restrict__ seg_off,
                                                       const int32_t* __restrict__ seg_cnt,
                                                       const int32_t* __restrict__ s_ra,
                                                       const int32_t* __restrict__ s_rb,
                                                       const uint32_t* __restrict__ s_gpos,
                                                    NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                    StageThr T, int optimistic,
                                                    int32_t* __restrict__ violation,
                                                    unsigned long long* __restrict__ stats) {
  const int lane = threadIdx.x;
  const int nseg = *num_segs;
  unsigned n_forced = 0, n_regular = 0, n_small = 0;   // counted on lane 0 (uniform decisions)
  unsigned dbg_iters = 0, dbg_hot = 0, dbg_internal = 0, dbg_batches = 0, dbg_chain = 0;
  unsigned long long cyc_load = 0, cyc_loop = 0;
  unsigned dbg_g[6] = {0, 0, 0, 0, 0, 0};
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int cnt = seg_cnt[seg];
    if (cnt <= kSmallSegment) continue;
    const int beg = seg_off[seg];
    const int end = beg + cnt;
    if (lane == 0) atomicAdd(&stats[3], (unsigned long long)cnt);
    const unsigned long long seg_t0 = __builtin_readcyclecounter();
    int hot = -1;          // wave-uniform
    RState H = {};         // wave-uniform state of region `hot` (authoritative while hot >= 0)
    for (int base = beg; base < end; base += 64) {
      const int p = base + lane;
      const bool valid = p < end;
      int ra = -1, rb = -2;
      uint32_t gpos = 0;
      RState A = {}, B = {};
      const unsigned long long bt0 = __builtin_readcyclecounter();
      if (valid) {
        ra = optimistic ? FindReadOnly(nodes.parent, s_ra[p]) : FindCompress(nodes.parent, s_ra[p]);
        rb = optimistic ? FindReadOnly(nodes.parent, s_rb[p]) : FindCompress(nodes.parent, s_rb[p]);
        gpos = s_gpos[p];
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
      const unsigned long long bt1 = __builtin_readcyclecounter();
      cyc_load += bt1 - bt0;
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
            if (lane == 0) { ++dbg_iters; ++dbg_hot; ++dbg_chain; }
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
        if (lane == 0) {   // debug classification of the generic iterations
          const bool h1 = (r1 == hot), h2 = (r2 == hot);
          const bool pl1 = (s1.cons < 0 && s1.flags == 0), pl2 = (s2.cons < 0 && s2.flags == 0);
          if (!h1 && !h2) {
            ++dbg_g[0];
            if (pl1 && pl2) ++dbg_g[4];
            if (s1.sz == 1 && s2.sz == 1) ++dbg_g[5];
          } else {
            const bool ppl = h1 ? pl2 : pl1;
            const int psz = h1 ? s2.sz : s1.sz;
            if (!ppl) ++dbg_g[1];
            else if (!(H.sz > psz)) ++dbg_g[2];
            else ++dbg_g[3];
          }
        }
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
      cyc_loop += __builtin_readcyclecounter() - bt1;
    }
    if (lane == 0) {
      atomicMax(&stats[16], __builtin_readcyclecounter() - seg_t0);   // slowest component
      atomicMax(&stats[17], (unsigned long long)cnt);                 // largest component
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
    atomicAdd(&stats[18], cyc_load);
    atomicAdd(&stats[19], cyc_loop);
    atomicAdd(&stats[20], (unsigned long long)dbg_chain);
    for (int k = 0; k < 6; ++k) atomicAdd(&stats[24 + k], (unsigned long long)dbg_g[k]);
  }
}

// ------------------------------------------------------------------------------------------
// Worker B: one wavefront replays one large component, 64 edges per batch, in *rounds*.
// ------------------------------------------------------------------------------------------
// A lone wavefront issues roughly one instruction every 4-8 cycles, so replaying the 64 edges of
// a batch one after the other (k_merge_wave_v1, ~130 instructions per edge) leaves the stage bound
// by its largest component.  This worker keeps the regions of the batch in an LDS table and
// commits as many edges per round as the sequential semantics allow:
//
//   * every pending lane reserves its two regions with its lane number (ds_min); a lane that
//     holds both reservations is the earliest pending edge on both regions, so executing it now
//     is what the sequential replay would do (deterministic reservations).  All such lanes run
//     DecideEdge at once, each on its own pair of regions;
//   * the region most edges of the batch touch is the batch's *hot* region and is not reserved.
//     The leading run of pending hot edges whose partner is a plain region (unconstrained,
//     unflagged) smaller than the hot region is committed as one *chain*: under the speculation
//     that every merge test passes, the sizes are a prefix sum and the weights ca/cb and the
//     products ca*p are lane-parallel; only  h = ca*p + cb*h  (two flops per channel) is replayed
//     in order, recording the pre-merge mean per lane, and the merge tests are then verified by
//     all lanes at once.  The chain is cut at the first failed test and that lane is replayed
//     by the generic code in the next round, so the result is exactly the sequential one;
//   * the first pending hot edge that does not qualify for the chain runs alone (generic code).
//
// The earliest pending lane always commits, so a batch takes at most 64 rounds.
constexpr int kTabSize = 256;     // >= 2 * 128 distinct regions of a batch at load factor 1/2
constexpr int kTabDirty = 0x100;

struct WaveTable {
  int32_t key[kTabSize];     // region id, -1: empty
  int32_t link[kTabSize];    // in-batch union-find over slots
  uint32_t res[kTabSize];    // reservation: (0xfffff - round) << 6 | lane, smaller wins
  int32_t cnt[kTabSize];     // number of pending endpoints on the slot at batch start
  float4 ds[kTabSize];
  int32_t cons[kTabSize];
  int32_t flags[kTabSize];   // region flags | kTabDirty
};

__device__ __forceinline__ int TabInsert(WaveTable& t, int r, bool& inserted) {
  unsigned h = ((unsigned)r * 2654435761u) >> 24;
  for (;;) {
    const int old = atomicCAS(&t.key[h], -1, r);
    if (old == -1) { inserted = true; return (int)h; }
    if (old == r) { inserted = false; return (int)h; }
    h = (h + 1) & (kTabSize - 1);
  }
}

__device__ __forceinline__ int SlotRoot(const WaveTable& t, int s) {
  int p;
  while ((p = t.link[s]) != s) s = p;
  return s;
}

__device__ __forceinline__ RState TabLoad(const WaveTable& t, int s) {
  const float4 ds = t.ds[s];
  RState r;
  r.d0 = ds.x;
  r.d1 = ds.y;
  r.d2 = ds.z;
  r.sz = __float_as_int(ds.w);
  r.cons = t.cons[s];
  r.flags = t.flags[s] & 0xff;
  return r;
}

__device__ __forceinline__ void TabStore(WaveTable& t, int s, const RState& r, int dirty) {
  t.ds[s] = make_float4(r.d0, r.d1, r.d2, __int_as_float(r.sz));
  t.cons[s] = r.cons;
  t.flags[s] = r.flags | dirty;
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

constexpr int kFill = 4;        // 64-edge chunks read per fill
constexpr int kQueue = 512;     // ring capacity >= kLag + kFill * 64, power of two
// The producer reads on while fewer than kLag edges wait in the ring: enough to keep the consumer
// busy for a batch or two, few enough that the roots it found are mostly still current.
constexpr int kLag = 128;

struct WaveQueue {
  int32_t ra[kQueue];
  int32_t rb[kQueue];
  uint32_t gpos[kQueue];
  int produced;   // entries pushed by the producer wave (monotonic)
  int consumed;   // entries taken by the consumer wave (monotonic)
  int done;       // the producer has read the whole component
};

// Live edges staged by the consumer (<= 63 left over + 64 new).
struct WaveStage {
  int32_t ra[128];
  int32_t rb[128];
  uint32_t gpos[128];
};

// Orders the LDS accesses of the lanes of ONE wavefront (they execute in order in hardware; this
// only keeps the compiler from moving them across the phase boundary).
__device__ __forceinline__ void WaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// Commit of a merge inside the wave worker: `ls` (loser slot) is linked to `ws` (winner slot).
// A region that stops being a representative keeps its own constraint field in the reference
// (MergeRegions only updates the survivor), and MergeConstrainedRegions later reads that field of
// every *node* -- so a constraint the loser acquired or lost earlier in this batch (its table
// entry is dirty and will never be written back) has to reach memory now.
__device__ __forceinline__ void CommitLoser(WaveTable& t, const NodeArrays& nodes, int ls, int ws) {
  t.link[ls] = ws;
  const int lid = t.key[ls];
  nodes.parent[lid] = t.key[ws];
  if (t.flags[ls] & kTabDirty) nodes.cons[lid] = t.cons[ls];
}

template <bool kDbg>
__global__ __launch_bounds__(128) void k_merge_wave(const int32_t* __restrict__ num_segs,
                                                    const int32_t* __restrict__ seg_off,
                                                    const int32_t* __restrict__ seg_cnt,
                                                    const int32_t* __restrict__ s_ra,
                                                    const int32_t* __restrict__ s_rb,
                                                    const uint32_t* __restrict__ s_gpos,
                                                    NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                    StageThr T, int optimistic,
                                                    int32_t* __restrict__ violation,
                                                    unsigned long long* __restrict__ stats,
                                                    int dbg_flags) {
  __shared__ WaveTable tab;
  auto Clock = []() -> unsigned long long { return kDbg ? __builtin_readcyclecounter() : 0ull; };
  __shared__ WaveQueue queue;
  __shared__ WaveStage stage;
  const int lane = threadIdx.x & 63;
  const bool producer = threadIdx.x >= 64;   // wave 1 reads ahead, wave 0 replays
  for (int s = threadIdx.x; s < kTabSize; s += 128) {
    tab.key[s] = -1;
    tab.res[s] = 0xffffffffu;
    tab.cnt[s] = 0;
  }
  __syncthreads();
  const int nseg = *num_segs;
  unsigned n_forced = 0, n_regular = 0, n_small = 0;   // per lane, reduced at the end
  unsigned dbg_rounds = 0, dbg_nwin = 0, dbg_chain = 0, dbg_solo = 0, dbg_batches = 0, dbg_cut = 0;
  unsigned long long dbg_taken = 0, dbg_live = 0;
  unsigned long long cyc_ph[5] = {0, 0, 0, 0, 0};
  unsigned long long dbg_x[6] = {0, 0, 0, 0, 0, 0};   // reserve+load, closure, masks, generic, chain
  unsigned long long cyc_load = 0, cyc_loop = 0, cyc_wait = 0;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int cnt = seg_cnt[seg];
    if (cnt <= kSmallSegment) continue;
    const int beg = seg_off[seg];
    const int end = beg + cnt;
    // Edges whose two ends already share a region are dropped when they are read (a large share
    // of a component's edges once its regions have grown): the producer wave reads 256 edges per
    // fill with all root searches in flight together and pushes the surviving edges into an LDS
    // ring; the consumer wave replays them 64 at a time, so the fixed cost of a batch is spent on
    // pending edges only and the global-memory latency of the reads is off the replay's path.
    if (threadIdx.x == 0) {
      queue.produced = 0;
      queue.consumed = 0;
      queue.done = 0;
    }
    __syncthreads();
    if (producer) {
      int produced = 0;
      for (int next = beg; next < end; next += kFill * 64) {
        const unsigned long long pt0 = Clock();
        while (produced - __hip_atomic_load(&queue.consumed, __ATOMIC_ACQUIRE,
                                            __HIP_MEMORY_SCOPE_WORKGROUP) > kLag) {
          __builtin_amdgcn_s_sleep(2);
        }
        const unsigned long long pt1 = Clock();
        cyc_wait += pt1 - pt0;
        int xa[kFill], xb[kFill], ca[kFill], cb[kFill];
        uint32_t gp[kFill];
        bool vd[kFill];
#pragma unroll
        for (int k = 0; k < kFill; ++k) {
          const int p = next + k * 64 + lane;
          vd[k] = p < end;
          xa[k] = vd[k] ? s_ra[p] : 0;
          xb[k] = vd[k] ? s_rb[p] : 0;
          gp[k] = vd[k] ? s_gpos[p] : 0u;
          ca[k] = xa[k];
          cb[k] = xb[k];
        }
        for (bool any = true; any;) {   // all root searches of the fill advance together
          int pa[kFill], pb[kFill];
#pragma unroll
          for (int k = 0; k < kFill; ++k) {
            pa[k] = nodes.parent[ca[k]];
            pb[k] = nodes.parent[cb[k]];
          }
          any = false;
#pragma unroll
          for (int k = 0; k < kFill; ++k) {
            if (pa[k] != ca[k]) { ca[k] = pa[k]; any = true; }
            if (pb[k] != cb[k]) { cb[k] = pb[k]; any = true; }
          }
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int k = 0; k < kFill; ++k) {
          // Path compression of the start node only (it is not a root, so the consumer never
          // writes it); an optimistic stage must stay undoable and does not compress.
          if (!optimistic && !(kDbg && (dbg_flags & 1024))) {
            if (vd[k] && ca[k] != xa[k]) nodes.parent[xa[k]] = ca[k];
            if (vd[k] && cb[k] != xb[k]) nodes.parent[xb[k]] = cb[k];
          }
          const bool pend = vd[k] && ca[k] != cb[k];
          const unsigned long long m = __ballot(pend);
          if (pend) {
            const int slot = (produced + (int)__popcll(m & lt)) & (kQueue - 1);
            queue.ra[slot] = ca[k];
            queue.rb[slot] = cb[k];
            queue.gpos[slot] = gp[k];
          }
          produced += (int)__popcll(m);
        }
        WaveSync();
        if (lane == 0) {
          __hip_atomic_store(&queue.produced, produced, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        cyc_load += Clock() - pt1;
      }
      if (lane == 0) {
        __hip_atomic_store(&queue.done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();   // end of the segment (matches the consumer's)
      continue;
    }

    // ---- consumer --------------------------------------------------------------------------------
    if (lane == 0) atomicAdd(&stats[3], (unsigned long long)cnt);
    const unsigned long long seg_t0 = Clock();
    int consumed = 0;   // wave-uniform
    int n_raw = 0;      // staged edges left over from the previous batch: roots to be re-validated
    for (;;) {
      const unsigned long long bt0 = Clock();
      // ---- stage 64 live edges -----------------------------------------------------------------------
      // The roots the producer found may be stale by now (the ring holds several batches); a large
      // share of the ring's edges is internal by the time it is taken.  A pass re-validates up
      // to 64 candidates (the left-overs of the previous batch first, then ring entries) and packs
      // the live ones in order; passes repeat until 64 live edges are staged or the component is
      // drained, so that the fixed cost of a batch is spent on live edges only.
      int n_valid = 0;
      bool drained = false;
      while (n_valid < 64 && !drained) {
        const int want = 64 - n_raw;
        int avail;
        for (;;) {   // `done` is read before `produced`: once done is set, produced is final
          const int done = __hip_atomic_load(&queue.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
          avail = __hip_atomic_load(&queue.produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) -
                  consumed;
          if (avail >= want || done) {
            if (done && avail <= want) drained = true;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        const int t = avail < want ? avail : want;
        if (n_raw + t == 0) break;
        int ca = 0, cb = 0;
        uint32_t cg = 0;
        const bool cand = lane < n_raw + t;
        if (lane < n_raw) {
          ca = stage.ra[lane];
          cb = stage.rb[lane];
          cg = stage.gpos[lane];
        } else if (cand) {
          const int slot = (consumed + lane - n_raw) & (kQueue - 1);
          ca = queue.ra[slot];
          cb = queue.rb[slot];
          cg = queue.gpos[slot];
        }
        for (bool more = cand; more;) {
          const int pa = nodes.parent[ca], pb = nodes.parent[cb];
          more = (pa != ca) || (pb != cb);
          ca = pa;
          cb = pb;
        }
        consumed += t;
        WaveSync();   // the candidates are in registers: ring slots and stage slots may be reused
        if (lane == 0 && t > 0) {
          __hip_atomic_store(&queue.consumed, consumed, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const bool live = cand && ca != cb;
        const unsigned long long lm = __ballot(live);
        if (live) {
          const int pos = n_valid + (int)__popcll(lm & ((1ull << lane) - 1ull));
          stage.ra[pos] = ca;
          stage.rb[pos] = cb;
          stage.gpos[pos] = cg;
        }
        if (kDbg && lane == 0) {
          dbg_taken += (unsigned)(n_raw + t);
          dbg_live += (unsigned)__popcll(lm);
        }
        n_valid += (int)__popcll(lm);
        n_raw = 0;
        WaveSync();
      }
      if (n_valid == 0) {
        if (drained) break;
        continue;
      }
      const unsigned long long bt0b = Clock();
      cyc_wait += bt0b - bt0;
      // ---- the batch: the first 64 staged edges (their roots are current) ------------------------
      const int take = n_valid < 64 ? n_valid : 64;
      const bool valid = lane < take;
      int ra = -1, rb = -1;
      uint32_t gpos = 0;
      if (valid) {
        ra = stage.ra[lane];
        rb = stage.rb[lane];
        gpos = stage.gpos[lane];
      }
      n_raw = n_valid - take;
      if (n_raw > 0) {   // move the rest to the front; it is re-validated by the next pass
        int xa = 0, xb = 0;
        uint32_t xg = 0;
        if (lane < n_raw) {
          xa = stage.ra[64 + lane];
          xb = stage.rb[64 + lane];
          xg = stage.gpos[64 + lane];
        }
        WaveSync();
        if (lane < n_raw) {
          stage.ra[lane] = xa;
          stage.rb[lane] = xb;
          stage.gpos[lane] = xg;
        }
      }
      WaveSync();
      bool pending = valid;
      int sa = 0, sb = 0;     // table slots of the current roots of the two end regions
      int mine_a = -1, mine_b = -1;   // slots this lane inserted (it writes them back and frees them)
      if (pending) {
        const RState A = LoadState(nodes, ra), B = LoadState(nodes, rb);   // both in flight
        bool ins_a, ins_b;
        sa = TabInsert(tab, ra, ins_a);
        sb = TabInsert(tab, rb, ins_b);
        if (ins_a) {
          tab.link[sa] = sa;
          TabStore(tab, sa, A, 0);
        }
        if (ins_b) {
          tab.link[sb] = sb;
          TabStore(tab, sb, B, 0);
        }
        atomicAdd(&tab.cnt[sa], 1);
        atomicAdd(&tab.cnt[sb], 1);
        if (ins_a) mine_a = sa;
        if (ins_b) mine_b = sb;
      }
      WaveSync();
      int hot = -1;   // wave-uniform slot of the hot region
      {
        int best = 0;
        if (pending) best = max((tab.cnt[sa] << 8) | sa, (tab.cnt[sb] << 8) | sb);
        best = WaveMax(best);
        if ((best >> 8) >= 3 && !(kDbg && (dbg_flags & 4))) hot = best & (kTabSize - 1);
      }
      if (kDbg && lane == 0) ++dbg_batches;
      const unsigned long long bt1 = Clock();
      cyc_load += bt1 - bt0b;

      bool my_kept = false;
      bool failed = false;    // this lane's chain test failed: replay it with the generic code
      for (unsigned round = 0;; ++round) {
        {   // current root slots (both ends and the hot region advance together)
          int h = hot;
          for (bool more = true; more;) {
            int pa = sa, pb = sb, ph = h;
            if (pending) {
              pa = tab.link[sa];
              pb = tab.link[sb];
            }
            if (h >= 0) ph = tab.link[h];
            more = (pa != sa) || (pb != sb) || (ph != h);
            sa = pa;
            sb = pb;
            h = ph;
          }
          hot = h;
          if (pending && sa == sb) pending = false;   // became internal
        }
        if (!__ballot(pending)) break;
        if (round > 140u) {   // cannot happen (the earliest pending lane commits, after at most one failed chain test): report
          if (lane == 0) atomicAdd(&stats[22], 1ull);
          break;
        }
        const unsigned long long ph0 = Clock();
        const bool a_hot = (sa == hot), b_hot = (sb == hot);
        const uint32_t key = ((0xfffffu - round) << 6) | (uint32_t)lane;
        if (pending) {
          if (!a_hot) atomicMin(&tab.res[sa], key);
          if (!b_hot) atomicMin(&tab.res[sb], key);
        }
        WaveSync();
        uint32_t res_a = 0, res_b = 0;
        RState A = {}, B = {};
        if (pending) {
          res_a = tab.res[sa];
          res_b = tab.res[sb];
          A = TabLoad(tab, sa);
          B = TabLoad(tab, sb);
        }
        // own_x: this lane is the earliest pending edge on region x (the hot region is not reserved)
        const unsigned long long ph1 = Clock();
        const bool own_a = pending && !a_hot && res_a == key;
        const bool own_b = pending && !b_hot && res_b == key;
        const int oa = (int)(res_a & 63u), ob = (int)(res_b & 63u);   // owners (earlier lanes)
        RState Hs = {}, P = {};
        int ps = 0;            // partner slot of a chain lane
        bool hot_lane = pending && (a_hot || b_hot);
        bool elig = false;     // chain lane
        bool both = false;     // both ends (will) belong to the hot region: internal once committed
        bool merging = false;
        bool case_s = false;
        bool fin = false;
        // A chain starts at the first edge that touches the hot region and only if that lane is the
        // earliest pending edge on its other end; otherwise nothing can be absorbed in this round
        // and the classification below is skipped (the other hot edges just wait).
        const unsigned long long lit_mask = __ballot(hot_lane);
        const bool chain_possible =
            lit_mask != 0 && ((__ballot(hot_lane && (own_a || own_b)) >> __builtin_ctzll(lit_mask)) & 1ull);
        if (chain_possible) {
          Hs = TabLoad(tab, hot);   // uniform
          fin = (Hs.flags & kFlagFinalized) != 0;
          const bool mode_ok = !(Hs.flags & kFlagNoDesc) && (!fin || Hs.sz >= T.min_size) &&
                               !(kDbg && (dbg_flags & 1));
          // A region is *effectively hot* for a lane when it is the hot region or when its owner
          // (an earlier lane) is a chain lane that absorbs it into the hot region: by the time this
          // lane is replayed the region is part of the hot one.  So a run of edges p1-p2, p2-p3, ...
          // hanging off the hot region joins the chain in one round.  The set of absorbing lanes
          // only grows, so the loop ends (no memory access inside).
          // Per end, evaluated once: would this end qualify as the partner of a chain edge
          // (plain, smaller, owned by this lane) and would the edge merge?
          // Case S (partner with the hot region's constraint) merges unless the descriptors are
          // further apart than the split threshold, whatever the sizes and flags; Case U
          // (unconstrained partner): regular test while the hot region is not finalized, a finalized
          // hot region (>= min size) absorbs small partners only.
          const bool base = pending && mode_ok && !failed;
          const bool part_a = base && own_a && A.flags == 0 && (A.cons < 0 || A.cons == Hs.cons) &&
                              A.sz < Hs.sz;
          const bool part_b = base && own_b && B.flags == 0 && (B.cons < 0 || B.cons == Hs.cons) &&
                              B.sz < Hs.sz;
          const bool merge_a = part_a && (A.cons >= 0 || !fin || A.sz < T.min_size);
          const bool merge_b = part_b && (B.cons >= 0 || !fin || B.sz < T.min_size);
          const bool abs_a = pending && !own_a, abs_b = pending && !own_b;   // may be absorbed
          unsigned long long em = 0;   // chain lanes that merge
          bool ea, eb;
          for (;;) {
            ea = a_hot || (abs_a && ((em >> oa) & 1ull));
            eb = b_hot || (abs_b && ((em >> ob) & 1ull));
            const unsigned long long em2 = __ballot((ea && !eb && merge_b) || (eb && !ea && merge_a));
            if (em2 == em) break;
            em = em2;
          }
          hot_lane = pending && (ea || eb);
          both = hot_lane && ea && eb;
          const bool pb_side = ea;   // the partner is the end that is not effectively hot
          P.d0 = pb_side ? B.d0 : A.d0;
          P.d1 = pb_side ? B.d1 : A.d1;
          P.d2 = pb_side ? B.d2 : A.d2;
          P.sz = pb_side ? B.sz : A.sz;
          P.cons = pb_side ? B.cons : A.cons;
          P.flags = 0;
          ps = pb_side ? sb : sa;
          elig = hot_lane && !both && (pb_side ? part_b : part_a);
          merging = hot_lane && !both && (pb_side ? merge_b : merge_a);
          case_s = P.cons >= 0;
        }
        const unsigned long long ph2 = Clock();
        const unsigned long long hot_mask = __ballot(hot_lane);
        const unsigned long long elig_mask = __ballot(elig);
        // hot lanes that are neither chain lanes nor internal end the chain
        const unsigned long long blocked = hot_mask & ~(elig_mask | __ballot(both));
        const unsigned long long prefix =
            blocked ? ((1ull << __builtin_ctzll(blocked)) - 1ull) : ~0ull;
        unsigned long long chain_mask = elig_mask & prefix;
        if (kDbg && (dbg_flags & 32)) {   // no jumping over earlier pending lanes
          const unsigned long long others = __ballot(pending) & ~chain_mask;
          if (others) chain_mask &= (1ull << __builtin_ctzll(others)) - 1ull;
        }
        if ((kDbg && (dbg_flags & 64)) && chain_mask) chain_mask = 1ull << __builtin_ctzll(chain_mask);
        // The first hot lane, when it is no chain lane, is replayed alone by the generic code (it
        // touches the hot region itself: nothing earlier can have absorbed one of its ends).
        const bool own = pending && (a_hot || own_a) && (b_hot || own_b);
        const bool solo = hot_lane && own && !elig && !both &&
                          lane == (int)__builtin_ctzll(hot_mask | (1ull << 63));
        bool n_win = pending && own && (!hot_lane || solo);
        if (kDbg && (dbg_flags & 8)) n_win = n_win && lane == (int)__builtin_ctzll(__ballot(pending));
        if constexpr (kDbg) {
          const unsigned long long nwin_mask = __ballot(n_win), solo_mask = __ballot(solo);
          const unsigned long long pend_mask = __ballot(pending);
          if (lane == 0) {
            ++dbg_rounds;
            dbg_nwin += (unsigned)__popcll(nwin_mask);
            dbg_solo += (unsigned)__popcll(solo_mask);
            dbg_x[0] += (unsigned)__popcll(pend_mask);                 // pending lanes per round
            dbg_x[1] += (unsigned)__popcll(hot_mask);                  // (effectively) hot lanes
            dbg_x[2] += (unsigned)__popcll(blocked);                   // hot lanes that end the chain
            dbg_x[3] += (chain_mask != 0);                             // rounds with a chain
            dbg_x[4] += (nwin_mask != 0);                              // rounds with generic commits
            dbg_x[5] += (unsigned)__popcll(pend_mask & ~hot_mask & ~nwin_mask);   // waiting non-hot lanes
          }
        }

        const unsigned long long ph3 = Clock();
        // ---- lanes that own both regions: generic edge ------------------------------------------
        if (n_win) {
          const RState& s1 = A;
          const RState& s2 = B;
          // Fast path, by far the most common generic edge: two plain regions (unconstrained,
          // not finalized, unmarked) that pass the regular test.  Same arithmetic as
          // DecideEdge / MergeStates for this case.
          if (s1.cons < 0 && s2.cons < 0 && (s1.flags | s2.flags) == 0 &&
              SquaredDistance(s1, s2) <= T.pass_s && !(kDbg && (dbg_flags & 256))) {
            const bool first = s1.sz > s2.sz;   // ties keep region 2
            const int ws = first ? sa : sb, ls = first ? sb : sa;
            RState m, o;
            m.d0 = first ? s1.d0 : s2.d0;
            m.d1 = first ? s1.d1 : s2.d1;
            m.d2 = first ? s1.d2 : s2.d2;
            m.sz = first ? s1.sz : s2.sz;
            o.d0 = first ? s2.d0 : s1.d0;
            o.d1 = first ? s2.d1 : s1.d1;
            o.d2 = first ? s2.d2 : s1.d2;
            o.sz = first ? s2.sz : s1.sz;
            const float denom = 1.0f / (float)(o.sz + m.sz);
            const float ca = (float)o.sz * denom;
            const float cb = (float)m.sz * denom;
            m.d0 = ca * o.d0 + cb * m.d0;
            m.d1 = ca * o.d1 + cb * m.d1;
            m.d2 = ca * o.d2 + cb * m.d2;
            m.sz += o.sz;
            m.cons = max(s1.cons, s2.cons);
            m.flags = 0;
            TabStore(tab, ws, m, kTabDirty);
            CommitLoser(tab, nodes, ls, ws);
            ++n_regular;
            pending = false;
            n_win = false;
          }
        }
        if (n_win) {
          RState s1 = A, s2 = B;
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
            my_kept = true;
            if (!SameState(o1, s1)) TabStore(tab, sa, s1, kTabDirty);
            if (!SameState(o2, s2)) TabStore(tab, sb, s2, kTabDirty);
          } else if (out == kOutMerge1) {
            TabStore(tab, sa, s1, kTabDirty);
            CommitLoser(tab, nodes, sb, sa);
          } else {
            TabStore(tab, sb, s2, kTabDirty);
            CommitLoser(tab, nodes, sa, sb);
          }
          pending = false;
        }

        const unsigned long long ph4 = Clock();
        // ---- the chain on the hot region -----------------------------------------------------
        if (chain_mask) {
          const bool in_chain = (chain_mask >> lane) & 1ull;
          merging = in_chain && merging;
          case_s = in_chain && case_s;
          const bool tested = case_s || (in_chain && !fin);
          const int v = merging ? P.sz : 0;
          const int incl = WaveInclusiveSum(v);
          const int S = Hs.sz + incl - v;     // size of the hot region before this lane's merge
          // MergeStates with o = partner, m = hot region
          const float denom = 1.0f / (float)(P.sz + S);
          const float ca = (float)P.sz * denom;
          const float cb = (float)S * denom;
          const float t0 = ca * P.d0, t1 = ca * P.d1, t2 = ca * P.d2;
          float h0 = Hs.d0, h1 = Hs.d1, h2 = Hs.d2;
          float r0 = 0.f, r1 = 0.f, r2 = 0.f;   // hot mean before this lane's merge
          const unsigned long long merging_mask = __ballot(merging);
          for (unsigned long long mm = merging_mask; mm; mm &= mm - 1) {
            const int k = (int)__builtin_ctzll(mm);
            if (lane == k) {
              r0 = h0;
              r1 = h1;
              r2 = h2;
            }
            const float cbk = ReadLaneF(cb, k);
            h0 = ReadLaneF(t0, k) + cbk * h0;
            h1 = ReadLaneF(t1, k) + cbk * h1;
            h2 = ReadLaneF(t2, k) + cbk * h2;
          }
          unsigned long long fail = 0;
          {
            const float x = r0 - P.d0, y = r1 - P.d1, z = r2 - P.d2;
            const float sd = (x * x + y * y + z * z) * (1.0f / 3.0f);
            const bool pass = case_s ? !(sd > T.split_s) : (sd <= T.pass_s);
            fail = __ballot(tested && !pass);
          }
          int fcut = 64;
          RState Hn = Hs;
          if (fail) {
            fcut = (int)__builtin_ctzll(fail);
            if (lane == fcut) failed = true;
            Hn.d0 = ReadLaneF(r0, fcut);
            Hn.d1 = ReadLaneF(r1, fcut);
            Hn.d2 = ReadLaneF(r2, fcut);
            Hn.sz = ReadLaneI(S, fcut);
            if (kDbg && lane == 0) ++dbg_cut;
          } else {
            Hn.d0 = h0;
            Hn.d1 = h1;
            Hn.d2 = h2;
            Hn.sz = Hs.sz + ReadLaneI(incl, 63);
          }
          const unsigned long long below = (fcut < 64) ? ((1ull << fcut) - 1ull) : ~0ull;
          const bool do_commit = in_chain && lane < fcut;
          if constexpr (kDbg) if (dbg_flags & 16) {   // self check: replay the committed chain with DecideEdge
            RState Hc = Hs;
            unsigned bad = 0;
            for (unsigned long long mm = chain_mask & below; mm; mm &= mm - 1) {
              const int k = (int)__builtin_ctzll(mm);
              RState a = Hc, b = ReadLaneState(P, k);
              const RState b0 = b;
              int st;
              const int out = DecideEdge(a, b, T, st);
              const bool km = (merging_mask >> k) & 1ull;
              if (km) {
                if (out != kOutMerge1 || st != (b0.cons >= 0 ? 1 : (fin ? 3 : 2))) ++bad;
                Hc = a;
              } else {
                if (out != kOutKeep || !SameState(a, Hc) || !SameState(b, b0)) ++bad;
              }
            }
            if (__float_as_int(Hc.d0) != __float_as_int(Hn.d0) || __float_as_int(Hc.d1) != __float_as_int(Hn.d1) ||
                __float_as_int(Hc.d2) != __float_as_int(Hn.d2) || !SameState(Hc, Hn)) ++bad;
            if (fail) {
              RState a = Hc, b = ReadLaneState(P, fcut);
              int st;
              DecideEdge(a, b, T, st);
              if (st == 2 || st == 1) ++bad;
            }
            if (lane == 0 && bad) atomicAdd(&stats[23], (unsigned long long)bad);
          }
          if (do_commit) {
            if (merging) {
              CommitLoser(tab, nodes, ps, hot);
              if (case_s) ++n_forced; else if (fin) ++n_small; else ++n_regular;
            } else {
              my_kept = true;   // both regions large, the hot one finalized: nothing changes
            }
            pending = false;
          }
          // An edge with both ends (by then) inside the hot region is internal: every lane that
          // absorbs one of its ends is an earlier chain lane, committed if this lane is below the
          // cut.
          // (the debug modes that shorten the chain leave these lanes to the next round's internal test)
          if (both && lane < fcut && ((prefix >> lane) & 1ull) && !(kDbg && (dbg_flags & (32 | 64)))) {
            pending = false;
          }
          if (lane == 0) {
            if (kDbg) dbg_chain += (unsigned)__popcll(merging_mask & below);
            if (merging_mask & below) TabStore(tab, hot, Hn, kTabDirty);
          }
        }
        if constexpr (kDbg) {
          const unsigned long long ph5 = Clock();
          cyc_ph[0] += ph1 - ph0;
          cyc_ph[1] += ph2 - ph1;
          cyc_ph[2] += ph3 - ph2;
          cyc_ph[3] += ph4 - ph3;
          cyc_ph[4] += ph5 - ph4;
        }
        if (!__ballot(pending)) break;   // nothing left: skip the next round's root resolution
        WaveSync();
      }
      WaveSync();

      if (valid && my_kept) kept_all[gpos] = 1;
      // ---- write the changed regions back, reset the table ---------------------------------------
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int s = e ? mine_b : mine_a;
        if (s >= 0) {
          if (tab.link[s] == s && (tab.flags[s] & kTabDirty)) {
            StoreState(nodes, tab.key[s], TabLoad(tab, s));
          }
          tab.key[s] = -1;
          tab.res[s] = 0xffffffffu;
          tab.cnt[s] = 0;
        }
      }
      // Make this batch's stores visible to the next batch's loads (same CU: L1 is shared).
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      WaveSync();
      cyc_loop += Clock() - bt1;
    }
    if (kDbg && lane == 0) {
      atomicMax(&stats[16], Clock() - seg_t0);            // slowest component
      atomicMax(&stats[17], (unsigned long long)cnt);     // largest component
    }
    __syncthreads();   // end of the segment (matches the producer's)
  }
  if (producer) {
    if (kDbg && lane == 0) {
      atomicAdd(&stats[27], cyc_load);   // producer: reading + root searches
      atomicAdd(&stats[28], cyc_wait);   // producer: ring full
    }
    return;
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_forced += __shfl_down(n_forced, off);
    n_regular += __shfl_down(n_regular, off);
    n_small += __shfl_down(n_small, off);
  }
  if (lane == 0) {
    if (n_forced) atomicAdd(&stats[0], (unsigned long long)n_forced);
    if (n_regular) atomicAdd(&stats[1], (unsigned long long)n_regular);
    if (n_small) atomicAdd(&stats[2], (unsigned long long)n_small);
  }
  if (kDbg && lane == 0) {
    atomicAdd(&stats[4], (unsigned long long)dbg_nwin);
    atomicAdd(&stats[5], (unsigned long long)dbg_rounds);
    atomicAdd(&stats[6], (unsigned long long)dbg_solo);
    atomicAdd(&stats[7], (unsigned long long)dbg_batches);
    atomicAdd(&stats[18], cyc_load);
    atomicAdd(&stats[26], cyc_wait);   // consumer: ring empty
    atomicAdd(&stats[19], cyc_loop);
    atomicAdd(&stats[20], (unsigned long long)dbg_chain);
    atomicAdd(&stats[21], (unsigned long long)dbg_cut);
    atomicAdd(&stats[29], dbg_taken);
    for (int k = 0; k < 5; ++k) atomicAdd(&stats[32 + k], cyc_ph[k]);
    for (int k = 0; k < 6; ++k) atomicAdd(&stats[38 + k], dbg_x[k]);
    atomicAdd(&stats[30], dbg_live);
  }
}

// ------------------------------------------------------------------------------------------
// Worker C: one WORKGROUP of four wavefronts replays one large component, 256 edges per batch.
// ------------------------------------------------------------------------------------------
// Same algorithm as k_merge_wave (reservations, transitive chain on the batch's hot region, see
// above), with a batch of 256 staged live edges spread over four wavefronts: the fixed cost of a
// batch and of a round is shared by four times the edges, four SIMDs issue the per-lane work, and
// the dependency chains of what would be neighbouring 64-edge batches overlap (tools/sched_sim.cpp:
// 1.9 instead of 4.25 rounds per 64 live edges on the 960x540 trace).  Wave-wide ballots become
// four 64-bit words exchanged through LDS, wave-local ordering becomes workgroup barriers, the
// chain recurrence runs on wavefront 0 with the per-lane inputs handed over through LDS.  There
// is no reader wave (it could not take part in a data-dependent number of barriers): the
// wavefronts read the component's edges themselves, two per thread and pass.
constexpr int kBlk = 256;
constexpr int kBTab = 1024;      // region table slots (<= 512 regions of a batch, load factor 1/2)
constexpr int kBStage = 768;     // staged live edges: < 256 left over + 512 of a pass

struct BlockShared {
  int32_t key[kBTab];
  int32_t link[kBTab];
  uint32_t res[kBTab];
  int32_t cnt[kBTab];
  float4 ds[kBTab];
  int32_t cons[kBTab];
  int32_t flags[kBTab];
  int32_t st_ra[kBStage];
  int32_t st_rb[kBStage];
  uint32_t st_gpos[kBStage];
  unsigned long long bm[4][4];   // ballot exchange (rotating slots)
  unsigned long long bmn[2][4][4];   // multi-predicate exchange (two alternating slots)
  float c_p0[kBlk], c_p1[kBlk], c_p2[kBlk];   // partner means by lane
  int32_t c_v[kBlk], c_incl[kBlk];            // partner size of merging lanes, wave-local prefix
  float c_r0[kBlk], c_r1[kBlk], c_r2[kBlk];               // hot mean before the lane's merge
  int32_t c_S[kBlk];                                      // hot size before the lane's merge
  float h_fin[4];
  int32_t wsum[4];
  int32_t rewind;
  int32_t fl_slot;   // slot of the first hot lane's other end (see the round loop)
};

struct Mask256 {
  unsigned long long w[4];
};
__device__ __forceinline__ bool MaskAny(const Mask256& m) { return (m.w[0] | m.w[1] | m.w[2] | m.w[3]) != 0; }
__device__ __forceinline__ bool MaskEq(const Mask256& a, const Mask256& b) {
  return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && a.w[3] == b.w[3];
}
__device__ __forceinline__ bool MaskBit(const Mask256& m, int i) {   // i in [0, 256)
  const int k = i >> 6;
  const unsigned long long w = k == 0 ? m.w[0] : k == 1 ? m.w[1] : k == 2 ? m.w[2] : m.w[3];
  return (w >> (i & 63)) & 1ull;
}
__device__ __forceinline__ int MaskFirst(const Mask256& m) {   // 256 if empty
  for (int k = 0; k < 4; ++k) {
    if (m.w[k]) return k * 64 + (int)__builtin_ctzll(m.w[k]);
  }
  return 256;
}
__device__ __forceinline__ int MaskCount(const Mask256& m) {
  return (int)(__popcll(m.w[0]) + __popcll(m.w[1]) + __popcll(m.w[2]) + __popcll(m.w[3]));
}
// number of set bits below position g
__device__ __forceinline__ int MaskRank(const Mask256& m, int g) {
  const int k = g >> 6;
  int r = 0;
  for (int j = 0; j < 4; ++j) {
    if (j < k) r += (int)__popcll(m.w[j]);
  }
  const unsigned long long w = k == 0 ? m.w[0] : k == 1 ? m.w[1] : k == 2 ? m.w[2] : m.w[3];
  return r + (int)__popcll(w & ((1ull << (g & 63)) - 1ull));
}

// Workgroup ballot: one barrier.  `slot` is a per-thread counter that advances identically in
// every thread; a slot is rewritten two calls later at the earliest, i.e. behind a barrier every
// wave can only reach after it has read the slot.
__device__ __forceinline__ Mask256 BlockBallot(BlockShared& sh, int& slot, bool pred) {
  const unsigned long long m = __ballot(pred);
  const int s = slot & 3;
  ++slot;
  if ((threadIdx.x & 63) == 0) sh.bm[s][threadIdx.x >> 6] = m;
  __syncthreads();
  Mask256 r;
  r.w[0] = sh.bm[s][0];
  r.w[1] = sh.bm[s][1];
  r.w[2] = sh.bm[s][2];
  r.w[3] = sh.bm[s][3];
  return r;
}

__device__ __forceinline__ int BTabInsert(BlockShared& t, int r, bool& inserted) {
  unsigned h = ((unsigned)r * 2654435761u) >> 22;
  for (;;) {
    const int old = atomicCAS(&t.key[h], -1, r);
    if (old == -1) { inserted = true; return (int)h; }
    if (old == r) { inserted = false; return (int)h; }
    h = (h + 1) & (kBTab - 1);
  }
}
__device__ __forceinline__ RState BTabLoad(const BlockShared& t, int s) {
  const float4 ds = t.ds[s];
  RState r;
  r.d0 = ds.x;
  r.d1 = ds.y;
  r.d2 = ds.z;
  r.sz = __float_as_int(ds.w);
  r.cons = t.cons[s];
  r.flags = t.flags[s] & 0xff;
  return r;
}
__device__ __forceinline__ void BTabStore(BlockShared& t, int s, const RState& r, int dirty) {
  t.ds[s] = make_float4(r.d0, r.d1, r.d2, __int_as_float(r.sz));
  t.cons[s] = r.cons;
  t.flags[s] = r.flags | dirty;
}
__device__ __forceinline__ void BCommitLoser(BlockShared& t, const NodeArrays& nodes, int ls, int ws) {
  t.link[ls] = ws;
  const int lid = t.key[ls];
  nodes.parent[lid] = t.key[ws];
  if (t.flags[ls] & kTabDirty) nodes.cons[lid] = t.cons[ls];   // see CommitLoser
}

// Workgroup ballot of up to four predicates with ONE barrier.
template <int N>
__device__ __forceinline__ void BlockBallotN(BlockShared& sh, int& slot, const bool (&pred)[N],
                                             Mask256 (&out)[N]) {
  const int s = slot & 1;
  ++slot;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned long long m = __ballot(pred[i]);
    if ((threadIdx.x & 63) == 0) sh.bmn[s][i][threadIdx.x >> 6] = m;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[i].w[k] = sh.bmn[s][i][k];
  }
}

// One staging pass of the block worker: R candidates per thread (the left-overs of the previous
// batch first, then edges read from the component's list), all root searches in flight together;
// the live candidates are packed in order behind the n_valid edges already staged.  When the stage
// is full the pass is cut there and `next` is rewound to the first edge that did not fit.
// Returns the fraction of live candidates in 1/256 units (the caller picks R for the next pass).
template <int R>
__device__ __forceinline__ int StagePass(BlockShared& sh, int& slot, const NodeArrays& nodes,
                                         const int32_t* __restrict__ s_ra,
                                         const int32_t* __restrict__ s_rb,
                                         const uint32_t* __restrict__ s_gpos, int end, int optimistic,
                                         int& next, int& n_raw, int& n_valid) {
  const int g = threadIdx.x;
  const int t_new = min(R * kBlk - n_raw, end - next);
  const int total = n_raw + t_new;
  const int next_base = next;
  int ca[R], cb[R], xa[R], xb[R];
  uint32_t cg[R];
  bool cand[R], fresh[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int i = g + k * kBlk;
    cand[k] = i < total;
    fresh[k] = cand[k] && i >= n_raw;
    ca[k] = cb[k] = 0;
    cg[k] = 0;
    if (cand[k] && !fresh[k]) {
      ca[k] = sh.st_ra[i];
      cb[k] = sh.st_rb[i];
      cg[k] = sh.st_gpos[i];
    } else if (fresh[k]) {
      const int p = next_base + (i - n_raw);
      ca[k] = s_ra[p];
      cb[k] = s_rb[p];
      cg[k] = s_gpos[p];
    }
    xa[k] = ca[k];
    xb[k] = cb[k];
  }
  for (bool more = true; more;) {   // the 2 R root searches of a thread advance together
    int pa[R], pb[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      pa[k] = nodes.parent[ca[k]];
      pb[k] = nodes.parent[cb[k]];
    }
    more = false;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      more = more || (pa[k] != ca[k]) || (pb[k] != cb[k]);
      ca[k] = pa[k];
      cb[k] = pb[k];
    }
  }
  if (!optimistic) {   // path compression of the start nodes (never representatives)
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (fresh[k] && ca[k] != xa[k]) nodes.parent[xa[k]] = ca[k];
      if (fresh[k] && cb[k] != xb[k]) nodes.parent[xb[k]] = cb[k];
    }
  }
  if (g == 0) sh.rewind = -1;
  int base = n_valid, live_total = 0;
#pragma unroll
  for (int k0 = 0; k0 < R; k0 += 2) {
    Mask256 lm[2];
    const bool pr[2] = {cand[k0] && ca[k0] != cb[k0], cand[k0 + 1] && ca[k0 + 1] != cb[k0 + 1]};
    BlockBallotN<2>(sh, slot, pr, lm);   // first barrier: every candidate is in registers
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + j;
      if (pr[j]) {
        const int pos = base + MaskRank(lm[j], g);
        if (pos < kBStage) {
          sh.st_ra[pos] = ca[k];
          sh.st_rb[pos] = cb[k];
          sh.st_gpos[pos] = cg[k];
        } else if (pos == kBStage) {
          // the first edge that does not fit (a fresh one: the left-overs always fit)
          sh.rewind = next_base + (g + k * kBlk - n_raw);
        }
      }
      const int c = MaskCount(lm[j]);
      base += c;
      live_total += c;
    }
  }
  __syncthreads();
  const int rw = sh.rewind;
  next = rw >= 0 ? rw : next_base + t_new;
  n_valid = base < kBStage ? base : kBStage;
  n_raw = 0;
  __syncthreads();   // sh.rewind is rewritten by the next pass
  return total > 0 ? (live_total * 256) / total : 256;
}

__global__ __launch_bounds__(256) void k_merge_block(const int32_t* __restrict__ num_segs,
                                                      const int32_t* __restrict__ seg_off,
                                                      const int32_t* __restrict__ seg_cnt,
                                                      const int32_t* __restrict__ s_ra,
                                                      const int32_t* __restrict__ s_rb,
                                                      const uint32_t* __restrict__ s_gpos,
                                                      NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                      StageThr T, int optimistic,
                                                      int32_t* __restrict__ violation,
                                                      unsigned long long* __restrict__ stats,
                                                      int dbg_flags) {
  __shared__ BlockShared sh;
  const int g = threadIdx.x;          // lane of the batch
  const int lane = g & 63, wave = g >> 6;
  for (int s = g; s < kBTab; s += kBlk) {
    sh.key[s] = -1;
    sh.res[s] = 0xffffffffu;
    sh.cnt[s] = 0;
  }
  __syncthreads();
  int slot = 0;   // BlockBallot slot counter
  const int nseg = *num_segs;
  unsigned n_forced = 0, n_regular = 0, n_small = 0;
  unsigned long long dbg_rounds = 0, dbg_batches = 0, dbg_generic = 0, dbg_chain = 0;
  unsigned long long cyc[6] = {0, 0, 0, 0, 0, 0};   // staging, table, round head, generic, chain, write-back
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int cnt = seg_cnt[seg];
    if (cnt <= kSmallSegment) continue;
    const int beg = seg_off[seg];
    const int end = beg + cnt;
    if (g == 0) atomicAdd(&stats[3], (unsigned long long)cnt);
    int next = beg;     // next edge of the component to be read (uniform)
    int live_frac = 256;   // live candidates of the last staging pass, in 1/256
    int n_raw = 0;      // staged edges left over from the previous batch (roots to be re-validated)
    for (;;) {
      // ---- stage up to 256 live edges ---------------------------------------------------------------
      const unsigned long long c0 = __builtin_readcyclecounter();
      int n_valid = 0;
      while (n_valid < kBlk && (n_raw > 0 || next < end)) {
        // eight candidates per thread when few of them are live (most of a grown component's
        // edges are internal), two otherwise
        live_frac = (live_frac < 77 && n_raw < kBlk)
                        ? StagePass<8>(sh, slot, nodes, s_ra, s_rb, s_gpos, end, optimistic, next, n_raw, n_valid)
                        : StagePass<2>(sh, slot, nodes, s_ra, s_rb, s_gpos, end, optimistic, next, n_raw, n_valid);
      }
      if (n_valid == 0) break;   // the component is drained
      const unsigned long long c1 = __builtin_readcyclecounter();
      cyc[0] += c1 - c0;
      // ---- the batch: the first 256 staged edges (their roots are current) ------------------------
      const int take = n_valid < kBlk ? n_valid : kBlk;
      const bool valid = g < take;
      int ra = -1, rb = -1;
      uint32_t gpos = 0;
      if (valid) {
        ra = sh.st_ra[g];
        rb = sh.st_rb[g];
        gpos = sh.st_gpos[g];
      }
      n_raw = n_valid - take;
      {   // move the rest to the front; it is re-validated by the next pass
        int xa[2] = {0, 0}, xb[2] = {0, 0};
        uint32_t xg[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = g + k * kBlk;
          if (i < n_raw) {
            xa[k] = sh.st_ra[kBlk + i];
            xb[k] = sh.st_rb[kBlk + i];
            xg[k] = sh.st_gpos[kBlk + i];
          }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = g + k * kBlk;
          if (i < n_raw) {
            sh.st_ra[i] = xa[k];
            sh.st_rb[i] = xb[k];
            sh.st_gpos[i] = xg[k];
          }
        }
      }
      bool pending = valid;
      int sa = 0, sb = 0;
      int mine_a = -1, mine_b = -1;
      if (pending) {
        const RState A0 = LoadState(nodes, ra), B0 = LoadState(nodes, rb);
        bool ins_a, ins_b;
        sa = BTabInsert(sh, ra, ins_a);
        sb = BTabInsert(sh, rb, ins_b);
        if (ins_a) {
          sh.link[sa] = sa;
          BTabStore(sh, sa, A0, 0);
          mine_a = sa;
        }
        if (ins_b) {
          sh.link[sb] = sb;
          BTabStore(sh, sb, B0, 0);
          mine_b = sb;
        }
        atomicAdd(&sh.cnt[sa], 1);
        atomicAdd(&sh.cnt[sb], 1);
      }
      __syncthreads();
      int hot = -1;   // block-uniform slot of the hot region
      {
        int best = 0;
        if (pending) best = max((sh.cnt[sa] << 10) | sa, (sh.cnt[sb] << 10) | sb);
        best = WaveMax(best);
        if (lane == 0) sh.wsum[wave] = best;
        __syncthreads();
        best = max(max(sh.wsum[0], sh.wsum[1]), max(sh.wsum[2], sh.wsum[3]));
        __syncthreads();
        if ((best >> 10) >= 3 && !(dbg_flags & 4)) hot = best & (kBTab - 1);
      }
      if (g == 0) ++dbg_batches;
      cyc[1] += __builtin_readcyclecounter() - c1;

      bool my_kept = false;
      bool failed = false;
      for (unsigned round = 0;; ++round) {
        const unsigned long long r0c = __builtin_readcyclecounter();
        {   // current root slots
          int h = hot;
          for (bool more = true; more;) {
            int pa = sa, pb = sb, ph = h;
            if (pending) {
              pa = sh.link[sa];
              pb = sh.link[sb];
            }
            if (h >= 0) ph = sh.link[h];
            more = (pa != sa) || (pb != sb) || (ph != h);
            sa = pa;
            sb = pb;
            h = ph;
          }
          hot = h;
          if (pending && sa == sb) pending = false;   // became internal
        }
        const bool a_hot = (sa == hot), b_hot = (sb == hot);
        bool hot_lane = pending && (a_hot || b_hot);
        // one exchange: who is still pending (loop exit) and which lanes touch the hot region
        Mask256 pl[2];
        {
          const bool pr[2] = {pending, hot_lane};
          BlockBallotN<2>(sh, slot, pr, pl);
        }
        if (!MaskAny(pl[0])) break;
        if (round > 600u) {   // cannot happen (the earliest pending lane commits): report
          if (g == 0) atomicAdd(&stats[22], 1ull);
          break;
        }
        const Mask256& lit = pl[1];
        const int first_lit = MaskFirst(lit);
        // A chain starts at the first edge that touches the hot region and only if that lane is the
        // earliest pending edge on its other end: it publishes that end's slot, everybody checks
        // the reservation word after the barrier.
        if (g == first_lit) sh.fl_slot = a_hot ? sb : sa;
        const uint32_t key = ((0xfffffu - round) << 8) | (uint32_t)g;
        if (pending) {
          if (!a_hot) atomicMin(&sh.res[sa], key);
          if (!b_hot) atomicMin(&sh.res[sb], key);
        }
        __syncthreads();
        uint32_t res_a = 0, res_b = 0;
        RState A = {}, B = {};
        if (pending) {
          res_a = sh.res[sa];
          res_b = sh.res[sb];
          A = BTabLoad(sh, sa);
          B = BTabLoad(sh, sb);
        }
        const bool own_a = pending && !a_hot && res_a == key;
        const bool own_b = pending && !b_hot && res_b == key;
        const int oa = (int)(res_a & 255u), ob = (int)(res_b & 255u);
        RState Hs = {}, P = {};
        int ps = 0;
        bool elig = false, both = false, merging = false, case_s = false, fin = false;
        const bool chain_possible =
            first_lit < 256 && !(dbg_flags & 1) &&
            sh.res[sh.fl_slot] == (((0xfffffu - round) << 8) | (uint32_t)first_lit);
        if (chain_possible) {
          Hs = BTabLoad(sh, hot);
          fin = (Hs.flags & kFlagFinalized) != 0;
          const bool mode_ok = !(Hs.flags & kFlagNoDesc) && (!fin || Hs.sz >= T.min_size);
          const bool base = pending && mode_ok && !failed;
          const bool part_a = base && own_a && A.flags == 0 && (A.cons < 0 || A.cons == Hs.cons) &&
                              A.sz < Hs.sz;
          const bool part_b = base && own_b && B.flags == 0 && (B.cons < 0 || B.cons == Hs.cons) &&
                              B.sz < Hs.sz;
          const bool merge_a = part_a && (A.cons >= 0 || !fin || A.sz < T.min_size);
          const bool merge_b = part_b && (B.cons >= 0 || !fin || B.sz < T.min_size);
          const bool abs_a = pending && !own_a, abs_b = pending && !own_b;
          Mask256 em = {{0, 0, 0, 0}};
          bool ea, eb;
          for (;;) {
            ea = a_hot || (abs_a && MaskBit(em, oa));
            eb = b_hot || (abs_b && MaskBit(em, ob));
            const Mask256 em2 =
                BlockBallot(sh, slot, (ea && !eb && merge_b) || (eb && !ea && merge_a));
            if (MaskEq(em2, em)) break;
            em = em2;
          }
          hot_lane = pending && (ea || eb);
          both = hot_lane && ea && eb;
          const bool pb_side = ea;
          P.d0 = pb_side ? B.d0 : A.d0;
          P.d1 = pb_side ? B.d1 : A.d1;
          P.d2 = pb_side ? B.d2 : A.d2;
          P.sz = pb_side ? B.sz : A.sz;
          P.cons = pb_side ? B.cons : A.cons;
          P.flags = 0;
          ps = pb_side ? sb : sa;
          elig = hot_lane && !both && (pb_side ? part_b : part_a);
          merging = hot_lane && !both && (pb_side ? merge_b : merge_a);
          case_s = P.cons >= 0;
        }
        // Without a chain the hot lanes are the literal ones and all of them wait.
        Mask256 hot_mask = lit, elig_mask = {{0, 0, 0, 0}}, mm_all = {{0, 0, 0, 0}};
        int cut = first_lit;   // first hot lane that ends the chain
        if (chain_possible) {
          Mask256 lb[4];
          const bool pr[4] = {hot_lane, elig, both, merging};
          BlockBallotN<4>(sh, slot, pr, lb);
          hot_mask = lb[0];
          elig_mask = lb[1];
          mm_all = lb[3];
          Mask256 blocked;
          for (int k = 0; k < 4; ++k) blocked.w[k] = lb[0].w[k] & ~(lb[1].w[k] | lb[2].w[k]);
          cut = MaskFirst(blocked);
        }
        const bool in_chain = elig && g < cut;
        const int first_hot = MaskFirst(hot_mask);
        const bool own = pending && (a_hot || own_a) && (b_hot || own_b);
        const bool solo = hot_lane && own && !elig && !both && g == first_hot;
        bool n_win = pending && own && (!hot_lane || solo);
        if (dbg_flags & 8) {
          const Mask256 pm = BlockBallot(sh, slot, pending);
          n_win = n_win && g == MaskFirst(pm);
        }
        if (g == 0) ++dbg_rounds;

        const unsigned long long r1c = __builtin_readcyclecounter();
        cyc[2] += r1c - r0c;
        // ---- lanes that own both regions: generic edge ------------------------------------------
        if (n_win) {
          if (A.cons < 0 && B.cons < 0 && (A.flags | B.flags) == 0 &&
              SquaredDistance(A, B) <= T.pass_s) {
            const bool first = A.sz > B.sz;   // ties keep region 2
            const int ws = first ? sa : sb, ls = first ? sb : sa;
            RState m, o;
            m.d0 = first ? A.d0 : B.d0;
            m.d1 = first ? A.d1 : B.d1;
            m.d2 = first ? A.d2 : B.d2;
            m.sz = first ? A.sz : B.sz;
            o.d0 = first ? B.d0 : A.d0;
            o.d1 = first ? B.d1 : A.d1;
            o.d2 = first ? B.d2 : A.d2;
            o.sz = first ? B.sz : A.sz;
            const float denom = 1.0f / (float)(o.sz + m.sz);
            const float ca = (float)o.sz * denom;
            const float cb = (float)m.sz * denom;
            m.d0 = ca * o.d0 + cb * m.d0;
            m.d1 = ca * o.d1 + cb * m.d1;
            m.d2 = ca * o.d2 + cb * m.d2;
            m.sz += o.sz;
            m.cons = max(A.cons, B.cons);
            m.flags = 0;
            BTabStore(sh, ws, m, kTabDirty);
            BCommitLoser(sh, nodes, ls, ws);
            ++n_regular;
            ++dbg_generic;
            pending = false;
            n_win = false;
          }
        }
        if (n_win) {
          RState s1 = A, s2 = B;
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
          ++dbg_generic;
          if (out == kOutKeep) {
            my_kept = true;
            if (!SameState(o1, s1)) BTabStore(sh, sa, s1, kTabDirty);
            if (!SameState(o2, s2)) BTabStore(sh, sb, s2, kTabDirty);
          } else if (out == kOutMerge1) {
            BTabStore(sh, sa, s1, kTabDirty);
            BCommitLoser(sh, nodes, sb, sa);
          } else {
            BTabStore(sh, sb, s2, kTabDirty);
            BCommitLoser(sh, nodes, sa, sb);
          }
          pending = false;
        }

        // ---- the chain on the hot region -----------------------------------------------------
        const unsigned long long r2c = __builtin_readcyclecounter();
        cyc[3] += r2c - r1c;
        // chain lanes = candidates below the cut (no exchange needed: both masks are known)
        Mask256 chain_mask, mm;
        for (int k = 0; k < 4; ++k) {
          const int lo = k * 64;
          const unsigned long long below_cut =
              cut >= lo + 64 ? ~0ull : (cut <= lo ? 0ull : ((1ull << (cut - lo)) - 1ull));
          chain_mask.w[k] = elig_mask.w[k] & below_cut;
          mm.w[k] = mm_all.w[k] & chain_mask.w[k];
        }
        if (MaskAny(chain_mask)) {
          merging = in_chain && merging;
          case_s = in_chain && case_s;
          const bool tested = case_s || (in_chain && !fin);
          // hand the per-lane inputs to wavefront 0: partner size and mean, wave-local size prefix
          const int v = merging ? P.sz : 0;
          const int incl = WaveInclusiveSum(v);
          sh.c_v[g] = v;
          sh.c_incl[g] = incl;
          sh.c_p0[g] = P.d0;
          sh.c_p1[g] = P.d1;
          sh.c_p2[g] = P.d2;
          if (lane == 63) sh.wsum[wave] = incl;
          __syncthreads();
          const int wtot = sh.wsum[0] + sh.wsum[1] + sh.wsum[2] + sh.wsum[3];
          if (wave == 0) {
            // MergeStates with o = partner, m = hot region: sizes are a prefix sum, the weights and
            // ca*p are lane-parallel, only h = ca*p + cb*h is replayed in lane order; the mean and
            // the size of the hot region before each merge are recorded for the verification.
            float h0 = Hs.d0, h1 = Hs.d1, h2 = Hs.d2;
            int woff = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int e = k * 64 + lane;
              const int ev = sh.c_v[e];
              const int S = Hs.sz + woff + sh.c_incl[e] - ev;   // hot size before this lane's merge
              const float denom = 1.0f / (float)(ev + S);
              const float ca = (float)ev * denom;
              const float x_cb = (float)S * denom;
              const float x_t0 = ca * sh.c_p0[e], x_t1 = ca * sh.c_p1[e], x_t2 = ca * sh.c_p2[e];
              sh.c_S[e] = S;
              float r0 = 0.f, r1 = 0.f, r2 = 0.f;
              for (unsigned long long w = mm.w[k]; w; w &= w - 1) {
                const int j = (int)__builtin_ctzll(w);
                if (lane == j) {
                  r0 = h0;
                  r1 = h1;
                  r2 = h2;
                }
                const float cbj = ReadLaneF(x_cb, j);
                h0 = ReadLaneF(x_t0, j) + cbj * h0;
                h1 = ReadLaneF(x_t1, j) + cbj * h1;
                h2 = ReadLaneF(x_t2, j) + cbj * h2;
              }
              sh.c_r0[e] = r0;
              sh.c_r1[e] = r1;
              sh.c_r2[e] = r2;
              woff += sh.wsum[k];
            }
            if (lane == 0) {
              sh.h_fin[0] = h0;
              sh.h_fin[1] = h1;
              sh.h_fin[2] = h2;
            }
          }
          __syncthreads();
          bool pass;
          {
            const float x = sh.c_r0[g] - P.d0, y = sh.c_r1[g] - P.d1, z = sh.c_r2[g] - P.d2;
            const float sd = (x * x + y * y + z * z) * (1.0f / 3.0f);
            pass = case_s ? !(sd > T.split_s) : (sd <= T.pass_s);
          }
          const Mask256 fail = BlockBallot(sh, slot, tested && !pass);
          const int fcut = MaskFirst(fail);
          if (g == fcut) failed = true;
          RState Hn = Hs;
          if (fcut < 256) {
            Hn.d0 = sh.c_r0[fcut];
            Hn.d1 = sh.c_r1[fcut];
            Hn.d2 = sh.c_r2[fcut];
            Hn.sz = sh.c_S[fcut];
          } else {
            Hn.d0 = sh.h_fin[0];
            Hn.d1 = sh.h_fin[1];
            Hn.d2 = sh.h_fin[2];
            Hn.sz = Hs.sz + wtot;
          }
          const bool do_commit = in_chain && g < fcut;
          if (do_commit) {
            if (merging) {
              BCommitLoser(sh, nodes, ps, hot);
              if (case_s) ++n_forced; else if (fin) ++n_small; else ++n_regular;
              ++dbg_chain;
            } else {
              my_kept = true;   // both regions large, the hot one finalized: nothing changes
            }
            pending = false;
          }
          // both ends (by then) inside the hot region: internal once the chain below it is committed
          if (both && g < fcut && g < cut) pending = false;
          bool any_merge = false;   // a merging lane below the failed test, if any
          for (int k = 0; k < 4; ++k) {
            const int lo = k * 64;
            const unsigned long long below_f =
                fcut >= lo + 64 ? ~0ull : (fcut <= lo ? 0ull : ((1ull << (fcut - lo)) - 1ull));
            any_merge = any_merge || (mm.w[k] & below_f) != 0;
          }
          if (g == 0 && any_merge) BTabStore(sh, hot, Hn, kTabDirty);
        }
        __syncthreads();
        cyc[4] += __builtin_readcyclecounter() - r2c;
      }
      const unsigned long long wb0 = __builtin_readcyclecounter();

      if (valid && my_kept) kept_all[gpos] = 1;
      // ---- write the changed regions back, free the table slots ---------------------------------------
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int s = e ? mine_b : mine_a;
        if (s >= 0) {
          if (sh.link[s] == s && (sh.flags[s] & kTabDirty)) StoreState(nodes, sh.key[s], BTabLoad(sh, s));
          sh.key[s] = -1;
          sh.res[s] = 0xffffffffu;
          sh.cnt[s] = 0;
        }
      }
      // Make this batch's stores visible to the next batch's loads (same CU: L1 is shared).
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __syncthreads();
      cyc[5] += __builtin_readcyclecounter() - wb0;
    }
    __syncthreads();   // end of the component
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_forced += __shfl_down(n_forced, off);
    n_regular += __shfl_down(n_regular, off);
    n_small += __shfl_down(n_small, off);
    dbg_generic += __shfl_down(dbg_generic, off);
    dbg_chain += __shfl_down(dbg_chain, off);
  }
  if (lane == 0) {
    if (n_forced) atomicAdd(&stats[0], (unsigned long long)n_forced);
    if (n_regular) atomicAdd(&stats[1], (unsigned long long)n_regular);
    if (n_small) atomicAdd(&stats[2], (unsigned long long)n_small);
    atomicAdd(&stats[4], dbg_generic);
    atomicAdd(&stats[20], dbg_chain);
  }
  if (g == 0) {
    atomicAdd(&stats[5], dbg_rounds);
    atomicAdd(&stats[7], dbg_batches);
    atomicAdd(&stats[18], cyc[0]);
    atomicAdd(&stats[26], cyc[1]);
    for (int k = 0; k < 4; ++k) atomicAdd(&stats[32 + k], cyc[2 + k]);
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
  // Active edges in component order; the scratch arrays of the earlier steps are free by now.
  int32_t* s_ra = reinterpret_cast<int32_t*>(S.a_comp);
  int32_t* s_rb = reinterpret_cast<int32_t*>(S.a_idx);
  uint32_t* s_gpos = reinterpret_cast<uint32_t*>(S.e_apos);
  hipLaunchKernelGGL(k_gather_sorted, dim3(Blocks(n_active)), dim3(256), 0, s, n_active, S.s_idx,
                     S.a_ra, S.a_rb, S.a_gpos, s_ra, s_rb, s_gpos);
  hipLaunchKernelGGL(k_merge_small, dim3(Blocks(n_active)), dim3(256), 0, s, S.num_segs, S.seg_off,
                     S.seg_cnt, s_ra, s_rb, s_gpos, nodes, kept_all, T,
                     optimistic ? 1 : 0, d_violation, S.stats);
  const int wave_grid = n_active / (kSmallSegment + 1) < 1 ? 1
                        : (n_active / (kSmallSegment + 1) > 8192 ? 8192
                                                                 : n_active / (kSmallSegment + 1));
  const int ew0 = NextEvent(S);
  if (ew0 >= 0) VSG_HIP(hipEventRecord((*S.ev_pool)[ew0], s));
  // Worker choice (both are exact): the four-wavefront worker pays off when the bucket's active
  // edges sit in few, large components that are mostly chains on one region (the fixed cost of a
  // batch is shared by four times the edges); many small clusters growing side by side are still
  // replayed faster by the one-wavefront worker (cheaper rounds).  S.block_worker: 0 never
  // (default), 1 always, 2 by the average component size of the bucket (VSG_BLOCK_WORKER).
  bool use_block = S.block_worker == 1;
  if (S.block_worker == 2 && n_active >= (1 << 20)) {
    int num_segs_host = 0;
    VSG_HIP(hipMemcpyAsync(&num_segs_host, S.num_segs, sizeof(int), hipMemcpyDeviceToHost, s));
    VSG_HIP(hipStreamSynchronize(s));
    use_block = num_segs_host > 0 && n_active / num_segs_host >= 64;
  }
  if (use_block) {
    hipLaunchKernelGGL(k_merge_block, dim3(wave_grid), dim3(256), 0, s, S.num_segs, S.seg_off,
                       S.seg_cnt, s_ra, s_rb, s_gpos, nodes, kept_all, T, optimistic ? 1 : 0,
                       d_violation, S.stats, S.wave_dbg);
  } else if (S.wave_v1) {
    hipLaunchKernelGGL(k_merge_wave_v1, dim3(wave_grid), dim3(64), 0, s, S.num_segs, S.seg_off,
                       S.seg_cnt, s_ra, s_rb, s_gpos, nodes, kept_all, T, optimistic ? 1 : 0,
                       d_violation, S.stats);
  } else if (S.wave_debug) {
    hipLaunchKernelGGL(k_merge_wave<true>, dim3(wave_grid), dim3(128), 0, s, S.num_segs, S.seg_off,
                       S.seg_cnt, s_ra, s_rb, s_gpos, nodes, kept_all, T, optimistic ? 1 : 0,
                       d_violation, S.stats, S.wave_dbg);
  } else {
    hipLaunchKernelGGL(k_merge_wave<false>, dim3(wave_grid), dim3(128), 0, s, S.num_segs, S.seg_off,
                       S.seg_cnt, s_ra, s_rb, s_gpos, nodes, kept_all, T, optimistic ? 1 : 0,
                       d_violation, S.stats, 0);
  }
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
