// readout_kernels.hip -- result read-out kernels (K7-K9), gfx950.
//
// Reference behaviour restated:
//   FlattenUnionFind + label image          segmentation_graph.h:596-629,
//                                            dense_segmentation_graph.h:512-518
//   EnforceN4Connectivity                    dense_segmentation_graph.h:1303-1337
//   run-length rasterisation                 dense_segmentation_graph.h:533-559
//   DetermineNeighborIdsImpl (edge walk)     segmentation_graph.h:466-496
#include "device_graph.h"

namespace vsg {

__device__ __forceinline__ int FindRO(const int32_t* __restrict__ parent, int x) {
  int p = parent[x];
  while (p != x) {
    x = p;
    p = parent[x];
  }
  return x;
}

// K7a: label_uf[i] = representative node of i.  (The reference re-creates representatives with
// ids >= N; only their identity matters, so the representative node id is used as the key.)
__global__ __launch_bounds__(256) void k_flatten(const int32_t* __restrict__ parent, size_t n,
                                                  int32_t* __restrict__ label_uf) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) label_uf[i] = FindRO(parent, (int)i);
}

void LaunchFlatten(NodeArrays nodes, size_t n, int32_t* label_uf, hipStream_t s) {
  hipLaunchKernelGGL(k_flatten, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nodes.parent, n,
                     label_uf);
  VSG_HIP(hipGetLastError());
}

// K7b: EnforceN4Connectivity.  The reference sweeps the image in raster order and rewrites the
// pixel *below* the current one, so row i+1 depends on the finished row i and, inside a row,
// pixel j depends on the rewritten below-left pixel j-1.  One workgroup owns a slice and walks its
// rows; inside a row all columns are evaluated in parallel from the previous iterate and
// re-evaluated until nothing changes.  The system is triangular (column j only depends on
// column j-1), so the fixed point is unique and equals the sequential result.
//
// Rewrites are rare (a pixel that hangs on its region by a diagonal only), and a row whose upper
// row was not rewritten and in which no pixel changes when evaluated on the original values stays
// as it is -- so k_n4_row_flags first marks, for all rows of all slices at once, the rows in which
// something would change given the ORIGINAL upper row, and the sweep only works on those rows and
// on the rows below a rewritten one; everything else it skips without touching the image (the
// sweep over all 1079 row pairs of a 1080p slice was 2.2 ms per chunk, one workgroup per slice).
constexpr int kN4Threads = 1024;

__global__ __launch_bounds__(256) void k_n4_row_flags(const int32_t* __restrict__ label_img, int W, int H,
                                                       const int32_t* __restrict__ frames,
                                                       int32_t* __restrict__ row_flags /* [slices][H] */) {
  const int i = blockIdx.x;       // upper row; the flag belongs to row i + 1
  const int32_t* img = label_img + (size_t)frames[blockIdx.y] * W * H;
  const int32_t* cur = img + (size_t)i * W;
  const int32_t* bel = cur + W;
  int changed = 0;
  for (int j = threadIdx.x; j < W; j += 256) {
    const int id = cur[j];
    const int left = (j > 0) ? cur[j - 1] : -1;
    const int right = (j < W - 1) ? cur[j + 1] : -1;
    const int bl = (j > 0) ? bel[j - 1] : -1;
    const int br = (j < W - 1) ? bel[j + 1] : -1;
    const int v = bel[j];
    changed |= (bl == id && left != id && v != id) || (br == id && right != id && v != id);
  }
  if (__syncthreads_or(changed) && threadIdx.x == 0) row_flags[(size_t)blockIdx.y * H + i + 1] = 1;
}

__global__ __launch_bounds__(kN4Threads) void k_enforce_n4(int32_t* __restrict__ label_img, int W,
                                                            int H, const int32_t* __restrict__ frames,
                                                            const int32_t* __restrict__ row_flags,
                                                            int32_t* __restrict__ adjust) {
  extern __shared__ int32_t srow[];   // cur[W], bel[W], it0[W], it1[W]
  int32_t* cur = srow;
  int32_t* bel = srow + W;
  int32_t* xa = srow + 2 * W;
  int32_t* xb = srow + 3 * W;
  const int tid = threadIdx.x;
  int32_t* img = label_img + (size_t)frames[blockIdx.x] * W * H;
  const int32_t* flags = row_flags + (size_t)blockIdx.x * H;
  bool have_cur = false;      // cur[] holds the (final) row i
  bool prev_rewritten = false;
  for (int i = 0; i < H - 1; ++i) {
    if (!prev_rewritten && !flags[i + 1]) {   // (uniform) nothing can change in row i + 1
      have_cur = false;
      continue;
    }
    int32_t* below = img + (size_t)(i + 1) * W;
    for (int j = tid; j < W; j += kN4Threads) {
      if (!have_cur) cur[j] = img[(size_t)i * W + j];
      const int v = below[j];
      bel[j] = v;
      xa[j] = v;
    }
    __syncthreads();
    int32_t* x_old = xa;
    int32_t* x_new = xb;
    for (;;) {
      int changed = 0;
      for (int j = tid; j < W; j += kN4Threads) {
        const int id = cur[j];
        const int left = (j > 0) ? cur[j - 1] : -1;
        const int right = (j < W - 1) ? cur[j + 1] : -1;
        const int bl = (j > 0) ? x_old[j - 1] : -1;      // below-left, already rewritten
        const int br = (j < W - 1) ? bel[j + 1] : -1;    // below-right, not yet visited
        int v = bel[j];
        if (bl == id && left != id && v != id) v = id;
        if (br == id && right != id && v != id) v = id;
        x_new[j] = v;
        changed |= (v != x_old[j]);
      }
      const int any = __syncthreads_or(changed);
      int32_t* t = x_old;
      x_old = x_new;
      x_new = t;
      if (!any) break;
    }
    int rewritten = 0;
    for (int j = tid; j < W; j += kN4Threads) {
      const int v = x_old[j];
      const int o = bel[j];
      if (v != o) {
        below[j] = v;
        atomicSub(&adjust[o], 1);
        atomicAdd(&adjust[v], 1);
        rewritten = 1;
      }
      cur[j] = v;
    }
    prev_rewritten = __syncthreads_or(rewritten) != 0;
    have_cur = true;
  }
}

void LaunchEnforceN4(int32_t* label_img, int W, int H, const int32_t* frames_dev, int num_frames,
                     int32_t* row_flags /* [num_frames * H], zeroed */, int32_t* adjust, hipStream_t s) {
  if (num_frames <= 0 || H < 2) return;
  hipLaunchKernelGGL(k_n4_row_flags, dim3(H - 1, num_frames), dim3(256), 0, s, label_img, W, H, frames_dev,
                     row_flags);
  hipLaunchKernelGGL(k_enforce_n4, dim3(num_frames), dim3(kN4Threads),
                     (size_t)4 * W * sizeof(int32_t), s, label_img, W, H, frames_dev, row_flags, adjust);
  VSG_HIP(hipGetLastError());
}

// K8a: number of runs in every row of a slice (one wavefront per row).
__global__ __launch_bounds__(64) void k_row_run_counts(const int32_t* __restrict__ img, int W, int H,
                                                        int32_t* __restrict__ row_counts) {
  const int y = blockIdx.x;
  const int lane = threadIdx.x;
  const int32_t* row = img + (size_t)y * W;
  int count = 0;
  for (int x0 = 0; x0 < W; x0 += 64) {
    const int x = x0 + lane;
    bool start = false;
    if (x < W) start = (x == 0) || (row[x] != row[x - 1]);
    count += __popcll(__ballot(start));
  }
  if (lane == 0) row_counts[y] = count;
}

void LaunchRowRunCounts(const int32_t* label_img, int W, int H, int frame, int32_t* row_counts,
                        hipStream_t s) {
  hipLaunchKernelGGL(k_row_run_counts, dim3(H), dim3(64), 0, s, label_img + (size_t)frame * W * H,
                     W, H, row_counts);
  VSG_HIP(hipGetLastError());
}

// K8b: write the intervals of every row at its global offset (row_offsets = exclusive scan of
// the run counts of all rows of all rasterised slices, in (slice, y) order).
__global__ __launch_bounds__(64) void k_write_intervals(const int32_t* __restrict__ img, int W, int H,
                                                         int frame,
                                                         const int32_t* __restrict__ row_offsets,
                                                         IntervalArrays out) {
  const int y = blockIdx.x;
  const int lane = threadIdx.x;
  const int32_t* row = img + (size_t)y * W;
  int base = row_offsets[y];
  for (int x0 = 0; x0 < W; x0 += 64) {
    const int x = x0 + lane;
    bool start = false;
    int lab = 0;
    if (x < W) {
      lab = row[x];
      start = (x == 0) || (lab != row[x - 1]);
    }
    const unsigned long long m = __ballot(start);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (start) {
      const int k = base + rank;
      out.label[k] = lab;
      out.ty[k] = ((uint32_t)frame << 16) | (uint32_t)y;
      out.lx[k] = x;
      if (x > 0) out.rx[k - 1] = x - 1;   // closes the previous run of this row
    }
    base += __popcll(m);
  }
  if (lane == 0) out.rx[base - 1] = W - 1;   // last run of the row
}

void LaunchWriteIntervals(const int32_t* label_img, int W, int H, int frame,
                          const int32_t* row_offsets, IntervalArrays out, hipStream_t s) {
  hipLaunchKernelGGL(k_write_intervals, dim3(H), dim3(64), 0, s,
                     label_img + (size_t)frame * W * H, W, H, frame, row_offsets, out);
  VSG_HIP(hipGetLastError());
}

// Tube splitting moves whole intervals to a new region key (dense_segmentation_graph.h:871-890).
__global__ __launch_bounds__(256) void k_relabel_intervals(const uint32_t* __restrict__ ty,
                                                            const int32_t* __restrict__ lxs,
                                                            const int32_t* __restrict__ rxs,
                                                            const int32_t* __restrict__ new_label,
                                                            int n, int W, int H,
                                                            int32_t* __restrict__ label_uf) {
  // one wavefront per interval
  const int iv = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (iv >= n) return;
  const uint32_t a = ty[iv];
  const int t = (int)(a >> 16), y = (int)(a & 0xFFFFu);
  const int lx = lxs[iv], rx = rxs[iv];
  const int lab = new_label[iv];
  int32_t* row = label_uf + ((size_t)t * H + y) * W;
  for (int x = lx + lane; x <= rx; x += 64) row[x] = lab;
}

void LaunchRelabelIntervals(const uint32_t* ty, const int32_t* lx, const int32_t* rx,
                            const int32_t* new_label, int n, int W, int H, int32_t* label_uf,
                            hipStream_t s) {
  if (n <= 0) return;
  const long long threads = (long long)n * 64;
  hipLaunchKernelGGL(k_relabel_intervals, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                     ty, lx, rx, new_label, n, W, H, label_uf);
  VSG_HIP(hipGetLastError());
}

// Open-addressing table of (pair -> smallest order key); empty = all ones.  *distinct counts the
// pairs entered (the caller falls back to listing every pair when the table gets too full).
// Once more than half the table is taken the caller's fall-back is certain (*distinct only grows),
// so inserts stop there: a pair is only ever dropped after that point, and nobody probes a table
// that is filling up (with 2^21 distinct pairs and more every insert would scan it end to end).
__device__ __forceinline__ void PairTableInsert(const PairTable& t, unsigned long long pair,
                                                unsigned long long order, int32_t* distinct) {
  unsigned long long x = pair * 0x9E3779B97F4A7C15ull;
  unsigned h = (unsigned)(x >> 40) & t.mask;
  const int half = (int)((t.mask >> 1) + 1u);
  for (unsigned probe = 0; probe <= t.mask; ++probe) {
    if ((probe & 63u) == 0 &&
        __hip_atomic_load(distinct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > half) {
      return;
    }
    unsigned long long k = t.key[h];
    if (k == ~0ull) {
      k = atomicCAS(&t.key[h], ~0ull, pair);
      if (k == ~0ull) {
        atomicAdd(distinct, 1);
        k = pair;
      }
    }
    if (k == pair) {
      if (order < t.order[h]) atomicMin(&t.order[h], order);
      return;
    }
    h = (h + 1) & t.mask;
  }
}

__global__ __launch_bounds__(256) void k_pair_table_compact(PairTable t, unsigned long long* __restrict__ pairs,
                                                             unsigned long long* __restrict__ order_keys,
                                                             int32_t* __restrict__ out_count) {
  const unsigned h = blockIdx.x * 256 + threadIdx.x;
  const bool full = h <= t.mask && t.key[h] != ~0ull;
  const unsigned long long m = __ballot(full);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(out_count, (int)__popcll(m));
  base = __shfl(base, (int)__builtin_ctzll(m));
  if (full) {
    const int idx = base + (int)__popcll(m & ((1ull << lane) - 1ull));
    pairs[idx] = t.key[h];
    order_keys[idx] = t.order[h];
  }
}

// K9: every kept edge whose end points carry different region keys yields one ordered pair
// (ka << 32 | kb) and an order key (bucket, list, position) for the first-appearance order that
// DetermineNeighborIdsImpl uses when it creates RegionInformation for unseen representatives.
__global__ __launch_bounds__(256) void k_neighbor_pairs(const ListDesc* __restrict__ lists,
                                                         const int32_t* __restrict__ label_uf, int W,
                                                         unsigned long long* __restrict__ pairs,
                                                         unsigned long long* __restrict__ order_keys,
                                                         int32_t* __restrict__ count, int capacity,
                                                         PairTable table) {
  const int l = blockIdx.y;
  const ListDesc L = lists[l];
  if (!L.offsets) return;
  const int n_valid = L.offsets[kNumBuckets + 1];
  const int lane = threadIdx.x & 63;
  // (whole wavefronts iterate together: the ballots below need every lane)
  for (int p0 = blockIdx.x * 256 + (threadIdx.x & ~63); p0 < n_valid; p0 += gridDim.x * 256) {
    const int p = p0 + lane;
    unsigned long long pair = 0;
    bool emit = false;
    if (p < n_valid && L.kept[p]) {
      const uint32_t slot = L.slots[p];
      int a, b;
      if (L.type == 0) {
        const uint32_t pix = slot >> 2;
        const int k = (int)(slot & 3u);
        a = L.base_a + (int)pix;
        b = a + ((k == 0) ? 1 : (k == 1) ? W : (k == 2) ? (W - 1) : (W + 1));
      } else {
        const uint32_t pix = slot / 9u;
        const int k = (int)(slot - pix * 9u);
        const int dy = k / 3 - 1, dx = k - (k / 3) * 3 - 1;
        a = L.base_a + (int)pix;
        b = L.base_b + L.prev_idx[pix] + dy * W + dx;
      }
      const int ka = label_uf[a], kb = label_uf[b];
      emit = ka != kb;
      pair = ((unsigned long long)(uint32_t)ka << 32) | (unsigned long long)(uint32_t)kb;
    }
    // The kept edges along the common boundary of two regions repeat the same pair: a lane whose
    // predecessor emits the same ordered pair stays silent (the earlier position carries the
    // smaller order key, which is the one that counts), and the wavefront takes its output slots
    // with one atomic.  Sort + unique on the rest do the exact job.
    const unsigned long long prev_pair = __shfl_up(pair, 1);
    const int prev_emit = __shfl_up((int)emit, 1);
    if (lane > 0 && emit && prev_emit && prev_pair == pair) emit = false;
    const unsigned long long m = __ballot(emit);
    if (m == 0) continue;
    if (table.key) {
      // Into the hash table: the pair with the smallest order key it has been seen with.  (Two large
      // regions meet along millions of kept edges: 43 M emitted pairs for 7 K distinct ones on a noisy
      // chunk, all of them sorted afterwards.)  The first look is a plain read: a pair that is there
      // already with a smaller order key costs no atomic.
      if (emit) {
        int lo = 0, hi = kNumBuckets + 1;   // bucket of position p
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (L.offsets[mid] <= p) lo = mid; else hi = mid;
        }
        const unsigned long long order = ((unsigned long long)lo << 48) | ((unsigned long long)l << 36) |
                                         (unsigned long long)(uint32_t)p;
        PairTableInsert(table, pair, order, count);
      }
      continue;
    }
    int base = 0;
    if (lane == 0) base = atomicAdd(count, (int)__popcll(m));
    base = __shfl(base, 0);
    if (!emit) continue;
    const int idx = base + (int)__popcll(m & ((1ull << lane) - 1ull));
    if (idx < capacity) {
      pairs[idx] = pair;
      // bucket of position p: binary search in the offsets
      int lo = 0, hi = kNumBuckets + 1;   // largest bkt with offsets[bkt] <= p
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (L.offsets[mid] <= p) lo = mid; else hi = mid;
      }
      order_keys[idx] = ((unsigned long long)lo << 48) | ((unsigned long long)l << 36) |
                        (unsigned long long)(uint32_t)p;
    }
  }
}

void LaunchNeighborPairs(const ListDesc* lists, int num_lists, const int32_t* label_uf, int W,
                         unsigned long long* pairs, unsigned long long* order_keys, int32_t* count,
                         int capacity, hipStream_t s) {
  VSG_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(k_neighbor_pairs, dim3(256, num_lists), dim3(256), 0, s, lists, label_uf, W,
                     pairs, order_keys, count, capacity, PairTable{nullptr, nullptr, 0});
  VSG_HIP(hipGetLastError());
}

void LaunchNeighborPairsHashed(const ListDesc* lists, int num_lists, const int32_t* label_uf, int W,
                               PairTable table, int32_t* distinct, hipStream_t s) {
  const size_t cap = (size_t)table.mask + 1;
  VSG_HIP(hipMemsetAsync(table.key, 0xFF, cap * sizeof(unsigned long long), s));
  VSG_HIP(hipMemsetAsync(table.order, 0xFF, cap * sizeof(unsigned long long), s));
  VSG_HIP(hipMemsetAsync(distinct, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(k_neighbor_pairs, dim3(256, num_lists), dim3(256), 0, s, lists, label_uf, W,
                     nullptr, nullptr, distinct, 0, table);
  VSG_HIP(hipGetLastError());
}

void LaunchPairTableCompact(PairTable table, unsigned long long* pairs, unsigned long long* order_keys,
                            int32_t* out_count, hipStream_t s) {
  VSG_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(k_pair_table_compact, dim3(((size_t)table.mask + 256) / 256), dim3(256), 0, s, table, pairs,
                     order_keys, out_count);
  VSG_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_gather_states(NodeArrays nodes, const int32_t* __restrict__ ids,
                                                        int n, float4* __restrict__ desc_sz_out,
                                                        int32_t* __restrict__ cons_out,
                                                        int32_t* __restrict__ flags_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r = ids[i];
  desc_sz_out[i] = nodes.desc_sz[r];
  cons_out[i] = nodes.cons[r];
  flags_out[i] = nodes.flags[r];
}

void LaunchGatherStates(NodeArrays nodes, const int32_t* ids, int n, float4* desc_sz_out,
                        int32_t* cons_out, int32_t* flags_out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_states, dim3((n + 255) / 256), dim3(256), 0, s, nodes, ids, n,
                     desc_sz_out, cons_out, flags_out);
  VSG_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_scatter_states(NodeArrays nodes, const int32_t* __restrict__ ids,
                                                         int n, const int32_t* __restrict__ parent_in,
                                                         const float4* __restrict__ desc_sz_in,
                                                         const int32_t* __restrict__ cons_in,
                                                         const int32_t* __restrict__ flags_in) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r = ids[i];
  nodes.parent[r] = parent_in[i];
  nodes.desc_sz[r] = desc_sz_in[i];
  nodes.cons[r] = cons_in[i];
  nodes.flags[r] = (uint8_t)flags_in[i];
}

void LaunchScatterStates(NodeArrays nodes, const int32_t* ids, int n, const int32_t* parent_in,
                         const float4* desc_sz_in, const int32_t* cons_in, const int32_t* flags_in,
                         hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_scatter_states, dim3((n + 255) / 256), dim3(256), 0, s, nodes, ids, n,
                     parent_in, desc_sz_in, cons_in, flags_in);
  VSG_HIP(hipGetLastError());
}

// For MergeConstrainedRegions (segmentation_graph.h:703-786): nodes whose *own* constraint field
// is >= 0, with their current representative.
__global__ __launch_bounds__(256) void k_constrained_roots(NodeArrays nodes, int begin, int end,
                                                            int32_t* __restrict__ flag_out,
                                                            int32_t* __restrict__ root_out) {
  const int i = begin + blockIdx.x * 256 + threadIdx.x;
  if (i >= end) return;
  const int c = nodes.cons[i];
  flag_out[i - begin] = (c >= 0) ? 1 : 0;
  root_out[i - begin] = (c >= 0) ? FindRO(nodes.parent, i) : -1;
}

void LaunchConstrainedRoots(NodeArrays nodes, int begin, int end, int32_t* flag_out,
                            int32_t* root_out, hipStream_t s) {
  const int n = end - begin;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_constrained_roots, dim3((n + 255) / 256), dim3(256), 0, s, nodes, begin, end,
                     flag_out, root_out);
  VSG_HIP(hipGetLastError());
}

// The node walk of MergeConstrainedRegions reduced to *runs*.  A node matters to the walk if its own
// constraint field is >= 0 or if it is a representative (whose field may become >= 0 during the
// walk).  Consecutive nodes with the same representative repeat the same step until it becomes a
// no-op, so only the first node of such a run (and the run's length, from the next entry) reaches
// the host.  value_out: the representative; kRunRepUnconstrained for a representative whose own
// field is < 0; kRunNone for a node that does not matter (it only terminates the run before it).
constexpr int kRunNone = -1;
constexpr int kRunRepUnconstrained = -2;
__global__ __launch_bounds__(256) void k_constrained_run_values(NodeArrays nodes, int begin, int end,
                                                                 int32_t* __restrict__ value_out) {
  const int i = begin + blockIdx.x * 256 + threadIdx.x;
  if (i >= end) return;
  const int c = nodes.cons[i];
  const int p = nodes.parent[i];
  int v = kRunNone;
  if (p == i) {
    v = c >= 0 ? i : kRunRepUnconstrained;
  } else if (c >= 0) {
    v = FindRO(nodes.parent, p);
  }
  value_out[i - begin] = v;
}

__global__ __launch_bounds__(256) void k_constrained_run_heads(int begin, int n,
                                                                const int32_t* __restrict__ value,
                                                                int32_t* __restrict__ flag_out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int v = value[j];
  const int i = begin + j;
  int head;
  if (j == 0) {
    head = 1;
  } else {
    const int pv = value[j - 1];
    const bool prev_plain = pv >= 0 && pv != i - 1;   // matters, is not a representative
    if (v == kRunNone) {
      head = pv != kRunNone;                           // terminates the run before it
    } else if (v == kRunRepUnconstrained || v == i) {
      head = 1;                                        // a representative is a run of its own
    } else {
      head = !(prev_plain && pv == v);
    }
  }
  flag_out[j] = head;
}

void LaunchConstrainedRuns(NodeArrays nodes, int begin, int end, int32_t* value_out,
                           int32_t* flag_out, hipStream_t s) {
  const int n = end - begin;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_constrained_run_values, dim3((n + 255) / 256), dim3(256), 0, s, nodes, begin,
                     end, value_out);
  hipLaunchKernelGGL(k_constrained_run_heads, dim3((n + 255) / 256), dim3(256), 0, s, begin, n,
                     value_out, flag_out);
  VSG_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_compact_i32(const int32_t* __restrict__ flags,
                                                      const int32_t* __restrict__ offsets,
                                                      const int32_t* __restrict__ values, int n,
                                                      int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) out[offsets[i]] = values[i];
}

void LaunchCompactI32(const int32_t* flags, const int32_t* offsets, const int32_t* values, int n,
                      int32_t* out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_compact_i32, dim3((n + 255) / 256), dim3(256), 0, s, flags, offsets, values,
                     n, out);
  VSG_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_gather_i32(const int32_t* __restrict__ src,
                                                     const int32_t* __restrict__ idx, int n,
                                                     int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

void LaunchGatherI32(const int32_t* src, const int32_t* idx, int n, int32_t* out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_i32, dim3((n + 255) / 256), dim3(256), 0, s, src, idx, n, out);
  VSG_HIP(hipGetLastError());
}

// Backward flow at the points the tube analysis reads (one (frame, y, x) request each).
__global__ __launch_bounds__(256) void k_gather_flow(const int32_t* __restrict__ req, int n,
                                                      const float* const* __restrict__ flows, int W,
                                                      float2* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int frame = req[3 * i], y = req[3 * i + 1], x = req[3 * i + 2];
  const float* f = flows[frame];
  out[i] = f ? reinterpret_cast<const float2*>(f)[(size_t)y * W + x] : make_float2(0.f, 0.f);
}

void LaunchGatherFlow(const int32_t* req, int n, const float* const* flows, int W, float2* out,
                      hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather_flow, dim3((n + 255) / 256), dim3(256), 0, s, req, n, flows, W, out);
  VSG_HIP(hipGetLastError());
}

// Small helpers for compacting sparse N-sized arrays.
__global__ __launch_bounds__(256) void k_nonzero_flags(const int32_t* __restrict__ a, int n,
                                                        int32_t* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flags[i] = (a[i] != 0) ? 1 : 0;
}

void LaunchNonzeroFlags(const int32_t* a, int n, int32_t* flags, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_nonzero_flags, dim3((n + 255) / 256), dim3(256), 0, s, a, n, flags);
  VSG_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_compact_index_value(const int32_t* __restrict__ flags,
                                                              const int32_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ values,
                                                              int n, int32_t* __restrict__ out_idx,
                                                              int32_t* __restrict__ out_val) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !flags[i]) return;
  const int p = offsets[i];
  out_idx[p] = i;
  out_val[p] = values[i];
}

void LaunchCompactIndexValue(const int32_t* flags, const int32_t* offsets, const int32_t* values,
                             int n, int32_t* out_idx, int32_t* out_val, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_compact_index_value, dim3((n + 255) / 256), dim3(256), 0, s, flags, offsets,
                     values, n, out_idx, out_val);
  VSG_HIP(hipGetLastError());
}

// For every key of a small sorted set: the smallest (order_key * 2 + side) over all emitted pairs
// that contain it (side 0 = first end point).  Used to reproduce the creation order of
// RegionInformation for representatives that own no pixel (segmentation_graph.h:471-493).
__global__ __launch_bounds__(256) void k_first_order_of_keys(
    const unsigned long long* __restrict__ pairs, const unsigned long long* __restrict__ order_keys,
    int m, const int32_t* __restrict__ keys_sorted, int num_keys,
    unsigned long long* __restrict__ out_min) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const unsigned long long pr = pairs[i];
  const int k2[2] = {(int)(uint32_t)(pr >> 32), (int)(uint32_t)(pr & 0xFFFFFFFFull)};
  for (int side = 0; side < 2; ++side) {
    const int key = k2[side];
    int lo = 0, hi = num_keys;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys_sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo < num_keys && keys_sorted[lo] == key) {
      atomicMin(&out_min[lo], order_keys[i] * 2ull + (unsigned long long)side);
    }
  }
}

void LaunchFirstOrderOfKeys(const unsigned long long* pairs, const unsigned long long* order_keys,
                            int m, const int32_t* keys_sorted, int num_keys,
                            unsigned long long* out_min, hipStream_t s) {
  if (m <= 0 || num_keys <= 0) return;
  hipLaunchKernelGGL(k_first_order_of_keys, dim3((m + 255) / 256), dim3(256), 0, s, pairs,
                     order_keys, m, keys_sorted, num_keys, out_min);
  VSG_HIP(hipGetLastError());
}

}  // namespace vsg
