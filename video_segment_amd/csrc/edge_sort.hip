// edge_sort.hip -- K3 / K4 / K5: edge weights -> bucket keys, and the stable bucket sort of the
// edge slots (gfx950).  HBM-bound integer / light f32 work: coalesced plane loads, LDS staging,
// wave ballots; no MFMA.
//
// Reference behaviour restated (paths relative to the reference root):
//   segmentation/pixel_distance.h:141-157                (ColorDiff3L1 / ColorDiff3L2)
//   segmentation/segmentation_graph.h:158-162, 336       (bucket index, per-bucket push_back)
//   segmentation/dense_segmentation_graph.h:956-1142     (edge enumeration order)
// The reference appends every edge to the vector of its bucket in (scan order, neighbour order);
// an edge here is its *slot* (pix*4+k spatial, pix*9+k temporal), so a bucket list is the stable
// sort of the slot ids by bucket key.  With 2050 possible keys (buckets 0..2047, 2048 = virtual
// edges, one bin for edges that do not exist at the frame border) that is ONE counting sort:
//
//   K3/K4  per tile of 2048 pixels: keys (u16 per slot, slot-major, written coalesced through an
//          LDS stage for the 9-key temporal case) + the tile's key histogram (LDS atomics),
//          stored bin-major:  hist[bin * T + tile].                      reads 12 (+12+8) B/px,
//                                                                        writes 8 / 18 B/px
//   scan   exclusive prefix over hist in that order = for every (bin, tile) the first output
//          position of the tile's slots with that key; offsets[bin] = value at tile 0.
//   K5     per tile: keys -> LDS, one histogram per wavefront over its contiguous quarter of the
//          tile, prefix over the four wavefronts + the scanned base, then every wavefront walks
//          its quarter 64 slots at a time: the rank of a slot among the equal keys of the step
//          comes from 12 ballots (one per key bit), the running position from the LDS counters.
//          Stable by construction: tiles, quarters, steps and lanes are all in slot order.
//                                                                        reads 2 B, writes 4 B/slot
// No iota value array, no multi-pass radix sort: 8 B of traffic per slot instead of ~24.
#include "merge_common.h"

namespace vsg {

// Pixels per tile (one workgroup); the temporal tile is smaller so that its 9 keys per pixel and
// the four per-wavefront counter arrays of K5 stay below 64 KiB of LDS.
constexpr int kTilePxSpatial = 2048;
constexpr int kTilePxTemporal = 1024;
constexpr int kBinInvalid = kBucketSlots - 1; // 2049: slots of edges that do not exist

__device__ __forceinline__ float ColorDist(float ab, float ag, float ar, float bb, float bg,
                                           float br, int l1) {
  const float d1 = ab - bb, d2 = ag - bg, d3 = ar - br;
  if (l1) {
    // (fabs(d1)+fabs(d2)+fabs(d3)) * (1.0f/3.0f) evaluated in double (double fabs overloads).
    return (float)((fabs((double)d1) + fabs((double)d2) + fabs((double)d3)) *
                   (double)(1.0f / 3.0f));
  }
  return sqrtf((d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 3.0f));
}

__device__ __forceinline__ uint16_t BucketOf(float w) {
  const float scale = 2048.0f / (1.0f + 1e-6f);   // segmentation_graph.h:336
  return (uint16_t)(int)fminf(2048.0f, w * scale);
}

// x86 cvttss2si semantics for int(float): out of range / NaN -> INT_MIN.
__device__ __forceinline__ int TruncToIntX86(float v) {
  if (!(v < 2147483648.0f && v >= -2147483648.0f)) return (int)0x80000000;
  return (int)v;
}

// The matrix is zeroed before the kernel (one coalesced memset); a tile only touches the few bins
// it has edges in -- 2050 scattered 4-byte stores per tile would cost more HBM write traffic than
// the keys themselves.
__device__ __forceinline__ void StoreTileHist(const int32_t* __restrict__ lds_hist, int tile,
                                              int num_tiles, int32_t* __restrict__ hist) {
  for (int b = threadIdx.x; b < kBucketSlots; b += 256) {
    const int c = lds_hist[b];
    if (c) hist[(size_t)b * num_tiles + tile] = c;
  }
}

// K3: slot = pix * 4 + k, k = 0 right, 1 bottom, 2 bottom-left, 3 bottom-right
// (AddSpatialEdgesImpl order, dense_segmentation_graph.h:971-996).
__global__ __launch_bounds__(256) void k_spatial_keys(const float* __restrict__ feat, int W, int H,
                                                       int l1, ushort4* __restrict__ keys,
                                                       int32_t* __restrict__ hist, int num_tiles) {
  __shared__ int32_t lds_hist[kBucketSlots];
  for (int b = threadIdx.x; b < kBucketSlots; b += 256) lds_hist[b] = 0;
  __syncthreads();
  const size_t n = (size_t)W * H;
  const float* fb = feat;
  const float* fg = feat + n;
  const float* fr = feat + 2 * n;
  const size_t base = (size_t)blockIdx.x * kTilePxSpatial;
#pragma unroll 2
  for (int i = 0; i < kTilePxSpatial / 256; ++i) {
    const size_t pix = base + (size_t)i * 256 + threadIdx.x;
    if (pix >= n) break;
    const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
    const float ab = fb[pix], ag = fg[pix], ar = fr[pix];
    ushort4 k4 = make_ushort4(kInvalidKey, kInvalidKey, kInvalidKey, kInvalidKey);
    const bool has_r = x < W - 1, has_b = y < H - 1, has_l = x > 0;
    if (has_r) {
      const size_t q = pix + 1;
      k4.x = BucketOf(ColorDist(ab, ag, ar, fb[q], fg[q], fr[q], l1));
    }
    if (has_b) {
      size_t q = pix + W;
      k4.y = BucketOf(ColorDist(ab, ag, ar, fb[q], fg[q], fr[q], l1));
      if (has_l) {
        q = pix + W - 1;
        k4.z = BucketOf(ColorDist(ab, ag, ar, fb[q], fg[q], fr[q], l1));
      }
      if (has_r) {
        q = pix + W + 1;
        k4.w = BucketOf(ColorDist(ab, ag, ar, fb[q], fg[q], fr[q], l1));
      }
    }
    keys[pix] = k4;
    atomicAdd(&lds_hist[k4.x == kInvalidKey ? kBinInvalid : k4.x], 1);
    atomicAdd(&lds_hist[k4.y == kInvalidKey ? kBinInvalid : k4.y], 1);
    atomicAdd(&lds_hist[k4.z == kInvalidKey ? kBinInvalid : k4.z], 1);
    atomicAdd(&lds_hist[k4.w == kInvalidKey ? kBinInvalid : k4.w], 1);
  }
  __syncthreads();
  StoreTileHist(lds_hist, blockIdx.x, num_tiles, hist);
}

// K4: slot = pix * 9 + (dy+1)*3 + (dx+1) around the (flow displaced) location in the previous
// slice (GetLocalEdges order TL,T,TR,L,C,R,BL,B,BR; dense_segmentation_graph.h:1011-1065,
// 1126-1135).  is_virtual: weight 1e10 -> bucket 2048 for every existing edge.
__global__ __launch_bounds__(256) void k_temporal_keys(const float* __restrict__ cur,
                                                        const float* __restrict__ prev,
                                                        const float* __restrict__ flow, int W,
                                                        int H, int l1, int is_virtual,
                                                        uint16_t* __restrict__ keys,
                                                        int32_t* __restrict__ prev_idx,
                                                        int32_t* __restrict__ hist, int num_tiles) {
  __shared__ int32_t lds_hist[kBucketSlots];
  __shared__ __attribute__((aligned(16))) uint16_t stage[256 * 9];
  for (int b = threadIdx.x; b < kBucketSlots; b += 256) lds_hist[b] = 0;
  __syncthreads();
  const size_t n = (size_t)W * H;
  const size_t base = (size_t)blockIdx.x * kTilePxTemporal;
  for (int i = 0; i < kTilePxTemporal / 256; ++i) {
    const size_t pix0 = base + (size_t)i * 256;
    if (pix0 >= n) break;
    const size_t pix = pix0 + threadIdx.x;
    if (pix < n) {
      const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
      int px = x, py = y;
      if (flow) {
        const float2 f = reinterpret_cast<const float2*>(flow)[pix];
        px = TruncToIntX86((float)x + f.x);
        py = TruncToIntX86((float)y + f.y);
        px = max(0, min(W - 1, px));
        py = max(0, min(H - 1, py));
      }
      prev_idx[pix] = py * W + px;
      float ab = 0, ag = 0, ar = 0;
      if (!is_virtual) {
        ab = cur[pix];
        ag = cur[n + pix];
        ar = cur[2 * n + pix];
      }
      int k = 0;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx, ++k) {
          uint16_t key = kInvalidKey;
          const int qy = py + dy, qx = px + dx;
          if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
            if (is_virtual) {
              key = (uint16_t)kNumBuckets;
            } else {
              const size_t q = (size_t)qy * W + qx;
              key = BucketOf(ColorDist(ab, ag, ar, prev[q], prev[n + q], prev[2 * n + q], l1));
            }
          }
          stage[threadIdx.x * 9 + k] = key;
          atomicAdd(&lds_hist[key == kInvalidKey ? kBinInvalid : key], 1);
        }
      }
    }
    __syncthreads();
    // 256 pixels x 9 keys x 2 B = 4608 contiguous bytes: coalesced 4-byte stores
    const int px_here = (int)min((size_t)256, n - pix0);
    const int words = (px_here * 9 + 1) / 2;
    uint32_t* out32 = reinterpret_cast<uint32_t*>(keys + pix0 * 9);   // pix0 * 18 B: 4-byte aligned
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(stage);
    if ((px_here * 9) & 1) {   // odd number of keys: the last one on its own
      for (int w = threadIdx.x; w < words - 1; w += 256) out32[w] = in32[w];
      if (threadIdx.x == 0) keys[pix0 * 9 + (size_t)px_here * 9 - 1] = stage[px_here * 9 - 1];
    } else {
      for (int w = threadIdx.x; w < words; w += 256) out32[w] = in32[w];
    }
    __syncthreads();
  }
  StoreTileHist(lds_hist, blockIdx.x, num_tiles, hist);
}

// ---- exclusive scan of the (bin-major) tile histograms -----------------------------------------
constexpr int kScanPerBlock = 2048;   // 256 threads x 8 elements

__device__ __forceinline__ int BlockExclusiveScan256(int v, int32_t* lds4, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = WaveInclusiveSum(v);
  if (lane == 63) lds4[wave] = incl;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += lds4[w];
  total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return before + incl - v;
}

__global__ __launch_bounds__(256) void k_scan_block_sums(const int32_t* __restrict__ a, int n,
                                                          int32_t* __restrict__ sums) {
  __shared__ int32_t lds4[4];
  const int base = blockIdx.x * kScanPerBlock + threadIdx.x * 8;
  int v = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) v += (base + k < n) ? a[base + k] : 0;
  int total;
  BlockExclusiveScan256(v, lds4, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_scan_sums(int32_t* __restrict__ sums, int m) {
  __shared__ int32_t lds4[4];
  int carry = 0;
  for (int base = 0; base < m; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < m ? sums[i] : 0;
    int total;
    const int ex = BlockExclusiveScan256(v, lds4, total);
    if (i < m) sums[i] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(256) void k_scan_apply(int32_t* __restrict__ a, int n,
                                                     const int32_t* __restrict__ sums,
                                                     int num_tiles, int32_t* __restrict__ offsets) {
  __shared__ int32_t lds4[4];
  const int base = blockIdx.x * kScanPerBlock + threadIdx.x * 8;
  int x[8];
  int v = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    x[k] = (base + k < n) ? a[base + k] : 0;
    v += x[k];
  }
  int total;
  int run = sums[blockIdx.x] + BlockExclusiveScan256(v, lds4, total);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = base + k;
    if (i < n) {
      a[i] = run;
      if (i % num_tiles == 0) offsets[i / num_tiles] = run;   // start of a bin
    }
    run += x[k];
  }
}

// ---- K5: stable scatter of the slot ids ----------------------------------------------------------
// Lanes of the wavefront whose 12-bit bin equals this lane's (one ballot per bit).
__device__ __forceinline__ unsigned long long SameBinMask(int bin) {
  unsigned long long m = ~0ull;
#pragma unroll
  for (int b = 0; b < 12; ++b) {
    const unsigned long long bal = __ballot((bin >> b) & 1);
    m &= ((bin >> b) & 1) ? bal : ~bal;
  }
  return m;
}

template <int kPerPx, int kTilePx>
__global__ __launch_bounds__(256) void k_scatter_slots(const uint16_t* __restrict__ keys,
                                                        size_t n_px,
                                                        const int32_t* __restrict__ scanned,
                                                        int num_tiles,
                                                        uint32_t* __restrict__ slots_out) {
  constexpr int kTileSlots = kTilePx * kPerPx;
  __shared__ __attribute__((aligned(16))) uint16_t lds_keys[kTileSlots];
  __shared__ int32_t cnt[4][kBucketSlots];
  const int tile = blockIdx.x;
  const size_t px0 = (size_t)tile * kTilePx;
  const int n_slots = (int)(min((size_t)kTilePx, n_px - px0) * kPerPx);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = threadIdx.x; b < 4 * kBucketSlots; b += 256) (&cnt[0][0])[b] = 0;
  {   // keys -> LDS (4-byte words; px0 * kPerPx * 2 B is 4-byte aligned)
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(keys + px0 * kPerPx);
    uint32_t* l32 = reinterpret_cast<uint32_t*>(lds_keys);
    const int words = n_slots >> 1;
    for (int w = threadIdx.x; w < words; w += 256) l32[w] = in32[w];
    if ((n_slots & 1) && threadIdx.x == 0) lds_keys[n_slots - 1] = keys[px0 * kPerPx + n_slots - 1];
  }
  __syncthreads();
  // the wavefront's contiguous quarter (a multiple of 64 slots)
  const int quarter = ((n_slots + 255) / 256) * 64;
  const int q_beg = min(n_slots, wave * quarter), q_end = min(n_slots, (wave + 1) * quarter);
  for (int i = q_beg + lane; i < q_end; i += 64) {
    const int key = lds_keys[i];
    atomicAdd(&cnt[wave][key == kInvalidKey ? kBinInvalid : key], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kBucketSlots; b += 256) {
    const int c0 = cnt[0][b], c1 = cnt[1][b], c2 = cnt[2][b], c3 = cnt[3][b];
    if ((c0 | c1 | c2 | c3) == 0) continue;   // the matrix is bin-major: a read per bin is a cache line
    const int run = scanned[(size_t)b * num_tiles + tile];
    cnt[0][b] = run;
    cnt[1][b] = run + c0;
    cnt[2][b] = run + c0 + c1;
    cnt[3][b] = run + c0 + c1 + c2;
  }
  __syncthreads();
  const uint32_t slot0 = (uint32_t)(px0 * kPerPx);
  for (int i0 = q_beg; i0 < q_end; i0 += 64) {
    const int i = i0 + lane;
    const bool valid = i < q_end;
    const int key = valid ? (int)lds_keys[i] : (int)kInvalidKey;
    const int bin = key == kInvalidKey ? kBinInvalid : key;
    const unsigned long long same = SameBinMask(bin);
    const int rank = (int)__popcll(same & ((1ull << lane) - 1ull));
    const int start = cnt[wave][bin];
    // Slots of edges that do not exist are not listed (nothing ever reads that tail).
    if (valid && bin != kBinInvalid) slots_out[start + rank] = slot0 + (uint32_t)i;
    __builtin_amdgcn_wave_barrier();
    if (rank == 0) cnt[wave][bin] = start + (int)__popcll(same);   // one lane per distinct bin
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- launchers -----------------------------------------------------------------------------------
static int NumTiles(size_t n_px, int per_px) {
  const int tile = per_px == 4 ? kTilePxSpatial : kTilePxTemporal;
  return (int)((n_px + tile - 1) / tile);
}
// Scratch sizes (ints) for the largest (temporal) list of a frame.
size_t EdgeSortHistInts(size_t n_px) { return (size_t)kBucketSlots * NumTiles(n_px, 9); }
size_t EdgeSortSumInts(size_t n_px) {
  return (EdgeSortHistInts(n_px) + kScanPerBlock - 1) / kScanPerBlock;
}

void LaunchSpatialKeys(const float* feat, int W, int H, int l1, uint16_t* keys, int32_t* hist,
                       hipStream_t s) {
  const int T = NumTiles((size_t)W * H, 4);
  VSG_HIP(hipMemsetAsync(hist, 0, (size_t)kBucketSlots * T * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_spatial_keys, dim3(T), dim3(256), 0, s, feat, W, H, l1,
                     reinterpret_cast<ushort4*>(keys), hist, T);
  VSG_HIP(hipGetLastError());
}

void LaunchTemporalKeys(const float* cur, const float* prev, const float* flow, int W, int H,
                        int l1, int is_virtual, uint16_t* keys, int32_t* prev_idx, int32_t* hist,
                        hipStream_t s) {
  const int T = NumTiles((size_t)W * H, 9);
  VSG_HIP(hipMemsetAsync(hist, 0, (size_t)kBucketSlots * T * sizeof(int32_t), s));
  hipLaunchKernelGGL(k_temporal_keys, dim3(T), dim3(256), 0, s, cur, prev, flow, W, H, l1,
                     is_virtual, keys, prev_idx, hist, T);
  VSG_HIP(hipGetLastError());
}

// hist (bin-major tile histograms) -> exclusive scan in place; offsets[0..kBucketSlots-1] = start
// of every bin; then the stable scatter of the slot ids of the existing edges.
void LaunchBucketSort(const uint16_t* keys, size_t n_px, int per_px, int32_t* hist, int32_t* sums,
                      int32_t* offsets, uint32_t* slots_out, hipStream_t s) {
  const int T = NumTiles(n_px, per_px);
  const int n = kBucketSlots * T;
  const int blocks = (n + kScanPerBlock - 1) / kScanPerBlock;
  hipLaunchKernelGGL(k_scan_block_sums, dim3(blocks), dim3(256), 0, s, hist, n, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, s, sums, blocks);
  hipLaunchKernelGGL(k_scan_apply, dim3(blocks), dim3(256), 0, s, hist, n, sums, T, offsets);
  if (per_px == 4) {
    hipLaunchKernelGGL((k_scatter_slots<4, kTilePxSpatial>), dim3(T), dim3(256), 0, s, keys, n_px,
                       hist, T, slots_out);
  } else {
    hipLaunchKernelGGL((k_scatter_slots<9, kTilePxTemporal>), dim3(T), dim3(256), 0, s, keys, n_px,
                       hist, T, slots_out);
  }
  VSG_HIP(hipGetLastError());
}

}  // namespace vsg
