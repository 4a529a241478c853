// radix_sort.hip -- a stable sort of (key, value) pairs for the merge path, hand-written for gfx950, and
// the choice between it and the library's.
//
// What the merge sorts are 32-bit keys of which only the low `end_bit` bits are in use (node ids: 26
// bits at 1080p x 21 slices; side-cluster keys: 21-23) with a 32-bit payload (the position in the
// stage's edge sequence), between a hundred and a hundred million pairs, ~100 times per chunk.  An LSD
// radix sort with digits of up to 9 bits -- three passes for up to 27 bits, two for up to 18:
//
//   n <= 4096           one launch: one workgroup keeps the pairs in registers / LDS through all passes
//   larger              per pass: tile histograms (k_rs_hist), exclusive scan of the bin-major histogram
//                       matrix (two launches, scan_device.h), k_rs_scatter
//
// A tile is 4096 pairs on 256 threads; a wavefront owns a contiguous quarter and walks it 64 pairs at
// a time: the rank of a pair among the equal digits of its step comes from one ballot per digit bit,
// the running count per (wavefront, digit) from LDS -- no atomics, stable by construction (tiles,
// quarters, steps and lanes are all in input order).  This is the scheme of edge_sort.hip's
// k_scatter_slots with the keys in registers.  HBM traffic per pass: keys twice, values once, both
// written once = 20 B per pair.
//
// Round 6 built this to take the last library call off the merge path and measured it against that
// call (bottom of the file): it wins between 0.4 and 2 million pairs and loses elsewhere, so the merge
// uses each where it is faster.  Two forms with fewer launches were built, measured and removed:
//   * one launch per pass with a chained scan over the tiles (decoupled look-back, tickets): with every
//     tile of a sort resident at once a tile walks back over half of its predecessors before it meets
//     a published prefix -- 770 us per launch on the bench input;
//   * one launch per pass for up to 128 tiles, every workgroup adding up the histogram columns in
//     front of its tile itself and counting the NEXT digit of every pair for the tile it scatters it
//     to (one atomic per pair): the keys of the merge come in runs of equal keys -- a component's
//     edges -- whose atomics hit one word (33 us per launch on average, 385 at worst).
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "device_graph.h"
#include "scan_device.h"

namespace vsg {

namespace {

constexpr int kRsTile = 4096;       // pairs per workgroup
constexpr int kRsSteps = 16;        // 64-pair steps per wavefront quarter
constexpr int kRsMaxBits = 9;
constexpr int kRsBins = 1 << kRsMaxBits;

__device__ __forceinline__ void RsWaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// Lanes of the wavefront that hold a pair with this lane's digit.
__device__ __forceinline__ unsigned long long RsMatch(int d, int bits, bool valid) {
  unsigned long long m = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const bool bit = (d >> b) & 1;
    const unsigned long long q = __ballot(bit);
    m &= bit ? q : ~q;
  }
  return m;
}

__device__ __forceinline__ unsigned long long RsLanesBelow() {
  return (1ull << (threadIdx.x & 63)) - 1ull;
}

// The wavefront's quarter, step by step: lr[j] = number of pairs with the same digit in front of pair
// j inside the quarter; cnt[d] (this wavefront's row, zeroed) = pairs per digit afterwards.
__device__ __forceinline__ void RsCountQuarter(const uint32_t (&k)[kRsSteps], const bool (&valid)[kRsSteps],
                                               int shift, int bits, int* cnt, int (&lr)[kRsSteps]) {
  const unsigned mask = (1u << bits) - 1u;
  const unsigned long long below = RsLanesBelow();
#pragma unroll
  for (int j = 0; j < kRsSteps; ++j) {
    const int d = (int)((k[j] >> shift) & mask);
    const unsigned long long m = RsMatch(d, bits, valid[j]);
    const int rank = __popcll(m & below);
    const int c = valid[j] ? cnt[d] : 0;
    lr[j] = c + rank;
    RsWaveSync();
    if (valid[j] && rank == 0) cnt[d] = c + __popcll(m);
    RsWaveSync();
  }
}

// cnt[w][b]: pairs of digit b in the quarter of wavefront w  ->  first output position of those
// pairs, given the first position g(b) of the tile's pairs of digit b.  Thread t owns bins 2t, 2t + 1.
template <class Base>
__device__ __forceinline__ void RsQuarterBases(int (*cnt)[kRsBins], int bins, Base g) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int b = 2 * (int)threadIdx.x + q;
    if (b < bins) {
      int at = g(b);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int c = cnt[w][b];
        cnt[w][b] = at;
        at += c;
      }
    }
  }
}

// ---- more than one tile: per pass a tile-histogram kernel, the exclusive scan of the bin-major
// histogram matrix (scan_device.h, two launches) and the scatter ---------------------------------------
__global__ __launch_bounds__(256) void k_rs_hist(const uint32_t* __restrict__ keys, int n, int shift, int bits,
                                                  int T, int32_t* __restrict__ hist /* [bins][T] */) {
  __shared__ int cnt[kRsBins];
  const int bins = 1 << bits;
  const unsigned mask = (unsigned)bins - 1u;
  for (int b = threadIdx.x; b < bins; b += 256) cnt[b] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * kRsTile;
#pragma unroll
  for (int j = 0; j < kRsSteps; ++j) {
    const long long i = base + j * 256 + threadIdx.x;
    if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & mask], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += 256) hist[(size_t)b * T + blockIdx.x] = cnt[b];
}

// hist: the exclusive scan of the tile histograms in bin-major order = the first output position of
// every (digit, tile).
__global__ __launch_bounds__(256) void k_rs_scatter(const uint32_t* __restrict__ keys_in,
                                                     const uint32_t* __restrict__ vals_in,
                                                     uint32_t* __restrict__ keys_out,
                                                     uint32_t* __restrict__ vals_out, int n, int shift, int bits,
                                                     int T, const int32_t* __restrict__ hist) {
  __shared__ int cnt[4][kRsBins];
  const int bins = 1 << bits;
  const unsigned mask = (unsigned)bins - 1u;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x;
  const long long qbase = (long long)tile * kRsTile + wave * (kRsTile / 4);
  uint32_t k[kRsSteps], v[kRsSteps];
  bool valid[kRsSteps];
  int lr[kRsSteps];
#pragma unroll
  for (int j = 0; j < kRsSteps; ++j) {
    const long long i = qbase + j * 64 + lane;
    valid[j] = i < n;
    k[j] = valid[j] ? keys_in[i] : 0u;
    v[j] = valid[j] ? vals_in[i] : 0u;
  }
  for (int b = threadIdx.x; b < 4 * kRsBins; b += 256) (&cnt[0][0])[b] = 0;
  __syncthreads();
  RsCountQuarter(k, valid, shift, bits, cnt[wave], lr);
  __syncthreads();
  RsQuarterBases(cnt, bins, [&](int b) { return hist[(size_t)b * T + tile]; });
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kRsSteps; ++j) {
    if (valid[j]) {
      const int pos = cnt[wave][(k[j] >> shift) & mask] + lr[j];
      keys_out[pos] = k[j];
      vals_out[pos] = v[j];
    }
  }
}

// n <= kRsTile: every pass inside one workgroup.
__global__ __launch_bounds__(256) void k_rs_small(const uint32_t* __restrict__ keys_in,
                                                   const uint32_t* __restrict__ vals_in,
                                                   uint32_t* __restrict__ keys_out,
                                                   uint32_t* __restrict__ vals_out, int n, int passes,
                                                   int digit_bits, int end_bit) {
  __shared__ int cnt[4][kRsBins];
  __shared__ int32_t lds4[4];
  __shared__ uint32_t lk[kRsTile], lv[kRsTile];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qbase = wave * (kRsTile / 4);
  uint32_t k[kRsSteps], v[kRsSteps];
  bool valid[kRsSteps];
  int lr[kRsSteps];
#pragma unroll
  for (int j = 0; j < kRsSteps; ++j) {
    const int i = qbase + j * 64 + lane;
    valid[j] = i < n;
    k[j] = valid[j] ? keys_in[i] : 0u;
    v[j] = valid[j] ? vals_in[i] : 0u;
  }
  for (int p = 0; p < passes; ++p) {
    const int shift = p * digit_bits;
    const int bits = min(digit_bits, end_bit - shift);   // (the last digit ends with the key)
    const int bins = 1 << bits;
    const unsigned mask = (unsigned)bins - 1u;
    for (int b = threadIdx.x; b < 4 * kRsBins; b += 256) (&cnt[0][0])[b] = 0;
    __syncthreads();
    RsCountQuarter(k, valid, shift, bits, cnt[wave], lr);
    __syncthreads();
    int tot[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int b = 2 * (int)threadIdx.x + q;
      if (b < bins) tot[q] = cnt[0][b] + cnt[1][b] + cnt[2][b] + cnt[3][b];
    }
    int total;
    const int ex = BlockExclusiveScan256(tot[0] + tot[1], lds4, total);
    const int g0 = ex, g1 = ex + tot[0];
    RsQuarterBases(cnt, bins, [&](int b) { return (b & 1) ? g1 : g0; });
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRsSteps; ++j) {
      if (valid[j]) {
        const int pos = cnt[wave][(k[j] >> shift) & mask] + lr[j];
        lk[pos] = k[j];
        lv[pos] = v[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRsSteps; ++j) {   // (the valid pairs are the first n positions again)
      const int i = qbase + j * 64 + lane;
      if (i < n) {
        k[j] = lk[i];
        v[j] = lv[i];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < kRsSteps; ++j) {
    const int i = qbase + j * 64 + lane;
    if (i < n) {
      keys_out[i] = k[j];
      vals_out[i] = v[j];
    }
  }
}

inline size_t RsAlign(size_t b) { return (b + 255) & ~(size_t)255; }
inline int RsTiles(int n) { return (int)(((long long)n + kRsTile - 1) / kRsTile); }

}  // namespace

// Temporary storage: the tile sums of the matrix scan, the histogram matrix, one (key, value) buffer.
size_t SortPairsU32HandTempBytes(int n) {
  const size_t T = (size_t)RsTiles(n > 0 ? n : 1);
  return RsAlign(kScanMaxTiles * sizeof(int32_t)) + RsAlign(T * kRsBins * sizeof(int32_t)) +
         2 * RsAlign((size_t)(n > 0 ? n : 1) * sizeof(uint32_t));
}

// Stable sort of the pairs by the low `end_bit` bits of the keys; the inputs are left as they are.
void SortPairsU32Hand(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                      const uint32_t* vals_in, uint32_t* vals_out, int n, int end_bit, hipStream_t s) {
  if (n <= 0) return;
  VSG_REQUIRE(end_bit >= 1 && end_bit <= 32, -4, "radix sort: key bits");
  const int passes = (end_bit + kRsMaxBits - 1) / kRsMaxBits;
  const int digit_bits = (end_bit + passes - 1) / passes;
  if (n <= kRsTile) {
    hipLaunchKernelGGL(k_rs_small, dim3(1), dim3(256), 0, s, keys_in, vals_in, keys_out, vals_out, n, passes,
                       digit_bits, end_bit);
    VSG_HIP(hipGetLastError());
    return;
  }
  VSG_REQUIRE(temp_bytes >= SortPairsU32HandTempBytes(n), -4, "radix sort: temporary storage too small");
  const int T = RsTiles(n);
  uint8_t* base = static_cast<uint8_t*>(temp);
  ScanScratch sc{reinterpret_cast<int32_t*>(base)};
  int32_t* hist = reinterpret_cast<int32_t*>(base + RsAlign(kScanMaxTiles * sizeof(int32_t)));
  uint8_t* pp = reinterpret_cast<uint8_t*>(hist) + RsAlign((size_t)T * kRsBins * sizeof(int32_t));
  uint32_t* tk = reinterpret_cast<uint32_t*>(pp);
  uint32_t* tv = reinterpret_cast<uint32_t*>(pp + RsAlign((size_t)n * sizeof(uint32_t)));
  const uint32_t* ki = keys_in;
  const uint32_t* vi = vals_in;
  for (int p = 0; p < passes; ++p) {
    // the last pass lands in the output, the passes before alternate between it and the temporary pair
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    uint32_t* ko = to_out ? keys_out : tk;
    uint32_t* vo = to_out ? vals_out : tv;
    const int shift = p * digit_bits;
    const int bits = std::min(digit_bits, end_bit - shift);   // (the last digit ends with the key)
    hipLaunchKernelGGL(k_rs_hist, dim3(T), dim3(256), 0, s, ki, n, shift, bits, T, hist);
    ExclusiveSum(sc, hist, hist, (1 << bits) * T, s);
    hipLaunchKernelGGL(k_rs_scatter, dim3(T), dim3(256), 0, s, ki, vi, ko, vo, n, shift, bits, T, hist);
    ki = ko;
    vi = vo;
  }
  VSG_HIP(hipGetLastError());
}

// ---- what the merge calls ------------------------------------------------------------------------------
// Measured on an MI355X (tools/sort_probe.py, 26-bit keys, us per sort): rocPRIM 8 / 16 / 39 /
// 98 / 131 / 164 / 180 / 567 at 1 K / 4 K / 64 K / 256 K / 512 K / 1 M / 4 M / 16 M pairs, this file 28 / 28 /
// 69 / 88 / 113 / 117 / 271 / 907: the library's single-workgroup sort and its merge sort win below a
// quarter of a million pairs, its one-sweep radix sort above a few million; in between three passes
// of 9 bits beat four of 8.  So the merge uses this file from 400 K to 2 M pairs and the library
// elsewhere (VSG_SORT_HAND=lo:hi moves the window; the tests run 0:2000000000 and an empty window).
static void SortHandWindow(int& lo, int& hi) {
  lo = 400000;
  hi = 2000000;
  if (const char* e = getenv("VSG_SORT_HAND")) {   // (read per call: the tests switch it inside one process)
    long long x = 0, y = 0;
    if (sscanf(e, "%lld:%lld", &x, &y) == 2) {
      lo = (int)std::min<long long>(std::max<long long>(x, 0), 0x7fffffff);
      hi = (int)std::min<long long>(std::max<long long>(y, -1), 0x7fffffff);
    }
  }
}

size_t SortPairsU32TempBytes(int n) {
  return std::max(SortPairsU32HandTempBytes(n), SortPairsU32LibTempBytes(n));
}

void SortPairsU32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                  const uint32_t* vals_in, uint32_t* vals_out, int n, int end_bit, hipStream_t s) {
  int lo, hi;
  SortHandWindow(lo, hi);
  if (n >= lo && n <= hi) {
    SortPairsU32Hand(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, s);
  } else {
    SortPairsU32Lib(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, s);
  }
}

}  // namespace vsg
