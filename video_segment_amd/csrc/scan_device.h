// scan_device.h -- device-wide exclusive prefix sums for the merge path, hand-written for gfx950, with
// the producer of the values and the consumer of the prefixes fused into the scan itself.
//
// The merge is a chain of "flag the edges that ..., number them, move them together" steps (tree
// edges, side-cluster edges, spine edges, run leaders, runs of equal component keys).  As separate
// kernels around a library scan each of them is four or five launches -- flag kernel, the scan's own
// two, a one-thread kernel that reports the total, the compaction -- in a path whose cost IS its
// launch count.  Here a step is two launches:
//
//   k_scan_sums  : tile t adds up value(i) over its elements                       -> sums[t]
//   k_scan_emit  : tile t adds up sums[0 .. t) (a few hundred loads, from L2), scans its own
//                  elements in order and hands every element its exclusive prefix: emit(i, value, prefix);
//                  the last tile knows the grand total: finish(total)  (device count, mailbox post)
//
// value(i) is evaluated twice (both kernels), so it has to be cheap and must not change in between.
// No look-back between workgroups, no spinning: nothing here can wait for a workgroup that has not
// been scheduled.  A tile is 2048 elements (256 threads x 8) times `chunks`; the
// launcher picks `chunks` so that there are at most a few thousand tiles (the sum over sums[0 .. t)
// is quadratic in their number).
#ifndef VSG_SCAN_DEVICE_H_
#define VSG_SCAN_DEVICE_H_

#include "merge_common.h"

namespace vsg {

constexpr int kScanItems = 8;
constexpr int kScanSub = 256 * kScanItems;   // elements per sub-tile

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

__device__ __forceinline__ int BlockSum256(int v, int32_t* lds4) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  const int total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return total;
}

template <class Value>
__global__ __launch_bounds__(256) void k_scan_sums(Value value, int n, int chunks, int32_t* __restrict__ sums) {
  __shared__ int32_t lds4[4];
  const long long base = (long long)blockIdx.x * chunks * kScanSub;
  int v = 0;
  for (int c = 0; c < chunks; ++c) {
    const long long sub = base + (long long)c * kScanSub;
    if (sub >= n) break;
    // (any order will do for a sum: strided, coalesced)
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      const long long i = sub + k * 256 + threadIdx.x;
      if (i < n) v += value((int)i);
    }
  }
  const int total = BlockSum256(v, lds4);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// LDS position of element j of a sub-tile: one pad word per 32 keeps both the striped (j = k * 256 +
// thread) and the blocked (j = thread * 8 + k) accesses spread over the banks.
__device__ __forceinline__ int ScanLdsAt(int j) { return j + (j >> 5); }

template <class Value, class Emit, class Finish>
__global__ __launch_bounds__(256) void k_scan_emit(Value value, Emit emit, Finish finish, int n, int chunks,
                                                    const int32_t* __restrict__ sums) {
  __shared__ int32_t lds4[4];
  __shared__ int32_t tile[kScanSub + kScanSub / 32];
  int before = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) before += sums[t];
  int run = BlockSum256(before, lds4);   // everything in front of this tile
  const long long base = (long long)blockIdx.x * chunks * kScanSub;
  for (int c = 0; c < chunks; ++c) {
    const long long sub = base + (long long)c * kScanSub;
    if (sub >= n) break;
    // value() and emit() see the elements striped over the threads (element k * 256 + thread: every
    // access they make is coalesced); the scan itself wants eight consecutive elements per thread:
    // the values go through LDS into that order and the prefixes come back the same way.
    int xs[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      const long long i = sub + k * 256 + threadIdx.x;
      xs[k] = i < n ? value((int)i) : 0;
      tile[ScanLdsAt(k * 256 + (int)threadIdx.x)] = xs[k];
    }
    __syncthreads();
    int xb[kScanItems];
    int v = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      xb[k] = tile[ScanLdsAt((int)threadIdx.x * kScanItems + k)];
      v += xb[k];
    }
    int total;
    int at = run + BlockExclusiveScan256(v, lds4, total);   // (two barriers: every thread has read its eight)
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      tile[ScanLdsAt((int)threadIdx.x * kScanItems + k)] = at;
      at += xb[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      const long long i = sub + k * 256 + threadIdx.x;
      if (i < n) emit((int)i, xs[k], tile[ScanLdsAt(k * 256 + (int)threadIdx.x)]);
    }
    __syncthreads();   // (the next sub-tile overwrites the tile)
    run += total;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) finish(run);
}

struct ScanNoFinish {
  __device__ void operator()(int) const {}
};

// emit(i, value(i), sum of value(j) for j < i) for every i < n; finish(sum of all) once.
template <class Value, class Emit, class Finish>
inline void FusedScan(const ScanScratch& sc, Value value, Emit emit, Finish finish, int n, hipStream_t s) {
  if (n <= 0) return;
  long long tiles = ((long long)n + kScanSub - 1) / kScanSub;
  int chunks = 1;
  while ((tiles + chunks - 1) / chunks > 2048) chunks *= 2;   // at most 2048 tiles (one million elements per tile at 2^31)
  const int grid = (int)((tiles + chunks - 1) / chunks);
  VSG_REQUIRE(grid <= kScanMaxTiles, -4, "scan: too many tiles");
  // (two scans in flight on different streams would share the tile sums)
  VSG_REQUIRE(sc.owner == nullptr || sc.owner == s, -4, "scan: issued on a stream that does not own the scratch");
  hipLaunchKernelGGL((k_scan_sums<Value>), dim3(grid), dim3(256), 0, s, value, n, chunks, sc.sums);
  hipLaunchKernelGGL((k_scan_emit<Value, Emit, Finish>), dim3(grid), dim3(256), 0, s, value, emit, finish, n, chunks,
                     sc.sums);
  VSG_HIP(hipGetLastError());
}

// ---- the plain forms -----------------------------------------------------------------------------
struct ScanLoadI32 {
  const int32_t* in;
  __device__ int operator()(int i) const { return in[i]; }
};
struct ScanStoreI32 {
  int32_t* out;
  __device__ void operator()(int i, int, int prefix) const { out[i] = prefix; }
};
// out[i] = in[0] + ... + in[i - 1]
inline void ExclusiveSum(const ScanScratch& sc, const int32_t* in, int32_t* out, int n, hipStream_t s) {
  FusedScan(sc, ScanLoadI32{in}, ScanStoreI32{out}, ScanNoFinish{}, n, s);
}

// Runs of equal keys in a sorted array: seg_off[r] = first position of run r, *num_runs = number of
// runs, then seg_cnt[r] = length of run r (a third launch, over the positions).  With the mailbox
// arguments the number of runs is posted as value `mail_idx` of the slot.
struct RunHeadValue {
  const uint32_t* keys;
  __device__ int operator()(int i) const { return (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0; }
};
struct RunHeadEmit {
  int32_t* seg_off;
  __device__ void operator()(int i, int head, int prefix) const {
    if (head) seg_off[prefix] = i;
  }
};
struct RunHeadFinish {
  int32_t* num_runs;
  __device__ void operator()(int total) const { *num_runs = total; }
};
void RunsOfSortedKeys(const ScanScratch& sc, const uint32_t* keys, int n, int32_t* seg_off, int32_t* seg_cnt,
                      int32_t* num_runs, hipStream_t s);

}  // namespace vsg

#endif  // VSG_SCAN_DEVICE_H_
