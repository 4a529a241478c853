// Microbenchmark (gfx950): the recurrence of k_spine_chain,  h = u + c*h,  with the coefficients
// streamed through SCALAR loads (one wavefront per channel: u and c are wave-uniform, so they can
// live in SGPRs and cost the chain wavefront no LDS / vector-memory issue slots), five register sets
// of eight steps, 1-4 of them in flight.  Result on MI355X (profiles/r5_micro_chain_sgpr.txt):
// 14.5 ns per step with 32 steps of look-ahead, 22.7 with 8 -- scalar loads return out of order, so
// every wait is s_waitcnt lgkmcnt(0) = a wait for the load issued LAST, and a scalar load that misses
// the scalar cache takes ~430 cycles.  The LDS form stays (6.2-7.4 ns per step, chain_lds.hip).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/chain_sgpr.hip -o /tmp/chain_sgpr && /tmp/chain_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kHalf = 8;
constexpr int kSets = 5;

struct CoefSet {
  float u[kHalf];
  float c[kHalf];
};

__device__ __forceinline__ void LoadSet(CoefSet& s, const float* __restrict__ u, const float* __restrict__ c, int at) {
#pragma unroll
  for (int q = 0; q < kHalf; ++q) {
    s.u[q] = u[at + q];
    s.c[q] = c[at + q];
  }
}

template <int kLook>
__global__ __launch_bounds__(64) void k_chain(const float* __restrict__ u_all, const float* __restrict__ c_all, int stride,
                                               int n, float* __restrict__ ck, float* __restrict__ out,
                                               long long* __restrict__ cyc) {
  const int ch = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  const float* __restrict__ u = u_all + (size_t)ch * stride;
  const float* __restrict__ c = c_all;
  float h = out[ch];
  CoefSet S[kSets];
#pragma unroll
  for (int r = 0; r < kLook; ++r) LoadSet(S[r], u, c, r * kHalf);
  const long long t0 = __builtin_readcyclecounter();
  const int nh = n / kHalf;
  for (int hb = 0; hb < nh; hb += kSets) {
#pragma unroll
    for (int r = 0; r < kSets; ++r) {
      LoadSet(S[(r + kLook) % kSets], u, c, (hb + r + kLook) * kHalf);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < kHalf; ++q) h = S[r].u[q] + S[r].c[q] * h;
      if (((hb + r) & 1) && threadIdx.x == 0) ck[(size_t)ch * (n / 16 + 8) + ((hb + r) >> 1)] = h;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[ch] = h;
  if (threadIdx.x == 0) cyc[ch] = t1 - t0;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class Kern>
int Run(Kern kern, const char* name, float* du, float* dc, int stride, int n, float* ck, float* out, long long* cyc) {
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(3), dim3(64), 0, 0, du, dc, stride, n, ck, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long hc3[3];
    CHECK(hipMemcpy(hc3, cyc, sizeof(hc3), hipMemcpyDeviceToHost));
    std::printf("%s: %.3f ms for %d steps x 3 channels -> %.2f ns/step (wall), %.1f counter ticks/step (channel 0)\n", name,
                ms, n, ms * 1e6 / n, (double)hc3[0] / n);
  }
  return 0;
}

int main() {
  const int n = 40 * 16 * 1000;   // steps (a multiple of 40)
  const int stride = n + 256;
  std::vector<float> hu(3 * (size_t)stride, 0.25f), hc(stride, 0.5f);
  float *du, *dc, *ck, *out;
  long long* cyc;
  CHECK(hipMalloc(&du, hu.size() * 4));
  CHECK(hipMalloc(&dc, hc.size() * 4));
  CHECK(hipMalloc(&ck, 3 * (size_t)(n / 16 + 8) * 4));
  CHECK(hipMalloc(&out, 64 * 4));
  CHECK(hipMalloc(&cyc, 3 * 8));
  CHECK(hipMemcpy(du, hu.data(), hu.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(out, 0, 64 * 4));
  if (Run(k_chain<4>, "lookahead 4 sets (32 steps)", du, dc, stride, n, ck, out, cyc)) return 1;
  if (Run(k_chain<3>, "lookahead 3 sets (24 steps)", du, dc, stride, n, ck, out, cyc)) return 1;
  if (Run(k_chain<2>, "lookahead 2 sets (16 steps)", du, dc, stride, n, ck, out, cyc)) return 1;
  if (Run(k_chain<1>, "lookahead 1 set  ( 8 steps)", du, dc, stride, n, ck, out, cyc)) return 1;
  float ho[3];
  CHECK(hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost));
  std::printf("h = %g %g %g (expect 0.5)\n", ho[0], ho[1], ho[2]);
  return 0;
}
