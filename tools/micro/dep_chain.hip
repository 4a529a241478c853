// Microbenchmark (gfx950): cycles per dependent  h = u + c*h  step (v_mul_f32 + v_add_f32, no
// contraction) for one wavefront alone on its SIMD, with the coefficients in registers.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/dep_chain.hip -o /tmp/dep_chain && /tmp/dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_chain(float* out, long long* cyc, int iters, float u, float c) {
  float h = out[threadIdx.x];
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) h = u + c * h;
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = h;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_chain2(float* out, long long* cyc, int iters, float u, float c) {   // two independent chains
  float h = out[threadIdx.x], g = h + 1.0f;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      h = u + c * h;
      g = u + c * g;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = h + g;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 4096);
  hipMalloc(&cyc, 64);
  hipMemset(out, 0, 4096);
  const int iters = 1 << 16;
  for (int rep = 0; rep < 2; ++rep) {
    for (int threads : {64, 3}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 0.5f, 0.999f);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long c;
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("one chain, %2d lanes: %.2f counter ticks / step, %.2f ns / step\n", threads, (double)c / (iters * 16.0),
             ms * 1e6 / (iters * 16.0));
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_chain2, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 0.5f, 0.999f);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("two chains, %2d lanes: %.2f ticks / step pair, %.2f ns / step pair\n", threads,
             (double)c / (iters * 16.0), ms * 1e6 / (iters * 16.0));
    }
  }
  return 0;
}
