// Microbenchmark (gfx950): the inner loop of k_spine_chain -- h = u + c*h with the coefficients
// read from LDS (ds_read_b128, one block of 16 steps ahead) -- in variants, to see what the
// 7.4 ns per step are made of.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off ... && ./chain_lds
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kRing = 4096;
struct Ring {
  alignas(16) float u[3][kRing];
  alignas(16) float c[kRing];
};

template <int kVariant>
__global__ __launch_bounds__(64) void k_chain(float* out, float* ck, int fills) {
  __shared__ Ring ring;
  for (int i = threadIdx.x; i < kRing; i += 64) {
    ring.u[0][i] = ring.u[1][i] = ring.u[2][i] = 0.25f;
    ring.c[i] = 0.5f;
  }
  __syncthreads();
  const int lane = threadIdx.x;
  const int ch = lane % 3;
  float h = out[lane];
  const float* urow = ring.u[ch];
  const float* crow = ring.c;
  if (kVariant & 8) __builtin_amdgcn_s_setprio(3);
  if ((kVariant & 16) && lane >= 3) return;   // only the three channel lanes stay
  for (int f = 0; f < fills; ++f) {
    const int slot0 = (f * 256) & (kRing - 1);
    float4 un[4], cn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      un[q] = *reinterpret_cast<const float4*>(urow + slot0 + 4 * q);
      cn[q] = *reinterpret_cast<const float4*>(crow + slot0 + 4 * q);
    }
    for (int b = 0; b < 16; ++b) {
      float4 u[4], c[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u[q] = un[q];
        c[q] = cn[q];
      }
      if (!(kVariant & 1) && b + 1 < 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          un[q] = *reinterpret_cast<const float4*>(urow + slot0 + 16 * (b + 1) + 4 * q);
          cn[q] = *reinterpret_cast<const float4*>(crow + slot0 + 16 * (b + 1) + 4 * q);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h = u[q].x + c[q].x * h;
        h = u[q].y + c[q].y * h;
        h = u[q].z + c[q].z * h;
        h = u[q].w + c[q].w * h;
      }
      if (!(kVariant & 2)) {
        if (lane < 3) ck[ch * 65536 + ((f * 16 + b) & 65535)] = h;
      }
    }
    if (!(kVariant & 4)) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_store(reinterpret_cast<int*>(&ring.c[0]) + 0, 0x3f000000, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  out[lane] = h;
}

template <int V>
void run(const char* what, float* out, float* ck) {
  const int fills = 4096;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_chain<V>, dim3(1), dim3(64), 0, 0, out, ck, 16);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_chain<V>, dim3(1), dim3(64), 0, 0, out, ck, fills);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-60s %.2f ns / step\n", what, ms * 1e6 / (fills * 256.0));
}

int main() {
  float *out, *ck;
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&ck, 3 * 65536 * 4);
  (void)hipMemset(out, 0, 4096);
  run<0>("as in k_spine_chain (LDS reads, store, fill hand-shake)", out, ck);
  run<2>("without the global store", out, ck);
  run<1>("without the LDS reads in the loop", out, ck);
  run<3>("without both", out, ck);
  run<7>("without both and without the fill hand-shake", out, ck);
  run<8>("as in k_spine_chain, s_setprio 3", out, ck);
  run<16>("as in k_spine_chain, lanes 3..63 retired", out, ck);
  run<18>("lanes 3..63 retired, without the global store", out, ck);
  return 0;
}
