// build_kernels.hip -- graph construction kernels (gfx950).
//
//   K0  k_minmax_u8          global min / max of the u8 frame (feeds the bilateral LUT scale)
//   K1  k_bilateral          u8 BGR -> f32 * (1/255) -> 49-tap LUT bilateral -> planar f32
//       k_convert_planar     PRESMOOTH_NONE variant
//       k_gaussian3          PRESMOOTH_GAUSSIAN variant (3x3, sigma 1.5)
//   K2  k_init_nodes / k_init_virtual_nodes
//   (K3 / K4 / K5 -- edge keys and the stable bucket sort -- are in edge_sort.hip)
//
// Reference behaviour restated (paths relative to the reference root):
//   imagefilter/image_filter.cpp:130-167, 184-277        (bilateral)
//   segmentation/dense_segmentation.cpp:164-198          (u8 -> f32 scale)
//   segmentation/pixel_distance.h:141-157                (ColorDiff3L1 / ColorDiff3L2)
//   segmentation/segmentation_graph.h:158-162, 336       (bucket index)
//   segmentation/dense_segmentation_graph.h:956-1142     (edge enumeration order)
//
// These are HBM-bound stencil / integer kernels: coalesced plane loads, LDS tile for the
// bilateral window, no MFMA.
#include <atomic>

#include "device_graph.h"

namespace vsg {

__constant__ float c_space_w[64];   // 49 spatial weights, tap order of image_filter.cpp:216-225

void UploadSpaceWeights(const float* w49, hipStream_t stream) {
  VSG_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_space_w), w49, 49 * sizeof(float), 0,
                                 hipMemcpyHostToDevice, stream));
}

// ------------------------------------------------------------------------------------------
// K0: min / max over all bytes of the frame (HBM bound: 3 B per pixel read once).
// One wavefront-wide 16-byte load per lane and step; rows may be padded (stride > 3 W), so a row
// is walked as [unaligned head bytes | 16-byte words | tail bytes].
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void MinMaxWord(uint32_t w, int& lo, int& hi) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int v = (int)((w >> (8 * k)) & 0xffu);
    lo = min(lo, v);
    hi = max(hi, v);
  }
}

__global__ __launch_bounds__(256) void k_minmax_u8(const uint8_t* __restrict__ bgr, size_t stride,
                                                    int row_bytes, int H, int* __restrict__ mm) {
  int lo = 255, hi = 0;
  // blockIdx.y walks the rows, the threads of blockIdx.x the 16-byte words of a row
  for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const uint8_t* row = bgr + (size_t)y * stride;
    const int head = (int)((16 - ((uintptr_t)row & 15)) & 15);          // bytes before alignment
    const int head_n = min(head, row_bytes);
    const int words = (row_bytes - head_n) >> 4;
    const int tail0 = head_n + (words << 4);
    const uint4* w16 = reinterpret_cast<const uint4*>(row + head_n);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) {
      const uint4 v = w16[i];
      MinMaxWord(v.x, lo, hi);
      MinMaxWord(v.y, lo, hi);
      MinMaxWord(v.z, lo, hi);
      MinMaxWord(v.w, lo, hi);
    }
    if (blockIdx.x == 0) {
      const int t = threadIdx.x;
      if (t < head_n) {
        const int v = row[t];
        lo = min(lo, v);
        hi = max(hi, v);
      }
      if (tail0 + t < row_bytes && t < 16) {
        const int v = row[tail0 + t];
        lo = min(lo, v);
        hi = max(hi, v);
      }
    }
  }
  // one pair of atomics per workgroup (same-address atomics serialise in L2)
  __shared__ int red[2][4];
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_down(lo, off));
    hi = max(hi, __shfl_down(hi, off));
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = lo;
    red[1][threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&mm[0], min(min(red[0][0], red[0][1]), min(red[0][2], red[0][3])));
    atomicMax(&mm[1], max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3])));
  }
}

void LaunchMinMax(const uint8_t* bgr, size_t stride, int W, int H, int* mm, hipStream_t s) {
  const int init[2] = {255, 0};
  VSG_HIP(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, s));
  const int row_bytes = W * 3;
  const int gx = max(1, min(8, (row_bytes / 16 + 255) / 256));
  const int gy = max(1, min(H, 512 / gx));
  hipLaunchKernelGGL(k_minmax_u8, dim3(gx, gy), dim3(256), 0, s, bgr, stride, row_bytes, H, mm);
  VSG_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// K1: bilateral filter.  One workgroup = 1024 threads = a 64x64 output tile, staged with its 4 px
// replicate halo in LDS next to the 12288-entry exp LUT (48 KiB + 61 KiB: one workgroup of sixteen
// wavefronts per CU, four per SIMD -- the kernel is a stream of LDS reads and LUT gathers whose
// latency needs that many wavefronts to hide; 256 threads with their own LUT copy left one or two
// per SIMD, the SIMDs half idle).  A thread filters two vertically adjacent pixels A (row y) and
// B (row y + 1) together: their 9x9 windows overlap in eight rows, so a tile value is read once for
// both (58 positions instead of 2 x 49 taps), and the arithmetic of the two runs as packed f32
// pairs {A, B} (v_pk_mul_f32 / v_pk_add_f32: IEEE single, no contraction -- the same roundings as
// two scalar operations).  The accumulation order over the 49 taps of EACH pixel (rows outer,
// columns inner) and every rounding step follow ParallelBilateralColor::operator() exactly; a
// position that is a tap of only one of the two contributes weight 0 to the other (x + 0 = x,
// l * 0 = +0: exact, the sums are non-negative).
// ------------------------------------------------------------------------------------------
constexpr int kTileW = 64, kTileH = 64, kRad = 4;
constexpr int kHaloW = kTileW + 2 * kRad, kHaloH = kTileH + 2 * kRad;
constexpr int kLutBins = 12288;
constexpr int kBilThreads = 1024;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Tap (i, j) of the circular window, and its index in the reference's enumeration (rows i = -4 .. 4
// outer, columns j inner, kept if i^2 + j^2 <= 16: image_filter.cpp:216-225).
constexpr bool TapIn(int i, int j) {
  return i >= -kRad && i <= kRad && j >= -kRad && j <= kRad && i * i + j * j <= kRad * kRad;
}
constexpr int TapIdx(int i, int j) {
  int k = 0;
  for (int a = -kRad; a <= kRad; ++a) {
    for (int b = -kRad; b <= kRad; ++b) {
      if (a == i && b == j) return k;
      if (a * a + b * b <= kRad * kRad) ++k;
    }
  }
  return -1;
}

__global__ __launch_bounds__(kBilThreads) void k_bilateral(const uint8_t* __restrict__ bgr, size_t stride,
                                                            int W, int H, const float* __restrict__ lut_g,
                                                            float scale, float* __restrict__ out_b,
                                                            float* __restrict__ out_g,
                                                            float* __restrict__ out_r, int tiles_x,
                                                            int num_tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* lut = smem;                         // [12288]
  float* tb = smem + kLutBins;               // [kHaloH * kHaloW]
  float* tg = tb + kHaloH * kHaloW;
  float* tr = tg + kHaloH * kHaloW;
  const int tid = threadIdx.x;
  for (int i = tid; i < kLutBins; i += kBilThreads) lut[i] = lut_g[i];
  const float c255 = (float)(1.0 / 255.0);   // convertTo(CV_32FC3, 1.0/255.0)
  const f32x2 scale2 = {scale, scale};

  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int x0 = (tile % tiles_x) * kTileW;
    const int y0 = (tile / tiles_x) * kTileH;
    __syncthreads();
    for (int idx = tid; idx < kHaloH * kHaloW; idx += kBilThreads) {
      const int ty = idx / kHaloW, tx = idx - ty * kHaloW;
      const int gy = min(max(y0 + ty - kRad, 0), H - 1);   // BORDER_REPLICATE
      const int gx = min(max(x0 + tx - kRad, 0), W - 1);
      const uint8_t* p = bgr + (size_t)gy * stride + (size_t)gx * 3;
      tb[idx] = (float)p[0] * c255;
      tg[idx] = (float)p[1] * c255;
      tr[idx] = (float)p[2] * c255;
    }
    __syncthreads();
    const int tx = tid & 63;
    const int x = x0 + tx;
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
      const int ty = 2 * ((tid >> 6) + 16 * r);   // rows ty (pixel A) and ty + 1 (pixel B)
      const int y = y0 + ty;
      if (x >= W || y >= H) continue;
      const int c = (ty + kRad) * kHaloW + (tx + kRad);   // A's centre; B's is one tile row below
      const f32x2 my_b = {tb[c], tb[c + kHaloW]};
      const f32x2 my_g = {tg[c], tg[c + kHaloW]};
      const f32x2 my_r = {tr[c], tr[c + kHaloW]};
      f32x2 weight_sum = {0.0f, 0.0f}, sum_b = {0.0f, 0.0f}, sum_g = {0.0f, 0.0f}, sum_r = {0.0f, 0.0f};
#pragma unroll
      for (int t = -kRad; t <= kRad + 1; ++t) {      // tile row relative to A: A's tap row t, B's tap row t - 1
#pragma unroll
        for (int j = -kRad; j <= kRad; ++j) {
          const bool in_a = TapIn(t, j), in_b = TapIn(t - 1, j);
          if (!in_a && !in_b) continue;
          const int o = c + t * kHaloW + j;
          const float lb = tb[o], lg = tg[o], lr = tr[o];
          const f32x2 l_b = {lb, lb}, l_g = {lg, lg}, l_r = {lr, lr};
          const f32x2 diff_b = my_b - l_b;
          const f32x2 diff_g = my_g - l_g;
          const f32x2 diff_r = my_r - l_r;
          const f32x2 d2 = (diff_b * diff_b + diff_g * diff_g + diff_r * diff_r) * scale2;
          f32x2 weight = {0.0f, 0.0f};
          if (in_a) weight.x = c_space_w[TapIdx(t, j)] * lut[(int)d2.x];
          if (in_b) weight.y = c_space_w[TapIdx(t - 1, j)] * lut[(int)d2.y];
          weight_sum += weight;
          sum_b += l_b * weight;
          sum_g += l_g * weight;
          sum_r += l_r * weight;
        }
      }
      const size_t pix = (size_t)y * W + x;
      // weight_sum = 1.0 / weight_sum in double, rounded to float == IEEE 1.0f / weight_sum.
      if (weight_sum.x > 0) {
        const float inv = 1.0f / weight_sum.x;
        out_b[pix] = sum_b.x * inv;
        out_g[pix] = sum_g.x * inv;
        out_r[pix] = sum_r.x * inv;
      } else {
        out_b[pix] = 0.0f;
        out_g[pix] = 0.0f;
        out_r[pix] = 0.0f;
      }
      if (y + 1 < H) {
        if (weight_sum.y > 0) {
          const float inv = 1.0f / weight_sum.y;
          out_b[pix + W] = sum_b.y * inv;
          out_g[pix + W] = sum_g.y * inv;
          out_r[pix + W] = sum_r.y * inv;
        } else {
          out_b[pix + W] = 0.0f;
          out_g[pix + W] = 0.0f;
          out_r[pix + W] = 0.0f;
        }
      }
    }
  }
}

constexpr int kMaxAttrDevices = 64;
size_t BilateralSmemBytes() { return (size_t)(kLutBins + 3 * kHaloH * kHaloW) * sizeof(float); }

void LaunchBilateral(const uint8_t* bgr, size_t stride, int W, int H, const float* lut,
                     float scale, float* planes /* 3*W*H */, hipStream_t s) {
  // The attribute belongs to the (function, device) pair: a process may hold handles on several
  // devices, and stream threads call this concurrently.
  static std::atomic<bool> attr_set[kMaxAttrDevices];
  const size_t smem = BilateralSmemBytes();
  int dev = 0;
  VSG_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxAttrDevices || !attr_set[dev].load(std::memory_order_acquire)) {
    VSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bilateral),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (dev >= 0 && dev < kMaxAttrDevices) attr_set[dev].store(true, std::memory_order_release);
  }
  const int tiles_x = (W + kTileW - 1) / kTileW;
  const int tiles_y = (H + kTileH - 1) / kTileH;
  const int num_tiles = tiles_x * tiles_y;
  const int grid = min(num_tiles, 256);   // one workgroup per CU (109 KiB LDS each), persistent
  const size_t n = (size_t)W * H;
  hipLaunchKernelGGL(k_bilateral, dim3(grid), dim3(kBilThreads), smem, s, bgr, stride, W, H, lut, scale,
                     planes, planes + n, planes + 2 * n, tiles_x, num_tiles);
  VSG_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_convert_planar(const uint8_t* __restrict__ bgr,
                                                         size_t stride, int W, int H,
                                                         float* __restrict__ planes) {
  const size_t n = (size_t)W * H;
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
  const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
  const uint8_t* p = bgr + (size_t)y * stride + (size_t)x * 3;
  const float c255 = (float)(1.0 / 255.0);
  planes[pix] = (float)p[0] * c255;
  planes[n + pix] = (float)p[1] * c255;
  planes[2 * n + pix] = (float)p[2] * c255;
}

void LaunchConvertPlanar(const uint8_t* bgr, size_t stride, int W, int H, float* planes,
                         hipStream_t s) {
  const size_t n = (size_t)W * H;
  hipLaunchKernelGGL(k_convert_planar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bgr,
                     stride, W, H, planes);
  VSG_HIP(hipGetLastError());
}

// PRESMOOTH_GAUSSIAN: cv::GaussianBlur(f32 BGR, Size(3, 3), sigma 1.5), dense_segmentation.cpp:186-188.
// OpenCV (2.4 line, un-vendored) filters separably with a symmetric 3-tap kernel {k1, k0, k1}: rows first,
//   r(y, x) = S(y, x) * k0 + (S(y, x - 1) + S(y, x + 1)) * k1,
// then columns with the same expression over r, every product and sum rounded to f32, border
// BORDER_REFLECT_101 (-1 -> 1, n -> n - 2).  One thread per pixel evaluates the three row values it
// needs itself: 27 bytes of input per pixel from L1/L2, no intermediate plane.  Parity unpinned.
__global__ __launch_bounds__(256) void k_gaussian3(const uint8_t* __restrict__ bgr, size_t stride, int W,
                                                    int H, float k0, float k1,
                                                    float* __restrict__ planes) {
  const size_t n = (size_t)W * H;
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
  const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
  const int xl = x > 0 ? x - 1 : (W > 1 ? 1 : 0), xr = x + 1 < W ? x + 1 : (W > 1 ? W - 2 : 0);
  const int yu = y > 0 ? y - 1 : (H > 1 ? 1 : 0), yd = y + 1 < H ? y + 1 : (H > 1 ? H - 2 : 0);
  const float c255 = (float)(1.0 / 255.0);
  const int rows[3] = {yu, y, yd};
  float r[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint8_t* p = bgr + (size_t)rows[k] * stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sl = (float)p[(size_t)xl * 3 + c] * c255;
      const float sc = (float)p[(size_t)x * 3 + c] * c255;
      const float sr = (float)p[(size_t)xr * 3 + c] * c255;
      r[k][c] = sc * k0 + (sl + sr) * k1;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) planes[(size_t)c * n + pix] = r[1][c] * k0 + (r[0][c] + r[2][c]) * k1;
}

void LaunchGaussian3(const uint8_t* bgr, size_t stride, int W, int H, float k0, float k1, float* planes,
                     hipStream_t s) {
  const size_t n = (size_t)W * H;
  hipLaunchKernelGGL(k_gaussian3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bgr, stride, W, H,
                     k0, k1, planes);
  VSG_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_interleaved_to_planar(const float* __restrict__ in,
                                                                size_t n,
                                                                float* __restrict__ planes) {
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
  planes[pix] = in[pix * 3];
  planes[n + pix] = in[pix * 3 + 1];
  planes[2 * n + pix] = in[pix * 3 + 2];
}

__global__ __launch_bounds__(256) void k_planar_to_interleaved(const float* __restrict__ planes,
                                                                size_t n, float* __restrict__ out) {
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
  out[pix * 3] = planes[pix];
  out[pix * 3 + 1] = planes[n + pix];
  out[pix * 3 + 2] = planes[2 * n + pix];
}

void LaunchInterleavedToPlanar(const float* in, size_t n, float* planes, hipStream_t s) {
  hipLaunchKernelGGL(k_interleaved_to_planar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     in, n, planes);
  VSG_HIP(hipGetLastError());
}

void LaunchPlanarToInterleaved(const float* planes, size_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_planar_to_interleaved, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     planes, n, out);
  VSG_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// K2: node initialisation (AddNodes[Constrained]WithDescriptors, AddVirtualNodesConstrained;
// dense_segmentation_graph.h:327-367, 1180-1228).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_nodes(const float* __restrict__ feat, size_t n,
                                                     int base, const int32_t* __restrict__ cons_in,
                                                     int32_t* __restrict__ parent,
                                                     float4* __restrict__ desc_sz,
                                                     int32_t* __restrict__ cons,
                                                     uint8_t* __restrict__ flags) {
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
  const size_t node = (size_t)base + pix;
  parent[node] = (int32_t)node;
  desc_sz[node] = make_float4(feat[pix], feat[n + pix], feat[2 * n + pix], __int_as_float(1));
  cons[node] = cons_in ? cons_in[pix] : -1;
  flags[node] = 0;
}

void LaunchInitNodes(const float* feat, size_t n, int base, const int32_t* cons_in,
                     NodeArrays nodes, hipStream_t s) {
  hipLaunchKernelGGL(k_init_nodes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, feat, n,
                     base, cons_in, nodes.parent, nodes.desc_sz, nodes.cons, nodes.flags);
  VSG_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_first_of_label(const int32_t* __restrict__ labels,
                                                         size_t n, int num_labels,
                                                         int32_t* __restrict__ first) {
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
  const int l = labels[pix];
  // The first pixel of a label in scan order starts a run of that label, so only run starts
  // compete (a few thousand atomics instead of one per pixel on a few hundred addresses).
  if (pix > 0 && labels[pix - 1] == l) return;
  if (l >= 0 && l < num_labels && first[l] > (int)pix) atomicMin(&first[l], (int)pix);
}

__global__ __launch_bounds__(256) void k_init_virtual_nodes(const int32_t* __restrict__ labels,
                                                             size_t n, int base, int num_labels,
                                                             const int32_t* __restrict__ first,
                                                             int32_t* __restrict__ parent,
                                                             float4* __restrict__ desc_sz,
                                                             int32_t* __restrict__ cons,
                                                             uint8_t* __restrict__ flags) {
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
  const size_t node = (size_t)base + pix;
  const int l = labels[pix];
  const int rep = (l >= 0 && l < num_labels) ? first[l] : (int)pix;
  parent[node] = base + rep;
  desc_sz[node] = make_float4(0.f, 0.f, 0.f, __int_as_float(0));   // size 0, descriptor unused
  cons[node] = l;
  flags[node] = kFlagNoDesc;
}

void LaunchInitVirtualNodes(const int32_t* labels, size_t n, int base, int num_labels,
                            int32_t* first_scratch, NodeArrays nodes, hipStream_t s) {
  VSG_HIP(hipMemsetAsync(first_scratch, 0x7f, (size_t)num_labels * sizeof(int32_t), s));
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_first_of_label, dim3(grid), dim3(256), 0, s, labels, n, num_labels,
                     first_scratch);
  hipLaunchKernelGGL(k_init_virtual_nodes, dim3(grid), dim3(256), 0, s, labels, n, base,
                     num_labels, first_scratch, nodes.parent, nodes.desc_sz, nodes.cons,
                     nodes.flags);
  VSG_HIP(hipGetLastError());
}

}  // namespace vsg
