// build_kernels.hip -- graph construction kernels (gfx950).
//
//   K0  k_minmax_u8          global min / max of the u8 frame (feeds the bilateral LUT scale)
//   K1  k_bilateral          u8 BGR -> f32 * (1/255) -> 49-tap LUT bilateral -> planar f32
//       k_convert_planar     PRESMOOTH_NONE variant
//   K2  k_init_nodes / k_init_virtual_nodes
//   K3  k_spatial_edges      4 spatial edges per pixel -> u16 bucket keys + slot ids
//   K4  k_temporal_edges     <=9 (flow displaced) temporal edges per pixel
//   K5  k_bucket_offsets     start of every bucket in a key-sorted list
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
#include "device_graph.h"

namespace vsg {

__constant__ float c_space_w[64];   // 49 spatial weights, tap order of image_filter.cpp:216-225

void UploadSpaceWeights(const float* w49, hipStream_t stream) {
  VSG_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_space_w), w49, 49 * sizeof(float), 0,
                                 hipMemcpyHostToDevice, stream));
}

// ------------------------------------------------------------------------------------------
// K0: min / max over all bytes of the frame.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax_u8(const uint8_t* __restrict__ bgr, size_t stride,
                                                    int row_bytes, int H, int* __restrict__ mm) {
  int lo = 255, hi = 0;
  const int total_threads = gridDim.x * blockDim.x;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)row_bytes * H;
  for (long long i = tid; i < n; i += total_threads) {
    const int y = (int)(i / row_bytes);
    const int x = (int)(i - (long long)y * row_bytes);
    const int v = bgr[(size_t)y * stride + x];
    lo = min(lo, v);
    hi = max(hi, v);
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, __shfl_down(lo, off));
    hi = max(hi, __shfl_down(hi, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&mm[0], lo);
    atomicMax(&mm[1], hi);
  }
}

void LaunchMinMax(const uint8_t* bgr, size_t stride, int W, int H, int* mm, hipStream_t s) {
  const int init[2] = {255, 0};
  VSG_HIP(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_minmax_u8, dim3(1024), dim3(256), 0, s, bgr, stride, W * 3, H, mm);
  VSG_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// K1: bilateral filter.  One workgroup = 256 threads = a 64x16 output tile (4 rows per thread),
// staged with its 4 px replicate halo in LDS next to the 12288-entry exp LUT (48 KiB).
// The accumulation order over the 49 taps and every rounding step follow
// ParallelBilateralColor::operator() exactly.
// ------------------------------------------------------------------------------------------
constexpr int kTileW = 64, kTileH = 16, kRad = 4;
constexpr int kHaloW = kTileW + 2 * kRad, kHaloH = kTileH + 2 * kRad;
constexpr int kLutBins = 12288;

__global__ __launch_bounds__(256) void k_bilateral(const uint8_t* __restrict__ bgr, size_t stride,
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
  for (int i = tid; i < kLutBins; i += 256) lut[i] = lut_g[i];
  const float c255 = (float)(1.0 / 255.0);   // convertTo(CV_32FC3, 1.0/255.0)

  for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const int x0 = (tile % tiles_x) * kTileW;
    const int y0 = (tile / tiles_x) * kTileH;
    __syncthreads();
    for (int idx = tid; idx < kHaloH * kHaloW; idx += 256) {
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
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ty = (tid >> 6) + 4 * r;
      const int y = y0 + ty;
      if (x >= W || y >= H) continue;
      const int c = (ty + kRad) * kHaloW + (tx + kRad);
      const float my_b = tb[c], my_g = tg[c], my_r = tr[c];
      float weight_sum = 0, sum_b = 0, sum_g = 0, sum_r = 0;
      int k = 0;
#pragma unroll
      for (int i = -kRad; i <= kRad; ++i) {
#pragma unroll
        for (int j = -kRad; j <= kRad; ++j) {
          if (i * i + j * j > kRad * kRad) continue;
          const int o = c + i * kHaloW + j;
          const float lb = tb[o], lg = tg[o], lr = tr[o];
          const float diff_b = my_b - lb;
          const float diff_g = my_g - lg;
          const float diff_r = my_r - lr;
          const int idx = (int)((diff_b * diff_b + diff_g * diff_g + diff_r * diff_r) * scale);
          const float weight = c_space_w[k] * lut[idx];
          weight_sum += weight;
          sum_b += lb * weight;
          sum_g += lg * weight;
          sum_r += lr * weight;
          ++k;
        }
      }
      const size_t pix = (size_t)y * W + x;
      if (weight_sum > 0) {
        // weight_sum = 1.0 / weight_sum in double, rounded to float == IEEE 1.0f / weight_sum.
        const float inv = 1.0f / weight_sum;
        out_b[pix] = sum_b * inv;
        out_g[pix] = sum_g * inv;
        out_r[pix] = sum_r * inv;
      } else {
        out_b[pix] = 0.0f;
        out_g[pix] = 0.0f;
        out_r[pix] = 0.0f;
      }
    }
  }
}

size_t BilateralSmemBytes() { return (size_t)(kLutBins + 3 * kHaloH * kHaloW) * sizeof(float); }

void LaunchBilateral(const uint8_t* bgr, size_t stride, int W, int H, const float* lut,
                     float scale, float* planes /* 3*W*H */, hipStream_t s) {
  static bool attr_set = false;
  const size_t smem = BilateralSmemBytes();
  if (!attr_set) {
    VSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bilateral),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  const int tiles_x = (W + kTileW - 1) / kTileW;
  const int tiles_y = (H + kTileH - 1) / kTileH;
  const int num_tiles = tiles_x * tiles_y;
  const int grid = min(num_tiles, 512);   // 2 workgroups per CU (70 KiB LDS each), persistent
  const size_t n = (size_t)W * H;
  hipLaunchKernelGGL(k_bilateral, dim3(grid), dim3(256), smem, s, bgr, stride, W, H, lut, scale,
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
// Edge weights -> bucket keys.
// ------------------------------------------------------------------------------------------
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

// K3: slot = pix * 4 + k, k = 0 right, 1 bottom, 2 bottom-left, 3 bottom-right
// (AddSpatialEdgesImpl order, dense_segmentation_graph.h:971-996).
__global__ __launch_bounds__(256) void k_spatial_edges(const float* __restrict__ feat, int W, int H,
                                                        int l1, ushort4* __restrict__ keys,
                                                        uint4* __restrict__ vals) {
  const size_t n = (size_t)W * H;
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
  const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
  const float* fb = feat;
  const float* fg = feat + n;
  const float* fr = feat + 2 * n;
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
  const unsigned s0 = (unsigned)pix * 4u;
  vals[pix] = make_uint4(s0, s0 + 1, s0 + 2, s0 + 3);
}

void LaunchSpatialEdges(const float* feat, int W, int H, int l1, uint16_t* keys, uint32_t* vals,
                        hipStream_t s) {
  const size_t n = (size_t)W * H;
  hipLaunchKernelGGL(k_spatial_edges, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, feat, W,
                     H, l1, reinterpret_cast<ushort4*>(keys), reinterpret_cast<uint4*>(vals));
  VSG_HIP(hipGetLastError());
}

// x86 cvttss2si semantics for int(float): out of range / NaN -> INT_MIN.
__device__ __forceinline__ int TruncToIntX86(float v) {
  if (!(v < 2147483648.0f && v >= -2147483648.0f)) return (int)0x80000000;
  return (int)v;
}

// K4: slot = pix * 9 + (dy+1)*3 + (dx+1) around the (flow displaced) location in the previous
// slice (GetLocalEdges order TL,T,TR,L,C,R,BL,B,BR; dense_segmentation_graph.h:1011-1065,
// 1126-1135).  is_virtual: weight 1e10 -> bucket 2048 for every existing edge.
__global__ __launch_bounds__(256) void k_temporal_edges(const float* __restrict__ cur,
                                                         const float* __restrict__ prev,
                                                         const float* __restrict__ flow, int W,
                                                         int H, int l1, int is_virtual,
                                                         uint16_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals,
                                                         int32_t* __restrict__ prev_idx) {
  const size_t n = (size_t)W * H;
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
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
  uint16_t* kp = keys + pix * 9;
  uint32_t* vp = vals + pix * 9;
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
      kp[k] = key;
      vp[k] = (uint32_t)pix * 9u + (uint32_t)k;
    }
  }
}

void LaunchTemporalEdges(const float* cur, const float* prev, const float* flow, int W, int H,
                         int l1, int is_virtual, uint16_t* keys, uint32_t* vals,
                         int32_t* prev_idx, hipStream_t s) {
  const size_t n = (size_t)W * H;
  hipLaunchKernelGGL(k_temporal_edges, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cur,
                     prev, flow, W, H, l1, is_virtual, keys, vals, prev_idx);
  VSG_HIP(hipGetLastError());
}

// K5: offsets[b] = first position in the key-sorted list with key >= b, b = 0..kBucketSlots-1.
__global__ __launch_bounds__(256) void k_bucket_offsets(const uint16_t* __restrict__ sorted_keys,
                                                         int n, int* __restrict__ offsets) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= kBucketSlots) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)sorted_keys[mid] < b) lo = mid + 1; else hi = mid;
  }
  offsets[b] = lo;
}

void LaunchBucketOffsets(const uint16_t* sorted_keys, int n, int* offsets, hipStream_t s) {
  hipLaunchKernelGGL(k_bucket_offsets, dim3((kBucketSlots + 255) / 256), dim3(256), 0, s,
                     sorted_keys, n, offsets);
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
