// vs_oracle.cpp -- CPU oracle for the dense over-segmentation hot path.
//
// TEST INFRASTRUCTURE ONLY (see vs_oracle.h).  A from-scratch restatement, in plain C++17, of the
// reference's algorithm.  Every block cites the reference file:line (relative to the reference
// root) whose behaviour it follows.  Data structures are our own; no reference source is copied.
//
// Floating point contract: the reference is built for baseline x86-64
// (SSE2 scalar f32, no FMA) and calls unqualified exp/sqrt/hypot on floats that resolve to the
// double libm functions, the result being rounded to float on assignment.  This file therefore
// spells those conversions explicitly and must be compiled with -ffp-contract=off and without
// -march / -ffast-math (see oracle/Makefile).
//
// PARITY PINNING: see vs_oracle.h header and tests/test_oracle_pins.py.

#include "vs_oracle.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

namespace vso {

#define VSO_CHECK(cond)                                                          \
  do {                                                                           \
    if (!(cond)) {                                                               \
      std::fprintf(stderr, "vs_oracle CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      std::abort();                                                              \
    }                                                                            \
  } while (0)

// ------------------------------------------------------------------------------------------
// Output data model (segment_util/segmentation.proto:55-172).
// ------------------------------------------------------------------------------------------
struct ScanInterval {
  int y = 0, left_x = 0, right_x = 0;
};
typedef std::vector<ScanInterval> Rasterization;
// (frame, rasterization) ordered by frame (segment_util/segmentation_util.h:230).
typedef std::vector<std::pair<int, std::shared_ptr<Rasterization>>> Rasterization3D;

struct ShapeMoments {
  float size = 0, mean_x = 0, mean_y = 0, moment_xx = 0, moment_xy = 0, moment_yy = 0;
};

struct Polygon {   // SegmentationDesc.Polygon
  std::vector<int> coord_idx;
  bool hole = false;
};

struct Region2D {
  int id = 0;
  Rasterization raster;
  ShapeMoments shape_moments;
  std::vector<Polygon> vectorization;   // Vectorization.polygon (empty: field absent)
};

struct CompoundRegion {
  int id = 0;
  int size = 0;
  std::vector<int> neighbor_id;
  bool has_parent = false;
  int parent_id = -1;
  std::vector<int> child_id;
  int start_frame = 0;
  int end_frame = 0;
};

struct HierarchyLevel {
  std::vector<CompoundRegion> region;
};

struct SegmentationDesc {
  std::vector<Region2D> region;
  std::vector<HierarchyLevel> hierarchy;
  int frame_width = 0, frame_height = 0;
  int chunk_size = 0, overlap_start = 0, chunk_id = -1, hierarchy_frame_idx = 0;
  int connectedness = 1;  // N4_CONNECT = 1, N8_CONNECT = 2
  bool has_vector_mesh = false;
  std::vector<float> vector_mesh;   // VectorMesh.coord: x, y pairs
  std::vector<uint32_t> features;   // SegmentationDesc.features: RegionFeatures.id (save_descriptors)
};

// proto2 wire encoding of SegmentationDesc (field numbers from segmentation.proto:55-172;
// serialization order = field-number order, as protobuf's C++ serializer emits it).
namespace wire {
static void Varint(std::string* s, uint64_t v) {
  while (v >= 0x80) {
    s->push_back(static_cast<char>((v & 0x7f) | 0x80));
    v >>= 7;
  }
  s->push_back(static_cast<char>(v));
}
static void Tag(std::string* s, int field, int wt) { Varint(s, (uint64_t)field << 3 | wt); }
static void Int32(std::string* s, int field, int32_t v) {
  Tag(s, field, 0);
  Varint(s, (uint64_t)(int64_t)v);  // negative int32 -> 10 byte varint
}
static void Float(std::string* s, int field, float f) {
  Tag(s, field, 5);
  uint32_t u;
  std::memcpy(&u, &f, 4);
  for (int i = 0; i < 4; ++i) s->push_back(static_cast<char>((u >> (8 * i)) & 0xff));
}
static void Bytes(std::string* s, int field, const std::string& b) {
  Tag(s, field, 2);
  Varint(s, b.size());
  s->append(b);
}
static std::string Encode(const SegmentationDesc& d) {
  std::string out;
  for (const Region2D& r : d.region) {
    std::string rs;
    Int32(&rs, 1, r.id);
    std::string raster;
    for (const ScanInterval& si : r.raster) {
      std::string sis;
      Int32(&sis, 1, si.y);
      Int32(&sis, 2, si.left_x);
      Int32(&sis, 3, si.right_x);
      Bytes(&raster, 1, sis);
    }
    Bytes(&rs, 3, raster);
    std::string sm;
    Float(&sm, 1, r.shape_moments.size);
    Float(&sm, 2, r.shape_moments.mean_x);
    Float(&sm, 3, r.shape_moments.mean_y);
    Float(&sm, 4, r.shape_moments.moment_xx);
    Float(&sm, 5, r.shape_moments.moment_xy);
    Float(&sm, 6, r.shape_moments.moment_yy);
    Bytes(&rs, 5, sm);
    if (!r.vectorization.empty()) {   // Region2D.vectorization = 6
      std::string vec;
      for (const Polygon& pg : r.vectorization) {
        std::string ps, packed;
        for (int idx : pg.coord_idx) Varint(&packed, (uint64_t)(int64_t)idx);
        if (!packed.empty()) Bytes(&ps, 1, packed);   // coord_idx = 1 [packed = true]
        Tag(&ps, 2, 0);                               // hole = 2: set_hole() is always called
        Varint(&ps, pg.hole ? 1 : 0);
        Bytes(&vec, 1, ps);
      }
      Bytes(&rs, 6, vec);
    }
    Bytes(&out, 2, rs);
  }
  for (const HierarchyLevel& h : d.hierarchy) {
    std::string hs;
    for (const CompoundRegion& c : h.region) {
      std::string cs;
      Int32(&cs, 1, c.id);
      Int32(&cs, 2, c.size);
      for (int n : c.neighbor_id) Int32(&cs, 3, n);
      if (c.has_parent) Int32(&cs, 4, c.parent_id);
      for (int n : c.child_id) Int32(&cs, 5, n);
      Int32(&cs, 6, c.start_frame);
      Int32(&cs, 7, c.end_frame);
      Bytes(&hs, 2, cs);
    }
    Bytes(&out, 3, hs);
  }
  Int32(&out, 4, d.frame_width);
  Int32(&out, 5, d.frame_height);
  Int32(&out, 6, d.chunk_size);
  Int32(&out, 7, d.overlap_start);
  Int32(&out, 8, d.chunk_id);
  Int32(&out, 9, d.hierarchy_frame_idx);
  for (uint32_t id : d.features) {   // SegmentationDesc.features = 10: RegionFeatures { required fixed32 id = 1 }
    std::string fs;
    Tag(&fs, 1, 5);
    for (int i = 0; i < 4; ++i) fs.push_back(static_cast<char>((id >> (8 * i)) & 0xff));
    Bytes(&out, 10, fs);
  }
  if (d.has_vector_mesh) {   // SegmentationDesc.vector_mesh = 11
    std::string mesh;
    if (!d.vector_mesh.empty()) {   // coord = 1 [packed = true]
      std::string packed;
      for (float f : d.vector_mesh) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        for (int i = 0; i < 4; ++i) packed.push_back(static_cast<char>((u >> (8 * i)) & 0xff));
      }
      Bytes(&mesh, 1, packed);
    }
    Bytes(&out, 11, mesh);
  }
  Int32(&out, 12, d.connectedness);
  return out;
}
}  // namespace wire

// ------------------------------------------------------------------------------------------
// Rasterization helpers (segment_util/segmentation_util.cpp).
// ------------------------------------------------------------------------------------------

// segmentation_util.cpp:644-650
static int RasterizationArea(const Rasterization& raster) {
  int area = 0;
  for (const ScanInterval& s : raster) area += s.right_x - s.left_x + 1;
  return area;
}

// segmentation_util.cpp:652-693.  All arithmetic in f32 except (n+m)*0.5 (double literal).
static void ShapeMomentsFromRasterization(const Rasterization& raster, ShapeMoments* moments) {
  float mean_x = 0, mean_y = 0, moment_xx = 0, moment_yy = 0, moment_xy = 0, area_sum = 0;
  for (const ScanInterval& s : raster) {
    const float m = (float)s.left_x;
    const float n = (float)s.right_x;
    const float curr_y = (float)s.y;
    const float len = (n - m + 1);
    area_sum += len;
    const float center_x = (float)((double)(n + m) * 0.5);
    const float sum_x = center_x * len;
    const float sum_y = curr_y * len;
    mean_x += sum_x;
    mean_y += sum_y;
    moment_xy += curr_y * sum_x;
    moment_yy += curr_y * sum_y;
    moment_xx += len * (-m + 2 * m * m + n + 2 * m * n + 2 * n * n) / 6.0f;
  }
  const float inv_area = 1.0f / area_sum;
  moments->size = area_sum;
  moments->mean_x = mean_x * inv_area;
  moments->mean_y = mean_y * inv_area;
  moments->moment_xx = moment_xx * inv_area;
  moments->moment_xy = moment_xy * inv_area;
  moments->moment_yy = moment_yy * inv_area;
}

// segmentation_util.cpp:484-570 (two-way merge of scan interval lists, joining abutting runs).
static void MergeRasterization(const Rasterization& lhs, const Rasterization& rhs,
                               Rasterization* merged_out) {
  size_t li = 0, ri = 0;
  const size_t lend = lhs.size(), rend = rhs.size();
  std::vector<int> offs;
  Rasterization merged;
  while (li != lend || ri != rend) {
    const int lhs_y = (li == lend ? 1 << 30 : lhs[li].y);
    const int rhs_y = (ri == rend ? 1 << 30 : rhs[ri].y);
    if (lhs_y < rhs_y) {
      merged.push_back(lhs[li++]);
    } else if (rhs_y < lhs_y) {
      merged.push_back(rhs[ri++]);
    } else {
      offs.clear();
      for (;;) {
        const bool left_cond = (li != lend && lhs[li].y == lhs_y);
        const bool right_cond = (ri != rend && rhs[ri].y == rhs_y);
        if (!(left_cond | right_cond)) break;
        const int lhs_x = left_cond ? lhs[li].left_x : std::numeric_limits<int>::max();
        const int rhs_x = right_cond ? rhs[ri].left_x : std::numeric_limits<int>::max();
        if (lhs_x < rhs_x) {
          offs.push_back(lhs[li].left_x);
          offs.push_back(lhs[li].right_x);
          ++li;
        } else {
          offs.push_back(rhs[ri].left_x);
          offs.push_back(rhs[ri].right_x);
          ++ri;
        }
      }
      int k = 0, l = 0;
      const int sz_k = (int)offs.size();
      while (k < sz_k) {
        if (k + 2 == sz_k) {
          merged.push_back(ScanInterval{lhs_y, offs[l], offs[k + 1]});
          break;
        } else if (offs[k + 2] - 1 == offs[k + 1]) {
          k += 2;
        } else {
          merged.push_back(ScanInterval{lhs_y, offs[l], offs[k + 1]});
          k += 2;
          l = k;
        }
      }
    }
  }
  merged_out->swap(merged);
}

// segmentation_util.cpp:1009-1101.  Partition of the intervals into N4/N8 connected components;
// components are emitted in order of their first interval (the disjoint-set internals of the
// reference (boost::disjoint_sets) do not influence the partition).
static bool ScanIntervalsNeighbored(const ScanInterval& a, const ScanInterval& b, bool n4) {
  if (std::abs(a.y - b.y) > 1) return false;
  if (n4) return std::max(a.left_x, b.left_x) <= std::min(a.right_x, b.right_x);
  return std::max(a.left_x, b.left_x) - std::min(a.right_x, b.right_x) <= 1;
}

static int ConnectedComponents(const Rasterization& raster, bool n4,
                               std::vector<Rasterization>* components) {
  const int n = (int)raster.size();
  std::vector<int> parent(n);
  auto find = [&parent](int i) {
    while (parent[i] != i) {
      parent[i] = parent[parent[i]];
      i = parent[i];
    }
    return i;
  };
  int last_change_idx = -1, last_y = -2, test_idx = 0;
  for (int i = 0; i < n; ++i) {
    parent[i] = i;
    const ScanInterval& cur = raster[i];
    if (cur.y != last_y) {
      test_idx = (last_y + 1 == cur.y) ? last_change_idx : i;
      last_y = cur.y;
      last_change_idx = i;
    }
    for (int k = test_idx; k < i; ++k) {
      if (ScanIntervalsNeighbored(cur, raster[k], n4)) {
        const int a = find(i), b = find(k);
        if (a != b) parent[a] = b;
      }
    }
  }
  int num_components = 0;
  for (int i = 0; i < n; ++i) num_components += (find(i) == i);
  if (num_components == 1) {
    if (components) components->push_back(raster);
    return 1;
  }
  if (components) {
    std::unordered_map<int, int> rep_to_comp;
    for (int i = 0; i < n; ++i) {
      const int rep = find(i);
      auto it = rep_to_comp.find(rep);
      if (it == rep_to_comp.end()) {
        rep_to_comp[rep] = (int)components->size();
        components->push_back(Rasterization{raster[i]});
      } else {
        (*components)[it->second].push_back(raster[i]);
      }
    }
  }
  return num_components;
}

// ------------------------------------------------------------------------------------------
// Shape descriptors (segment_util/segmentation_util.h:138-151, .cpp:243-410).
// ------------------------------------------------------------------------------------------
struct Pt {
  float x = 0, y = 0;
};
static inline Pt operator+(Pt a, Pt b) { return Pt{a.x + b.x, a.y + b.y}; }
static inline Pt operator-(Pt a, Pt b) { return Pt{a.x - b.x, a.y - b.y}; }
static inline Pt operator*(Pt a, float s) { return Pt{a.x * s, a.y * s}; }
static inline float HypotF(float a, float b) { return (float)std::hypot((double)a, (double)b); }

struct ShapeDescriptor {
  Pt center;
  float mag_major = 0, mag_minor = 0;
  Pt dir_major{1.0f, 0.0f};
  Pt dir_minor{0.0f, 1.0f};
  int size = 0;
};

// segmentation_util.cpp:243-340, single-moment form (:342-345).
static bool GetShapeDescriptorFromShapeMoment(const ShapeMoments& moment, ShapeDescriptor* sd) {
  float mixed_x = 0, mixed_y = 0, mixed_xx = 0, mixed_xy = 0, mixed_yy = 0, area_sum = 0;
  {
    const float area = moment.size;
    area_sum += area;
    mixed_x += moment.mean_x * area;
    mixed_y += moment.mean_y * area;
    mixed_xx += moment.moment_xx * area;
    mixed_xy += moment.moment_xy * area;
    mixed_yy += moment.moment_yy * area;
  }
  VSO_CHECK(area_sum > 0);
  const float inv_area_sum = 1.0f / area_sum;
  mixed_x *= inv_area_sum;
  mixed_y *= inv_area_sum;
  mixed_xx *= inv_area_sum;
  mixed_xy *= inv_area_sum;
  mixed_yy *= inv_area_sum;
  sd->center = Pt{mixed_x, mixed_y};
  sd->size = (int)area_sum;
  if (area_sum < 10) return false;

  const float var_xx = mixed_xx - mixed_x * mixed_x;
  const float var_xy = mixed_xy - mixed_x * mixed_y;
  const float var_yy = mixed_yy - mixed_y * mixed_y;
  const float trace = var_xx + var_yy;
  const float det = var_xx * var_yy - var_xy * var_xy;
  // 0.25 is a double literal: 0.25 * trace * trace - det is evaluated in double.
  float discriminant = (float)(0.25 * (double)trace * (double)trace - (double)det);
  discriminant = std::max(0.0f, discriminant);
  const float sqrt_disc = (float)std::sqrt((double)discriminant);
  const float e_1 = (float)((double)trace * 0.5 - (double)sqrt_disc);
  const float e_2 = (float)((double)trace * 0.5 + (double)sqrt_disc);
  // std::min(fabs(e_1), fabs(e_2)) < 1 with double fabs.
  if (std::min(std::fabs((double)e_1), std::fabs((double)e_2)) < 1) return false;

  Pt ev_1{1.0f, 0.0f};
  Pt ev_2{0.0f, 1.0f};
  const Pt v_1{e_1 - var_yy, var_xy};
  const Pt v_2{e_2 - var_yy, var_xy};
  const float v_1_norm = HypotF(v_1.y, v_1.x);
  const float v_2_norm = HypotF(v_2.y, v_2.x);
  if (v_1_norm > 1e-6f && v_2_norm > 1e-6f && (double)discriminant > 0.1) {
    ev_1 = v_1 * (1.0f / v_1_norm);
    ev_2 = v_2 * (1.0f / v_2_norm);
  }
  float e_1_sigma = (float)std::sqrt(std::fabs((double)e_1));
  float e_2_sigma = (float)std::sqrt(std::fabs((double)e_2));
  if (e_1_sigma < e_2_sigma) {
    std::swap(e_1_sigma, e_2_sigma);
    std::swap(ev_1, ev_2);
  }
  const Pt ev_1_normal{-ev_1.y, ev_1.x};
  if (ev_2.x * ev_1_normal.x + ev_2.y * ev_1_normal.y < 0) {
    ev_2 = Pt{-ev_2.x, -ev_2.y};
  }
  sd->center = Pt{mixed_x, mixed_y};
  sd->mag_major = e_1_sigma;
  sd->mag_minor = e_2_sigma;
  sd->dir_major = ev_1;
  sd->dir_minor = ev_2;
  return true;
}

// segmentation_util.cpp:366-380
static void ShapeDescriptorBox(const ShapeDescriptor& shape, float border, Pt* coords /*4*/) {
  const Pt major = shape.dir_major * (shape.mag_major * 1.65f + border);
  const Pt minor = shape.dir_minor * (shape.mag_minor * 1.65f + border);
  const Pt center = shape.center;
  coords[0] = center - major + minor;
  coords[1] = center - major - minor;
  coords[2] = center + major - minor;
  coords[3] = center + major + minor;
}

// segmentation_util.cpp:382-410 (segment/segment intersection in double, inv_kross in float).
static bool ShapeDescriptorBoxesIntersect(const Pt* lhs, const Pt* rhs) {
  for (int k = 0; k < 4; ++k) {
    const Pt ld = lhs[(k + 1) % 4] - lhs[k];
    const double ldx = ld.x, ldy = ld.y;
    for (int l = 0; l < 4; ++l) {
      const Pt rd = rhs[(l + 1) % 4] - rhs[l];
      const double rdx = rd.x, rdy = rd.y;
      const Pt dl = rhs[l] - lhs[k];
      const double dx = dl.x, dy = dl.y;
      const double kross = ldx * rdy - ldy * rdx;
      if (std::fabs(kross) < 1e-6) continue;
      const float inv_kross = (float)(1.0f / kross);
      const double t = (dx * rdy - dy * rdx) * inv_kross;
      const double s = (dx * ldy - dy * ldx) * inv_kross;
      if (t > -1e-6f && t < 1.0f + 1e-6f && s > -1e-6f && s < 1.0f + 1e-6f) return true;
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------------
// Bilateral pre-filter (imagefilter/image_filter.cpp:106-180, 184-277) and feature conversion
// (segmentation/dense_segmentation.cpp:164-198).
// ------------------------------------------------------------------------------------------
static const int kBilateralRadius = 4;     // int(3.0f * 1.5f), image_filter.cpp:197
static const int kBilateralBins = 12288;   // (1 << 12) * 3, image_filter.cpp:237

// image_filter.cpp:208-250.  Returns scale; fills lut[12288] and space_w[<=81] (49 used).
static float BilateralTables(double min_val, double max_val, float sigma_space, float sigma_color,
                             int cn, float* lut, float* space_w, int* space_dy, int* space_dx,
                             int* space_sz_out) {
  const int radius = (int)(sigma_space * 1.5f);
  int space_sz = 0;
  const float space_coeff = -0.5f / (sigma_space * sigma_space);
  for (int i = -radius; i <= radius; ++i) {
    for (int j = -radius; j <= radius; ++j) {
      const int r2 = i * i + j * j;
      if (r2 > radius * radius) continue;
      if (space_dy) {
        space_dy[space_sz] = i;
        space_dx[space_sz] = j;
      }
      space_w[space_sz++] = (float)std::exp((double)(space_coeff * (float)r2));
    }
  }
  if (space_sz_out) *space_sz_out = space_sz;
  const float diff_range = std::max<float>(
      1e-3f, (float)((max_val - min_val) * (max_val - min_val) * cn * (double)1.02f));
  const int num_bins = (1 << 12) * cn;
  const float scale = (float)num_bins / diff_range;
  const float color_coeff = (float)(-0.5 / (double)(sigma_color * sigma_color));
  bool zero_reached = false;
  for (int i = 0; i < num_bins; ++i) {
    if (!zero_reached) {
      lut[i] = (float)std::exp((double)((float)i / scale * color_coeff));
      zero_reached = ((double)lut[i] < 1e-10);
    } else {
      lut[i] = 0;
    }
  }
  return scale;
}

// image: H*W*3 f32 interleaved.  output likewise.
// Threading of the reference's defaults (timing baseline only; results are identical):
// parallel_graph_construction (dense_segmentation_graph.cpp:31: one std::thread per Add*Edges
// call, joined in FinishBuildingGraph) and the row-parallel bilateral filter (ParallelFor over
// rows, base/base.h:155-158).  g_threads <= 1: everything on the calling thread.
static int g_threads = 1;

template <class F>
static void ParallelRows(int rows, const F& fn) {
  const int nt = std::max(1, std::min(g_threads, rows));
  if (nt == 1) {
    fn(0, rows);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) {
    const int r0 = (int)((int64_t)rows * t / nt), r1 = (int)((int64_t)rows * (t + 1) / nt);
    pool.emplace_back([=, &fn] { fn(r0, r1); });
  }
  for (std::thread& th : pool) th.join();
}

static void BilateralFilter3(const float* image, int W, int H, float sigma_space,
                             float sigma_color, float* output) {
  const int radius = (int)(sigma_space * 1.5f);
  const int BW = W + 2 * radius, BH = H + 2 * radius;
  // cv::copyMakeBorder(BORDER_REPLICATE), image_filter.cpp:203-207.
  std::vector<float> border((size_t)BW * BH * 3);
  for (int y = 0; y < BH; ++y) {
    const int sy = std::min(std::max(y - radius, 0), H - 1);
    for (int x = 0; x < BW; ++x) {
      const int sx = std::min(std::max(x - radius, 0), W - 1);
      const float* s = image + ((size_t)sy * W + sx) * 3;
      float* d = border.data() + ((size_t)y * BW + x) * 3;
      d[0] = s[0];
      d[1] = s[1];
      d[2] = s[2];
    }
  }
  // cv::minMaxLoc over all channels (image_filter.cpp:228-230).
  double min_val = image[0], max_val = image[0];
  for (size_t i = 0, n = (size_t)W * H * 3; i < n; ++i) {
    min_val = std::min(min_val, (double)image[i]);
    max_val = std::max(max_val, (double)image[i]);
  }
  std::vector<float> lut(kBilateralBins);
  float space_w[81];
  int sdy[81], sdx[81], space_sz = 0;
  const float scale = BilateralTables(min_val, max_val, sigma_space, sigma_color, 3, lut.data(),
                                      space_w, sdy, sdx, &space_sz);
  std::vector<ptrdiff_t> space_ofs(space_sz);
  for (int k = 0; k < space_sz; ++k) space_ofs[k] = ((ptrdiff_t)sdy[k] * BW + sdx[k]) * 3;

  // ParallelBilateralColor::operator(), image_filter.cpp:130-167.
  ParallelRows(H, [&](int row_begin, int row_end) {
  for (int i = row_begin; i < row_end; ++i) {
    const float* src_ptr = border.data() + ((size_t)(i + radius) * BW + radius) * 3;
    float* dst_ptr = output + (size_t)i * W * 3;
    for (int j = 0; j < W; ++j, src_ptr += 3, dst_ptr += 3) {
      const float my_b = src_ptr[0], my_g = src_ptr[1], my_r = src_ptr[2];
      float weight_sum = 0, sum_r = 0, sum_g = 0, sum_b = 0;
      for (int k = 0; k < space_sz; ++k) {
        const float* local_ptr = src_ptr + space_ofs[k];
        const float diff_b = my_b - local_ptr[0];
        const float diff_g = my_g - local_ptr[1];
        const float diff_r = my_r - local_ptr[2];
        const int idx = (int)((diff_b * diff_b + diff_g * diff_g + diff_r * diff_r) * scale);
        const float weight = space_w[k] * lut[idx];
        weight_sum += weight;
        sum_b += local_ptr[0] * weight;
        sum_g += local_ptr[1] * weight;
        sum_r += local_ptr[2] * weight;
      }
      if (weight_sum > 0) {
        weight_sum = (float)(1.0 / (double)weight_sum);
        dst_ptr[0] = sum_b * weight_sum;
        dst_ptr[1] = sum_g * weight_sum;
        dst_ptr[2] = sum_r * weight_sum;
      } else {
        dst_ptr[0] = dst_ptr[1] = dst_ptr[2] = 0.0f;
      }
    }
  }
  });
}

// cv::GaussianBlur(src, dst, Size(3, 3), sigma) for CV_32FC3 -- OpenCV, un-vendored third-party
// arithmetic: the published algorithm of the 2.4 line (imgproc/src/smooth.cpp getGaussianKernel,
// filter.cpp SymmRowSmallFilter / SymmColumnSmallFilter for a symmetric 3-tap float kernel), restated;
// PARITY UNPINNED (no OpenCV in this image).
//   kernel: t_i = exp(-0.5 / sigma^2 * (i - 1)^2) in double, stored as float; normalised by the double
//           sum of the stored floats: cf_i = float(cf_i * (1 / sum))
//   rows:   r[x] = S[x] * k0 + (S[x - 1] + S[x + 1]) * k1          (k0 = cf[1], k1 = cf[0] = cf[2])
//   cols:   d[y] = r[y] * k0 + (r[y - 1] + r[y + 1]) * k1          every operation rounded to f32
//   border: BORDER_REFLECT_101 (-1 -> 1, n -> n - 2; a single row / column reflects onto itself)
static void GaussianBlur3x3(const float* src, int W, int H, double sigma, float* dst) {
  float cf[3];
  double sum = 0;
  const double scale2x = -0.5 / (sigma * sigma);
  for (int i = 0; i < 3; ++i) {
    const double x = i - 1.0;
    cf[i] = (float)std::exp(scale2x * x * x);
    sum += cf[i];
  }
  sum = 1.0 / sum;
  for (int i = 0; i < 3; ++i) cf[i] = (float)(cf[i] * sum);
  const float k0 = cf[1], k1 = cf[0];
  std::vector<float> rows((size_t)W * H * 3);
  for (int y = 0; y < H; ++y) {
    const float* s = src + (size_t)y * W * 3;
    float* r = rows.data() + (size_t)y * W * 3;
    for (int x = 0; x < W; ++x) {
      const int xl = x > 0 ? x - 1 : (W > 1 ? 1 : 0), xr = x + 1 < W ? x + 1 : (W > 1 ? W - 2 : 0);
      for (int c = 0; c < 3; ++c) {
        const float side = s[xl * 3 + c] + s[xr * 3 + c];
        const float centre = s[x * 3 + c] * k0;
        r[x * 3 + c] = centre + side * k1;
      }
    }
  }
  for (int y = 0; y < H; ++y) {
    const int yu = y > 0 ? y - 1 : (H > 1 ? 1 : 0), yd = y + 1 < H ? y + 1 : (H > 1 ? H - 2 : 0);
    const float* r0 = rows.data() + (size_t)yu * W * 3;
    const float* r1 = rows.data() + (size_t)y * W * 3;
    const float* r2 = rows.data() + (size_t)yd * W * 3;
    float* d = dst + (size_t)y * W * 3;
    for (int i = 0; i < W * 3; ++i) {
      const float side = r0[i] + r2[i];
      const float centre = r1[i] * k0;
      d[i] = centre + side * k1;
    }
  }
}

// dense_segmentation.cpp:164-198.  convertTo(CV_32FC3, 1.0/255.0) is OpenCV (un-vendored);
// assumed float(u8) * float(1.0/255.0)  -- parity unpinned for this one step.
static void PreprocessFeatures(const uint8_t* bgr, size_t stride, int W, int H, int presmoothing,
                               float* out) {
  std::vector<float> tmp((size_t)W * H * 3);
  const float scale = (float)(1.0 / 255.0);
  for (int y = 0; y < H; ++y) {
    const uint8_t* row = bgr + (size_t)y * stride;
    float* d = tmp.data() + (size_t)y * W * 3;
    for (int x = 0; x < W * 3; ++x) d[x] = (float)row[x] * scale;
  }
  if (presmoothing == 2) {
    BilateralFilter3(tmp.data(), W, H, 3.0f, 0.25f, out);
  } else if (presmoothing == 1) {
    GaussianBlur3x3(tmp.data(), W, H, 1.5, out);   // dense_segmentation.cpp:186-188
  } else {
    VSO_CHECK(presmoothing == 0);
    std::memcpy(out, tmp.data(), tmp.size() * sizeof(float));
  }
}

// ------------------------------------------------------------------------------------------
// Pixel distances (segmentation/pixel_distance.h:141-157).
// ------------------------------------------------------------------------------------------
static inline float ColorDiff3L1(const float* p1, const float* p2) {
  const float d1 = p1[0] - p2[0], d2 = p1[1] - p2[1], d3 = p1[2] - p2[2];
  return (float)((std::fabs((double)d1) + std::fabs((double)d2) + std::fabs((double)d3)) *
                 (double)(1.0f / 3.0f));
}
static inline float ColorDiff3L2(const float* p1, const float* p2) {
  const float d1 = p1[0] - p2[0], d2 = p1[1] - p2[1], d3 = p1[2] - p2[2];
  return (float)std::sqrt((double)((d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 3.0f)));
}

// ------------------------------------------------------------------------------------------
// RegionInformation (segmentation/segmentation_common.h:39-116), over-seg subset.
// ------------------------------------------------------------------------------------------
struct RegionInformation {
  int index = -1;
  int size = 0;
  bool flagged_for_removal = false;
  std::vector<int> neighbor_idx;
  std::unique_ptr<Rasterization3D> raster;
  int constrained_id = -1;
  int region_id = -1;
};
typedef std::vector<std::unique_ptr<RegionInformation>> RegionInfoList;
typedef std::unordered_map<int, RegionInformation*> RegionInfoPtrMap;

// segmentation_common.h:144-152
static bool InsertSortedUniquely(int t, std::vector<int>* array) {
  auto pos = std::lower_bound(array->begin(), array->end(), t);
  if (pos == array->end() || *pos != t) {
    array->insert(pos, t);
    return true;
  }
  return false;
}

// ------------------------------------------------------------------------------------------
// Tube helpers (segmentation/dense_segmentation_graph.h:581-655, .cpp:35-209).
// ------------------------------------------------------------------------------------------
struct TubeSlice {
  int frame = -1;
  Rasterization raster;
  ShapeDescriptor shape;

  void ComputeShapeDescriptor() {  // dense_segmentation_graph.h:637-641
    ShapeMoments moment;
    ShapeMomentsFromRasterization(raster, &moment);
    GetShapeDescriptorFromShapeMoment(moment, &shape);
  }
  void MergeFrom(const TubeSlice& other) {  // :631-635
    VSO_CHECK(frame == other.frame);
    MergeRasterization(raster, other.raster, &raster);
    ComputeShapeDescriptor();
  }
};
typedef std::vector<TubeSlice> Tube3D;

// dense_segmentation_graph.h:601-629.  flow: W*H*2 f32 or null.
static std::pair<int, float> FindPreviousTube(const TubeSlice& slice,
                                              const std::vector<Tube3D>& tubes, int frame,
                                              const float* flow, int W) {
  Pt prev_center = slice.shape.center;
  if (flow) {
    const float* flow_ptr = flow + ((size_t)(int)prev_center.y * W) * 2 + 2 * (int)prev_center.x;
    prev_center = prev_center + Pt{flow_ptr[0], flow_ptr[1]};
  }
  float closest_dist = std::numeric_limits<float>::max();
  float closest_idx = -1;  // float in the reference
  for (int k = 0; k < (int)tubes.size(); ++k) {
    if (tubes[k].empty() || tubes[k].back().frame >= frame) continue;
    const Pt diff = tubes[k].back().shape.center - prev_center;
    const float dist = HypotF(diff.y, diff.x);
    if (dist < closest_dist) {
      closest_dist = dist;
      closest_idx = (float)k;
    }
  }
  return std::make_pair((int)closest_idx, closest_dist);
}

// dense_segmentation_graph.cpp:35-45
static float AverageTubeSliceSize(const Tube3D& ts) {
  if (ts.empty()) return 0;
  float area_sum = 0;
  for (const TubeSlice& s : ts) area_sum += (float)s.shape.size;
  return area_sum / (float)ts.size();
}

// dense_segmentation_graph.cpp:47-88
static void MergeTube3D(const Tube3D& lhs, const Tube3D& rhs, Tube3D* result) {
  size_t li = 0, ri = 0;
  if (lhs.empty()) {
    *result = rhs;
    return;
  }
  if (rhs.empty()) {
    *result = lhs;
    return;
  }
  while (li < lhs.size() && ri < rhs.size()) {
    if (lhs[li].frame < rhs[ri].frame) {
      result->push_back(lhs[li++]);
    } else if (lhs[li].frame > rhs[ri].frame) {
      result->push_back(rhs[ri++]);
    } else {
      TubeSlice merged = lhs[li];
      merged.MergeFrom(rhs[ri]);
      result->push_back(merged);
      ++li;
      ++ri;
    }
  }
  while (li < lhs.size()) result->push_back(lhs[li++]);
  while (ri < rhs.size()) result->push_back(rhs[ri++]);
}

// dense_segmentation_graph.cpp:90-110
static bool AreTubesTemporalNeighbors(const Tube3D& lhs, const Tube3D& rhs) {
  if (lhs.empty() || rhs.empty()) return false;
  ShapeDescriptor a, b;
  if (lhs[0].frame - 1 == rhs.back().frame) {
    a = lhs[0].shape;
    b = rhs.back().shape;
  } else if (lhs.back().frame + 1 == rhs[0].frame) {
    a = lhs.back().shape;
    b = rhs[0].shape;
  } else {
    return false;
  }
  const float size_ratio = (float)std::min(a.size, b.size) * (1.0f / (float)std::max(a.size, b.size));
  const Pt diff = a.center - b.center;
  return (double)size_ratio > 0.9 && std::hypot((double)diff.y, (double)diff.x) < 20;
}

// dense_segmentation_graph.cpp:112-148
static float AverageTubeDistance(const Tube3D& lhs, const Tube3D& rhs) {
  if (lhs.empty() || rhs.empty()) return std::numeric_limits<float>::max();
  const int start_frame = std::max(lhs[0].frame, rhs[0].frame);
  const int end_frame = std::min(lhs.back().frame, rhs.back().frame);
  int li = 0, ri = 0;
  float diff_sum = 0;
  int weight = 0;
  for (int f = start_frame; f <= end_frame; ++f) {
    while (lhs[li].frame < f) ++li;
    while (rhs[ri].frame < f) ++ri;
    if (lhs[li].frame != f || rhs[ri].frame != f) continue;
    const Pt cd = lhs[li].shape.center - rhs[ri].shape.center;
    diff_sum = (float)((double)diff_sum + std::hypot((double)cd.y, (double)cd.x));
    ++weight;
  }
  if (weight > 0) return diff_sum / (float)weight;
  return std::numeric_limits<float>::max();
}

// dense_segmentation_graph.cpp:150-191
static float Tube3DIntersection(const Tube3D& lhs, const Tube3D& rhs) {
  if (lhs.empty() || rhs.empty()) return std::numeric_limits<float>::max();
  const int start_frame = std::max(lhs[0].frame, rhs[0].frame);
  const int end_frame = std::min(lhs.back().frame, rhs.back().frame);
  int li = 0, ri = 0, intersect_count = 0, weight = 0;
  for (int f = start_frame; f <= end_frame; ++f) {
    while (lhs[li].frame < f) ++li;
    while (rhs[ri].frame < f) ++ri;
    if (lhs[li].frame != f || rhs[ri].frame != f) continue;
    Pt lb[4], rb[4];
    ShapeDescriptorBox(lhs[li].shape, 10, lb);
    ShapeDescriptorBox(rhs[ri].shape, 10, rb);
    if (ShapeDescriptorBoxesIntersect(lb, rb)) ++intersect_count;
    ++weight;
  }
  if (weight > 0) return (float)intersect_count * (1.0f / (float)weight);
  return std::numeric_limits<float>::max();
}

// dense_segmentation_graph.cpp:193-210
static int GetClosestTube3D(const Tube3D& tube, const std::vector<Tube3D>& tubes, int ignore) {
  float min_dist = std::numeric_limits<float>::max();
  int min_idx = -1;
  for (int k = 0; k < (int)tubes.size(); ++k) {
    if (k == ignore) continue;
    const float d = AverageTubeDistance(tube, tubes[k]);
    if (d < min_dist) {
      min_dist = d;
      min_idx = k;
    }
  }
  return min_idx;
}

// ------------------------------------------------------------------------------------------
// The dense segmentation graph = FastSegmentationGraph<ColorMeanDescriptorTraits>
// (segmentation/segmentation_graph.h) + DenseSegmentationGraph (dense_segmentation_graph.h).
// ------------------------------------------------------------------------------------------
struct BucketCensus {
  int64_t edges = 0, internal = 0, regular = 0, fail = 0, small = 0, kept = 0, forced = 0;
};

class DenseGraph {
 public:
  ~DenseGraph() {
    for (std::thread& t : add_edges_tasks_) {
      if (t.joinable()) t.join();
    }
  }
  // dense_segmentation_graph.h:290-312; segmentation_graph.h:322-337.
  DenseGraph(int W, int H, int max_frames, bool l1)
      : W_(W), H_(H), max_frames_(max_frames), l1_(l1) {
    num_buckets_ = 2048;
    bucket_lists_.resize(2 * max_frames - 1);
    for (auto& bl : bucket_lists_) bl.resize(num_buckets_ + 1);
    scale_ = (float)num_buckets_ / (1.0f + 1e-6f);
    force_merge_weight_ = l1 ? 0.002f : 0.001f;  // dense_segmentation.cpp:259-264
    region_ids_.assign((size_t)(H + 2) * (W + 2), 0);
    regions_.reserve((size_t)((float)((size_t)W * H * max_frames) * 1.02f));
    census_.resize(num_buckets_);
  }

  int num_frames() const { return num_frames_; }
  int W() const { return W_; }
  int H() const { return H_; }

  // AddNodesAndSpatialEdges[Constrained] (dense_segmentation_graph.h:83-105, 906-930).
  void AddFrame(const float* feat, const int32_t* constraint_ids) {
    AddNodesWithDescriptors(feat, constraint_ids);
    const int frame_idx = num_frames_;
    if (g_threads > 1) {   // every task fills its own bucket list
      add_edges_tasks_.emplace_back([this, feat, frame_idx] { AddSpatialEdgesImpl(feat, frame_idx); });
    } else {
      AddSpatialEdgesImpl(feat, frame_idx);
    }
    ++num_frames_;
    VSO_CHECK(num_frames_ <= max_frames_);
  }

  // AddVirtualNodesConstrained (dense_segmentation_graph.h:327-367).
  void AddVirtualFrame(const int32_t* ids) {
    const int base_idx = num_frames_ * H_ * W_;
    VSO_CHECK((int)regions_.size() == base_idx);
    virtual_slices_.push_back(num_frames_);
    std::unordered_map<int, int> constraint_to_rep;
    int region_idx = base_idx;
    for (int i = 0; i < H_; ++i) {
      for (int j = 0; j < W_; ++j, ++region_idx) {
        const int cid = ids[(size_t)i * W_ + j];
        Region r;
        r.my_id = region_idx;
        r.sz = 0;
        r.constraint_id = cid;
        r.virtual_no_desc = true;
        regions_.push_back(r);
        auto pos = constraint_to_rep.find(cid);
        if (pos == constraint_to_rep.end()) {
          constraint_to_rep.insert(std::make_pair(cid, region_idx));
        } else {
          regions_[region_idx].my_id = regions_[pos->second].my_id;
        }
      }
    }
    virtual_nodes_.push_back(std::make_pair(base_idx, base_idx + H_ * W_));
    ++num_frames_;
    VSO_CHECK(num_frames_ <= max_frames_);
  }

  // AddTemporalEdges / AddTemporalFlowEdges / virtual variants
  // (dense_segmentation_graph.h:369-395, 932-954, 1068-1142).  frame_idx = num_frames_ (already
  // incremented for the current slice).
  void AddTemporal(const float* cur, const float* prev, const float* flow, bool is_virtual) {
    const int frame_idx = num_frames_;
    if (g_threads > 1) {
      add_edges_tasks_.emplace_back(
          [=] { AddTemporalImpl(cur, prev, flow, is_virtual, frame_idx); });
    } else {
      AddTemporalImpl(cur, prev, flow, is_virtual, frame_idx);
    }
  }

  // FinishBuildingGraph, dense_segmentation_graph.h:396-403.
  void FinishBuildingGraph() {
    for (std::thread& t : add_edges_tasks_) t.join();
    add_edges_tasks_.clear();
  }

  void AddTemporalImpl(const float* cur, const float* prev, const float* flow, bool is_virtual,
                       int frame_idx) {
    const int base_diff = W_ * H_;
    const int base_idx = (frame_idx - 1) * W_ * H_;
    const int bucket_list_idx = 2 * (frame_idx - 1) - 1;
    VSO_CHECK(bucket_list_idx >= 0);
    int curr_idx = base_idx;
    for (int i = 0; i < H_; ++i) {
      for (int j = 0; j < W_; ++j, ++curr_idx) {
        int prev_x = j, prev_y = i;
        if (flow) {
          const float* flow_ptr = flow + ((size_t)i * W_ + j) * 2;
          prev_x = (int)((float)j + flow_ptr[0]);
          prev_y = (int)((float)i + flow_ptr[1]);
          prev_x = std::max(0, std::min(W_ - 1, prev_x));
          prev_y = std::max(0, std::min(H_ - 1, prev_y));
        }
        const int prev_idx = base_idx - base_diff + prev_y * W_ + prev_x;
        // GetLocalEdges, dense_segmentation_graph.h:1002-1066.
        const float* a = is_virtual ? nullptr : cur + ((size_t)i * W_ + j) * 3;
        for (int dy = -1; dy <= 1; ++dy) {
          if (dy < 0 && !(prev_y > 0)) continue;
          if (dy > 0 && !(prev_y + 1 < H_)) continue;
          for (int dx = -1; dx <= 1; ++dx) {
            if (dx < 0 && !(prev_x > 0)) continue;
            if (dx > 0 && !(prev_x + 1 < W_)) continue;
            float w;
            if (is_virtual) {
              w = 1e10f;  // ConstantPixelDistance(1e10), :372
            } else {
              const float* b = prev + ((size_t)(prev_y + dy) * W_ + (prev_x + dx)) * 3;
              w = l1_ ? ColorDiff3L1(a, b) : ColorDiff3L2(a, b);
            }
            AddEdge(curr_idx, prev_idx + dy * W_ + dx, w, bucket_list_idx);
          }
        }
      }
    }
  }

  // DenseSegmentationGraph::SegmentGraphSpatially, dense_segmentation_graph.h:406-416: the spatial
  // bucket lists (2 t) only, min_region_size 0, no constraint merge.  Edges that are not kept
  // leave their bucket list (segmentation_graph.h:442), so the full pass that follows
  // (two_stage_segmentation, segmentation.cpp:280-283) only sees what this pass kept.
  void SegmentGraphSpatially() {
    std::vector<int> spatial_lists;
    for (int i = 0; i < num_frames_; ++i) spatial_lists.push_back(2 * i);
    SegmentGraph(0, false, &spatial_lists, true);
  }

  // FastSegmentationGraph::SegmentGraph, segmentation_graph.h:339-463.  bucket_list_ids: the
  // bucket lists to walk (null: all).  The merge statistics add up over the calls on one graph
  // unless reset_stats is set.
  void SegmentGraph(int min_region_size, bool force_constraints,
                    const std::vector<int>* bucket_list_ids = nullptr, bool reset_stats = false) {
    FinishBuildingGraph();
    const float inv_scale = (float)(1.0 / (double)scale_);
    if (reset_stats || !segmented_once_) {
      num_forced_merges_ = num_regular_merges_ = num_small_region_merges_ = 0;
    }
    segmented_once_ = true;
    std::vector<int> all_lists;
    if (!bucket_list_ids) {
      for (size_t bl = 0; bl < bucket_lists_.size(); ++bl) all_lists.push_back((int)bl);
      bucket_list_ids = &all_lists;
    }
    const float merge_thr = 0.05f;   // pixel_distance.h:471
    const float split_thr = 0.15f;   // pixel_distance.h:472
    // Analysis aid (VSO_DEPTH_STATS=1, stderr): per bucket, the longest chain of edges that have
    // to be evaluated one after the other -- an edge waits for the last edge that CHANGED one of
    // its two regions (merge, finalisation, dropped constraint); kept edges change nothing.  That
    // is the floor of any schedule that keeps the reference's order exactly.
    const bool depth_stats = getenv("VSO_DEPTH_STATS") != nullptr;
    std::vector<int> dep;
    if (depth_stats) dep.assign(regions_.size(), 0);
    for (int bucket_idx = 0; bucket_idx < num_buckets_; ++bucket_idx) {
      const float weight = (float)bucket_idx * inv_scale;
      BucketCensus& cs = census_[bucket_idx];
      cs = BucketCensus();
      long long d_edges = 0, d_changing = 0;
      int d_max = 0;
      if (depth_stats) std::fill(dep.begin(), dep.end(), 0);
      for (int bl : *bucket_list_ids) {
        if (bl >= (int)bucket_lists_.size()) continue;
        EdgeList remaining;
        EdgeList& edges = bucket_lists_[bl][bucket_idx];
        cs.edges += (int64_t)edges.size();
        for (const Edge& e : edges) {
          Region* rep_1 = GetRegion(e.region_1);
          Region* rep_2 = GetRegion(e.region_2);
          if (rep_1 == rep_2) {
            ++cs.internal;
            continue;
          }
          int d_id1 = 0, d_id2 = 0, d_here = 0;
          long long d_before = 0;
          if (depth_stats) {
            d_id1 = rep_1->my_id;
            d_id2 = rep_2->my_id;
            d_here = 1 + std::max(dep[d_id1], dep[d_id2]);
            d_before = cs.regular + cs.small + cs.forced + cs.fail;
            ++d_edges;
          }
          struct DepthNote {   // runs when the edge has been decided
            std::function<void()> f;
            ~DepthNote() { if (f) f(); }
          } note;
          if (depth_stats) {
            const int c1 = rep_1->constraint_id, c2 = rep_2->constraint_id;
            note.f = [&, c1, c2]() {
              const bool changed = (cs.regular + cs.small + cs.forced + cs.fail) != d_before ||
                                   rep_1->constraint_id != c1 || rep_2->constraint_id != c2;
              if (!changed) return;
              ++d_changing;
              d_max = std::max(d_max, d_here);
              dep[d_id1] = dep[d_id2] = d_here;
              dep[GetRegion(d_id1)->my_id] = d_here;   // the survivor of a merge
            };
          }
          if (rep_1->constraint_id < 0 || rep_2->constraint_id < 0) {
            if (!rep_1->region_finalized && !rep_2->region_finalized) {
              const float d = DescriptorDistance(rep_1->descriptor, rep_2->descriptor, weight);
              if (d < merge_thr) {
                MergeRegions(rep_1, rep_2);
                ++num_regular_merges_;
                ++cs.regular;
              } else {
                rep_1->region_finalized = true;
                rep_2->region_finalized = true;
                ++cs.fail;
              }
            }
            if (rep_1->region_finalized || rep_2->region_finalized) {
              if (rep_1->sz < min_region_size || rep_2->sz < min_region_size) {
                MergeRegions(rep_1, rep_2);
                ++num_small_region_merges_;
                ++cs.small;
              } else {
                remaining.push_back(e);
                ++cs.kept;
              }
            }
          } else if (rep_1->constraint_id == rep_2->constraint_id) {
            const float d = DescriptorDistance(rep_1->descriptor, rep_2->descriptor, weight);
            if (d > split_thr) {
              if ((double)rep_1->sz < (double)rep_2->sz * 0.3) {
                rep_1->constraint_id = -1;
              } else if ((double)rep_2->sz < (double)rep_1->sz * 0.3) {
                rep_2->constraint_id = -1;
              } else {
                rep_1->constraint_id = -1;
                rep_2->constraint_id = -1;
              }
              remaining.push_back(e);
              ++cs.kept;
            } else {
              MergeRegions(rep_1, rep_2);
              ++num_forced_merges_;
              ++cs.forced;
            }
          } else {
            remaining.push_back(e);
            ++cs.kept;
          }
        }
        edges.swap(remaining);
      }
      if (depth_stats && d_edges >= 10000) {
        std::fprintf(stderr, "[vso] bucket %d: %lld edges between different regions, %lld change a state, "
                     "longest chain %d\n", bucket_idx, d_edges, d_changing, d_max);
      }
    }
    if (force_constraints) MergeConstrainedRegions();
  }

  void NodeRoots(int32_t* out) {
    for (int i = 0, n = (int)regions_.size(); i < n; ++i) out[i] = GetRegion(i)->my_id;
  }

  // DenseSegmentationGraph::ObtainResults, dense_segmentation_graph.h:468-579.
  // flows: per slice W*H*2 f32 (null entries allowed for slices without flow) or null.
  void ObtainResults(RegionInfoList* region_list, RegionInfoPtrMap* region_map,
                     const std::vector<const float*>* flows, bool enforce_n4,
                     bool enforce_spatial_connectedness) {
    if (enforce_spatial_connectedness) {
      FlattenUnionFind(true);
      if (flows) VSO_CHECK((int)flows->size() == num_frames_);
    }
    const int lda = W_ + 2;
    std::fill(region_ids_.begin(), region_ids_.begin() + lda, -1);
    std::fill(region_ids_.end() - lda, region_ids_.end(), -1);
    for (int i = 0; i < H_ + 2; ++i) {
      region_ids_[(size_t)i * lda] = -1;
      region_ids_[(size_t)i * lda + W_ + 1] = -1;
    }
    int* id_view = region_ids_.data() + lda + 1;  // pixel (0,0)
    std::unordered_map<int, int> size_adjust_map;

    for (int t = 0; t < num_frames_; ++t) {
      const int base_idx = W_ * H_ * t;
      if (std::binary_search(virtual_slices_.begin(), virtual_slices_.end(), t)) continue;
      // constrained_slices_ is never filled on the live path -> N4 always runs.
      for (int i = 0, idx = base_idx; i < H_; ++i) {
        int* region_ptr = id_view + (size_t)i * lda;
        for (int j = 0; j < W_; ++j, ++idx) region_ptr[j] = GetRegion(idx)->my_id;
      }
      if (enforce_n4) EnforceN4Connectivity(id_view, lda, &size_adjust_map);
      for (int i = 0; i < H_; ++i) {
        const int* region_ptr = id_view + (size_t)i * lda;
        int prev_id = region_ptr[0];
        int left_x = 0;
        for (int j = 1; j < W_; ++j) {
          const int curr_id = region_ptr[j];
          if (prev_id != curr_id) {
            AddIntervalToRasterization(t, i, left_x, j - 1, prev_id, region_list, region_map);
            left_x = j;
            prev_id = curr_id;
          }
          if (j + 1 == W_) {
            AddIntervalToRasterization(t, i, left_x, j, prev_id, region_list, region_map);
          }
        }
        // NOTE: a frame of width 1 emits no interval in the reference (loop starts at j = 1).
      }
    }
    if (enforce_spatial_connectedness) {
      EnforceSpatialConnectedness(region_list, region_map, flows, &size_adjust_map);
    }
    for (const auto& kv : size_adjust_map) {
      auto pos = region_map->find(kv.first);
      if (pos == region_map->end()) {
        regions_[kv.first].sz = 0;
        continue;
      }
      pos->second->size += kv.second;
    }
  }

  // FastSegmentationGraph::DetermineNeighborIdsImpl, segmentation_graph.h:466-496.
  void DetermineNeighborIds(RegionInfoList* region_list, RegionInfoPtrMap* map) {
    for (int bucket_idx = 0; bucket_idx <= num_buckets_; ++bucket_idx) {
      for (size_t bl = 0; bl < bucket_lists_.size(); ++bl) {
        for (const Edge& e : bucket_lists_[bl][bucket_idx]) {
          const Region* r1 = GetRegion(e.region_1);
          const Region* r2 = GetRegion(e.region_2);
          const int r1_id = r1->my_id, r2_id = r2->my_id;
          if (r1_id == r2_id) continue;
          RegionInformation* r1_info = GetCreateRegionInformation(*r1, region_list, map);
          RegionInformation* r2_info = GetCreateRegionInformation(*r2, region_list, map);
          InsertSortedUniquely(r2_info->index, &r1_info->neighbor_idx);
          InsertSortedUniquely(r1_info->index, &r2_info->neighbor_idx);
        }
      }
    }
  }

  void MergeStats(int64_t* s3) const {
    s3[0] = num_forced_merges_;
    s3[1] = num_regular_merges_;
    s3[2] = num_small_region_merges_;
  }
  const std::vector<BucketCensus>& census() const { return census_; }

 private:
  struct Edge {
    int region_1, region_2;
  };
  typedef std::vector<Edge> EdgeList;
  // segmentation_graph.h:242-261.  virtual_no_desc marks regions built with the descriptor-less
  // constructor (indeterminate descriptor in the reference; never read with a non-zero weight,
  // SURVEY A.7-5).
  struct Region {
    int my_id = -1;
    int sz = 0;
    int constraint_id = -1;
    bool region_finalized = false;
    bool virtual_no_desc = false;
    float descriptor[3] = {0, 0, 0};
  };

  // segmentation_graph.h:158-162
  inline void AddEdge(int r1, int r2, float weight, int bucket_list) {
    const int bucket_index = (int)(std::min<float>((float)num_buckets_, weight * scale_));
    bucket_lists_[bucket_list][bucket_index].push_back(Edge{r1, r2});
  }

  // segmentation_graph.h:651-669 (recursive path compression, restated iteratively: every node
  // on the path ends up pointing at the representative).
  inline Region* GetRegion(int id) {
    int root = id;
    for (;;) {
      const int p = regions_[root].my_id;
      if (regions_[p].my_id == p) {
        root = p;
        break;
      }
      root = p;
    }
    int cur = id;
    while (regions_[cur].my_id != root) {
      const int next = regions_[cur].my_id;
      regions_[cur].my_id = root;
      cur = next;
    }
    return &regions_[root];
  }

  // pixel_distance.h:479-493
  inline float DescriptorDistance(const float* lhs, const float* rhs, float edge_distance) const {
    const float d1 = lhs[0] - rhs[0], d2 = lhs[1] - rhs[1], d3 = lhs[2] - rhs[2];
    const float dist = (float)std::sqrt((double)((d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 3.0f)));
    if (edge_distance < force_merge_weight_ && (double)dist < 0.2) return 0.0f;
    return dist;
  }

  // segmentation_graph.h:671-701 + pixel_distance.h:495-505
  inline Region* MergeRegions(Region* rep_1, Region* rep_2) {
    Region *merged, *other;
    if (rep_1->sz > rep_2->sz) {
      merged = rep_1;
      other = rep_2;
    } else {
      merged = rep_2;
      other = rep_1;
    }
    if (!merged->virtual_no_desc && !other->virtual_no_desc) {
      const float denom = 1.0f / (float)(other->sz + merged->sz);
      const float a = (float)other->sz * denom;
      const float b = (float)merged->sz * denom;
      merged->descriptor[0] = a * other->descriptor[0] + b * merged->descriptor[0];
      merged->descriptor[1] = a * other->descriptor[1] + b * merged->descriptor[1];
      merged->descriptor[2] = a * other->descriptor[2] + b * merged->descriptor[2];
    } else if (merged->virtual_no_desc && !other->virtual_no_desc) {
      // Only reachable with sz(other) <= sz(merged) == 0, i.e. two size-0 regions; descriptor
      // indeterminate in the reference and never read afterwards.
    }
    // A size-0 virtual region merged into a real one contributes a = 0 in exact arithmetic; the
    // reference multiplies an indeterminate value by 0 here.  We leave merged's descriptor
    // untouched (it is never read again after MergeConstrainedRegions' virtual pass).
    merged->sz += other->sz;
    merged->constraint_id = std::max(rep_1->constraint_id, rep_2->constraint_id);
    other->my_id = merged->my_id;
    return merged;
  }

  // segmentation_graph.h:703-786
  void MergeConstrainedRegions() {
    std::unordered_map<int, int> constraint_to_region;
    std::vector<std::pair<int, int>> vnodes(virtual_nodes_);
    vnodes.push_back(std::make_pair(0, 0));
    vnodes.push_back(std::make_pair((int)regions_.size(), (int)regions_.size()));
    std::sort(vnodes.begin(), vnodes.end());
    const float split_thr = 0.15f;
    for (size_t k = 1; k < vnodes.size(); ++k) {
      for (int ri = vnodes[k - 1].second; ri != vnodes[k].first; ++ri) {
        if (regions_[ri].constraint_id < 0) continue;
        Region* my_rep = GetRegion(regions_[ri].my_id);
        auto pos = constraint_to_region.find(my_rep->constraint_id);
        if (pos == constraint_to_region.end()) {
          constraint_to_region.insert(std::make_pair(my_rep->constraint_id, my_rep->my_id));
        } else {
          Region* constraint_rep = GetRegion(pos->second);
          if (constraint_rep != my_rep) {
            const float distance =
                DescriptorDistance(my_rep->descriptor, constraint_rep->descriptor, 1.0f);
            if (distance > split_thr) {
              if ((double)my_rep->sz < (double)constraint_rep->sz * 0.3) {
                my_rep->constraint_id = -1;
              } else if ((double)constraint_rep->sz < (double)my_rep->sz * 0.3) {
                constraint_rep->constraint_id = -1;
                pos->second = my_rep->my_id;
              } else {
                my_rep->constraint_id = -1;
                constraint_rep->constraint_id = -1;
                constraint_to_region.erase(pos);
              }
            } else {
              MergeRegions(my_rep, constraint_rep);
            }
          }
        }
      }
    }
    for (size_t k = 0; k < vnodes.size(); ++k) {
      for (int ri = vnodes[k].first; ri != vnodes[k].second; ++ri) {
        VSO_CHECK(regions_[ri].constraint_id >= 0);
        Region* my_rep = GetRegion(regions_[ri].my_id);
        auto pos = constraint_to_region.find(my_rep->constraint_id);
        if (pos == constraint_to_region.end()) {
          constraint_to_region.insert(std::make_pair(my_rep->constraint_id, my_rep->my_id));
        } else {
          Region* constraint_rep = GetRegion(pos->second);
          if (constraint_rep != my_rep) MergeRegions(my_rep, constraint_rep);
        }
      }
    }
  }

  // segmentation_graph.h:596-629 with separate_representatives = true.
  void FlattenUnionFind(bool separate_representatives) {
    VSO_CHECK(!flattened_);
    flattened_ = true;
    const int region_offset = (int)regions_.size();
    int new_region_id = region_offset;
    VSO_CHECK(separate_representatives);
    for (int i = 0; i < region_offset; ++i) {
      Region* r = GetRegion(i);
      int flattened_id = r->my_id;
      if (flattened_id < region_offset) {
        r->my_id = new_region_id;
        flattened_id = new_region_id;
        Region nr;
        nr.my_id = new_region_id++;
        nr.sz = r->sz;
        nr.constraint_id = r->constraint_id;
        nr.virtual_no_desc = true;
        regions_.push_back(nr);
      }
      regions_[i].my_id = flattened_id;
    }
  }

  // segmentation_graph.h:498-522
  RegionInformation* GetCreateRegionInformation(const Region& region, RegionInfoList* region_list,
                                                RegionInfoPtrMap* map) {
    auto it = map->find(region.my_id);
    if (it != map->end()) return it->second;
    RegionInformation* info = new RegionInformation;
    info->index = max_region_id_++;
    info->size = region.sz;
    info->constrained_id = region.constraint_id;
    region_list->emplace_back(info);
    map->insert(std::make_pair(region.my_id, info));
    return info;
  }

  // dense_segmentation_graph.h:432-466
  void AddIntervalToRasterization(int frame, int y, int left_x, int right_x, int region_id,
                                  RegionInfoList* region_list, RegionInfoPtrMap* map) {
    RegionInformation* ri = GetCreateRegionInformation(regions_[region_id], region_list, map);
    if (ri->raster == nullptr) ri->raster.reset(new Rasterization3D);
    if (ri->raster->empty() || ri->raster->back().first < frame) {
      ri->raster->push_back(std::make_pair(frame, std::make_shared<Rasterization>()));
    }
    ri->raster->back().second->push_back(ScanInterval{y, left_x, right_x});
  }

  // dense_segmentation_graph.h:1303-1337
  void EnforceN4Connectivity(int* id_view, int lda, std::unordered_map<int, int>* size_adjust) {
    for (int i = 0; i < H_ - 1; ++i) {
      int* region_ptr = id_view + (size_t)i * lda;
      for (int j = 0; j < W_; ++j, ++region_ptr) {
        const int region_id = *region_ptr;
        if (region_ptr[lda - 1] == region_id && region_ptr[-1] != region_id &&
            region_ptr[lda] != region_id) {
          --(*size_adjust)[region_ptr[lda]];
          ++(*size_adjust)[region_id];
          region_ptr[lda] = region_id;
        }
        if (region_ptr[lda + 1] == region_id && region_ptr[1] != region_id &&
            region_ptr[lda] != region_id) {
          --(*size_adjust)[region_ptr[lda]];
          ++(*size_adjust)[region_id];
          region_ptr[lda] = region_id;
        }
      }
    }
  }

  // dense_segmentation_graph.h:666-904
  void EnforceSpatialConnectedness(RegionInfoList* region_list, RegionInfoPtrMap* region_map,
                                   const std::vector<const float*>* flows,
                                   std::unordered_map<int, int>* size_adjust_map) {
    const int num_regions = (int)region_list->size();
    for (int r = 0; r < num_regions; ++r) {
      RegionInformation& ri = *(*region_list)[r];
      if (ri.raster == nullptr) continue;
      Rasterization3D& raster = *ri.raster;
      std::vector<Tube3D> result_tubes;
      std::vector<Tube3D> active_tubes;
      const float inv_frame_diam = (float)(1.0f / std::hypot((double)W_, (double)H_));

      for (const auto& raster_slice : raster) {
        const int frame = raster_slice.first;
        std::vector<Rasterization> components;
        ConnectedComponents(*raster_slice.second, /*n4=*/true, &components);
        std::vector<TubeSlice> slices;
        slices.reserve(components.size());
        for (auto& comp : components) {
          TubeSlice slice;
          slice.frame = frame;
          slice.raster.swap(comp);
          slice.ComputeShapeDescriptor();
          slices.push_back(std::move(slice));
        }
        components.clear();

        if (active_tubes.empty()) {
          for (auto& slice : slices) active_tubes.push_back(Tube3D{std::move(slice)});
        } else {
          std::vector<Tube3D> new_active_tubes;
          std::vector<int> used_indices(active_tubes.size(), 0);
          for (auto& slice : slices) {
            const float* flow = nullptr;
            if (flows) flow = (*flows)[frame];
            const auto match = FindPreviousTube(slice, active_tubes, frame, flow, W_);
            const int prev_idx = match.first;
            if (prev_idx < 0) {
              new_active_tubes.push_back(Tube3D{std::move(slice)});
              continue;
            }
            const float diff_dist = match.second;
            const int sa = active_tubes[prev_idx].back().shape.size;
            const int sb = slice.shape.size;
            // int / (int + 1e-6) -> double; stored to a float in the reference.
            const float area_ratio = (float)((double)std::min(sa, sb) /
                                             ((double)std::max(sa, sb) + 1e-6));
            if ((double)area_ratio > 0.75 && diff_dist * inv_frame_diam < 0.04f) {
              ++used_indices[prev_idx];
              active_tubes[prev_idx].push_back(std::move(slice));
              new_active_tubes.push_back(Tube3D());
              new_active_tubes.back().swap(active_tubes[prev_idx]);
            } else {
              new_active_tubes.push_back(Tube3D{std::move(slice)});
            }
          }
          for (size_t k = 0; k < active_tubes.size(); ++k) {
            if (used_indices[k] == 0) result_tubes.push_back(std::move(active_tubes[k]));
          }
          new_active_tubes.swap(active_tubes);
        }
      }
      for (auto& t : active_tubes) result_tubes.push_back(std::move(t));
      if (result_tubes.size() <= 1) continue;

      auto merge_with_closest_tube = [&result_tubes](int k) -> bool {
        const int idx = GetClosestTube3D(result_tubes[k], result_tubes, k);
        if (idx < 0) return false;
        Tube3D merged;
        MergeTube3D(result_tubes[idx], result_tubes[k], &merged);
        result_tubes[idx].swap(merged);
        result_tubes.erase(result_tubes.begin() + k);
        return true;
      };

      for (int k = 0; k < (int)result_tubes.size();) {
        bool merge = AverageTubeSliceSize(result_tubes[k]) < 20;
        if (!merge) {
          for (int l = 0; l < (int)result_tubes.size(); ++l) {
            if (l == k) continue;
            if ((double)Tube3DIntersection(result_tubes[k], result_tubes[l]) > 0.8) {
              merge = true;
              break;
            }
          }
        }
        if (merge && merge_with_closest_tube(k)) {
        } else {
          ++k;
        }
      }

      for (int k = 0; k < (int)result_tubes.size();) {
        bool is_merged = false;
        for (int l = 0; l < (int)result_tubes.size(); ++l) {
          if (l == k) continue;
          if (AreTubesTemporalNeighbors(result_tubes[k], result_tubes[l])) {
            Tube3D merged;
            MergeTube3D(result_tubes[k], result_tubes[l], &merged);
            result_tubes[l].swap(merged);
            result_tubes.erase(result_tubes.begin() + k);
            is_merged = true;
            break;
          }
        }
        if (!is_merged) ++k;
      }

      int tube_to_keep = -1;
      int tube_to_keep_score = 0;
      std::vector<float> tube_areas(result_tubes.size());
      for (int k = 0; k < (int)result_tubes.size(); ++k) {
        float area = 0;
        for (const auto& slice : result_tubes[k]) area += (float)slice.shape.size;
        tube_areas[k] = area;
        const float tube_score = area;
        if (tube_score > (float)tube_to_keep_score) {
          tube_to_keep_score = (int)tube_score;
          tube_to_keep = k;
        }
      }

      for (int k = 0; k < (int)result_tubes.size(); ++k) {
        int first_idx = result_tubes[k][0].frame * W_ * H_;
        const ScanInterval& first_scanline = result_tubes[k][0].raster[0];
        first_idx += first_scanline.y * W_ + first_scanline.left_x;
        Region* rep = GetRegion(first_idx);
        if (k != tube_to_keep) {
          // int -= float: evaluated in float, then truncated (matters above 2^24).
          int& adj = (*size_adjust_map)[rep->my_id];
          adj = (int)((float)adj - tube_areas[k]);
          Region nr;
          nr.my_id = (int)regions_.size();
          nr.sz = (int)tube_areas[k];
          nr.constraint_id = -1;
          nr.virtual_no_desc = true;
          regions_.push_back(nr);
          rep = &regions_.back();
          const int region_id = rep->my_id;
          for (const auto& slice : result_tubes[k]) {
            const int base_idx = slice.frame * W_ * H_;
            for (const ScanInterval& si : slice.raster) {
              const int row_idx = base_idx + si.y * W_;
              for (int x = si.left_x; x <= si.right_x; ++x) regions_[row_idx + x].my_id = region_id;
            }
          }
        }
        RegionInformation* nri = GetCreateRegionInformation(*rep, region_list, region_map);
        nri->raster.reset(new Rasterization3D);
        for (auto& slice : result_tubes[k]) {
          auto new_raster = std::make_shared<Rasterization>();
          new_raster->swap(slice.raster);
          nri->raster->push_back(std::make_pair(slice.frame, new_raster));
        }
      }
    }
  }

  // dense_segmentation_graph.h:1180-1228
  void AddNodesWithDescriptors(const float* feat, const int32_t* constraint_ids) {
    const int base_idx = num_frames_ * H_ * W_;
    VSO_CHECK((int)regions_.size() == base_idx);
    for (int i = 0; i < H_; ++i) {
      for (int j = 0; j < W_; ++j) {
        Region r;
        r.my_id = base_idx + i * W_ + j;
        r.sz = 1;
        r.constraint_id = constraint_ids ? constraint_ids[(size_t)i * W_ + j] : -1;
        const float* p = feat + ((size_t)i * W_ + j) * 3;
        r.descriptor[0] = p[0];
        r.descriptor[1] = p[1];
        r.descriptor[2] = p[2];
        regions_.push_back(r);
      }
    }
  }

  // dense_segmentation_graph.h:956-1000
  void AddSpatialEdgesImpl(const float* feat, int frame_idx) {
    const int base_idx = frame_idx * H_ * W_;
    const int bl = 2 * frame_idx;
    int cur_idx = base_idx;
    const int end_y = H_ - 1, end_x = W_ - 1;
    for (int i = 0; i <= end_y; ++i) {
      for (int j = 0; j <= end_x; ++j, ++cur_idx) {
        const float* a = feat + ((size_t)i * W_ + j) * 3;
        auto dist = [&](int dx, int dy) {
          const float* b = feat + ((size_t)(i + dy) * W_ + (j + dx)) * 3;
          return l1_ ? ColorDiff3L1(a, b) : ColorDiff3L2(a, b);
        };
        if (j < end_x) AddEdge(cur_idx, cur_idx + 1, dist(1, 0), bl);
        if (i < end_y) {
          AddEdge(cur_idx, cur_idx + W_, dist(0, 1), bl);
          if (j > 0) AddEdge(cur_idx, cur_idx + W_ - 1, dist(-1, 1), bl);
          if (j < end_x) AddEdge(cur_idx, cur_idx + W_ + 1, dist(1, 1), bl);
        }
      }
    }
  }

  int W_, H_, max_frames_;
  bool l1_;
  int num_frames_ = 0;
  int num_buckets_ = 2048;
  float scale_ = 1.0f;
  float force_merge_weight_ = 0.001f;
  std::vector<Region> regions_;
  std::vector<std::pair<int, int>> virtual_nodes_;
  std::vector<std::vector<EdgeList>> bucket_lists_;
  std::vector<int> region_ids_;
  std::vector<int> virtual_slices_;
  bool flattened_ = false;
  int max_region_id_ = 0;
  int64_t num_forced_merges_ = 0, num_regular_merges_ = 0, num_small_region_merges_ = 0;
  bool segmented_once_ = false;
  std::vector<std::thread> add_edges_tasks_;
  std::vector<BucketCensus> census_;
};

// segment_util/segmentation_util.cpp:741-770 (level 0).  out: W*H, untouched where uncovered.
static void SegmentationDescToIdImage(const SegmentationDesc& seg, int W, int32_t* out) {
  for (const Region2D& region : seg.region) {
    for (const ScanInterval& s : region.raster) {
      int32_t* p = out + (size_t)s.y * W + s.left_x;
      for (int j = 0, len = s.right_x - s.left_x + 1; j < len; ++j) p[j] = region.id;
    }
  }
}

#include "vs_oracle_boundary.inc"
#include "vs_oracle_region.inc"

// ------------------------------------------------------------------------------------------
// Segmentation (over-segmentation half), segmentation/segmentation.cpp.
// ------------------------------------------------------------------------------------------
struct SegOptions {
  int min_region_size = 200;
  bool compute_vectorization = false;    // segmentation.h: compute_vectorization (seg_tree --over_segment)
  bool two_stage_segmentation = false;   // segmentation.h:53-55
  bool enforce_n4_connectivity = true;
  bool enforce_spatial_connectedness = true;
};

class Segmentation {
 public:
  Segmentation(const SegOptions& o, int W, int H, int chunk_id, int max_frames, bool l1)
      : options_(o), W_(W), H_(H), chunk_id_(chunk_id), graph_(new DenseGraph(W, H, max_frames, l1)) {}

  DenseGraph* graph() { return graph_.get(); }

  // segmentation.cpp:272-303
  void RunOverSegmentation(const std::vector<const float*>* flows) {
    region_infos_.reset(new RegionInfoList());
    if (options_.two_stage_segmentation) graph_->SegmentGraphSpatially();   // segmentation.cpp:280-283
    graph_->SegmentGraph(options_.min_region_size, true);
    graph_->MergeStats(merge_stats_);
    RegionInfoPtrMap map;
    graph_->ObtainResults(region_infos_.get(), &map, flows, options_.enforce_n4_connectivity,
                          options_.enforce_spatial_connectedness);
    graph_->DetermineNeighborIds(region_infos_.get(), &map);
    graph_.reset();
  }

  // segmentation.cpp:392-420 (level 0 only)
  void ConstrainSegmentationToFrameInterval(int lhs, int rhs) {
    for (auto& rp : *region_infos_) {
      if (rp->raster == nullptr || rp->raster->empty() || rp->raster->front().first >= rhs ||
          rp->raster->back().first < lhs) {
        rp->flagged_for_removal = true;
      }
    }
  }

  // segmentation.cpp:422-456 (level 0 only)
  void AdjustRegionAreaToFrameInterval(int lhs, int rhs) {
    for (auto& rp : *region_infos_) {
      int size_increment = 0;
      if (rp->raster == nullptr) continue;
      for (const auto& slice : *rp->raster) {
        if (slice.first < lhs || slice.first >= rhs) size_increment -= RasterizationArea(*slice.second);
      }
      rp->size += size_increment;
    }
  }

  // segmentation.cpp:537-582
  void AssignUniqueRegionIds(bool use_constrained_ids, int region_id_offset, int* max_region_id) {
    assigned_constrained_ids_ = use_constrained_ids;
    int max_id = -1;
    for (auto& rp : *region_infos_) {
      if (use_constrained_ids && rp->constrained_id >= 0) {
        rp->region_id = rp->constrained_id;
      } else {
        rp->region_id = rp->index + region_id_offset;
      }
      max_id = std::max(max_id, rp->region_id);
    }
    if (max_region_id) *max_region_id = std::max(region_id_offset, max_id + 1);
  }

  // segmentation.cpp:458-533, 671-773 (save_descriptors = false, no vectorization)
  void RetrieveSegmentation3D(int frame_number, bool output_hierarchy, SegmentationDesc* desc) {
    desc->frame_width = W_;
    desc->frame_height = H_;
    desc->chunk_id = chunk_id_;
    desc->connectedness = options_.enforce_n4_connectivity ? 1 : 2;
    for (const auto& rp : *region_infos_) {
      const RegionInformation& ri = *rp;
      if (ri.raster == nullptr) continue;
      // LocateRasterization: first slice with frame >= frame_number.
      auto it = std::lower_bound(
          ri.raster->begin(), ri.raster->end(), frame_number,
          [](const std::pair<int, std::shared_ptr<Rasterization>>& a, int f) { return a.first < f; });
      if (it == ri.raster->end() || it->first != frame_number) continue;
      VSO_CHECK(!it->second->empty());
      Region2D r;
      r.id = ri.region_id;
      r.raster = *it->second;
      ShapeMomentsFromRasterization(r.raster, &r.shape_moments);
      desc->region.push_back(std::move(r));
    }
    if (assigned_constrained_ids_) {
      // std::sort by id (segmentation_util.cpp:187-191); ids are unique per frame.
      std::sort(desc->region.begin(), desc->region.end(),
                [](const Region2D& a, const Region2D& b) { return a.id < b.id; });
    }
    if (output_hierarchy) {
      desc->hierarchy.emplace_back();
      HierarchyLevel& hier = desc->hierarchy.back();
      for (const auto& rp : *region_infos_) {
        const RegionInformation& ri = *rp;
        if (ri.flagged_for_removal) continue;
        CompoundRegion c;
        c.id = ri.region_id;
        c.size = ri.size;
        for (int n : ri.neighbor_idx) {
          if ((*region_infos_)[n]->flagged_for_removal) continue;
          c.neighbor_id.push_back((*region_infos_)[n]->region_id);
        }
        if (assigned_constrained_ids_) std::sort(c.neighbor_id.begin(), c.neighbor_id.end());
        VSO_CHECK(ri.raster != nullptr);
        c.start_frame = ri.raster->front().first;
        c.end_frame = ri.raster->back().first;
        hier.region.push_back(std::move(c));
      }
      if (assigned_constrained_ids_) {
        std::sort(hier.region.begin(), hier.region.end(),
                  [](const CompoundRegion& a, const CompoundRegion& b) { return a.id < b.id; });
      }
    }
    if (options_.compute_vectorization) ComputeFrameVectorization(desc);   // segmentation.cpp:527-532
  }

  const int64_t* merge_stats() const { return merge_stats_; }

 private:
  SegOptions options_;
  int W_, H_, chunk_id_;
  std::unique_ptr<DenseGraph> graph_;
  std::unique_ptr<RegionInfoList> region_infos_;
  bool assigned_constrained_ids_ = false;
  int64_t merge_stats_[3] = {0, 0, 0};
};

// ------------------------------------------------------------------------------------------
// DenseSegmentation (segmentation/dense_segmentation.{h,cpp}).
// ------------------------------------------------------------------------------------------
class DenseSegmentation {
 public:
  DenseSegmentation(const vso_options& o, int W, int H) : options_(o), W_(W), H_(H) {
    VSO_CHECK(options_.chunk_size >= 3);                                        // cpp:54
    overlap_frames_ = (int)(options_.chunk_overlap_ratio * (float)options_.chunk_size + 0.5f);  // :59
    overlap_frames_ = std::min(overlap_frames_, 2);                             // :62
    VSO_CHECK(overlap_frames_ < options_.chunk_size);
    VSO_CHECK(options_.num_constraint_frames >= 1);
    constraint_frames_ = std::min(options_.num_constraint_frames, overlap_frames_ - 1);  // :71
  }

  // dense_segmentation.cpp:108-162.  features == nullptr <=> no new frame.
  int ProcessFrame(bool flush, const uint8_t* bgr, size_t stride, const float* flow,
                   bool has_flow_stream) {
    if (seg_ == nullptr && !pending_import_) NewSegmentation(options_.chunk_size);
    if (bgr && pending_import_) {
      // First frame of a stream that continues another one: it is the constrained overlap frame;
      // rebuild exactly the state ChunkBoundaryOutput leaves behind (cpp:291-315).
      auto feat = std::make_shared<std::vector<float>>((size_t)W_ * H_ * 3);
      PreprocessFeatures(bgr, stride, W_, H_, options_.presmoothing, feat->data());
      feature_buffer_.push_back(nullptr);
      feature_buffer_.push_back(feat);
      if (has_flow_stream) {
        VSO_CHECK(flow != nullptr);
        flow_buffer_.push_back(nullptr);
        flow_buffer_.push_back(std::make_shared<std::vector<float>>(flow, flow + (size_t)W_ * H_ * 2));
      }
      curr_chunk_start_ = 1;
      NewSegmentation(curr_chunk_start_ + options_.chunk_size);
      seg_->graph()->AddVirtualFrame(halo_[0].data());
      seg_->graph()->AddFrame(feature_buffer_[1]->data(), halo_[1].data());
      seg_->graph()->AddTemporal(nullptr, nullptr,
                                 flow_buffer_.empty() ? nullptr : flow_buffer_[1]->data(), true);
      pending_import_ = false;
      ++input_frames_;
      bgr = nullptr;   // consumed
    }
    if (bgr) {
      auto feat = std::make_shared<std::vector<float>>((size_t)W_ * H_ * 3);
      PreprocessFeatures(bgr, stride, W_, H_, options_.presmoothing, feat->data());
      feature_buffer_.push_back(feat);
      if (has_flow_stream) {
        if (input_frames_ == 0) {
          flow_buffer_.push_back(nullptr);
        } else {
          VSO_CHECK(flow != nullptr);
          auto fc = std::make_shared<std::vector<float>>(flow, flow + (size_t)W_ * H_ * 2);
          flow_buffer_.push_back(fc);
          VSO_CHECK(flow_buffer_.size() == feature_buffer_.size());
        }
      }
      seg_->graph()->AddFrame(feature_buffer_.back()->data(), nullptr);  // :145
      if (feature_buffer_.size() > 1) {                                   // :147-153
        const float* fl = flow_buffer_.empty() ? nullptr : flow_buffer_.back()->data();
        seg_->graph()->AddTemporal(feature_buffer_.end()[-1]->data(),
                                   feature_buffer_.end()[-2]->data(), fl, false);
      }
      ++input_frames_;
    }
    if (flush || (int)feature_buffer_.size() - curr_chunk_start_ >= options_.chunk_size) {
      ChunkBoundaryOutput(flush);
      return (int)results_.size();
    }
    results_.clear();
    return 0;
  }

  const std::vector<std::unique_ptr<SegmentationDesc>>& results() const { return results_; }
  const int64_t* last_merge_stats() const { return last_merge_stats_; }

  // Multi-GPU hand-off (not in the reference, where the state simply lives on in the object):
  // everything the next chunk needs from this one -- the id images of the two overlap
  // segmentations (dense_segmentation.cpp:300-308) and the running counters.
  bool ExportHalo(int32_t* virt, int32_t* cons, int64_t scalars[4]) const {
    if (halo_[0].empty()) return false;
    std::memcpy(virt, halo_[0].data(), halo_[0].size() * sizeof(int32_t));
    std::memcpy(cons, halo_[1].data(), halo_[1].size() * sizeof(int32_t));
    scalars[0] = max_region_id_;
    scalars[1] = chunk_id_;
    scalars[2] = num_output_frames_;
    scalars[3] = input_frames_;
    return true;
  }
  void ImportHalo(const int32_t* virt, const int32_t* cons, const int64_t scalars[4]) {
    VSO_CHECK(input_frames_ == 0 && seg_ == nullptr);
    halo_[0].assign(virt, virt + (size_t)W_ * H_);
    halo_[1].assign(cons, cons + (size_t)W_ * H_);
    max_region_id_ = (int)scalars[0];
    chunk_id_ = (int)scalars[1];
    num_output_frames_ = (int)scalars[2];
    input_frames_ = (int)scalars[3] - 1;   // the constrained overlap frame is fed again
    pending_import_ = true;
  }
  const float* last_smoothed() const {
    return (feature_buffer_.empty() || !feature_buffer_.back()) ? nullptr : feature_buffer_.back()->data();
  }
  int W() const { return W_; }
  int H() const { return H_; }

 private:
  // dense_segmentation.cpp:268-279
  void NewSegmentation(int max_frames) {
    SegOptions so;
    so.min_region_size = (int)(options_.frac_min_region_size * (float)W_ *
                               options_.frac_min_region_size * (float)H_ * (float)options_.chunk_size);
    so.enforce_n4_connectivity = options_.enforce_n4_connectivity != 0;
    so.enforce_spatial_connectedness = options_.enforce_spatial_connectedness != 0;
    so.two_stage_segmentation = options_.two_stage_oversegment != 0;   // dense_segmentation.cpp:274
    so.compute_vectorization = options_.compute_vectorization != 0;
    seg_.reset(new Segmentation(so, W_, H_, chunk_id_, max_frames, options_.color_distance == 0));
  }

  // dense_segmentation.cpp:281-331
  void ChunkBoundaryOutput(bool flush) {
    SegmentAndOutputChunk(flush);
    if (flush) {
      seg_.reset();
      return;
    }
    NewSegmentation(curr_chunk_start_ + options_.chunk_size);
    VSO_CHECK((int)overlap_segmentations_.size() == constraint_frames_ + 1);
    VSO_CHECK(overlap_segmentations_.size() >= 2);
    // The id image is the graph's persistent region_ids_ buffer in the reference; every pixel is
    // covered by exactly one Region2D, so the previous contents never show through.
    for (int k = 0; k < 2; ++k) {
      halo_[k].assign((size_t)W_ * H_, -1);
      SegmentationDescToIdImage(*overlap_segmentations_[k], W_, halo_[k].data());
    }
    seg_->graph()->AddVirtualFrame(halo_[0].data());                            // :305
    seg_->graph()->AddFrame(feature_buffer_[1]->data(), halo_[1].data());       // :307-308
    if (!flow_buffer_.empty()) {                                                // :311-315
      seg_->graph()->AddTemporal(nullptr, nullptr, flow_buffer_[1]->data(), true);
    } else {
      seg_->graph()->AddTemporal(nullptr, nullptr, nullptr, true);
    }
    // Loop :318-328 never runs (overlap_frames_ <= 2).
    overlap_segmentations_.clear();
  }

  // dense_segmentation.cpp:333-432
  void SegmentAndOutputChunk(bool flush) {
    std::vector<const float*> flows;
    if (!flow_buffer_.empty()) {
      for (const auto& f : flow_buffer_) flows.push_back(f ? f->data() : nullptr);
    }
    seg_->RunOverSegmentation(flow_buffer_.empty() ? nullptr : &flows);
    std::memcpy(last_merge_stats_, seg_->merge_stats(), sizeof(last_merge_stats_));

    const int buffered = (int)feature_buffer_.size();
    const int overlap_start = buffered - (flush ? 0 : overlap_frames_);
    const int last_output_frame = std::min<int>(buffered - 1, overlap_start);
    VSO_CHECK(overlap_start > curr_chunk_start_);
    const int max_result_frame = std::min<int>(buffered - 1, last_output_frame + constraint_frames_);

    seg_->ConstrainSegmentationToFrameInterval(0, last_output_frame + 1);
    seg_->AdjustRegionAreaToFrameInterval(0, last_output_frame + 1);
    int new_max_region_id = 0;
    const bool use_constraints = chunk_id_ > 0;
    seg_->AssignUniqueRegionIds(use_constraints, max_region_id_, &new_max_region_id);
    max_region_id_ = new_max_region_id;

    const int chunk_size = last_output_frame - curr_chunk_start_ + 1;
    results_.clear();
    overlap_segmentations_.clear();
    const int hierarchy_frame_idx = num_output_frames_;
    for (int frame_idx = curr_chunk_start_; frame_idx <= max_result_frame; ++frame_idx) {
      std::unique_ptr<SegmentationDesc> desc(new SegmentationDesc());
      const bool output_hierarchy = frame_idx == curr_chunk_start_;
      seg_->RetrieveSegmentation3D(frame_idx, output_hierarchy, desc.get());
      desc->chunk_size = chunk_size;
      desc->overlap_start = chunk_size;
      desc->hierarchy_frame_idx = hierarchy_frame_idx;
      if (frame_idx <= last_output_frame) {
        if (frame_idx < last_output_frame) {
          results_.push_back(std::move(desc));
          ++num_output_frames_;
          continue;  // moved-from desc is not buffered (frame_idx < last_output_frame)
        } else {
          results_.push_back(std::unique_ptr<SegmentationDesc>(new SegmentationDesc(*desc)));
        }
        ++num_output_frames_;
      }
      if (frame_idx >= last_output_frame) overlap_segmentations_.push_back(std::move(desc));
    }

    feature_buffer_.erase(feature_buffer_.begin(), feature_buffer_.begin() + last_output_frame);
    if (!flow_buffer_.empty()) {
      flow_buffer_.erase(flow_buffer_.begin(), flow_buffer_.begin() + last_output_frame);
    }
    curr_chunk_start_ = flush ? 0 : 1;
    if (!flush) {
      VSO_CHECK(overlap_frames_ == (int)feature_buffer_.size());
      feature_buffer_[0].reset();
      if (!flow_buffer_.empty()) {
        VSO_CHECK(overlap_frames_ == (int)flow_buffer_.size());
        flow_buffer_[0].reset();
      }
    }
    ++chunk_id_;
  }

  vso_options options_;
  int W_, H_;
  int input_frames_ = 0;
  int chunk_id_ = 0;
  int overlap_frames_ = 2;
  int constraint_frames_ = 1;
  int max_region_id_ = 0;
  int num_output_frames_ = 0;
  std::vector<std::shared_ptr<std::vector<float>>> feature_buffer_;
  std::vector<std::shared_ptr<std::vector<float>>> flow_buffer_;
  int curr_chunk_start_ = 0;
  std::vector<std::unique_ptr<SegmentationDesc>> overlap_segmentations_;
  std::vector<std::unique_ptr<SegmentationDesc>> results_;
  std::unique_ptr<Segmentation> seg_;
  int64_t last_merge_stats_[3] = {0, 0, 0};
  std::vector<int32_t> halo_[2];
  bool pending_import_ = false;
};

}  // namespace vso

// ==========================================================================================
// C interface
// ==========================================================================================
struct vso_stream {
  std::unique_ptr<vso::DenseSegmentation> ds;
  std::vector<std::string> encoded;
};

struct vso_region {
  std::unique_ptr<vso::RegionSegmentation> rs;
  std::vector<std::unique_ptr<vso::SegmentationDesc>> results;
  std::vector<std::string> encoded;
};

struct vso_graph {
  std::unique_ptr<vso::DenseGraph> g;
  vso::RegionInfoList regions;
  vso::RegionInfoPtrMap map;
  int W = 0, H = 0;
};

extern "C" {

void vso_set_threads(int n) { vso::g_threads = n < 1 ? 1 : n; }

void vso_region_default_options(vso_region_options* o) {
  vso::RegionSegOptions d;
  o->min_region_num = d.min_region_num;
  o->max_region_num = d.max_region_num;
  o->level_cutoff_fraction = d.level_cutoff_fraction;
  o->small_region_penalizer = d.small_region_penalizer;
  o->luminance_bins = d.luminance_bins;
  o->color_bins = d.color_bins;
  o->flow_bins = d.flow_bins;
  o->chunk_set_size = d.chunk_set_size;
  o->chunk_set_overlap = d.chunk_set_overlap;
  o->constraint_chunks = d.constraint_chunks;
  o->use_appearance = d.use_appearance;
  o->use_flow = d.use_flow;
  o->use_size_penalizer = d.use_size_penalizer;
  o->compute_vectorization = d.compute_vectorization;
  o->save_descriptors = d.save_descriptors;
}

vso_region* vso_region_create(const vso_region_options* o, int width, int height) {
  vso::RegionSegOptions d;
  d.min_region_num = o->min_region_num;
  d.max_region_num = o->max_region_num;
  d.level_cutoff_fraction = o->level_cutoff_fraction;
  d.small_region_penalizer = o->small_region_penalizer;
  d.luminance_bins = o->luminance_bins;
  d.color_bins = o->color_bins;
  d.flow_bins = o->flow_bins;
  d.chunk_set_size = o->chunk_set_size;
  d.chunk_set_overlap = o->chunk_set_overlap;
  d.constraint_chunks = o->constraint_chunks;
  d.use_appearance = o->use_appearance != 0;
  d.use_flow = o->use_flow != 0;
  d.use_size_penalizer = o->use_size_penalizer != 0;
  d.compute_vectorization = o->compute_vectorization != 0;
  d.save_descriptors = o->save_descriptors != 0;
  vso_region* r = new vso_region;
  r->rs.reset(new vso::RegionSegmentation(d, width, height));
  return r;
}

void vso_region_destroy(vso_region* r) { delete r; }

int vso_region_process_frame(vso_region* r, int flush, const uint8_t* seg_desc, size_t seg_len,
                             const uint8_t* bgr, size_t stride, const float* flow) {
  r->results.clear();
  r->encoded.clear();
  try {
    if (seg_desc) {
      vso::SegmentationDesc d;
      if (!vso::wire::Decode(seg_desc, seg_len, &d)) return -1;
      r->rs->ProcessFrame(flush != 0, &d, bgr, stride, flow, &r->results);
    } else {
      r->rs->ProcessFrame(flush != 0, nullptr, nullptr, 0, nullptr, &r->results);
    }
  } catch (const vso::ReferenceCheckFailed& e) {
    std::fprintf(stderr, "vs_oracle: the reference aborts here: %s\n", e.what());
    return -2;
  }
  for (const auto& d : r->results) r->encoded.push_back(vso::wire::Encode(*d));
  return (int)r->results.size();
}

int vso_region_result_bytes(const vso_region* r, int i, const uint8_t** data, size_t* len) {
  if (i < 0 || i >= (int)r->encoded.size()) return -1;
  *data = reinterpret_cast<const uint8_t*>(r->encoded[(size_t)i].data());
  *len = r->encoded[(size_t)i].size();
  return 0;
}

void vso_bgr_to_lab(const uint8_t* bgr, size_t stride, int width, int height, uint8_t* lab) {
  vso::cvlab::BgrToLab8(bgr, stride, width, height, lab);
}

int vso_vectorize_id_image(const int32_t* ids, int width, int height, const uint8_t** data, size_t* len) {
  // Region2D list as RetrieveSegmentation3D emits it after SortRegions2DById: one region per id,
  // scan intervals in scan order.
  thread_local std::string wire;
  std::map<int, vso::Region2D> by_id;
  for (int y = 0; y < height; ++y) {
    const int32_t* row = ids + (size_t)y * width;
    for (int x = 0; x < width;) {
      int x2 = x;
      while (x2 + 1 < width && row[x2 + 1] == row[x]) ++x2;
      if (row[x] < 0) return -1;
      vso::Region2D& r = by_id[row[x]];
      r.id = row[x];
      r.raster.push_back(vso::ScanInterval{y, x, x2});
      x = x2 + 1;
    }
  }
  vso::SegmentationDesc d;
  d.frame_width = width;
  d.frame_height = height;
  for (auto& kv : by_id) {
    vso::ShapeMomentsFromRasterization(kv.second.raster, &kv.second.shape_moments);
    d.region.push_back(std::move(kv.second));
  }
  vso::ComputeFrameVectorization(&d);
  wire = vso::wire::Encode(d);
  *data = reinterpret_cast<const uint8_t*>(wire.data());
  *len = wire.size();
  return 0;
}

void vso_default_options(vso_options* o) {
  o->presmoothing = 2;
  o->frac_min_region_size = 0.01f;
  o->chunk_size = 20;
  o->chunk_overlap_ratio = 0.2f;
  o->num_constraint_frames = 1;
  o->enforce_n4_connectivity = 1;
  o->enforce_spatial_connectedness = 1;
  o->color_distance = 1;
  o->two_stage_oversegment = 0;
  o->compute_vectorization = 0;
}

vso_stream* vso_stream_create(const vso_options* o, int width, int height) {
  vso_stream* s = new vso_stream;
  s->ds.reset(new vso::DenseSegmentation(*o, width, height));
  return s;
}
void vso_stream_destroy(vso_stream* s) { delete s; }

int vso_stream_process_frame(vso_stream* s, int flush, const uint8_t* bgr, size_t stride,
                             const float* flow, int has_flow_stream) {
  const int n = s->ds->ProcessFrame(flush != 0, bgr, stride, flow, has_flow_stream != 0);
  s->encoded.clear();
  for (const auto& d : s->ds->results()) s->encoded.push_back(vso::wire::Encode(*d));
  return n;
}
int vso_stream_num_results(const vso_stream* s) { return (int)s->ds->results().size(); }
int vso_stream_result_bytes(const vso_stream* s, int i, const uint8_t** data, size_t* len) {
  if (i < 0 || i >= (int)s->encoded.size()) return -1;
  *data = reinterpret_cast<const uint8_t*>(s->encoded[i].data());
  *len = s->encoded[i].size();
  return 0;
}
int vso_stream_result_id_image(const vso_stream* s, int i, int32_t* out) {
  if (i < 0 || i >= (int)s->ds->results().size()) return -1;
  const size_t n = (size_t)s->ds->W() * s->ds->H();
  for (size_t k = 0; k < n; ++k) out[k] = -1;
  vso::SegmentationDescToIdImage(*s->ds->results()[i], s->ds->W(), out);
  return 0;
}
int vso_stream_result_num_regions(const vso_stream* s, int i) {
  if (i < 0 || i >= (int)s->ds->results().size()) return -1;
  return (int)s->ds->results()[i]->region.size();
}
int vso_stream_result_hierarchy_regions(const vso_stream* s, int i) {
  if (i < 0 || i >= (int)s->ds->results().size()) return -1;
  const auto& d = *s->ds->results()[i];
  return d.hierarchy.empty() ? 0 : (int)d.hierarchy[0].region.size();
}
int vso_stream_result_first_region(const vso_stream* s, int i, int* id, float* m) {
  if (i < 0 || i >= (int)s->ds->results().size()) return -1;
  const auto& d = *s->ds->results()[i];
  if (d.region.empty()) return -1;
  const vso::Region2D& r = d.region[0];
  *id = r.id;
  m[0] = r.shape_moments.size;
  m[1] = r.shape_moments.mean_x;
  m[2] = r.shape_moments.mean_y;
  m[3] = r.shape_moments.moment_xx;
  m[4] = r.shape_moments.moment_xy;
  m[5] = r.shape_moments.moment_yy;
  return 0;
}
void vso_stream_last_merge_stats(const vso_stream* s, int64_t* stats3) {
  std::memcpy(stats3, s->ds->last_merge_stats(), 3 * sizeof(int64_t));
}
int vso_stream_last_smoothed(const vso_stream* s, float* out) {
  const float* p = s->ds->last_smoothed();
  if (!p) return -1;
  std::memcpy(out, p, (size_t)s->ds->W() * s->ds->H() * 3 * sizeof(float));
  return 0;
}

int vso_stream_export_halo(const vso_stream* s, int32_t* virt, int32_t* cons, int64_t* scalars4) {
  return s->ds->ExportHalo(virt, cons, scalars4) ? 0 : -1;
}
void vso_stream_import_halo(vso_stream* s, const int32_t* virt, const int32_t* cons,
                            const int64_t* scalars4) {
  s->ds->ImportHalo(virt, cons, scalars4);
}

void vso_preprocess(const uint8_t* bgr, size_t stride, int width, int height, int presmoothing,
                    float* out) {
  vso::PreprocessFeatures(bgr, stride, width, height, presmoothing, out);
}

float vso_bilateral_tables(float min_val, float max_val, float* lut, float* space_w) {
  return vso::BilateralTables((double)min_val, (double)max_val, 3.0f, 0.25f, 3, lut, space_w,
                              nullptr, nullptr, nullptr);
}

static inline uint16_t BucketOf(float w) {
  const float scale = 2048.0f / (1.0f + 1e-6f);
  return (uint16_t)(int)(std::min<float>(2048.0f, w * scale));
}

void vso_spatial_buckets(const float* feat, int W, int H, int l1, uint16_t* out) {
  const size_t n = (size_t)W * H;
  for (size_t i = 0; i < 4 * n; ++i) out[i] = 0xFFFF;
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) {
      const float* a = feat + ((size_t)y * W + x) * 3;
      auto d = [&](int dx, int dy) {
        const float* b = feat + ((size_t)(y + dy) * W + (x + dx)) * 3;
        return l1 ? vso::ColorDiff3L1(a, b) : vso::ColorDiff3L2(a, b);
      };
      const size_t p = (size_t)y * W + x;
      if (x < W - 1) out[0 * n + p] = BucketOf(d(1, 0));
      if (y < H - 1) {
        out[1 * n + p] = BucketOf(d(0, 1));
        if (x > 0) out[2 * n + p] = BucketOf(d(-1, 1));
        if (x < W - 1) out[3 * n + p] = BucketOf(d(1, 1));
      }
    }
  }
}

void vso_temporal_buckets(const float* cur, const float* prev, const float* flow, int W, int H,
                          int l1, uint16_t* out, int32_t* prev_idx) {
  const size_t n = (size_t)W * H;
  for (size_t i = 0; i < 9 * n; ++i) out[i] = 0xFFFF;
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) {
      int px = x, py = y;
      if (flow) {
        const float* f = flow + ((size_t)y * W + x) * 2;
        px = (int)((float)x + f[0]);
        py = (int)((float)y + f[1]);
        px = std::max(0, std::min(W - 1, px));
        py = std::max(0, std::min(H - 1, py));
      }
      const size_t p = (size_t)y * W + x;
      prev_idx[p] = py * W + px;
      const float* a = cur + p * 3;
      int k = 0;
      for (int dy = -1; dy <= 1; ++dy) {
        for (int dx = -1; dx <= 1; ++dx, ++k) {
          if (py + dy < 0 || py + dy >= H || px + dx < 0 || px + dx >= W) continue;
          const float* b = prev + ((size_t)(py + dy) * W + (px + dx)) * 3;
          out[(size_t)k * n + p] = BucketOf(l1 ? vso::ColorDiff3L1(a, b) : vso::ColorDiff3L2(a, b));
        }
      }
    }
  }
}

vso_graph* vso_graph_create(int width, int height, int max_frames, int l1) {
  vso_graph* g = new vso_graph;
  g->g.reset(new vso::DenseGraph(width, height, max_frames, l1 != 0));
  g->W = width;
  g->H = height;
  return g;
}
void vso_graph_destroy(vso_graph* g) { delete g; }
void vso_graph_add_frame(vso_graph* g, const float* feat, const int32_t* constraint_ids) {
  g->g->AddFrame(feat, constraint_ids);
}
void vso_graph_add_virtual_frame(vso_graph* g, const int32_t* constraint_ids) {
  g->g->AddVirtualFrame(constraint_ids);
}
void vso_graph_add_temporal(vso_graph* g, const float* cur, const float* prev, const float* flow,
                            int is_virtual) {
  g->g->AddTemporal(cur, prev, flow, is_virtual != 0);
}
void vso_graph_segment_spatially(vso_graph* g) { g->g->SegmentGraphSpatially(); }
void vso_graph_segment(vso_graph* g, int min_region_size, int force_constraints) {
  g->g->SegmentGraph(min_region_size, force_constraints != 0);
}
void vso_graph_obtain_results(vso_graph* g, const float* const* flows, int enforce_n4,
                              int enforce_spatial_connectedness) {
  std::vector<const float*> fl;
  if (flows) fl.assign(flows, flows + g->g->num_frames());
  g->g->ObtainResults(&g->regions, &g->map, flows ? &fl : nullptr, enforce_n4 != 0,
                      enforce_spatial_connectedness != 0);
  g->g->DetermineNeighborIds(&g->regions, &g->map);
}
int vso_graph_num_regions(const vso_graph* g) { return (int)g->regions.size(); }
int64_t vso_graph_num_neighbor_links(const vso_graph* g) {
  int64_t n = 0;
  for (const auto& r : g->regions) n += (int64_t)r->neighbor_idx.size();
  return n;
}
void vso_graph_node_roots(vso_graph* g, int32_t* out) { g->g->NodeRoots(out); }
void vso_graph_index_image(const vso_graph* g, int t, int32_t* out) {
  const size_t n = (size_t)g->W * g->H;
  for (size_t k = 0; k < n; ++k) out[k] = -1;
  for (const auto& r : g->regions) {
    if (!r->raster) continue;
    for (const auto& slice : *r->raster) {
      if (slice.first != t) continue;
      for (const vso::ScanInterval& s : *slice.second) {
        for (int x = s.left_x; x <= s.right_x; ++x) out[(size_t)s.y * g->W + x] = r->index;
      }
    }
  }
}
void vso_graph_region_sizes(const vso_graph* g, int32_t* sizes, int32_t* constrained) {
  for (size_t i = 0; i < g->regions.size(); ++i) {
    sizes[i] = g->regions[i]->size;
    constrained[i] = g->regions[i]->constrained_id;
  }
}
int64_t vso_graph_get_regions(const vso_graph* g, int32_t* regions5, int32_t* nbr_ptr, int32_t* nbr_idx) {
  int64_t total = 0;
  for (size_t i = 0; i < g->regions.size(); ++i) {
    const vso::RegionInformation& r = *g->regions[i];
    if (regions5) {
      const bool has = r.raster && !r.raster->empty();
      regions5[5 * i + 0] = r.index;
      regions5[5 * i + 1] = r.size;
      regions5[5 * i + 2] = r.constrained_id;
      regions5[5 * i + 3] = has ? r.raster->front().first : -1;
      regions5[5 * i + 4] = has ? r.raster->back().first : -1;
    }
    if (nbr_ptr) nbr_ptr[i] = (int32_t)total;
    if (nbr_idx) {
      for (int n : r.neighbor_idx) nbr_idx[total++] = n;
    } else {
      total += (int64_t)r.neighbor_idx.size();
    }
  }
  if (nbr_ptr) nbr_ptr[g->regions.size()] = (int32_t)total;
  return total;
}
int64_t vso_graph_get_intervals(const vso_graph* g, int frame, int32_t* out4) {
  int64_t n = 0;
  for (const auto& r : g->regions) {
    if (!r->raster) continue;
    for (const auto& slice : *r->raster) {
      if (slice.first != frame) continue;
      for (const vso::ScanInterval& s : *slice.second) {
        if (out4) {
          out4[4 * n + 0] = r->index;
          out4[4 * n + 1] = s.y;
          out4[4 * n + 2] = s.left_x;
          out4[4 * n + 3] = s.right_x;
        }
        ++n;
      }
    }
  }
  return n;
}
void vso_graph_merge_stats(const vso_graph* g, int64_t* stats3) { g->g->MergeStats(stats3); }
void vso_graph_bucket_census(const vso_graph* g, int64_t* out) {
  const auto& c = g->g->census();
  for (size_t b = 0; b < c.size(); ++b) {
    out[b * 7 + 0] = c[b].edges;
    out[b * 7 + 1] = c[b].internal;
    out[b * 7 + 2] = c[b].regular;
    out[b * 7 + 3] = c[b].fail;
    out[b * 7 + 4] = c[b].small;
    out[b * 7 + 5] = c[b].kept;
    out[b * 7 + 6] = c[b].forced;
  }
}

}  // extern "C"
