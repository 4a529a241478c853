// region_segmentation.cpp -- see region_segmentation.h.
//
// Reference behaviour restated (own data layout: plain structs per region, one descriptor record
// instead of a class hierarchy of descriptors / extractors / updaters, index-linked edge buckets):
//   segmentation/region_segmentation.cpp:97-365        chunk sets, overlap, constraints, output
//   segmentation/segmentation.cpp:80-389, 392-773      base level, hierarchy levels, ids, retrieval
//   segmentation/region_segmentation_graph.cpp:45-503  agglomerative clustering
//   segmentation/region_descriptor.cpp:90-145, 376-589 appearance / flow / size-penalizer
//   segmentation/histograms.cpp:104-404, 464-598       sparse colour histogram, flow histogram
//
// Results have to be the reference's, including where they depend on things outside its tree:
// sums over a sparse histogram run in the iteration order of std::unordered_map<int, float>
// (constructed and filled exactly as the reference does it), the skeleton of constrained levels
// is walked in the order of std::unordered_map<int, std::vector<int>>, atan2 / hypot / log are
// libm's double functions.  Parity for this stage is unpinned.
#include "region_segmentation.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <exception>
#include <ctime>
#include <functional>
#include <limits>
#include <thread>
#include <unordered_map>

#include "common.h"

namespace vsg {

// ---------------------------------------------------------------------------------------------
// BGR -> Lab, 8 bit (OpenCV 2.4 color.cpp RGB2Lab_b, srgb, blueIdx 0; cvCbrt of mathfuncs.cpp)
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int kXyzShift = 12, kGammaBits = 3, kLabShift2 = kXyzShift + kGammaBits;
constexpr int kCbrtEntries = 256 * 3 / 2 * (1 << kGammaBits);

float FastCbrt(float value) {   // OpenCV's cvCbrt: exponent / 3 + quartic rational polynomial
  uint32_t bits;
  std::memcpy(&bits, &value, 4);
  const uint32_t ix = bits & 0x7fffffffu, sign = bits & 0x80000000u;
  int ex = (int)(ix >> 23) - 127;
  int shx = ex % 3;
  shx -= shx >= 0 ? 3 : 0;
  ex = (ex - shx) / 3;
  const uint32_t mant = (ix & ((1u << 23) - 1u)) | ((uint32_t)(shx + 127) << 23);
  float frf;
  std::memcpy(&frf, &mant, 4);
  const double fr = frf;   // in [0.125, 1)
  const float r = (float)(((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr +
                             119.1654824285581628956914143) * fr + 13.43250139086239872172837314) * fr +
                           0.1636161226585754240958355063) /
                          ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr +
                             168.5254414101568283957668343) * fr + 33.9905941350215598754191872) * fr + 1.0));
  uint32_t rb;
  std::memcpy(&rb, &r, 4);
  rb = (uint32_t)((int32_t)rb + (ex << 23) + (int32_t)sign) & ((bits << 1) != 0 ? 0xffffffffu : 0u);
  float out;
  std::memcpy(&out, &rb, 4);
  return out;
}

struct LabTables {
  // per input byte and XYZ row: gamma(v) * coefficient, so that a pixel is three adds per row
  int contrib[3][3][256];   // [xyz row][source channel b,g,r][value]
  uint16_t cbrt[kCbrtEntries];
  LabTables() {
    auto sat16 = [](float v) {
      const long iv = std::lrint((double)v);
      return (uint16_t)(iv < 0 ? 0 : (iv > 65535 ? 65535 : iv));
    };
    uint16_t gamma[256];
    for (int i = 0; i < 256; ++i) {
      const float x = i * (1.f / 255.f);
      gamma[i] = sat16(255.f * (1 << kGammaBits) *
                       (x <= 0.04045f ? x * (1.f / 12.92f) : (float)std::pow((double)(x + 0.055) * (1. / 1.055), 2.4)));
    }
    for (int i = 0; i < kCbrtEntries; ++i) {
      const float x = i * (1.f / (255.f * (1 << kGammaBits)));
      cbrt[i] = sat16((1 << kLabShift2) * (x < 0.008856f ? x * 7.787f + 0.13793103448275862f : FastCbrt(x)));
    }
    static const float m[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f,
                               0.072169f, 0.019334f, 0.119193f, 0.950227f};   // sRGB -> XYZ (D65), rows X Y Z, columns R G B
    static const float white[3] = {0.950456f, 1.f, 1.088754f};
    for (int row = 0; row < 3; ++row) {
      const float scale = row == 1 ? (float)(1 << kXyzShift) : (1 << kXyzShift) / white[row];
      const int c_r = (int)std::lrint((double)(m[row * 3 + 0] * scale));
      const int c_g = (int)std::lrint((double)(m[row * 3 + 1] * scale));
      const int c_b = (int)std::lrint((double)(m[row * 3 + 2] * scale));
      for (int v = 0; v < 256; ++v) {
        contrib[row][0][v] = gamma[v] * c_b;   // source byte 0 is blue
        contrib[row][1][v] = gamma[v] * c_g;
        contrib[row][2][v] = gamma[v] * c_r;
      }
    }
  }
};

inline int DescaleBy(int x, int n) { return (x + (1 << (n - 1))) >> n; }
inline uint8_t ClampByte(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

}  // namespace

// VSG_DEBUG_STATS: CPU time of the descriptor passes summed over the host threads, and the longest
// single region per frame (what bounds the frame when the regions are few and large)
static std::atomic<long long> g_cpu_color_us(0), g_cpu_flow_us(0), g_longest_us(0);
static bool g_debug_stats = getenv("VSG_DEBUG_STATS") != nullptr;
static long long ThreadCpuUs() {
  timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return (long long)ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
}

static double NowMs() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// fn(i) for i in [0, n), on up to 16 host threads (32 on hosts with 64 or more: a 4K frame has 24
// descriptor tasks); dynamic, the items are of very different sizes; fn must not throw.
static int ParallelSlots(int hw) { return hw >= 64 ? 32 : 16; }
static int ParallelSlots() { return ParallelSlots((int)std::thread::hardware_concurrency()); }

static void ParallelItems(int n, size_t work, const std::function<void(int, int)>& fn) {
  const int hw = (int)std::thread::hardware_concurrency();
  const size_t min_work = getenv("VSG_PARALLEL_MIN_WORK") ? (size_t)atoll(getenv("VSG_PARALLEL_MIN_WORK")) : 65536;
  const int threads = (int)std::min<size_t>((size_t)std::max(1, std::min(std::max(hw, 2), ParallelSlots(hw))),
                                            std::max<size_t>(1, work / std::max<size_t>(min_work, 1)));
  if (threads <= 1 || n <= 1) {
    for (int i = 0; i < n; ++i) fn(i, 0);
    return;
  }
  std::atomic<int> next(0);
  std::vector<std::thread> pool;
  auto body = [&](int t) {
    for (int i; (i = next.fetch_add(1)) < n;) fn(i, t);
  };
  for (int t = 1; t < threads; ++t) pool.emplace_back(body, t);
  body(0);
  for (std::thread& t : pool) t.join();
}


void BgrToLab8(const uint8_t* src, size_t stride, int W, int H, uint8_t* dst) {
  static const LabTables T;
  const int l_scale = (116 * 255 + 50) / 100;
  const int l_shift = -((16 * 255 * (1 << kLabShift2) + 50) / 100);
  const int band = 32;   // rows per work item
  ParallelItems((H + band - 1) / band, (size_t)W * H, [&](int item, int) {
  for (int y = item * band; y < std::min(H, (item + 1) * band); ++y) {
    const uint8_t* s = src + (size_t)y * stride;
    uint8_t* d = dst + (size_t)y * W * 3;
    for (int x = 0; x < W; ++x, s += 3, d += 3) {
      int f[3];
      for (int row = 0; row < 3; ++row) {
        const int acc = T.contrib[row][0][s[0]] + T.contrib[row][1][s[1]] + T.contrib[row][2][s[2]];
        f[row] = T.cbrt[DescaleBy(acc, kXyzShift)];
      }
      d[0] = ClampByte(DescaleBy(l_scale * f[1] + l_shift, kLabShift2));
      d[1] = ClampByte(DescaleBy(500 * (f[0] - f[1]) + 128 * (1 << kLabShift2), kLabShift2));
      d[2] = ClampByte(DescaleBy(200 * (f[1] - f[2]) + 128 * (1 << kLabShift2), kLabShift2));
    }
  }
  });
}

namespace {

[[noreturn]] void ReferenceAborts(const char* what) {
  Throw(-1 /* VSG_ERR_INVALID */, std::string("the reference aborts on this input (glog CHECK): ") + what);
}

// ---------------------------------------------------------------------------------------------
// Descriptors of a region: sparse Lab histogram, per-frame flow histograms, size penalizer.
// ---------------------------------------------------------------------------------------------
struct ColorHist {
  typedef std::unordered_map<int, float> Bins;
  Bins bins;
  double weight_sum = 0.0;
  bool normalized = false;
  int lum_bins, color_bins;
  ColorHist(int lb, int cb) : bins((size_t)(lb * cb * cb / 10)), lum_bins(lb), color_bins(cb) {}

  void AddLabPixel(const uint8_t* px) {   // AddPixelInterpolated + AddValueInterpolated, weight 1
    const float pos[3] = {(float)px[0] * (1.0f / 255.f) * (lum_bins - 1),
                          (float)px[1] * (1.0f / 255.f) * (color_bins - 1),
                          (float)px[2] * (1.0f / 255.f) * (color_bins - 1)};
    int lo[3], hi[3];
    float w_lo[3], w_hi[3];
    for (int c = 0; c < 3; ++c) {
      lo[c] = (int)pos[c];
      const float frac = pos[c] - (float)lo[c];
      hi[c] = lo[c] + (frac >= 1e-6f);
      w_lo[c] = 1.0f - frac;
      w_hi[c] = frac;
    }
    const int sq = color_bins * color_bins;
    for (int a = 0; a < 2; ++a) {
      const int slice = (a ? hi[0] : lo[0]) * sq;
      const float wa = a ? w_hi[0] : w_lo[0];
      for (int b = 0; b < 2; ++b) {
        const int row = slice + (b ? hi[1] : lo[1]) * color_bins;
        const float wb = b ? w_hi[1] : w_lo[1];
        for (int c = 0; c < 2; ++c) {
          const float value = wa * wb * (c ? w_hi[2] : w_lo[2]) * 1.0f;
          bins[row + (c ? hi[2] : lo[2])] += value;
        }
      }
    }
    weight_sum += 1.0f;
  }
  // The pixels of one region in one frame.  Same sums, bin by bin in pixel order, and the same
  // insertion order of new bins (first touch) as AddLabPixel pixel by pixel -- accumulated in a
  // dense scratch table instead of eight hash lookups per pixel (66 M per 4K frame).
  struct Scratch {
    std::vector<float> acc;
    std::vector<uint32_t> stamp;
    std::vector<int> touched;
    uint32_t epoch = 0;
  };
  void AddLabPixels(const uint8_t* lab, int W, const Raster& raster, Scratch* sc) {
    const size_t total = (size_t)lum_bins * color_bins * color_bins;
    if (sc->acc.size() != total) {
      sc->acc.assign(total, 0.f);
      sc->stamp.assign(total, 0u);
      sc->epoch = 0;
    }
    if (++sc->epoch == 0) {
      std::fill(sc->stamp.begin(), sc->stamp.end(), 0u);
      sc->epoch = 1;
    }
    sc->touched.clear();
    const uint32_t epoch = sc->epoch;
    float* acc = sc->acc.data();
    uint32_t* stamp = sc->stamp.data();
    const int sq = color_bins * color_bins, cb = color_bins;
    const float s0 = (float)(lum_bins - 1), s12 = (float)(color_bins - 1);
    size_t pixels = 0;
    // In batches: first the bins and the eight weights of every pixel (no dependence between
    // pixels: the compiler vectorizes these loops), then the sums, pixel by pixel and bin by bin in
    // the order of AddLabPixel.  Neighbouring pixels mostly fall between the same bins: over such a
    // run the eight sums stay in registers.
    constexpr int kBatch = 128;
    float pos[3][kBatch], frac[3][kBatch], value[8][kBatch];
    int lo[3][kBatch], base[kBatch], steps[kBatch];
    auto touch = [&](int k) {
      if (stamp[k] != epoch) {
        stamp[k] = epoch;
        sc->touched.push_back(k);
        auto it = bins.find(k);
        acc[k] = it != bins.end() ? it->second : 0.0f;
      }
    };
    for (const Interval& iv : raster) {
      for (int x0 = iv.lx; x0 <= iv.rx; x0 += kBatch) {
        const int m = std::min(kBatch, iv.rx - x0 + 1);
        const uint8_t* px = lab + ((size_t)iv.y * W + x0) * 3;
        for (int i = 0; i < m; ++i) {
          pos[0][i] = (float)px[3 * i] * (1.0f / 255.f) * s0;
          pos[1][i] = (float)px[3 * i + 1] * (1.0f / 255.f) * s12;
          pos[2][i] = (float)px[3 * i + 2] * (1.0f / 255.f) * s12;
        }
        for (int c = 0; c < 3; ++c) {
          for (int i = 0; i < m; ++i) {
            const int l = (int)pos[c][i];
            lo[c][i] = l;
            frac[c][i] = pos[c][i] - (float)l;
          }
        }
        for (int i = 0; i < m; ++i) {
          base[i] = lo[0][i] * sq + lo[1][i] * cb + lo[2][i];
          // hi = lo + (frac >= 1e-6) per channel
          steps[i] = (frac[0][i] >= 1e-6f ? 4 : 0) | (frac[1][i] >= 1e-6f ? 2 : 0) | (frac[2][i] >= 1e-6f ? 1 : 0);
        }
        for (int i = 0; i < m; ++i) {
          const float wa[2] = {1.0f - frac[0][i], frac[0][i]};
          const float wb[2] = {1.0f - frac[1][i], frac[1][i]};
          const float wc[2] = {1.0f - frac[2][i], frac[2][i]};
          for (int j = 0; j < 8; ++j) value[j][i] = wa[j >> 2] * wb[(j >> 1) & 1] * wc[j & 1] * 1.0f;
        }
        for (int i = 0; i < m;) {
          int e = i + 1;
          while (e < m && base[e] == base[i] && steps[e] == steps[i]) ++e;
          const int st = steps[i];
          int k[8];
          for (int j = 0; j < 8; ++j) {
            k[j] = base[i] + ((j & 4) && (st & 4) ? sq : 0) + ((j & 2) && (st & 2) ? cb : 0) + ((j & 1) && (st & 1) ? 1 : 0);
          }
          if (st == 7) {   // eight different bins
            for (int j = 0; j < 8; ++j) touch(k[j]);
            float a0 = acc[k[0]], a1 = acc[k[1]], a2 = acc[k[2]], a3 = acc[k[3]];
            float a4 = acc[k[4]], a5 = acc[k[5]], a6 = acc[k[6]], a7 = acc[k[7]];
            for (int q = i; q < e; ++q) {
              a0 += value[0][q];
              a1 += value[1][q];
              a2 += value[2][q];
              a3 += value[3][q];
              a4 += value[4][q];
              a5 += value[5][q];
              a6 += value[6][q];
              a7 += value[7][q];
            }
            acc[k[0]] = a0;
            acc[k[1]] = a1;
            acc[k[2]] = a2;
            acc[k[3]] = a3;
            acc[k[4]] = a4;
            acc[k[5]] = a5;
            acc[k[6]] = a6;
            acc[k[7]] = a7;
          } else {         // a channel exactly on a bin: two weights go to the same bin, one after the other
            for (int q = i; q < e; ++q) {
              for (int j = 0; j < 8; ++j) {
                touch(k[j]);
                acc[k[j]] += value[j][q];
              }
            }
          }
          i = e;
        }
        pixels += (size_t)m;
      }
    }
    for (int k : sc->touched) bins[k] = acc[k];   // (a new bin is inserted here: first-touch order)
    weight_sum += (double)pixels;   // (+= 1.0f per pixel: exact in a double)
  }
  void Normalize() {
    normalized = true;
    if (weight_sum == 0) return;
    const float denom = (float)(1.0f / weight_sum);
    for (auto& b : bins) b.second *= denom;
  }
  void Absorb(const ColorHist& o) {   // MergeWithHistogram
    const double n = weight_sum + o.weight_sum;
    if (n == 0) return;
    const float wl = (float)(weight_sum / n), wr = (float)(o.weight_sum / n);
    weight_sum = n;
    if (!normalized) {
      for (auto& b : bins) {
        auto it = o.bins.find(b.first);
        if (it != o.bins.end()) b.second += it->second;
      }
      for (const auto& ob : o.bins) {
        if (bins.find(ob.first) == bins.end()) bins.insert(ob);
      }
      return;
    }
    double total = 0;
    for (auto& b : bins) {
      auto it = o.bins.find(b.first);
      b.second = it != o.bins.end() ? b.second * wl + it->second * wr : b.second * wl;
      total += b.second;
    }
    for (const auto& ob : o.bins) {
      if (bins.find(ob.first) == bins.end()) total += (bins[ob.first] = ob.second * wr);
    }
    const float denom = (float)(1.0f / total);
    for (auto& b : bins) b.second *= denom;
  }
  float ChiSquare(const ColorHist& o) const {
    auto term = [](float a, float b) -> float {
      const float add = a + b;
      if (std::fabs((double)add) > 1e-12) {
        const float sub = a - b;
        return sub * sub / add;
      }
      return 0.0f;
    };
    double sum = 0;
    for (const auto& b : bins) {
      auto it = o.bins.find(b.first);
      sum += term(b.second, it != o.bins.end() ? it->second : 0.0f);
    }
    for (const auto& ob : o.bins) {
      if (bins.find(ob.first) == bins.end()) sum += term(0, ob.second);
    }
    return (float)(0.5 * sum);
  }
};

struct FlowHist {
  std::vector<float> bins;
  int num_vectors = 0;
  explicit FlowHist(int n) : bins((size_t)n, 0.f) {}
  void Add(float x, float y) {
    const int n = (int)bins.size();
    const float angle = (float)(std::atan2((double)y, (double)x) / (2.0 * M_PI + 1e-4) + 0.5);
    float& b = bins[(size_t)(angle * n)];
    b = (float)((double)b + std::hypot((double)x, (double)y));
    ++num_vectors;
  }
  void NormalizeToOne() {
    float sum = 0;
    for (float v : bins) sum += v;
    if (sum > 0) {
      sum = (float)(1.0 / sum);
      for (float& v : bins) v *= sum;
    }
  }
  void Absorb(const FlowHist& o) {
    const float nl = (float)num_vectors, nr = (float)o.num_vectors;
    if (nl + nr > 0) {
      const float inv = 1.0f / (nl + nr);
      for (size_t i = 0; i < bins.size(); ++i) bins[i] = (bins[i] * nl + o.bins[i] * nr) * inv;
      num_vectors += o.num_vectors;
      NormalizeToOne();
    }
  }
  float ChiSquare(const FlowHist& o) const {
    float sum = 0;
    for (size_t i = 0; i < bins.size(); ++i) {
      const float add = bins[i] + o.bins[i];
      if (add) {
        const float sub = bins[i] - o.bins[i];
        sum += sub * sub / add;
      }
    }
    return (float)(0.5 * sum);
  }
};

// What FlowHist::Add computes from a flow vector -- its bin and its magnitude -- for every pixel of
// a frame, on all host threads: a region's histogram is a sequential float sum over its pixels, a
// frame of a few very large regions would otherwise leave most threads idle while one of them
// evaluates atan2 and hypot 700 K times.  The sums stay sequential, per region (ChunkSet::AddFrame).
// The bin FlowHist::Add puts a vector in -- (size_t)((float)(atan2(y, x) / (2 pi + 1e-4) + 0.5) * n)
// -- from a polynomial arctangent (Abramowitz & Stegun 4.4.49, |error| <= 2e-8 rad) where that
// cannot be wrong: the bin position it gives has to be at least 1e-4 of a bin away from a bin edge,
// 25 times what the polynomial, the float rounding of the angle and of the product can move it
// together.  Returns -1 otherwise (edges, every vector with a zero component -- signed zeros --, zero
// and non-finite vectors: the caller evaluates atan2).
inline int FastFlowBin(float x, float y, int num_bins) {
  const double ax = std::fabs((double)x), ay = std::fabs((double)y);
  const double mx = std::max(ax, ay), mn = std::min(ax, ay);
  if (!(mx > 0) || !(mx <= std::numeric_limits<double>::max())) return -1;
  // An axis-aligned vector has a zero component whose SIGN atan2 honours (atan2(-0.0, x < 0) = -pi,
  // not +pi) and `y < 0` below cannot see: the caller evaluates atan2.
  if (mn == 0) return -1;
  const double z = mn / mx, z2 = z * z;
  double p = 0.0028662257;
  p = p * z2 - 0.0161657367;
  p = p * z2 + 0.0429096138;
  p = p * z2 - 0.0752896400;
  p = p * z2 + 0.1065626393;
  p = p * z2 - 0.1420889944;
  p = p * z2 + 0.1999355085;
  p = p * z2 - 0.3333314528;
  p = p * z2 + 1.0;
  double a = z * p;
  if (ay > ax) a = M_PI / 2 - a;
  if (x < 0) a = M_PI - a;
  if (y < 0) a = -a;
  const double t = (a / (2.0 * M_PI + 1e-4) + 0.5) * num_bins;
  const double cell = std::floor(t), d = t - cell;
  if (d < 1e-4 || d > 1.0 - 1e-4 || cell < 0 || cell >= num_bins) return -1;
  return (int)cell;
}

struct FlowSamples {
  std::vector<uint16_t> bin;
  std::vector<double> magnitude;
  bool valid = false;
  void Compute(const float* flow, int W, int H, int num_bins) {
    bin.resize((size_t)W * H);
    magnitude.resize((size_t)W * H);
    const int band = 16;
    ParallelItems((H + band - 1) / band, (size_t)W * H, [&](int item, int) {
      uint32_t last_x = 0, last_y = 0;
      uint16_t last_bin = 0;
      double last_mag = 0;
      bool have_last = false;
      for (int y = item * band; y < std::min(H, (item + 1) * band); ++y) {
        const float* p = flow + (size_t)y * W * 2;
        uint16_t* b = bin.data() + (size_t)y * W;
        double* m = magnitude.data() + (size_t)y * W;
        for (int x = 0; x < W; ++x, p += 2) {
          uint32_t ux, uy;
          std::memcpy(&ux, p, 4);
          std::memcpy(&uy, p + 1, 4);
          if (!have_last || ux != last_x || uy != last_y) {   // (the same bits give the same values)
            const int fast = FastFlowBin(p[0], p[1], num_bins);
            if (fast >= 0) {
              last_bin = (uint16_t)fast;
            } else {
              const float angle = (float)(std::atan2((double)p[1], (double)p[0]) / (2.0 * M_PI + 1e-4) + 0.5);
              last_bin = (uint16_t)(size_t)(angle * num_bins);
            }
            last_mag = std::hypot((double)p[0], (double)p[1]);
            last_x = ux;
            last_y = uy;
            have_last = true;
          }
          b[x] = last_bin;
          m[x] = last_mag;
        }
      }
    });
    valid = true;
  }
};

// FlowHist::Add for every pixel of a raster, in pixel order, from the samples of the frame; the sum
// of a run of one bin stays in a register.
void AccumulateFlow(const FlowSamples& samples, int W, const Raster& raster, FlowHist* F) {
  for (const Interval& iv : raster) {
    const size_t at = (size_t)iv.y * W + iv.lx;
    const uint16_t* b = samples.bin.data() + at;
    const double* m = samples.magnitude.data() + at;
    const int len = iv.rx - iv.lx + 1;
    for (int x = 0; x < len;) {
      const int bin = b[x];
      float sum = F->bins[(size_t)bin];
      do {
        sum = (float)((double)sum + m[x]);
        ++x;
      } while (x < len && b[x] == bin);
      F->bins[(size_t)bin] = sum;
    }
    F->num_vectors += len;
  }
}

struct Descriptors {
  bool present = false;                          // false: a fresh super-region before its first merge
  std::unique_ptr<ColorHist> color;              // use_appearance
  bool color_done = false;
  std::vector<std::unique_ptr<FlowHist>> flow;   // use_flow: histogram of frame flow_start + i (or null)
  int flow_start = -1;
  bool flow_done = false;
  float inv_median_size = 1.0f;                  // use_size_penalizer (set per level)

  void CloneFrom(const Descriptors& o) {
    present = true;
    color.reset(o.color ? new ColorHist(*o.color) : nullptr);
    color_done = o.color_done;
    flow.clear();
    for (const auto& h : o.flow) flow.emplace_back(h ? new FlowHist(*h) : nullptr);
    flow_start = o.flow_start;
    flow_done = o.flow_done;
    inv_median_size = o.inv_median_size;
  }
  int flow_end() const { return flow_start + (int)flow.size(); }
  void AbsorbFlow(const Descriptors& o) {   // FlowDescriptor::MergeWithDescriptor
    while (flow_start > o.flow_start) {
      flow.emplace(flow.begin(), nullptr);
      --flow_start;
    }
    while (o.flow_end() > flow_end()) flow.emplace_back(nullptr);
    for (int k = flow_start; k < flow_end(); ++k) {
      const int li = k - flow_start, ri = k - o.flow_start;
      if (ri < 0 || ri >= (int)o.flow.size() || !o.flow[(size_t)ri]) continue;
      if (!flow[(size_t)li]) flow[(size_t)li].reset(new FlowHist(*o.flow[(size_t)ri]));
      else flow[(size_t)li]->Absorb(*o.flow[(size_t)ri]);
    }
    while (!flow.empty() && !flow.front()) {
      flow.erase(flow.begin());
      ++flow_start;
    }
    while (!flow.empty() && !flow.back()) flow.pop_back();
  }
};

// RegionInformation (segmentation_common.h:39-116).
struct Node {
  int index = -1, size = 0, parent_idx = -1;
  bool removed = false;
  std::vector<int> neighbors;
  std::unique_ptr<Raster3D> raster;
  std::unique_ptr<std::vector<int>> children;
  Node* counterpart = nullptr;
  int constrained_id = -1, region_id = -1;
  std::unique_ptr<std::vector<int>> counterpart_ids;
  Descriptors desc;
};
typedef std::vector<std::unique_ptr<Node>> Level;

struct Setup {   // what of the options the descriptor code needs
  bool appearance, flow, size_penalizer;
  float penalizer;
  int lum_bins, color_bins, flow_bins;
};

// MergeDescriptorsFrom: clone into an empty record, merge otherwise.
void TakeDescriptors(Node* dst, const Node& src, const Setup& S) {
  if (!dst->desc.present) {
    dst->desc.CloneFrom(src.desc);
    return;
  }
  if (S.appearance) dst->desc.color->Absorb(*src.desc.color);
  if (S.flow) dst->desc.AbsorbFlow(src.desc);
}

bool InsertSorted(int v, std::vector<int>* a) {
  auto it = std::lower_bound(a->begin(), a->end(), v);
  if (it != a->end() && *it == v) return false;
  a->insert(it, v);
  return true;
}

// Region distance: SquaredORDistance over the appearance / flow distances, scaled by the size
// penalizer (region_descriptor.h:181-217, region_descriptor.cpp:376-388, 460-493).
float NodeDistance(const Node& a, const Node& b, const Setup& S) {
  float or_product = 1.0f;
  if (S.appearance) or_product *= (1.0f - a.desc.color->ChiSquare(*b.desc.color));
  if (S.flow) {
    const Descriptors &l = a.desc, &r = b.desc;
    const int begin = std::max(l.flow_start, r.flow_start), end = std::min(l.flow_end(), r.flow_end());
    double sum = 0, weight_sum = 0;
    for (int f = begin; f < end; ++f) {
      const FlowHist* hl = l.flow[(size_t)(f - l.flow_start)].get();
      const FlowHist* hr = r.flow[(size_t)(f - r.flow_start)].get();
      if (!hl || !hr) continue;
      const float weight = (float)std::min(hl->num_vectors, hr->num_vectors);
      sum += hl->ChiSquare(*hr) * weight;
      weight_sum += weight;
    }
    const float d = weight_sum > 0 ? (float)(sum / weight_sum) : 0.f;
    or_product *= (1.0f - d);
  }
  const float or_dist = 1.0f - or_product;
  const float base = or_dist * or_dist;
  if (!S.size_penalizer) return base;
  const int min_sz = std::min(a.size, b.size);
  const float scale = std::min(
      1.0f, (float)(1.0f + S.penalizer * std::log((double)(min_sz * a.desc.inv_median_size)) / std::log(2.0)));
  return std::max(0.f, std::min(1.f, base * scale));
}

void MergeRaster3D(const Raster3D& a, const Raster3D& b, Raster3D* out) {   // segmentation_util.cpp:607-642
  size_t i = 0, j = 0;
  while (i < a.size() || j < b.size()) {
    const int fa = i < a.size() ? a[i].frame : std::numeric_limits<int>::max();
    const int fb = j < b.size() ? b[j].frame : std::numeric_limits<int>::max();
    if (fa < fb) {
      out->push_back(a[i++]);
    } else if (fb < fa) {
      out->push_back(b[j++]);
    } else {
      out->push_back(RasterSlice{fa, Raster()});
      MergeRasters(a[i].raster, b[j].raster, &out->back().raster);
      ++i;
      ++j;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Agglomerative clustering of one hierarchy level (region_segmentation_graph.cpp).
// Edge buckets are index-linked lists over one edge array: append at the tail, unlink anywhere,
// take the head -- the operations (and therefore the order) of the reference's std::list buckets.
// ---------------------------------------------------------------------------------------------
typedef std::unordered_map<uint64_t, float> WeightMap;   // (lookups only: any map does)
inline uint64_t PairKey(int a, int b) {
  if (a > b) std::swap(a, b);
  return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
}

class Clustering {
 public:
  Clustering(int num_buckets, const Setup& S) : S_(S), num_buckets_(num_buckets), scale_(num_buckets * (1.0f / 1.0f)) {
    head_.assign((size_t)num_buckets + 1, -1);
    tail_.assign((size_t)num_buckets + 1, -1);
  }

  // AddRegionEdgesImpl (+ the skeleton of AddRegionEdgesConstrained).
  void Build(const Level& level, const std::vector<int>& constraints, const WeightMap* cached,
             const std::unordered_map<int, std::vector<int>>* skeleton) {
    VSG_REQUIRE(!level.empty(), -4, "empty hierarchy level");
    // (all regions before the first edge: AddEdge tests both ends' constraints.  The reference
    // interleaves the two and reads the later region out of merely reserved vector storage --
    // undefined behaviour for a constrained region; the intent is what is implemented here.)
    nodes_.reserve(level.size());
    for (size_t i = 0; i < level.size(); ++i) {
      VSG_REQUIRE(level[i]->index == (int)i, -4, "level out of order");
      nodes_.push_back(GNode{(int)i, constraints[i], 1, level[i].get(), nullptr});
    }
    for (size_t i = 0; i < level.size(); ++i) {
      const Node& n = *level[i];
      for (int nb : n.neighbors) {
        if (pos_.find(PairKey((int)i, nb)) != pos_.end()) continue;
        float w;
        if (cached) {
          auto it = cached->find(PairKey((int)i, nb));
          VSG_REQUIRE(it != cached->end(), -4, "edge weight of the level below is missing");
          w = it->second;
        } else {
          w = NodeDistance(n, *level[(size_t)nb], S_);
        }
        AddEdge((int)i, nb, w);
      }
    }
    if (skeleton) {   // regions of one constraint are chained by virtual edges (weight 2 -> last bucket)
      for (const auto& entry : *skeleton) {
        for (size_t k = 1; k < entry.second.size(); ++k) AddEdge(entry.second[k - 1], entry.second[k], 2.0f);
      }
    }
  }

  // SegmentGraph: lowest-cost merges until the level holds about cutoff_fraction of its regions.
  int Run(bool merge_rasters, float cutoff_fraction) {
    merge_rasters_ = merge_rasters;
    VSG_REQUIRE(cutoff_fraction > 0 && cutoff_fraction <= 1, -1, "cutoff fraction outside (0, 1]");
    int budget = (int)(nodes_.size() * (1.0f - cutoff_fraction));
    budget -= (int)(BucketSize(num_buckets_) * cutoff_fraction);
    budget = std::min<int>(budget, (int)nodes_.size() - 1);
    int lowest = 0;
    while (lowest < num_buckets_ && head_[(size_t)lowest] < 0) ++lowest;
    int merges = 0;
    for (int m = 0; m < budget && lowest < num_buckets_; ++m) {
      for (bool merged = false; !merged;) {
        int e = head_[(size_t)lowest];
        GNode* a = Find(edges_[(size_t)e].a);
        GNode* b = Find(edges_[(size_t)e].b);
        if (!Mergeable(*a, *b)) {
          Unlink(e);   // stays in the position map, marked as not listed
          e = head_[(size_t)lowest];
        } else {
          const int min_bucket = (int)(Merge(a, b) * scale_);
          ++merges;
          if (min_bucket < lowest) {
            lowest = min_bucket;
            break;
          }
          e = head_[(size_t)lowest];
          merged = true;
        }
        if (e < 0) {
          do {
            ++lowest;
          } while (lowest < num_buckets_ && head_[(size_t)lowest] < 0);
          if (lowest >= num_buckets_) break;
        }
      }
    }
    // forced merges along the virtual edges: the last bucket in list order, including what a merge
    // appends to it while it is walked (the reference iterates the live list)
    for (int e = head_[(size_t)num_buckets_]; e >= 0; e = edges_[(size_t)e].next) {
      GNode* a = Find(edges_[(size_t)e].a);
      GNode* b = Find(edges_[(size_t)e].b);
      if (a == b) continue;
      if (!(a->constraint == b->constraint && a->constraint >= 0)) {
        // region_segmentation_graph.cpp:165.  Reachable from plain input: a distance of exactly
        // 1.0 falls into the bucket of the virtual edges ((int)(1.0f * 2048) == num_buckets).
        ReferenceAborts("RegionAgglomerationGraph::SegmentGraph: forced merge of regions without a common constraint "
                        "(two neighbouring regions at distance exactly 1.0)");
      }
      Merge(a, b);
      ++merges;
    }
    return merges;
  }

  // ObtainSegmentationResult: the next level's regions (first-touch order over the children),
  // parent / child links, mapped neighbours, and the edge weights between the results.
  void Collect(Level* below, Level* above, WeightMap* weights) {
    std::unordered_map<int, Node*> made;
    std::vector<int> rep_of;
    for (size_t c = 0; c < below->size(); ++c) {
      GNode* g = Find((int)c);
      auto it = made.find(g->id);
      if (it == made.end()) {
        if (g->info != g->merged.get()) {   // never merged: a copy without ids / counterparts
          std::unique_ptr<Node> copy(new Node());
          copy->size = g->info->size;
          copy->neighbors = g->info->neighbors;
          TakeDescriptors(copy.get(), *g->info, S_);
          if (merge_rasters_) copy->raster.reset(new Raster3D(*g->info->raster));
          g->merged = std::move(copy);
          g->info = g->merged.get();
        }
        Node* n = g->merged.get();
        n->index = (int)above->size();
        n->constrained_id = g->constraint;
        n->children.reset(new std::vector<int>);
        it = made.emplace(g->id, n).first;
        above->push_back(std::move(g->merged));
        rep_of.push_back(g->id);
      }
      it->second->children->push_back((int)c);
      (*below)[c]->parent_idx = it->second->index;
    }
    weights->clear();
    const float inv_scale = 1.0f / scale_;
    for (auto& n : *above) {
      std::vector<int> mapped;
      for (int nb : n->neighbors) {
        const GNode* g = Find(nb);
        const int nb_index = g->info->index;
        VSG_REQUIRE(nb_index >= 0, -4, "neighbour without a result");
        auto p = pos_.find(PairKey(rep_of[(size_t)n->index], g->id));
        const int bucket = p != pos_.end() ? edges_[(size_t)p->second].bucket : -1;   // operator[]: default -1
        (*weights)[PairKey(n->index, nb_index)] = inv_scale * bucket;
        InsertSorted(nb_index, &mapped);
      }
      n->neighbors.swap(mapped);
    }
  }

 private:
  struct GNode {
    int id, constraint, sz;
    const Node* info;
    std::unique_ptr<Node> merged;
  };
  struct EdgeRec {
    int a, b, bucket, prev, next;
    bool listed;
  };

  static bool Mergeable(const GNode& a, const GNode& b) {
    return a.constraint < 0 || b.constraint < 0 || a.constraint == b.constraint;
  }
  GNode* Find(int i) {
    int root = i;
    while (nodes_[(size_t)root].id != root) root = nodes_[(size_t)root].id;
    while (nodes_[(size_t)i].id != root) {
      const int next = nodes_[(size_t)i].id;
      nodes_[(size_t)i].id = root;
      i = next;
    }
    return &nodes_[(size_t)root];
  }
  int BucketSize(int b) const {
    int n = 0;
    for (int e = head_[(size_t)b]; e >= 0; e = edges_[(size_t)e].next) ++n;
    return n;
  }
  void Unlink(int e) {
    EdgeRec& r = edges_[(size_t)e];
    if (r.prev >= 0) edges_[(size_t)r.prev].next = r.next; else head_[(size_t)r.bucket] = r.next;
    if (r.next >= 0) edges_[(size_t)r.next].prev = r.prev; else tail_[(size_t)r.bucket] = r.prev;
    r.listed = false;
    r.prev = r.next = -1;
  }
  bool AddEdge(int a, int b, float weight) {
    const int bucket = std::min(num_buckets_, (int)(weight * scale_));
    if (bucket < 0) {
      // (an edge weight cached from the level below as inv_scale * -1: the reference indexes its
      // bucket array with it -- undefined behaviour; it only arises after a distance of exactly 1.0)
      ReferenceAborts("RegionAgglomerationGraph::AddEdge: negative bucket");
    }
    const bool mergeable = Mergeable(nodes_[(size_t)a], nodes_[(size_t)b]);
    const int e = (int)edges_.size();
    edges_.push_back(EdgeRec{std::min(a, b), std::max(a, b), bucket, -1, -1, false});
    if (mergeable) {
      EdgeRec& r = edges_.back();
      r.prev = tail_[(size_t)bucket];
      if (r.prev >= 0) edges_[(size_t)r.prev].next = e; else head_[(size_t)bucket] = e;
      tail_[(size_t)bucket] = e;
      r.listed = true;
    }
    if (bucket != num_buckets_) {
      const bool fresh = pos_.emplace(PairKey(a, b), e).second;
      if (!fresh) ReferenceAborts("RegionAgglomerationGraph::AddEdge: edge exists");
    } else if (!mergeable) {
      ReferenceAborts("RegionAgglomerationGraph::AddEdge: unmergeable virtual edge");
    }
    return mergeable;
  }
  void DropEdgesOf(int region, const std::vector<int>& neighbors, int other, std::vector<int>* kept) {
    for (int n : neighbors) {
      const int rep = Find(n)->id;
      auto p = pos_.find(PairKey(region, rep));
      if (p == pos_.end()) continue;
      if (edges_[(size_t)p->second].listed) Unlink(p->second);
      pos_.erase(p);
      if (rep != other) InsertSorted(rep, kept);
    }
  }
  // MergeRegions: returns the smallest weight among the re-inserted mergeable edges.
  float Merge(GNode* a, GNode* b) {
    const Node& ia = *a->info;
    const Node& ib = *b->info;
    const int id_a = a->id, id_b = b->id;
    std::unique_ptr<Node> fresh(new Node());
    DropEdgesOf(id_a, ia.neighbors, id_b, &fresh->neighbors);
    DropEdgesOf(id_b, ib.neighbors, id_a, &fresh->neighbors);
    GNode* keep = a->sz > b->sz ? a : b;
    keep->sz = a->sz + b->sz;
    a->id = keep->id;
    b->id = keep->id;
    keep->constraint = std::max(a->constraint, b->constraint);
    fresh->size = ia.size + ib.size;
    TakeDescriptors(fresh.get(), ia, S_);
    TakeDescriptors(fresh.get(), ib, S_);
    if (merge_rasters_) {
      fresh->raster.reset(new Raster3D());
      MergeRaster3D(*ia.raster, *ib.raster, fresh->raster.get());
    }
    std::vector<float> dist(fresh->neighbors.size());
    for (size_t k = 0; k < dist.size(); ++k) {
      dist[k] = NodeDistance(*fresh, *nodes_[(size_t)fresh->neighbors[k]].info, S_);
    }
    float lowest = 1.e6f;
    for (size_t k = 0; k < dist.size(); ++k) {
      if (AddEdge(keep->id, fresh->neighbors[k], dist[k])) lowest = std::min(lowest, dist[k]);
    }
    keep->merged = std::move(fresh);
    keep->info = keep->merged.get();
    return lowest;
  }

  Setup S_;
  int num_buckets_;
  float scale_;
  bool merge_rasters_ = false;
  std::vector<GNode> nodes_;
  std::vector<EdgeRec> edges_;
  std::vector<int> head_, tail_;
  std::unordered_map<uint64_t, int> pos_;   // edge -> record (lookups only)
};

// ---------------------------------------------------------------------------------------------
// One chunk set: base level from the over-segmentation, hierarchy, ids, retrieval
// (Segmentation, hierarchical half).
// ---------------------------------------------------------------------------------------------
class ChunkSet {
 public:
  ChunkSet(const RegionSegOptions& o, const Setup& S, int W, int H, int chunk_set_id)
      : o_(o), S_(S), W_(W), H_(H), id_(chunk_set_id) {}

  int frames() const { return frames_; }
  int levels() const { return (int)levels_.size(); }

  // InitializeBaseHierarchyLevel.  in: ids -> nodes of the previous chunk set (counterparts);
  // out: ids -> nodes of this one.
  void AddBaseLevel(const std::vector<CompoundOut>& regions, const std::unordered_map<int, Node*>* in,
                    std::unordered_map<int, Node*>* out) {
    if (levels_.size() != 1) {
      levels_.clear();
      levels_.emplace_back(new Level());
    }
    if (out) out->clear();
    Level& base = *levels_[0];
    for (const CompoundOut& r : regions) {
      auto it = by_id_.find(r.id);
      Node* n;
      if (it == by_id_.end()) {
        base.emplace_back(new Node());
        n = base.back().get();
        n->index = (int)base.size() - 1;
        n->size = r.size;
        n->raster.reset(new Raster3D);
        n->desc.present = true;
        if (S_.appearance) n->desc.color.reset(new ColorHist(S_.lum_bins, S_.color_bins));
        if (in) {
          auto cp = in->find(r.id);
          if (cp != in->end()) n->counterpart = cp->second;
        }
        by_id_.emplace(r.id, n);
      } else {
        n = it->second;
        n->size += r.size;
      }
      if (out) (*out)[r.id] = n;
    }
    for (const CompoundOut& r : regions) {
      Node* n = by_id_.at(r.id);
      for (int nb : r.neighbor_ids) {
        auto it = by_id_.find(nb);
        VSG_REQUIRE(it != by_id_.end(), -1, "hierarchy(0) names a neighbour that is not a region of the chunk");
        InsertSorted(it->second->index, &n->neighbors);
      }
    }
  }

  // AddOverSegmentation: rasters and descriptor samples of one frame.  The regions of a frame are
  // independent (one node each): their pixels are visited on several host threads.
  void AddFrame(const SegDesc& d, const uint8_t* lab, const float* flow, const FlowSamples* samples) {
    std::vector<Node*> nodes;
    nodes.reserve(d.regions.size());
    size_t pixels = 0;
    for (const Region2DOut& r : d.regions) {
      auto it = by_id_.find(r.id);
      VSG_REQUIRE(it != by_id_.end(), -1, "Region2D without a CompoundRegion in hierarchy(0)");
      Node* n = it->second;
      VSG_REQUIRE(n->raster->empty() || n->raster->back().frame < frames_, -1, "rasterization slices out of order");
      n->raster->push_back(RasterSlice{frames_, r.raster});
      nodes.push_back(n);
      for (const Interval& iv : r.raster) pixels += (size_t)(iv.rx - iv.lx + 1);
      if (S_.flow && flow) {
        Descriptors& D = n->desc;
        if (D.flow_start < 0) D.flow_start = frames_;
        const int fi = frames_ - D.flow_start;
        while (fi >= (int)D.flow.size()) D.flow.emplace_back(nullptr);
        if (!D.flow[(size_t)fi]) D.flow[(size_t)fi].reset(new FlowHist(S_.flow_bins));
      }
    }
    if (scratch_.size() < (size_t)ParallelSlots()) scratch_.resize((size_t)ParallelSlots());
    const int frame = frames_;
    std::atomic<long long> longest_us(0);
    // One task per region and descriptor, the largest first: a frame of a few very large regions
    // is bounded by its longest task.
    struct Task {
      int region;
      bool flow;
      size_t cost;
    };
    std::vector<Task> tasks;
    for (int i = 0; i < (int)nodes.size(); ++i) {
      size_t px = 0;
      for (const Interval& iv : d.regions[(size_t)i].raster) px += (size_t)(iv.rx - iv.lx + 1);
      if (S_.appearance) tasks.push_back(Task{i, false, 2 * px});
      if (S_.flow && flow) tasks.push_back(Task{i, true, px});
    }
    std::stable_sort(tasks.begin(), tasks.end(), [](const Task& a, const Task& b) { return a.cost > b.cost; });
    ParallelItems((int)tasks.size(), pixels, [&](int ti, int t) {
      const Task& task = tasks[(size_t)ti];
      const Region2DOut& r = d.regions[(size_t)task.region];
      Node* n = nodes[(size_t)task.region];
      const long long c0 = g_debug_stats ? ThreadCpuUs() : 0;
      if (!task.flow) {
        n->desc.color->AddLabPixels(lab, W_, r.raster, &scratch_[(size_t)t]);
      } else {
        AccumulateFlow(*samples, W_, r.raster, n->desc.flow[(size_t)(frame - n->desc.flow_start)].get());
      }
      if (g_debug_stats) {
        const long long c2 = ThreadCpuUs();
        (task.flow ? g_cpu_flow_us : g_cpu_color_us) += c2 - c0;
        long long longest = longest_us.load();
        while (c2 - c0 > longest && !longest_us.compare_exchange_weak(longest, c2 - c0)) {}
      }
    });
    g_longest_us += longest_us.load();
    ++frames_;
  }

  // PullCounterpartSegmentationResult.
  void PullConstraints(const ChunkSet& prev) {
    const int n_levels = prev.levels();
    for (auto& n : *levels_[0]) {
      if (!n->counterpart) continue;
      n->constrained_id = n->counterpart->region_id;
      std::unique_ptr<std::vector<int>> ids(new std::vector<int>((size_t)(n_levels - 1)));
      int idx = n->counterpart->parent_idx;
      for (int l = 1; l < n_levels; ++l) {
        const Node& up = *(*prev.levels_[(size_t)l])[(size_t)idx];
        (*ids)[(size_t)(l - 1)] = up.region_id;
        idx = up.parent_idx;
      }
      n->counterpart_ids = std::move(ids);
    }
    constrained_ = true;
  }

  // RunHierarchicalSegmentation(distance, enforce_max_region_num = true).
  void BuildHierarchy() {
    VSG_REQUIRE(!levels_.empty(), -3, "no base hierarchy level");
    for (auto& n : *levels_[0]) {
      Descriptors& D = n->desc;
      if (S_.appearance && !D.color_done) {
        D.color->Normalize();
        D.color_done = true;
      }
      if (S_.flow && !D.flow_done) {
        for (auto& h : D.flow) {
          if (h) h->NormalizeToOne();
        }
        D.flow_done = true;
      }
    }
    int level = 0;
    int count = (int)levels_[0]->size();
    WeightMap weights;
    while (count > o_.min_region_num) {
      Level& cur = *levels_[(size_t)level];
      if (S_.size_penalizer && !cur.empty()) {   // RegionSizePenalizerUpdater: median region size
        std::vector<int> sizes;
        sizes.reserve(cur.size());
        for (const auto& n : cur) sizes.push_back(n->size);
        auto mid = sizes.begin() + sizes.size() / 2;
        std::nth_element(sizes.begin(), mid, sizes.end());
        const float inv = *mid > 0 ? 1.0f / *mid : 1.f;
        for (auto& n : cur) n->desc.inv_median_size = inv;
      }
      Clustering graph(2048, S_);   // SegmentationOptions::num_domain_buckets
      if (constrained_) {
        std::vector<int> constraints;
        std::unordered_map<int, std::vector<int>> skeleton;
        Constraints(level, &constraints, &skeleton);
        graph.Build(cur, constraints, level == 0 ? nullptr : &weights, &skeleton);
      } else {
        graph.Build(cur, std::vector<int>(cur.size(), -1), level == 0 ? nullptr : &weights, nullptr);
      }
      if (level == 0) {
        const float cutoff = std::min(1.0f, o_.max_region_num * (1.0f / levels_[0]->size()));
        graph.Run(true, cutoff);
      } else if (!graph.Run(false, o_.level_cutoff_fraction)) {
        break;   // no merge possible
      }
      levels_.emplace_back(new Level());
      graph.Collect(&cur, levels_.back().get(), &weights);
      count = (int)levels_[(size_t)level]->size();
      ++level;
    }
  }

  void ClipToFrames(int keep_until, int area_until) {   // Constrain... + AdjustRegionArea...ToFrameInterval
    for (auto& n : *levels_[0]) {
      if (!n->raster || n->raster->empty() || n->raster->front().frame >= keep_until || n->raster->back().frame < 0) {
        n->removed = true;
      }
    }
    for (size_t l = 1; l < levels_.size(); ++l) {
      for (auto& n : *levels_[l]) {
        bool all_gone = true;
        for (int c : *n->children) {
          if (!(*levels_[l - 1])[(size_t)c]->removed) {
            all_gone = false;
            break;
          }
        }
        n->removed = all_gone;
      }
    }
    std::vector<int> below;
    for (size_t l = 0; l < levels_.size(); ++l) {
      std::vector<int> delta(levels_[l]->size(), 0);
      for (auto& n : *levels_[l]) {
        int d = 0;
        if (l == 0) {
          if (!n->raster) continue;
          for (const RasterSlice& s : *n->raster) {
            if (s.frame < 0 || s.frame >= area_until) d -= RasterArea(s.raster);
          }
        } else {
          for (int c : *n->children) d += below[(size_t)c];
        }
        n->size += d;
        delta[(size_t)n->index] = d;
      }
      below.swap(delta);
    }
  }

  void AssignIds(bool use_constraints, const std::vector<int>& offsets, std::vector<int>* next_offsets) {
    sorted_output_ = use_constraints;
    VSG_REQUIRE(offsets.size() >= levels_.size() && next_offsets->size() >= levels_.size(), -4, "id offsets");
    for (size_t l = 0; l < levels_.size(); ++l) {
      int max_id = -1;
      for (auto& n : *levels_[l]) {
        n->region_id = (use_constraints && n->constrained_id >= 0) ? n->constrained_id : n->index + offsets[l];
        max_id = std::max(max_id, n->region_id);
      }
      (*next_offsets)[l] = std::max(offsets[l], max_id + 1);
    }
  }

  void DropBaseLevel() {   // DiscardBottomLevel
    if (levels_.size() < 2) return;
    for (auto& n : *levels_[1]) n->children.reset();
    levels_.erase(levels_.begin());
  }

  void Retrieve(int frame, bool with_hierarchy, SegDesc* d) const {   // RetrieveSegmentation3D
    d->frame_width = W_;
    d->frame_height = H_;
    d->chunk_id = id_;
    d->connectedness = 1;   // enforce_n4_connectivity keeps its default
    for (const auto& n : *levels_[0]) {
      if (!n->raster) continue;
      auto it = std::lower_bound(n->raster->begin(), n->raster->end(), frame,
                                 [](const RasterSlice& s, int f) { return s.frame < f; });
      if (it == n->raster->end() || it->frame != frame) continue;
      VSG_REQUIRE(!it->raster.empty(), -4, "empty rasterization slice");
      d->regions.emplace_back();
      Region2DOut& r = d->regions.back();
      r.id = n->region_id;
      r.raster = it->raster;
      MomentsFromRaster(r.raster, &r.moments);
    }
    auto by_id = [](const auto& a, const auto& b) { return a.id < b.id; };
    if (sorted_output_) std::sort(d->regions.begin(), d->regions.end(), by_id);
    if (with_hierarchy && o_.save_descriptors) {
      // segmentation.cpp:490-501: one RegionFeatures per region that is not flagged for removal, in
      // list order.  The descriptors of this path add no extension to it (AddToRegionFeatures is
      // empty for the appearance and the flow descriptor, region_descriptor.cpp:137-138, .h:382).
      for (const auto& n : *levels_[0]) {
        if (!n->removed) d->feature_ids.push_back((uint32_t)n->region_id);
      }
    }
    if (with_hierarchy) {
      d->has_hierarchy = true;
      std::vector<std::pair<int, int>> span_below, span;
      for (size_t l = 0; l < levels_.size(); ++l) {
        std::vector<CompoundOut>* out = l == 0 ? &d->hierarchy0 : (d->upper_levels.emplace_back(), &d->upper_levels.back());
        span.assign(levels_[l]->size(), std::make_pair(0, 0));
        for (const auto& np : *levels_[l]) {
          const Node& n = *np;
          if (n.removed) continue;
          out->emplace_back();
          CompoundOut& c = out->back();
          c.id = n.region_id;
          c.size = n.size;
          for (int nb : n.neighbors) {
            const Node& o = *(*levels_[l])[(size_t)nb];
            if (!o.removed) c.neighbor_ids.push_back(o.region_id);
          }
          if (sorted_output_) std::sort(c.neighbor_ids.begin(), c.neighbor_ids.end());
          if (l + 1 < levels_.size()) {
            c.has_parent = true;
            c.parent_id = (*levels_[l + 1])[(size_t)n.parent_idx]->region_id;
          }
          int lo = std::numeric_limits<int>::max(), hi = 0;
          if (l > 0) {
            VSG_REQUIRE(n.children != nullptr, -4, "super-region without children");
            for (int ch : *n.children) {
              const Node& child = *(*levels_[l - 1])[(size_t)ch];
              if (child.removed) continue;
              c.child_ids.push_back(child.region_id);
              lo = std::min(lo, span_below[(size_t)ch].first);
              hi = std::max(hi, span_below[(size_t)ch].second);
            }
            if (sorted_output_) std::sort(c.child_ids.begin(), c.child_ids.end());
          } else {
            VSG_REQUIRE(n.raster && !n.raster->empty(), -4, "base region without rasterization");
            lo = n.raster->front().frame;
            hi = n.raster->back().frame;
          }
          c.start_frame = lo;
          c.end_frame = hi;
          span[(size_t)n.index] = std::make_pair(lo, hi);
        }
        span_below.swap(span);
        if (sorted_output_) std::sort(out->begin(), out->end(), by_id);
      }
    }
    if (o_.compute_vectorization) ComputeFrameVectorization(d);
  }

 private:
  // SetupRegionConstraints: the constraint of every region of `level` (the id its counterpart's
  // ancestor got in the previous chunk set) and the regions per constraint.
  void Constraints(int level, std::vector<int>* ids, std::unordered_map<int, std::vector<int>>* skeleton) const {
    ids->clear();
    ids->reserve(levels_[(size_t)level]->size());
    for (const auto& np : *levels_[(size_t)level]) {
      int base = np->index;
      if (level > 0) {
        for (int l = level; l > 0; --l) {
          const Node& n = *(*levels_[(size_t)l])[(size_t)base];
          int found = -1;
          for (int ch : *n.children) {
            if ((*levels_[(size_t)l - 1])[(size_t)ch]->constrained_id >= 0) {
              found = ch;
              break;
            }
          }
          base = found;
          if (found < 0) break;
        }
      } else if (np->constrained_id < 0) {
        base = -1;
      }
      int constraint = -1;
      if (base >= 0) {
        const Node& b = *(*levels_[0])[(size_t)base];
        if (!b.counterpart_ids) ReferenceAborts("Segmentation::SetupRegionConstraints: lack of counterparts");
        if (level < (int)b.counterpart_ids->size()) {
          constraint = (*b.counterpart_ids)[(size_t)level];
          (*skeleton)[constraint].push_back(np->index);
        }
      }
      ids->push_back(constraint);
    }
  }

  const RegionSegOptions& o_;
  Setup S_;
  int W_, H_, id_;
  int frames_ = 0;
  std::vector<ColorHist::Scratch> scratch_;   // per host thread (AddFrame)
  std::vector<std::unique_ptr<Level>> levels_;
  std::unordered_map<int, Node*> by_id_;
  bool constrained_ = false;
  bool sorted_output_ = false;
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// RegionSegmentationHost: chunk sets with overlap (region_segmentation.cpp)
// ---------------------------------------------------------------------------------------------
struct RegionSegmentationHost::Impl {
  double ms_lab = 0, ms_add = 0, ms_out = 0;   // VSG_DEBUG_STATS
  double ms_build = 0, ms_ids = 0, ms_retrieve = 0;
  int frames_in = 0;
  RegionSegOptions o;
  Setup S;
  int W, H;
  int chunk_sets = 0, read_chunks = 0, overlap_start = -1, lookahead_start = -1, output_frames = 0;
  std::unique_ptr<ChunkSet> cur, next;
  std::vector<int> id_offsets;
  std::vector<uint8_t> lab;
  FlowSamples flow_samples;   // of the frame being added

  void Output(bool flush, std::vector<std::unique_ptr<SegDesc>>* results) {   // ChunkBoundaryOutput
    if (!flush) {
      Segment(overlap_start, lookahead_start > 0 ? lookahead_start : cur->frames(), results);
    } else {
      Segment(cur->frames(), cur->frames(), results);
    }
    overlap_start = lookahead_start = -1;
    if (!flush) {
      cur = std::move(next);
    } else {
      cur.reset();
    }
    next.reset();
  }

  void Segment(int overlap_at, int lookahead_at, std::vector<std::unique_ptr<SegDesc>>* results) {   // SegmentAndOutputChunk
    const double t0 = NowMs();
    cur->BuildHierarchy();
    const double t1 = NowMs();
    if (cur->levels() > (int)id_offsets.size()) id_offsets.resize((size_t)cur->levels(), 0);
    cur->ClipToFrames(lookahead_at, overlap_at);
    std::vector<int> next_offsets(id_offsets.size());
    cur->AssignIds(chunk_sets > 0, id_offsets, &next_offsets);
    id_offsets.swap(next_offsets);
    if (next) next->PullConstraints(*cur);
    cur->DropBaseLevel();
    const double t2 = NowMs();
    const int first_frame = output_frames;
    // The frames of the chunk set are independent (rasters of the frame, moments, boundary
    // vectorization): one task per frame.
    std::vector<std::unique_ptr<SegDesc>> frames((size_t)std::max(overlap_at, 0));
    std::exception_ptr failure;
    std::atomic<bool> failed(false);
    ParallelItems(overlap_at, (size_t)overlap_at * (size_t)W * H, [&](int f, int) {
      try {
        std::unique_ptr<SegDesc> d(new SegDesc());
        cur->Retrieve(f, f == 0, d.get());
        d->hierarchy_frame_idx = first_frame;
        d->chunk_size = lookahead_at;
        d->overlap_start = overlap_at;
        frames[(size_t)f] = std::move(d);
      } catch (...) {
        if (!failed.exchange(true)) failure = std::current_exception();
      }
    });
    if (failed) std::rethrow_exception(failure);
    for (auto& d : frames) {
      results->push_back(std::move(d));
      ++output_frames;
    }
    ms_build += t1 - t0;
    ms_ids += t2 - t1;
    ms_retrieve += NowMs() - t2;
    ++chunk_sets;
  }
};

RegionSegmentationHost::RegionSegmentationHost(const RegionSegOptions& options, int frame_width, int frame_height)
    : impl_(new Impl()) {
  VSG_REQUIRE(options.chunk_set_size > 1, -1, "At least two chunks per chunk_set required.");
  VSG_REQUIRE(options.chunk_set_overlap > 0, -1, "At least one chunk in overlap expected.");
  VSG_REQUIRE(options.chunk_set_overlap < options.chunk_set_size, -1,
              "Overlap has to be strictly smaller than a chunk set.");
  VSG_REQUIRE(options.constraint_chunks <= options.chunk_set_overlap, -1,
              "Constraints must be smaller or equal to overlap");
  VSG_REQUIRE(options.use_appearance || options.use_flow, -1, "At least apperance or flow need to be set.");
  VSG_REQUIRE(frame_width >= 1 && frame_height >= 1, -1, "frame size");
  impl_->o = options;
  impl_->S = Setup{options.use_appearance, options.use_flow, options.use_size_penalizer, options.small_region_penalizer,
                   options.luminance_bins, options.color_bins, options.flow_bins};
  impl_->W = frame_width;
  impl_->H = frame_height;
}

RegionSegmentationHost::~RegionSegmentationHost() {
  if (getenv("VSG_DEBUG_STATS") && impl_) {
    std::fprintf(stderr, "[vsg] region segmentation: lab %.1f ms, descriptors %.1f ms, hierarchy + output %.1f ms over %d frames\n",
                 impl_->ms_lab, impl_->ms_add, impl_->ms_out, impl_->frames_in);
    std::fprintf(stderr, "[vsg]   hierarchy %.1f ms, clip + ids %.1f ms, retrieve %.1f ms\n", impl_->ms_build, impl_->ms_ids,
                 impl_->ms_retrieve);
    std::fprintf(stderr, "[vsg]   descriptor passes, CPU over all threads: colour %.1f ms, flow %.1f ms; longest region of every "
                 "frame summed: %.1f ms\n", g_cpu_color_us.exchange(0) * 1e-3, g_cpu_flow_us.exchange(0) * 1e-3,
                 g_longest_us.exchange(0) * 1e-3);
  }
}

int RegionSegmentationHost::ProcessFrame(bool flush, const SegDesc* overseg, const uint8_t* bgr, size_t stride,
                                         const float* flow) {
  Impl& I = *impl_;
  results_.clear();
  encoded_.clear();
  VSG_REQUIRE((overseg == nullptr) == (bgr == nullptr), -1,
              "Requring both segmentation and features to be either set or null.");
  if (!I.cur) I.cur.reset(new ChunkSet(I.o, I.S, I.W, I.H, I.chunk_sets));
  const int overlap_from = I.o.chunk_set_size - I.o.chunk_set_overlap;
  const int lookahead_from = overlap_from + I.o.constraint_chunks;
  if (overseg) {
    VSG_REQUIRE(stride >= (size_t)I.W * 3, -1, "stride smaller than a row");
    const double t_lab = NowMs();
    if (I.S.appearance) {
      I.lab.resize((size_t)I.W * I.H * 3);
      BgrToLab8(bgr, stride, I.W, I.H, I.lab.data());
    }
    if (I.S.flow && flow) I.flow_samples.Compute(flow, I.W, I.H, I.S.flow_bins);
    I.ms_lab += NowMs() - t_lab;
    ++I.frames_in;
    const bool starts_chunk = overseg->has_hierarchy;
    if (starts_chunk) ++I.read_chunks;
    if (starts_chunk && I.read_chunks > 0 && I.read_chunks % I.o.chunk_set_size == 0) {
      const double t_out = NowMs();
      I.Output(false, &results_);
      I.ms_out += NowMs() - t_out;
    }
    const double t_add = NowMs();
    const int phase = I.read_chunks % I.o.chunk_set_size;
    if (phase >= overlap_from) {
      if (!I.next) I.next.reset(new ChunkSet(I.o, I.S, I.W, I.H, I.chunk_sets + 1));
      if (I.overlap_start < 0) I.overlap_start = I.cur->frames();
      if (starts_chunk) {
        std::unordered_map<int, Node*> mapping;
        std::unordered_map<int, Node*>* shared = phase < lookahead_from ? &mapping : nullptr;
        I.cur->AddBaseLevel(overseg->hierarchy0, nullptr, shared);
        I.next->AddBaseLevel(overseg->hierarchy0, shared, nullptr);
      }
      I.cur->AddFrame(*overseg, I.lab.data(), flow, &I.flow_samples);
      I.next->AddFrame(*overseg, I.lab.data(), flow, &I.flow_samples);
    } else {
      if (starts_chunk) I.cur->AddBaseLevel(overseg->hierarchy0, nullptr, nullptr);
      I.cur->AddFrame(*overseg, I.lab.data(), flow, &I.flow_samples);
    }
    if (phase >= lookahead_from && I.lookahead_start < 0) I.lookahead_start = I.cur->frames();
    I.ms_add += NowMs() - t_add;
  }
  if (flush) {
    const double t_out = NowMs();
    I.Output(true, &results_);
    I.ms_out += NowMs() - t_out;
  }
  return (int)results_.size();
}

const std::string& RegionSegmentationHost::result_bytes(int i) {
  if (encoded_.size() != results_.size()) {
    encoded_.clear();
    for (const auto& r : results_) encoded_.push_back(EncodeSegDesc(*r));
  }
  return encoded_[(size_t)i];
}

}  // namespace vsg

#ifdef VSG_TEST_MODELS
// tests/test_descriptor_passes.py compiles this file with a test entry that compares the batched
// passes above with the per-pixel ones (AddLabPixel, FlowHist::Add)
#include "../../tests/host/descriptor_model.inc"
#endif
