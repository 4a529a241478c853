// tube_harness.cpp -- TEST PROGRAM for the host tube analysis (postprocess.cpp: TubeSplitter), CPU only.
// Built by tests/test_tube_analysis.py with -DVSG_TEST_MODELS, which adds TubeSplitter::FinishPlain
// (tests/host/tube_plain_model.inc, the analysis written the plain way, in the reference's order).
//
//   tube_harness random <cases> <seed>   random regions (many small components per slice, moving blobs,
//                                        frames with gaps, with and without flow): Finish and FinishPlain
//                                        have to return the same tubes, areas and kept tube; exit 1 if not
//   tube_harness file <dump> [plain]     the input of one chunk, dumped by the library under
//                                        VSG_DUMP_TUBES=<dump> (dense_graph.cpp): timing and a checksum
//                                        of the result ("plain": of FinishPlain)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "host_model.h"

using namespace vsg;

static double Now() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool SameResult(const TubeResult& a, const TubeResult& b) {
  if (a.tubes.size() != b.tubes.size() || a.tube_to_keep != b.tube_to_keep || a.tubes_matched != b.tubes_matched)
    return false;
  if (a.areas.size() != b.areas.size() ||
      (a.areas.size() && std::memcmp(a.areas.data(), b.areas.data(), 4 * a.areas.size()) != 0))
    return false;
  for (size_t k = 0; k < a.tubes.size(); ++k) {
    if (a.tubes[k].size() != b.tubes[k].size()) return false;
    for (size_t i = 0; i < a.tubes[k].size(); ++i) {
      const RasterSlice &x = a.tubes[k][i], &y = b.tubes[k][i];
      if (x.frame != y.frame || x.raster.size() != y.raster.size()) return false;
      if (x.raster.size() && std::memcmp(x.raster.data(), y.raster.data(), sizeof(Interval) * x.raster.size()) != 0)
        return false;
    }
  }
  return true;
}

static void MaskToRaster(const std::vector<char>& mask, int w, int h, int ox, int oy, Raster* out) {
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w;) {
      if (!mask[(size_t)y * w + x]) { ++x; continue; }
      int e = x;
      while (e + 1 < w && mask[(size_t)y * w + e + 1]) ++e;
      out->push_back(Interval{oy + y, ox + x, e});
      out->back().rx = ox + e;
      x = e + 1;
    }
  }
}

// N4 components of a raster by testing every pair of intervals (adjacent rows, overlapping in x),
// ordered by first interval: what SplitComponentsN4 (a row sweep) has to return.
static void ComponentsByAllPairs(const Raster& r, std::vector<Raster>* comps) {
  const int n = (int)r.size();
  std::vector<int> uf((size_t)n);
  for (int i = 0; i < n; ++i) uf[i] = i;
  auto root = [&](int i) { while (uf[i] != i) i = uf[i]; return i; };
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < i; ++k) {
      if (std::abs(r[i].y - r[k].y) == 1 && std::max(r[i].lx, r[k].lx) <= std::min(r[i].rx, r[k].rx)) {
        const int a = root(i), b = root(k);
        if (a != b) uf[std::max(a, b)] = std::min(a, b);
      }
    }
  }
  std::vector<int> comp_of((size_t)n, -1);
  for (int i = 0; i < n; ++i) {
    const int rt = root(i);
    if (comp_of[rt] < 0) {
      comp_of[rt] = (int)comps->size();
      comps->emplace_back();
    }
    (*comps)[(size_t)comp_of[rt]].push_back(r[i]);
  }
}

static bool SameComponents(const std::vector<Raster>& a, const std::vector<Raster>& b) {
  if (a.size() != b.size()) return false;
  for (size_t k = 0; k < a.size(); ++k) {
    if (a[k].size() != b[k].size() || std::memcmp(a[k].data(), b[k].data(), sizeof(Interval) * a[k].size()) != 0) return false;
  }
  return true;
}

static int RunRandom(int cases, unsigned seed) {
  const int W = 320, H = 200;
  int splits = 0, joins = 0;
  for (int c = 0; c < cases; ++c) {
    std::mt19937 rng(seed * 7919u + (unsigned)c);
    auto uni = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
    auto real = [&]() { return (double)(rng() >> 8) / (double)(1u << 24); };
    const int kind = c % 6;
    const int w = kind == 5 ? uni(120, 200) : uni(6, 70), h = uni(5, 50), ox = uni(0, W - 201), oy = uni(0, H - 51);
    const int num_frames = uni(1, 14);
    Raster3D raster;
    int frame = uni(0, 3);
    double bx = real() * w, by = real() * h, vx = real() * 6 - 3, vy = real() * 4 - 2;
    const double fill = kind == 0 ? 0.08 + real() * 0.5 : kind >= 4 ? real() * 0.01 : 0.02 + real() * 0.1;
    const int jump_at = uni(1, 8);
    for (int f = 0; f < num_frames; ++f) {
      std::vector<char> mask((size_t)w * h, 0);
      for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
          bool on = real() < fill;
          if (kind >= 1) {   // a moving blob; 2: with a second one; 3: that shrinks away; 4: that jumps
                             // once, too far to be matched; 5: two blobs far apart
            const double r1 = (kind == 3 ? std::max(0.0, 9.0 - f) : 7.0);
            on = on || (x - bx) * (x - bx) + (y - by) * (y - by) < r1 * r1;
            if (kind == 5) on = on || (x - (w - 8)) * (x - (w - 8)) + (y - by) * (y - by) < 36;
            if (kind == 2) on = on || (x - (w - bx)) * (x - (w - bx)) + (y - (h - by)) * (y - (h - by)) < 16;
            if (kind < 4 && real() < 0.03) on = !on;
          }
          mask[(size_t)y * w + x] = on;
        }
      }
      RasterSlice sl;
      sl.frame = frame;
      MaskToRaster(mask, w, h, ox, oy, &sl.raster);
      if (!sl.raster.empty()) raster.push_back(std::move(sl));
      frame += real() < 0.15 ? uni(2, 3) : 1;   // a region can miss frames
      bx += vx; by += vy;
      if (kind == 4 && f + 1 == jump_at) bx += bx < w / 2 ? 17.5 : -17.5;
    }
    if (raster.empty()) continue;
    for (const RasterSlice& sl : raster) {
      std::vector<Raster> sweep, pairs;
      SplitComponentsN4(sl.raster, &sweep);
      ComponentsByAllPairs(sl.raster, &pairs);
      if (!SameComponents(sweep, pairs)) {
        std::printf("case %d: SplitComponentsN4 differs from the all-pairs components (%zu vs %zu)\n", c, sweep.size(), pairs.size());
        return 1;
      }
    }
    TubeSplitter fast, plain;
    std::vector<FlowRequest> req, req2;
    fast.Prepare(raster, &req);
    plain.Prepare(raster, &req2);
    if (req.size() != req2.size()) return 1;
    std::vector<float> flow(2 * req.size());
    for (float& v : flow) v = (float)(real() * 8 - 4);
    const bool with_flow = (c / 6) % 2 == 0;
    if (fast.MaySplit() != plain.MaySplit()) return 1;
    if (!fast.MaySplit()) continue;
    TubeResult a, b;
    fast.Finish(W, H, with_flow ? flow.data() : nullptr, &a);
    plain.FinishPlain(W, H, with_flow ? flow.data() : nullptr, &b);
    if (!SameResult(a, b)) {
      std::printf("case %d (kind %d, %d frames, %dx%d, flow %d): results differ (%zu vs %zu tubes, keep %d vs %d)\n",
                  c, kind, num_frames, w, h, (int)with_flow, a.tubes.size(), b.tubes.size(), a.tube_to_keep, b.tube_to_keep);
      return 1;
    }
    splits += a.tubes.size() > 1;
    joins += a.tubes_matched - (int)a.tubes.size();
  }
  std::printf("%d cases identical (%d regions split, %d joins)\n", cases, splits, joins);
  return 0;
}

static int RunFile(const char* path, bool use_plain) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return 2;
  auto get = [&]() { int v = 0; if (std::fread(&v, 4, 1, f) != 1) std::abort(); return v; };
  const int W = get(), H = get(), n = get(), have_flows = get();
  double t_prep = 0, t_fin = 0, worst = 0;
  unsigned long long sum = 1469598103934665603ull;
  auto mix = [&](unsigned long long v) { sum = (sum ^ v) * 1099511628211ull; };
  long long tubes = 0;
  for (int r = 0; r < n; ++r) {
    const int ns = get();
    if (ns < 0) continue;
    Raster3D raster((size_t)ns);
    for (RasterSlice& sl : raster) {
      sl.frame = get();
      sl.raster.resize((size_t)get());
      if (!sl.raster.empty() && std::fread(sl.raster.data(), sizeof(Interval), sl.raster.size(), f) != sl.raster.size()) std::abort();
    }
    const int nreq = get();
    std::vector<float> samples;
    if (have_flows) {
      samples.resize(2 * (size_t)nreq);
      if (nreq && std::fread(samples.data(), 8, (size_t)nreq, f) != (size_t)nreq) std::abort();
    }
    TubeSplitter ts;
    std::vector<FlowRequest> req;
    const double t0 = Now();
    ts.Prepare(raster, have_flows ? &req : nullptr);
    const double t1 = Now();
    t_prep += t1 - t0;
    if (have_flows && (int)req.size() != nreq) { std::fprintf(stderr, "request count differs\n"); return 1; }
    if (!ts.MaySplit()) continue;
    TubeResult out;
    if (use_plain) ts.FinishPlain(W, H, have_flows ? samples.data() : nullptr, &out);
    else ts.Finish(W, H, have_flows ? samples.data() : nullptr, &out);
    const double t2 = Now();
    t_fin += t2 - t1;
    worst = std::max(worst, t2 - t1);
    tubes += out.tubes_matched;
    mix((unsigned long long)out.tubes.size()); mix((unsigned long long)(out.tube_to_keep + 1));
    for (size_t k = 0; k < out.tubes.size(); ++k) {
      unsigned int a;
      std::memcpy(&a, &out.areas[k], 4);
      mix(a);
      for (const RasterSlice& sl : out.tubes[k]) {
        mix((unsigned long long)sl.frame);
        for (const Interval& iv : sl.raster) mix(((unsigned long long)iv.y << 40) ^ ((unsigned long long)iv.lx << 20) ^ (unsigned long long)iv.rx);
      }
    }
  }
  std::fclose(f);
  std::printf("%dx%d, %d regions, %lld matched tubes: prepare %.1f ms, finish %.1f ms (longest region %.1f ms), checksum %016llx\n",
              W, H, n, tubes, t_prep, t_fin, worst, sum);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 4 && !std::strcmp(argv[1], "random")) return RunRandom(std::atoi(argv[2]), (unsigned)std::atoi(argv[3]));
  if (argc >= 3 && !std::strcmp(argv[1], "file")) return RunFile(argv[2], argc > 3 && !std::strcmp(argv[3], "plain"));
  std::fprintf(stderr, "usage: tube_harness random <cases> <seed> | file <dump> [plain]\n");
  return 2;
}
