// postprocess.cpp -- host-side post-processing of the GPU over-segmentation: scan-interval
// utilities, shape moments, spatial-connectedness tube analysis and the SegmentationDesc wire
// encoder.  These stages work on a few 10^4..10^6 scan intervals (not pixels), are full of
// order-dependent float decisions, and stay on the host (north_star: "host code stays C++").
//
// Reference behaviour restated (never copied); numerics follow the reference's C++ promotion
// rules on baseline x86-64 (float expressions, double where a double literal / double libm
// overload appears).  Compile with -ffp-contract=off.
#include "host_model.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace vsg {

int RasterArea(const Raster& r) {
  int area = 0;
  for (const Interval& s : r) area += s.rx - s.lx + 1;
  return area;
}

// segment_util/segmentation_util.cpp:652-693
void MomentsFromRaster(const Raster& r, Moments* out) {
  float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0, area = 0;
  for (const Interval& s : r) {
    const float m = (float)s.lx, n = (float)s.rx, y = (float)s.y;
    const float len = (n - m + 1);
    area += len;
    const float cx = (float)((double)(n + m) * 0.5);
    const float row_x = cx * len;
    const float row_y = y * len;
    sx += row_x;
    sy += row_y;
    sxy += y * row_x;
    syy += y * row_y;
    sxx += len * (-m + 2 * m * m + n + 2 * m * n + 2 * n * n) / 6.0f;
  }
  const float inv = 1.0f / area;
  out->size = area;
  out->mean_x = sx * inv;
  out->mean_y = sy * inv;
  out->xx = sxx * inv;
  out->xy = sxy * inv;
  out->yy = syy * inv;
}

// segment_util/segmentation_util.cpp:484-570
void MergeRasters(const Raster& a, const Raster& b, Raster* out) {
  Raster res;
  res.reserve(a.size() + b.size());
  size_t ia = 0, ib = 0;
  std::vector<int> ends;   // left, right, left, right ... of one scanline, ordered by left
  const int kInf = 1 << 30;
  while (ia < a.size() || ib < b.size()) {
    const int ya = ia < a.size() ? a[ia].y : kInf;
    const int yb = ib < b.size() ? b[ib].y : kInf;
    if (ya < yb) {
      res.push_back(a[ia++]);
      continue;
    }
    if (yb < ya) {
      res.push_back(b[ib++]);
      continue;
    }
    const int y = ya;
    ends.clear();
    while (true) {
      const bool ha = ia < a.size() && a[ia].y == y;
      const bool hb = ib < b.size() && b[ib].y == y;
      if (!ha && !hb) break;
      const int xa = ha ? a[ia].lx : std::numeric_limits<int>::max();
      const int xb = hb ? b[ib].lx : std::numeric_limits<int>::max();
      if (xa < xb) {
        ends.push_back(a[ia].lx);
        ends.push_back(a[ia].rx);
        ++ia;
      } else {
        ends.push_back(b[ib].lx);
        ends.push_back(b[ib].rx);
        ++ib;
      }
    }
    // join runs that abut (next.left - 1 == cur.right)
    const int n = (int)ends.size();
    int start = 0, k = 0;
    while (k < n) {
      if (k + 2 == n) {
        res.push_back(Interval{y, ends[start], ends[k + 1]});
        break;
      }
      if (ends[k + 2] - 1 == ends[k + 1]) {
        k += 2;
      } else {
        res.push_back(Interval{y, ends[start], ends[k + 1]});
        k += 2;
        start = k;
      }
    }
  }
  out->swap(res);
}

// segment_util/segmentation_util.cpp:1009-1101 (N4_CONNECT branch)
void SplitComponentsN4(const Raster& r, std::vector<Raster>* comps) {
  const int n = (int)r.size();
  std::vector<int> uf(n);
  auto root = [&uf](int i) {
    while (uf[i] != i) {
      uf[i] = uf[uf[i]];
      i = uf[i];
    }
    return i;
  };
  // Intervals of one row are disjoint and ordered by lx, so the intervals of the row above that
  // overlap interval i in x are consecutive and the first of them never moves left as i advances
  // (which pairs are united first does not matter: the components come out the same).
  int row_start = 0, prev_start = 0, prev_end = 0, cursor = 0;
  for (int i = 0; i < n; ++i) {
    uf[i] = i;
    if (i == 0 || r[i].y != r[i - 1].y) {
      const bool adjacent = i > 0 && r[i - 1].y + 1 == r[i].y;
      prev_start = adjacent ? row_start : i;
      prev_end = i;
      row_start = i;
      cursor = prev_start;
    }
    while (cursor < prev_end && r[cursor].rx < r[i].lx) ++cursor;
    for (int k = cursor; k < prev_end && r[k].lx <= r[i].rx; ++k) {
      const int a = root(i), b = root(k);
      if (a != b) uf[a] = b;
    }
  }
  int num = 0;
  for (int i = 0; i < n; ++i) num += (root(i) == i);
  if (num == 1) {
    comps->push_back(r);
    return;
  }
  std::vector<int> comp_of_root(n, -1);
  for (int i = 0; i < n; ++i) {
    const int rt = root(i);
    if (comp_of_root[rt] < 0) {
      comp_of_root[rt] = (int)comps->size();
      comps->emplace_back();
    }
    (*comps)[comp_of_root[rt]].push_back(r[i]);
  }
}

// ---------------------------------------------------------------------------------------
// Shape descriptor (segment_util/segmentation_util.h:138-151, .cpp:243-410).
// ---------------------------------------------------------------------------------------
namespace {

struct V2 {
  float x = 0, y = 0;
};
inline V2 Add(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
inline V2 Sub(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
inline V2 Mul(V2 a, float s) { return V2{a.x * s, a.y * s}; }
inline float HypotYX(float y, float x) { return (float)std::hypot((double)y, (double)x); }

struct Shape {
  V2 center;
  float mag_major = 0, mag_minor = 0;
  V2 dir_major{1.0f, 0.0f};
  V2 dir_minor{0.0f, 1.0f};
  int size = 0;
};

// Updates center and size always; axes only when they are reliable (return value true).
bool ShapeFromMoments(const Moments& mo, Shape* sh) {
  float mx = 0, my = 0, mxx = 0, mxy = 0, myy = 0, area_sum = 0;
  const float area = mo.size;
  area_sum += area;
  mx += mo.mean_x * area;
  my += mo.mean_y * area;
  mxx += mo.xx * area;
  mxy += mo.xy * area;
  myy += mo.yy * area;
  const float inv = 1.0f / area_sum;
  mx *= inv;
  my *= inv;
  mxx *= inv;
  mxy *= inv;
  myy *= inv;
  sh->center = V2{mx, my};
  sh->size = (int)area_sum;
  if (area_sum < 10) return false;
  const float vxx = mxx - mx * mx;
  const float vxy = mxy - mx * my;
  const float vyy = myy - my * my;
  const float trace = vxx + vyy;
  const float det = vxx * vyy - vxy * vxy;
  float disc = (float)(0.25 * (double)trace * (double)trace - (double)det);
  disc = std::max(0.0f, disc);
  const float sq = (float)std::sqrt((double)disc);
  const float e1 = (float)((double)trace * 0.5 - (double)sq);
  const float e2 = (float)((double)trace * 0.5 + (double)sq);
  if (std::min(std::fabs((double)e1), std::fabs((double)e2)) < 1) return false;
  V2 ev1{1.0f, 0.0f}, ev2{0.0f, 1.0f};
  const V2 v1{e1 - vyy, vxy};
  const V2 v2{e2 - vyy, vxy};
  const float n1 = HypotYX(v1.y, v1.x);
  const float n2 = HypotYX(v2.y, v2.x);
  if (n1 > 1e-6f && n2 > 1e-6f && (double)disc > 0.1) {
    ev1 = Mul(v1, 1.0f / n1);
    ev2 = Mul(v2, 1.0f / n2);
  }
  float s1 = (float)std::sqrt(std::fabs((double)e1));
  float s2 = (float)std::sqrt(std::fabs((double)e2));
  if (s1 < s2) {
    std::swap(s1, s2);
    std::swap(ev1, ev2);
  }
  const V2 normal{-ev1.y, ev1.x};
  if (ev2.x * normal.x + ev2.y * normal.y < 0) ev2 = V2{-ev2.x, -ev2.y};
  sh->center = V2{mx, my};
  sh->mag_major = s1;
  sh->mag_minor = s2;
  sh->dir_major = ev1;
  sh->dir_minor = ev2;
  return true;
}

void ShapeBox(const Shape& s, float border, V2* c) {
  const V2 major = Mul(s.dir_major, s.mag_major * 1.65f + border);
  const V2 minor = Mul(s.dir_minor, s.mag_minor * 1.65f + border);
  c[0] = Add(Sub(s.center, major), minor);
  c[1] = Sub(Sub(s.center, major), minor);
  c[2] = Sub(Add(s.center, major), minor);
  c[3] = Add(Add(s.center, major), minor);
}

bool BoxesIntersect(const V2* p, const V2* q) {
  for (int k = 0; k < 4; ++k) {
    const V2 pd = Sub(p[(k + 1) % 4], p[k]);
    const double pdx = pd.x, pdy = pd.y;
    for (int l = 0; l < 4; ++l) {
      const V2 qd = Sub(q[(l + 1) % 4], q[l]);
      const double qdx = qd.x, qdy = qd.y;
      const V2 dl = Sub(q[l], p[k]);
      const double dx = dl.x, dy = dl.y;
      const double kross = pdx * qdy - pdy * qdx;
      if (std::fabs(kross) < 1e-6) continue;
      const float inv_kross = (float)(1.0f / kross);
      const double t = (dx * qdy - dy * qdx) * inv_kross;
      const double s = (dx * pdy - dy * pdx) * inv_kross;
      if (t > -1e-6f && t < 1.0f + 1e-6f && s > -1e-6f && s < 1.0f + 1e-6f) return true;
    }
  }
  return false;
}

struct TSlice {
  int frame = -1;
  Raster raster;
  Shape shape;
  void Recompute() {
    Moments m;
    MomentsFromRaster(raster, &m);
    ShapeFromMoments(m, &shape);   // keeps previous axes if unreliable (reference behaviour)
  }
};
typedef std::vector<TSlice> Tube;

float MeanSliceSize(const Tube& t) {   // dense_segmentation_graph.cpp:35-45
  if (t.empty()) return 0;
  float s = 0;
  for (const TSlice& sl : t) s += (float)sl.shape.size;
  return s / (float)t.size();
}

// Joins b into a; where both have a slice of a frame, a's slice is the base (its axes stay when the
// joined shape is unreliable).  The slices are moved, not copied: a and b are left empty.  Frames
// that only b had are appended to *gained (may be null).
void JoinTubes(Tube* a, Tube* b, Tube* out, std::vector<int>* gained) {   // .cpp:47-88
  out->clear();
  out->reserve(a->size() + b->size());
  const bool a_empty = a->empty();
  size_t i = 0, j = 0;
  while (i < a->size() && j < b->size()) {
    if ((*a)[i].frame < (*b)[j].frame) {
      out->push_back(std::move((*a)[i++]));
    } else if ((*a)[i].frame > (*b)[j].frame) {
      if (gained) gained->push_back((*b)[j].frame);
      out->push_back(std::move((*b)[j++]));
    } else {
      TSlice& m = (*a)[i];
      MergeRasters(m.raster, (*b)[j].raster, &m.raster);
      m.Recompute();
      out->push_back(std::move(m));
      ++i;
      ++j;
    }
  }
  while (i < a->size()) out->push_back(std::move((*a)[i++]));
  while (j < b->size()) {
    if (gained && !a_empty) gained->push_back((*b)[j].frame);
    out->push_back(std::move((*b)[j++]));
  }
  if (gained && a_empty) {
    for (const TSlice& sl : *out) gained->push_back(sl.frame);
  }
  a->clear();
  b->clear();
}

bool TemporalNeighbors(const Tube& a, const Tube& b) {   // .cpp:90-110
  if (a.empty() || b.empty()) return false;
  const Shape *p, *q;
  if (a[0].frame - 1 == b.back().frame) {
    p = &a[0].shape;
    q = &b.back().shape;
  } else if (a.back().frame + 1 == b[0].frame) {
    p = &a.back().shape;
    q = &b[0].shape;
  } else {
    return false;
  }
  const float ratio = (float)std::min(p->size, q->size) * (1.0f / (float)std::max(p->size, q->size));
  const V2 d = Sub(p->center, q->center);
  return (double)ratio > 0.9 && std::hypot((double)d.y, (double)d.x) < 20;
}

float BoxOverlapFraction(const Tube& a, const Tube& b) {   // .cpp:150-191
  if (a.empty() || b.empty()) return std::numeric_limits<float>::max();
  const int f0 = std::max(a[0].frame, b[0].frame);
  const int f1 = std::min(a.back().frame, b.back().frame);
  int i = 0, j = 0, w = 0, hits = 0;
  for (int f = f0; f <= f1; ++f) {
    while (a[i].frame < f) ++i;
    while (b[j].frame < f) ++j;
    if (a[i].frame != f || b[j].frame != f) continue;
    V2 pa[4], pb[4];
    ShapeBox(a[i].shape, 10, pa);
    ShapeBox(b[j].shape, 10, pb);
    if (BoxesIntersect(pa, pb)) ++hits;
    ++w;
  }
  return w > 0 ? (float)hits * (1.0f / (float)w) : std::numeric_limits<float>::max();
}

}  // namespace

// dense_segmentation_graph.h:666-861, in two steps so that the backward flow can be sampled
// where it lives (on the device): Prepare computes what does not depend on the flow -- the N4
// components of every slice with their shape descriptors -- and lists the points the matching
// will read (FindPreviousTube reads flow[frame](int(cy), int(cx)) for every component of every
// slice after the region's first one, whatever the earlier decisions were); Finish does the
// temporal matching with those samples.
struct TubeSplitter::Impl {
  std::vector<std::vector<TSlice>> slices;   // per raster slice: components, ordered by first interval
};

TubeSplitter::TubeSplitter() : impl_(new Impl) {}
TubeSplitter::~TubeSplitter() {}
TubeSplitter::TubeSplitter(TubeSplitter&& o) noexcept : impl_(std::move(o.impl_)) {}

void TubeSplitter::Prepare(const Raster3D& raster, std::vector<FlowRequest>* requests) {
  impl_->slices.clear();
  impl_->slices.reserve(raster.size());
  for (const RasterSlice& rs : raster) {
    std::vector<Raster> comps;
    SplitComponentsN4(rs.raster, &comps);
    impl_->slices.emplace_back(comps.size());
    std::vector<TSlice>& sl = impl_->slices.back();
    for (size_t c = 0; c < comps.size(); ++c) {
      sl[c].frame = rs.frame;
      sl[c].raster.swap(comps[c]);
      sl[c].Recompute();
      if (impl_->slices.size() > 1 && requests) {
        requests->push_back(FlowRequest{rs.frame, (int)sl[c].shape.center.y, (int)sl[c].shape.center.x});
      }
    }
  }
}

bool TubeSplitter::MaySplit() const {
  // One component per slice and consecutive frames can still split (a tube is only continued if
  // sizes and centres agree), so every region with more than one slice component goes through
  // Finish; a region with a single slice has exactly one tube.
  return impl_->slices.size() > 1 || (impl_->slices.size() == 1 && impl_->slices[0].size() > 1);
}

void TubeSplitter::Finish(int W, int H, const float* flow_xy, TubeResult* out) {
  out->tubes.clear();
  out->areas.clear();
  out->tube_to_keep = -1;
  std::vector<Tube> done, active;
  std::vector<float> last_x, last_y;
  std::vector<char> candidate;
  const float inv_diam = (float)(1.0f / std::hypot((double)W, (double)H));
  size_t sample = 0;

  bool first_slice = true;
  for (std::vector<TSlice>& slices : impl_->slices) {
    const int frame = slices.empty() ? -1 : slices[0].frame;
    // Prepare listed one flow sample per component of every slice after the first, whether or
    // not the matching reads it: the cursor advances with the slices, not with the decisions.
    const bool had_requests = !first_slice;
    first_slice = false;
    if (active.empty()) {
      if (had_requests) sample += slices.size();
      for (TSlice& s : slices) {
        active.emplace_back();
        active.back().push_back(std::move(s));
      }
      continue;
    }
    std::vector<Tube> next;
    std::vector<char> continued(active.size(), 0);
    // centre of every active tube's last slice (NaN frame marker: tubes already continued or
    // ending in this frame are not candidates)
    const int num_active = (int)active.size();
    last_x.resize((size_t)num_active);
    last_y.resize((size_t)num_active);
    candidate.resize((size_t)num_active);
    for (int k = 0; k < num_active; ++k) {
      candidate[k] = !active[k].empty() && active[k].back().frame < frame;
      if (candidate[k]) {
        last_x[k] = active[k].back().shape.center.x;
        last_y[k] = active[k].back().shape.center.y;
      }
    }
    for (TSlice& s : slices) {
      // FindPreviousTube, dense_segmentation_graph.h:601-629
      V2 c = s.shape.center;
      if (flow_xy) {
        const float* fp = flow_xy + 2 * sample;
        c = Add(c, V2{fp[0], fp[1]});
      }
      ++sample;
      float best = std::numeric_limits<float>::max();
      float best_idx = -1;   // float in the reference
      double skip_above = std::numeric_limits<double>::infinity();
      for (int k = 0; k < num_active; ++k) {
        if (!candidate[k]) continue;
        const float dx = last_x[k] - c.x, dy = last_y[k] - c.y;
        // the distance is only evaluated where it can be below the best one so far
        if ((double)dx * (double)dx + (double)dy * (double)dy > skip_above) continue;
        const float dist = HypotYX(dy, dx);
        if (dist < best) {
          best = dist;
          best_idx = (float)k;
          skip_above = (double)best * (double)best * (1.0 + 1e-6);
        }
      }
      const int prev = (int)best_idx;
      if (prev < 0) {
        next.emplace_back();
        next.back().push_back(std::move(s));
        continue;
      }
      const int sa = active[prev].back().shape.size, sb = s.shape.size;
      const float ratio = (float)((double)std::min(sa, sb) / ((double)std::max(sa, sb) + 1e-6));
      if ((double)ratio > 0.75 && best * inv_diam < 0.04f) {
        continued[prev] = 1;
        candidate[prev] = 0;
        active[prev].push_back(std::move(s));
        next.emplace_back();
        next.back().swap(active[prev]);
      } else {
        next.emplace_back();
        next.back().push_back(std::move(s));
      }
    }
    for (size_t k = 0; k < active.size(); ++k) {
      if (!continued[k]) done.push_back(std::move(active[k]));
    }
    next.swap(active);
  }
  for (Tube& t : active) done.push_back(std::move(t));
  out->tubes_matched = (int)done.size();

  if (done.size() > 1) {
    // The reference keeps the tubes in a vector, erases the joined one and puts the join in the
    // other one's place, so the order of the survivors never changes.  Here the tubes keep their
    // index: a doubly linked list of the live ones gives the reference's order, and the tubes
    // that have a slice in a frame are listed per frame (the only ones at a finite distance).
    const int T = (int)done.size();
    std::vector<int> nxt((size_t)T + 1), prv((size_t)T + 1);   // T = list head
    for (int k = 0; k <= T; ++k) {
      nxt[k] = k == T ? 0 : k + 1;
      prv[k] = k == 0 ? T : k - 1;
    }
    std::vector<char> live((size_t)T, 1);
    auto unlink = [&](int k) {
      live[k] = 0;
      nxt[prv[k]] = nxt[k];
      prv[nxt[k]] = prv[k];
    };
    int frame_lo = std::numeric_limits<int>::max(), frame_hi = -1;
    for (const Tube& t : done) {
      if (t.empty()) continue;
      frame_lo = std::min(frame_lo, t[0].frame);
      frame_hi = std::max(frame_hi, t.back().frame);
    }
    std::vector<std::vector<int>> in_frame(frame_hi >= frame_lo ? (size_t)(frame_hi - frame_lo + 1) : 0);
    for (int k = 0; k < T; ++k) {
      for (const TSlice& sl : done[k]) in_frame[(size_t)(sl.frame - frame_lo)].push_back(k);
    }
    std::vector<int> seen((size_t)T, -1), gained;
    // centre of tube l's slice of frame f (x = NaN: l has no slice there), so that a distance
    // is evaluated without visiting the other tube
    const size_t num_frames = in_frame.size();
    const float kNone = std::numeric_limits<float>::quiet_NaN();
    std::vector<V2> centre_at(num_frames * (size_t)T, V2{kNone, kNone});
    auto note_centres = [&](int k) {
      for (const TSlice& sl : done[k]) centre_at[(size_t)(sl.frame - frame_lo) * T + k] = sl.shape.center;
    };
    for (int k = 0; k < T; ++k) note_centres(k);
    // ClosestTube (.cpp:193-210): the first tube, in order, at the smallest mean centre distance
    // (MeanCentreDistance, .cpp:112-148: over the common frames, in frame order, summed in float); tubes without a common frame
    // are at distance max and never chosen.  The exact distance is only evaluated where a bound
    // computed with plain square roots says it can be below the best one so far.
    auto closest = [&](int k) {
      float best = std::numeric_limits<float>::max();
      int best_idx = -1;
      const Tube& tk = done[k];
      double skip_above = std::numeric_limits<double>::infinity();
      for (const TSlice& sl : tk) {
        std::vector<int>& ids = in_frame[(size_t)(sl.frame - frame_lo)];
        size_t w = 0;
        for (size_t q = 0; q < ids.size(); ++q) {
          const int l = ids[q];
          if (!live[l]) continue;          // joined away: drop the entry
          ids[w++] = l;
          if (l == k || seen[l] == k) continue;
          seen[l] = k;
          double bound = 0;
          int common = 0;
          for (const TSlice& a : tk) {
            const V2 cl = centre_at[(size_t)(a.frame - frame_lo) * T + l];
            if (cl.x != cl.x) continue;
            const V2 dc = Sub(a.shape.center, cl);
            bound += std::sqrt((double)dc.x * (double)dc.x + (double)dc.y * (double)dc.y);
            ++common;
          }
          if (bound > skip_above * (double)common) continue;
          float sum = 0;
          for (const TSlice& a : tk) {
            const V2 cl = centre_at[(size_t)(a.frame - frame_lo) * T + l];
            if (cl.x != cl.x) continue;
            const V2 dc = Sub(a.shape.center, cl);
            sum = (float)((double)sum + std::hypot((double)dc.y, (double)dc.x));
          }
          const float d = sum / (float)common;
          if (d < best || (d == best && best_idx >= 0 && l < best_idx)) {
            best = d;
            best_idx = l;
            // the float sums round at every frame: 1e-4 covers more frames than a chunk has
            skip_above = (double)best * (1.0 + 1e-4);
          }
        }
        ids.resize(w);
      }
      return best_idx;
    };
    // pass 1: small or overlapping tubes join their closest tube (:795-816)
    for (int k = nxt[T]; k != T;) {
      bool merge = MeanSliceSize(done[k]) < 20;
      if (!merge) {
        // a tube without a common frame counts as overlapping (the fraction of no frames is
        // max): look for one whose frames lie before or after this tube's first
        const int k0 = done[k][0].frame, k1 = done[k].back().frame;
        for (int l = nxt[T]; l != T && !merge; l = nxt[l]) {
          merge = l != k && (done[l].back().frame < k0 || done[l][0].frame > k1);
        }
      }
      if (!merge) {
        for (int l = nxt[T]; l != T; l = nxt[l]) {
          if (l != k && (double)BoxOverlapFraction(done[k], done[l]) > 0.8) {
            merge = true;
            break;
          }
        }
      }
      const int after = nxt[k];
      if (merge) {
        const int idx = closest(k);
        if (idx >= 0) {
          Tube j;
          gained.clear();
          JoinTubes(&done[idx], &done[k], &j, &gained);
          done[idx].swap(j);
          for (int f : gained) in_frame[(size_t)(f - frame_lo)].push_back(idx);
          note_centres(idx);
          unlink(k);
        }
      }
      k = after;
    }
    // pass 2: tubes abutting in time (:819-839)
    for (int k = nxt[T]; k != T;) {
      const int after = nxt[k];
      for (int l = nxt[T]; l != T; l = nxt[l]) {
        if (l == k) continue;
        if (TemporalNeighbors(done[k], done[l])) {
          Tube j;
          JoinTubes(&done[k], &done[l], &j, nullptr);
          done[l].swap(j);
          unlink(k);
          break;
        }
      }
      k = after;
    }
    size_t w = 0;
    for (int k = nxt[T]; k != T; k = nxt[k]) {
      if ((size_t)k != w) done[w].swap(done[k]);
      ++w;
    }
    done.resize(w);
  }

  // largest tube keeps the region (:841-861)
  int keep = -1, keep_score = 0;
  out->areas.resize(done.size());
  for (int k = 0; k < (int)done.size(); ++k) {
    float area = 0;
    for (const TSlice& s : done[k]) area += (float)s.shape.size;
    out->areas[k] = area;
    if (area > (float)keep_score) {
      keep_score = (int)area;
      keep = k;
    }
  }
  out->tube_to_keep = keep;
  out->tubes.resize(done.size());
  for (size_t k = 0; k < done.size(); ++k) {
    for (TSlice& s : done[k]) {
      out->tubes[k].push_back(RasterSlice{s.frame, Raster()});
      out->tubes[k].back().raster.swap(s.raster);
    }
  }
  impl_->slices.clear();
}

void SplitRegionIntoTubes(const Raster3D& raster, int W, int H,
                          const std::vector<const float*>& flows, bool have_flows,
                          TubeResult* out) {
  TubeSplitter ts;
  std::vector<FlowRequest> req;
  ts.Prepare(raster, &req);
  std::vector<float> samples;
  if (have_flows) {
    samples.resize(2 * req.size());
    for (size_t i = 0; i < req.size(); ++i) {
      const float* fp = flows[(size_t)req[i].frame] + ((size_t)req[i].y * W + req[i].x) * 2;
      samples[2 * i] = fp[0];
      samples[2 * i + 1] = fp[1];
    }
  }
  ts.Finish(W, H, have_flows ? samples.data() : nullptr, out);
}

// ---------------------------------------------------------------------------------------
// SegmentationDesc wire encoding (segment_util/segmentation.proto).
// ---------------------------------------------------------------------------------------
namespace {
inline void PutVarint(std::string* s, uint64_t v) {
  while (v >= 0x80) {
    s->push_back((char)((v & 0x7f) | 0x80));
    v >>= 7;
  }
  s->push_back((char)v);
}
inline void PutInt(std::string* s, int field, int32_t v) {
  PutVarint(s, ((uint64_t)field << 3) | 0);
  PutVarint(s, (uint64_t)(int64_t)v);
}
inline void PutFloat(std::string* s, int field, float f) {
  PutVarint(s, ((uint64_t)field << 3) | 5);
  uint32_t u;
  std::memcpy(&u, &f, 4);
  char b[4] = {(char)(u & 0xff), (char)((u >> 8) & 0xff), (char)((u >> 16) & 0xff), (char)(u >> 24)};
  s->append(b, 4);
}
inline void PutMsg(std::string* s, int field, const std::string& m) {
  PutVarint(s, ((uint64_t)field << 3) | 2);
  PutVarint(s, m.size());
  s->append(m);
}
inline size_t VarintLen(uint64_t v) {
  size_t n = 1;
  while (v >= 0x80) {
    v >>= 7;
    ++n;
  }
  return n;
}
}  // namespace

std::string EncodeSegDesc(const SegDesc& d) {
  std::string out;
  std::string region, raster, tmp;
  for (const Region2DOut& r : d.regions) {
    region.clear();
    PutInt(&region, 1, r.id);                       // Region2D.id = 1
    raster.clear();
    for (const Interval& iv : r.raster) {           // Rasterization.scan_inter = 1
      tmp.clear();
      PutInt(&tmp, 1, iv.y);
      PutInt(&tmp, 2, iv.lx);
      PutInt(&tmp, 3, iv.rx);
      PutMsg(&raster, 1, tmp);
    }
    PutMsg(&region, 3, raster);                     // Region2D.raster = 3
    tmp.clear();
    PutFloat(&tmp, 1, r.moments.size);
    PutFloat(&tmp, 2, r.moments.mean_x);
    PutFloat(&tmp, 3, r.moments.mean_y);
    PutFloat(&tmp, 4, r.moments.xx);
    PutFloat(&tmp, 5, r.moments.xy);
    PutFloat(&tmp, 6, r.moments.yy);
    PutMsg(&region, 5, tmp);                        // Region2D.shape_moments = 5
    if (!r.polygons.empty()) {                      // Region2D.vectorization = 6
      std::string vec, poly, packed;
      for (const PolygonOut& pg : r.polygons) {
        poly.clear();
        packed.clear();
        for (int idx : pg.coord_idx) PutVarint(&packed, (uint64_t)(int64_t)idx);
        if (!packed.empty()) PutMsg(&poly, 1, packed);   // Polygon.coord_idx = 1 [packed]
        PutVarint(&poly, ((uint64_t)2 << 3) | 0);       // Polygon.hole = 2 (always set)
        PutVarint(&poly, pg.hole ? 1 : 0);
        PutMsg(&vec, 1, poly);                          // Vectorization.polygon = 1
      }
      PutMsg(&region, 6, vec);
    }
    PutMsg(&out, 2, region);                        // SegmentationDesc.region = 2
  }
  if (d.has_hierarchy) {
    std::string level, c;
    auto put_level = [&](const std::vector<CompoundOut>& regions) {
      level.clear();
      for (const CompoundOut& cr : regions) {
        c.clear();
        PutInt(&c, 1, cr.id);
        PutInt(&c, 2, cr.size);
        for (int n : cr.neighbor_ids) PutInt(&c, 3, n);
        if (cr.has_parent) PutInt(&c, 4, cr.parent_id);
        for (int n : cr.child_ids) PutInt(&c, 5, n);
        PutInt(&c, 6, cr.start_frame);
        PutInt(&c, 7, cr.end_frame);
        PutMsg(&level, 2, c);                       // HierarchyLevel.region = 2
      }
      PutMsg(&out, 3, level);                       // SegmentationDesc.hierarchy = 3
    };
    put_level(d.hierarchy0);
    for (const auto& lv : d.upper_levels) put_level(lv);
  }
  PutInt(&out, 4, d.frame_width);
  PutInt(&out, 5, d.frame_height);
  PutInt(&out, 6, d.chunk_size);
  PutInt(&out, 7, d.overlap_start);
  PutInt(&out, 8, d.chunk_id);
  PutInt(&out, 9, d.hierarchy_frame_idx);
  for (uint32_t id : d.feature_ids) {               // SegmentationDesc.features = 10
    std::string f;
    PutVarint(&f, ((uint64_t)1 << 3) | 5);          // RegionFeatures.id = 1, fixed32
    for (int i = 0; i < 4; ++i) f.push_back((char)((id >> (8 * i)) & 0xff));
    PutMsg(&out, 10, f);
  }
  if (d.has_vector_mesh) {                          // SegmentationDesc.vector_mesh = 11
    std::string mesh;
    if (!d.vector_mesh.empty()) {                   // VectorMesh.coord = 1 [packed]
      std::string packed((const char*)d.vector_mesh.data(), d.vector_mesh.size() * sizeof(float));
      PutMsg(&mesh, 1, packed);
    }
    PutMsg(&out, 11, mesh);
  }
  PutInt(&out, 12, d.connectedness);
  (void)VarintLen;
  return out;
}

namespace {
// Cursor over a proto2 message.
struct WireIn {
  const uint8_t* p;
  const uint8_t* end;
  bool bad = false;
  bool more() const { return !bad && p < end; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; p < end && shift < 64; shift += 7) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    bad = true;
    return 0;
  }
  WireIn sub() {
    const uint64_t n = varint();
    if (bad || n > (uint64_t)(end - p)) {
      bad = true;
      return WireIn{p, p, true};
    }
    WireIn s{p, p + n, false};
    p += n;
    return s;
  }
  void skip(int wire_type) {
    switch (wire_type) {
      case 0: (void)varint(); break;
      case 1: if (end - p < 8) bad = true; else p += 8; break;
      case 2: (void)sub(); break;
      case 5: if (end - p < 4) bad = true; else p += 4; break;
      default: bad = true;
    }
  }
};
}  // namespace

bool DecodeSegDesc(const uint8_t* data, size_t len, SegDesc* d) {
  WireIn in{data, data + len};
  bool first_level = true;
  while (in.more()) {
    const uint64_t tag = in.varint();
    const int field = (int)(tag >> 3), wt = (int)(tag & 7);
    if (field == 2 && wt == 2) {                      // Region2D
      WireIn r = in.sub();
      d->regions.emplace_back();
      Region2DOut& reg = d->regions.back();
      while (r.more()) {
        const uint64_t t = r.varint();
        if ((t >> 3) == 1 && (t & 7) == 0) {
          reg.id = (int)(int64_t)r.varint();
        } else if ((t >> 3) == 3 && (t & 7) == 2) {   // Rasterization
          WireIn ra = r.sub();
          while (ra.more()) {
            const uint64_t t2 = ra.varint();
            if ((t2 >> 3) != 1 || (t2 & 7) != 2) {
              ra.skip((int)(t2 & 7));
              continue;
            }
            WireIn si = ra.sub();
            Interval iv{0, 0, 0};
            while (si.more()) {
              const uint64_t t3 = si.varint();
              if ((t3 & 7) != 0) {
                si.skip((int)(t3 & 7));
                continue;
              }
              const int v = (int)(int64_t)si.varint();
              if ((t3 >> 3) == 1) iv.y = v; else if ((t3 >> 3) == 2) iv.lx = v; else if ((t3 >> 3) == 3) iv.rx = v;
            }
            ra.bad = ra.bad || si.bad;
            reg.raster.push_back(iv);
          }
          r.bad = r.bad || ra.bad;
        } else {
          r.skip((int)(t & 7));
        }
      }
      in.bad = in.bad || r.bad;
    } else if (field == 3 && wt == 2) {               // HierarchyLevel
      WireIn h = in.sub();
      std::vector<CompoundOut>* level;
      if (first_level) {
        d->has_hierarchy = true;
        level = &d->hierarchy0;
        first_level = false;
      } else {
        d->upper_levels.emplace_back();
        level = &d->upper_levels.back();
      }
      while (h.more()) {
        const uint64_t t = h.varint();
        if ((t >> 3) != 2 || (t & 7) != 2) {
          h.skip((int)(t & 7));
          continue;
        }
        WireIn c = h.sub();
        level->emplace_back();
        CompoundOut& cr = level->back();
        while (c.more()) {
          const uint64_t t2 = c.varint();
          if ((t2 & 7) != 0) {
            c.skip((int)(t2 & 7));
            continue;
          }
          const int v = (int)(int64_t)c.varint();
          switch ((int)(t2 >> 3)) {
            case 1: cr.id = v; break;
            case 2: cr.size = v; break;
            case 3: cr.neighbor_ids.push_back(v); break;
            case 4: cr.has_parent = true; cr.parent_id = v; break;
            case 5: cr.child_ids.push_back(v); break;
            case 6: cr.start_frame = v; break;
            case 7: cr.end_frame = v; break;
            default: break;
          }
        }
        h.bad = h.bad || c.bad;
      }
      in.bad = in.bad || h.bad;
    } else if (wt == 0) {
      const int v = (int)(int64_t)in.varint();
      switch (field) {
        case 4: d->frame_width = v; break;
        case 5: d->frame_height = v; break;
        case 6: d->chunk_size = v; break;
        case 7: d->overlap_start = v; break;
        case 8: d->chunk_id = v; break;
        case 9: d->hierarchy_frame_idx = v; break;
        case 12: d->connectedness = v; break;
        default: break;
      }
    } else {
      in.skip(wt);
    }
  }
  return !in.bad;
}

void RenderIdImage(const SegDesc& d, int W, int32_t* out) {
  for (const Region2DOut& r : d.regions) {
    for (const Interval& iv : r.raster) {
      int32_t* p = out + (size_t)iv.y * W;
      for (int x = iv.lx; x <= iv.rx; ++x) p[x] = r.id;
    }
  }
}

}  // namespace vsg

#ifdef VSG_TEST_MODELS
// tests/test_tube_analysis.py compiles this file with the plain restatement of Finish beside it
#include "../../tests/host/tube_plain_model.inc"
#endif
