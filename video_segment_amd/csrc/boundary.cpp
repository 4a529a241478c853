// boundary.cpp -- joint boundaries of a frame's regions and their vectorization (host, C++).
//
// Reference behaviour restated: segmentation/boundary.h:40-175, segmentation/boundary.cpp:52-640
// (BoundaryComputation::ComputeBoundary / TraceBoundary / NextDirection / VertexOrder /
// SetSegmentRegions / ComputeVectorization, BoundarySegmentKey), called from
// Segmentation::RetrieveSegmentation3D when compute_vectorization is set
// (segmentation/segmentation.cpp:527-532: min_hole_length 10, min_segment_length 4, max_error 1.0),
// i.e. what `seg_tree_sample --over_segment` asks of the dense unit (seg_tree.cpp:202-204).
// The tracing scheme is Liow's common-boundary contour tracing (CVGIP 53(3), 1991): boundaries
// live on the pixel *corners* [0, W] x [0, H]; a boundary is cut into segments at the corners
// where three or four regions (or the frame) meet, so that the segment between two regions is
// traced -- and later simplified -- once and shared by both.
//
// Two things this path inherits from outside the reference tree (parity unpinned):
//  * cv::approxPolyDP (OpenCV 2.4.x, un-vendored): the Douglas-Peucker variant below restates
//    the published algorithm of that release for integer points -- start point from three
//    farthest-point sweeps for closed curves, explicit stack, the final pass that drops points on
//    almost straight joints;
//  * the order in which unmatched hole segments are traced is the iteration order of an
//    std::unordered_map with the reference's hasher and bucket hint; the same container is used
//    here, so the order is the platform's, as it would be for the reference built on it.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <unordered_map>

#include "common.h"
#include "host_model.h"

namespace vsg {

namespace {

struct Pt {
  int x = 0, y = 0;
  bool operator==(const Pt& o) const { return x == o.x && y == o.y; }
  bool operator!=(const Pt& o) const { return !(*this == o); }
};

struct BSegment {
  Pt start, end;
  int start_order = 0;
  int left_region = -1, right_region = -1;
  std::vector<Pt> points;   // from start to end, both included
};

struct Boundary {
  std::vector<BSegment> segments;
  int region = -1;
  bool is_hole = false;
  bool IsSimple() const { return segments.size() == 1 && segments[0].start_order == 1; }
  int Length() const {
    int n = 0;
    for (const BSegment& s : segments) n += (int)s.points.size() - 1;
    return n;
  }
};

// Key of a segment shared by two regions (boundary.cpp:611-633): end points in lexicographic
// order with the regions swapped accordingly; closed segments order their two regions.
struct SegmentKey {
  Pt p1, p2;
  int r1 = -1, r2 = -1;
  explicit SegmentKey(const BSegment& s) {
    if (s.start.x < s.end.x || (s.start.x == s.end.x && s.start.y < s.end.y)) {
      p1 = s.start;
      p2 = s.end;
      r1 = s.left_region;
      r2 = s.right_region;
    } else if (s.start == s.end) {
      p1 = p2 = s.start;
      r1 = std::min(s.left_region, s.right_region);
      r2 = std::max(s.left_region, s.right_region);
    } else {
      p1 = s.end;
      p2 = s.start;
      r1 = s.right_region;
      r2 = s.left_region;
    }
  }
  bool operator==(const SegmentKey& o) const {
    return p1 == o.p1 && p2 == o.p2 && r1 == o.r1 && r2 == o.r2;
  }
};

struct SegmentKeyHasher {   // boundary.h:186-197
  explicit SegmentKeyHasher(int frame_width) : frame_width_(frame_width) {}
  size_t operator()(const SegmentKey& k) const {
    return (size_t)((k.p1.y * frame_width_ + k.p1.x) * 10 + (k.r1 % 7 + k.r2 % 3));
  }
  int frame_width_;
};

// Freeman directions: 3 2 1 / 4 X 0 / 5 6 7.  Only the four axis directions occur (N4 input).
enum Dir { D_R = 0, D_TR = 1, D_T = 2, D_TL = 3, D_L = 4, D_BL = 5, D_B = 6, D_BR = 7 };
const int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

Dir VectorToDir(int dx, int dy) {
  for (int d = 0; d < 8; ++d) {
    if (kDx[d] == dx && kDy[d] == dy) return (Dir)d;
  }
  Throw(-4, "boundary: unexpected step");
}

// ---- cv::approxPolyDP for integer points (OpenCV 2.4.x, restated; see the file header) --------
void ApproxPolyDP(const std::vector<Pt>& src, double eps, bool closed, std::vector<Pt>* dst) {
  dst->clear();
  const int count = (int)src.size();
  if (count == 0) return;
  struct Slice {
    int start, end;
  };
  std::vector<Slice> stack;
  auto at = [&](int i) -> const Pt& { return src[(size_t)(i % count)]; };
  eps *= eps;
  bool is_closed = closed;
  int init_iters = 3;
  Slice slice{0, 0}, right{0, 0};
  Pt start_pt{-1000000, -1000000}, end_pt, pt;
  bool le_eps = false;
  int pos = 0;   // reader position

  if (!is_closed) {
    right.start = count;
    end_pt = src[0];
    start_pt = src[(size_t)count - 1];
    if (start_pt != end_pt) {
      slice.start = 0;
      slice.end = count - 1;
      stack.push_back(slice);
    } else {
      is_closed = true;
      init_iters = 1;
    }
  }
  if (is_closed) {
    // 1. approximately the two farthest points of the contour
    right.start = 0;
    for (int i = 0; i < init_iters; ++i) {
      double max_dist = 0;
      pos = right.start % count;
      start_pt = at(pos);
      ++pos;
      for (int j = 1; j < count; ++j) {
        pt = at(pos);
        ++pos;
        const double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
        const double dist = dx * dx + dy * dy;
        if (dist > max_dist) {
          max_dist = dist;
          right.start = j;
        }
      }
      le_eps = max_dist <= eps;
      pos %= count;   // the reader is back where the sweep started
      right.start = (right.start);   // offset relative to the sweep's start point (made absolute below)
      if (i + 1 < init_iters) right.start = (pos + right.start) % count;
    }
    // 2. the stack
    if (!le_eps) {
      slice.start = pos;
      slice.end = right.start += slice.start;
      right.start -= right.start >= count ? count : 0;
      right.end = slice.start;
      if (right.end < right.start) right.end += count;
      stack.push_back(right);
      stack.push_back(slice);
    } else {
      dst->push_back(start_pt);
    }
  }
  // 3. the recursion
  while (!stack.empty()) {
    slice = stack.back();
    stack.pop_back();
    end_pt = at(slice.end);
    start_pt = at(slice.start);
    if (slice.end > slice.start + 1) {
      double max_dist = 0;
      const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
      for (int i = slice.start + 1; i < slice.end; ++i) {
        pt = at(i);
        const double dist = std::fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
        if (dist > max_dist) {
          max_dist = dist;
          right.start = i;
        }
      }
      le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
    } else {
      le_eps = true;
    }
    if (le_eps) {
      dst->push_back(start_pt);
    } else {
      right.end = slice.end;
      slice.end = right.start;
      stack.push_back(right);
      stack.push_back(slice);
    }
  }
  if (!closed) dst->push_back(end_pt);

  // last stage: drop points on [almost] straight joints (in place: reads run ahead of the
  // writes, and for a closed contour the final read wraps around to what was written first)
  std::vector<Pt>& d = *dst;
  const int cnt = (int)d.size();
  int new_count = cnt;
  int rd = closed ? cnt - 1 : 0;
  auto read = [&](int& r) -> Pt {
    const Pt v = d[(size_t)r];
    if (++r >= cnt) r = 0;
    return v;
  };
  start_pt = read(rd);
  int wr = rd;
  pt = read(rd);
  for (int i = !closed; i < cnt - !closed && new_count > 2; ++i) {
    end_pt = read(rd);
    const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
    const double dist = std::fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
    const double successive_inner_product =
        (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
    if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 &&
        successive_inner_product >= 0) {
      --new_count;
      d[(size_t)wr] = start_pt = end_pt;
      if (++wr >= cnt) wr = 0;
      pt = read(rd);
      ++i;
      continue;
    }
    d[(size_t)wr] = start_pt = pt;
    if (++wr >= cnt) wr = 0;
    pt = end_pt;
  }
  if (!closed) d[(size_t)wr] = pt;
  if (new_count < cnt) d.resize((size_t)new_count);
}

// N8 connected components of a rasterization, ordered by first interval
// (segment_util/segmentation_util.cpp:1009-1101 with N8_CONNECT).
void SplitComponentsN8(const Raster& r, std::vector<Raster>* comps) {
  const int n = (int)r.size();
  std::vector<int> parent((size_t)n);
  auto find = [&](int i) {
    while (parent[(size_t)i] != i) i = parent[(size_t)i] = parent[(size_t)parent[(size_t)i]];
    return i;
  };
  int last_change = -1, last_y = -2, test_idx = 0;
  for (int i = 0; i < n; ++i) {
    parent[(size_t)i] = i;
    if (r[(size_t)i].y != last_y) {
      test_idx = (last_y + 1 == r[(size_t)i].y) ? last_change : i;
      last_y = r[(size_t)i].y;
      last_change = i;
    }
    for (int k = test_idx; k < i; ++k) {
      const Interval &a = r[(size_t)i], &b = r[(size_t)k];
      if (std::abs(a.y - b.y) <= 1 && std::max(a.lx, b.lx) - std::min(a.rx, b.rx) <= 1) {
        const int ra = find(i), rb = find(k);
        if (ra != rb) parent[(size_t)ra] = rb;
      }
    }
  }
  comps->clear();
  std::unordered_map<int, int> rep_to_comp;
  for (int i = 0; i < n; ++i) {
    const int rep = find(i);
    auto it = rep_to_comp.find(rep);
    if (it == rep_to_comp.end()) {
      rep_to_comp[rep] = (int)comps->size();
      comps->push_back(Raster{r[(size_t)i]});
    } else {
      (*comps)[(size_t)it->second].push_back(r[(size_t)i]);
    }
  }
}

class BoundaryComputation {
 public:
  BoundaryComputation(int W, int H, int min_hole_length)
      : W_(W), H_(H), lda_(W + 2), min_hole_length_(min_hole_length),
        ids_((size_t)(W + 2) * (H + 2), -1) {
    for (int d = 0; d < 8; ++d) off_[d] = kDx[d] + kDy[d] * lda_;
  }

  void ComputeBoundary(const SegDesc& seg, std::vector<Boundary>* boundaries) {
    VSG_REQUIRE(seg.connectedness == 1, -1, "Requires N4 connected segmentation.");
    for (const Region2DOut& r : seg.regions) {
      for (const Interval& iv : r.raster) {
        int32_t* row = &ids_[(size_t)(iv.y + 1) * lda_ + 1];
        for (int x = iv.lx; x <= iv.rx; ++x) row[x] = r.id;
      }
    }
    for (const Region2DOut& r : seg.regions) {
      std::vector<Raster> comps;
      SplitComponentsN8(r.raster, &comps);
      for (const Raster& comp : comps) {
        const Pt start{comp[0].lx, comp[0].y};   // top-left pixel of the component
        Boundary b;
        TraceBoundary(r.id, start, D_B, &b);
        if (b.IsSimple() && b.Length() < min_hole_length_) continue;   // small holes
        boundaries->push_back(std::move(b));
      }
    }
    // Holes: a segment that only one boundary produced belongs to a hole of the region on its
    // right.  The map is the reference's (same hasher, same bucket hint); see the file header.
    typedef std::unordered_map<SegmentKey, std::pair<int, int>, SegmentKeyHasher> Hash;
    Hash hash(boundaries->size() * 20, SegmentKeyHasher(W_));
    const std::pair<int, int> kNone(-1, -1);
    for (int bi = 0; bi < (int)boundaries->size(); ++bi) {
      const Boundary& b = (*boundaries)[(size_t)bi];
      for (int si = 0; si < (int)b.segments.size(); ++si) {
        const BSegment& s = b.segments[(size_t)si];
        if (s.points.size() < 3 || IsFrameSegment(s)) continue;
        const SegmentKey key(s);
        auto it = hash.find(key);
        if (it == hash.end()) hash[key] = std::make_pair(bi, si);
        else hash[key] = kNone;
      }
    }
    // Hole boundaries are appended while the map is walked; entries refer to boundaries by index.
    for (auto& elem : hash) {
      if (elem.second == kNone) continue;
      const BSegment s = (*boundaries)[(size_t)elem.second.first].segments[(size_t)elem.second.second];
      Boundary hole;
      const Pt& last = s.points.back();
      const Pt& before = s.points[s.points.size() - 2];
      TraceBoundary(s.right_region, last, VectorToDir(before.x - last.x, before.y - last.y), &hole);
      hole.is_hole = true;
      for (const BSegment& hs : hole.segments) {
        if (hs.points.size() < 3) continue;
        auto it = hash.find(SegmentKey(hs));
        if (it != hash.end()) it->second = kNone;
      }
      boundaries->push_back(std::move(hole));
    }
  }

  void ComputeVectorization(const std::vector<Boundary>& boundaries, int min_segment_length,
                            float max_error, SegDesc* seg) {
    std::vector<std::vector<Pt>> polygon_segments;
    polygon_segments.reserve(20 * boundaries.size());
    std::unordered_map<SegmentKey, int, SegmentKeyHasher> seg_hash(boundaries.size() * 20,
                                                                  SegmentKeyHasher(W_));
    min_segment_length = std::max(3, min_segment_length);
    std::unordered_map<long long, int> mesh_index;   // point -> index into vector_mesh
    seg->has_vector_mesh = true;
    for (const Boundary& b : boundaries) {
      std::vector<Pt> polygon;
      polygon.reserve((size_t)b.Length());
      for (const BSegment& s : b.segments) {
        const bool is_closed = s.start == s.end;
        if (!is_closed && (int)s.points.size() < min_segment_length) {
          polygon.push_back(s.points[0]);   // collapsed to its start
          continue;
        }
        const SegmentKey key(s);
        auto pos = seg_hash.find(key);
        if (pos == seg_hash.end()) {
          std::vector<Pt> result;
          ApproxPolyDP(s.points, (double)max_error, is_closed, &result);
          if (is_closed) result.push_back(result[0]);
          polygon.insert(polygon.end(), result.begin(), result.end() - 1);
          polygon_segments.push_back(result);
          seg_hash[key] = (int)polygon_segments.size() - 1;
        } else {   // the neighbour's polyline, walked the other way
          const std::vector<Pt>& ps = polygon_segments[(size_t)pos->second];
          polygon.insert(polygon.end(), ps.rbegin(), ps.rend() - 1);
        }
      }
      polygon.push_back(polygon[0]);
      if (polygon.size() == 3 && polygon[0] == polygon[2]) continue;   // no interior
      // GetMutableRegion2DFromId: lower_bound over the regions (sorted by id)
      auto rit = std::lower_bound(seg->regions.begin(), seg->regions.end(), b.region,
                                  [](const Region2DOut& r, int id) { return r.id < id; });
      VSG_REQUIRE(rit != seg->regions.end() && rit->id == b.region, -4, "boundary of an unknown region");
      rit->polygons.emplace_back();
      PolygonOut& poly = rit->polygons.back();
      poly.hole = b.is_hole;
      for (const Pt& pt : polygon) {
        const long long k = (long long)pt.y * (W_ + 1) + pt.x;
        auto mi = mesh_index.find(k);
        if (mi != mesh_index.end()) {
          poly.coord_idx.push_back(mi->second);
        } else {
          const int idx = (int)seg->vector_mesh.size();
          seg->vector_mesh.push_back((float)pt.x);
          seg->vector_mesh.push_back((float)pt.y);
          poly.coord_idx.push_back(idx);
          mesh_index[k] = idx;
        }
      }
    }
  }

 private:
  const int32_t* At(const Pt& p) const { return &ids_[(size_t)(p.y + 1) * lda_ + p.x + 1]; }

  bool IsFrameSegment(const BSegment& s) const {
    for (const Pt& p : s.points) {
      if (!(p.x == 0 || p.y == 0 || p.x == W_ || p.y == H_)) return false;
    }
    return true;
  }

  // Number of boundaries meeting at a corner (boundary.cpp:420-452).
  int VertexOrder(const int32_t* c) const {
    const int curr = c[0], left = c[off_[D_L]], top = c[off_[D_T]], top_left = c[off_[D_TL]];
    if (curr < 0) {
      if (left >= 0) return left != top_left ? 2 : 1;   // right border
      return top_left != top ? 2 : 1;                   // bottom border
    } else if (left < 0) {
      return top != curr ? 2 : 1;
    } else if (top < 0) {
      return left != curr ? 2 : 1;
    }
    const int changes = (int)(curr != left) + (int)(left != top_left) + (int)(top_left != top) +
                        (int)(top != curr);
    return changes > 2 ? changes : 1;
  }

  void SetSegmentRegions(const int32_t* c, Dir prev, BSegment* s) const {   // boundary.cpp:454-482
    switch (prev) {
      case D_R: s->left_region = c[off_[D_TL]]; s->right_region = c[off_[D_L]]; break;
      case D_T: s->left_region = c[off_[D_L]]; s->right_region = c[0]; break;
      case D_L: s->left_region = c[0]; s->right_region = c[off_[D_T]]; break;
      case D_B: s->left_region = c[off_[D_T]]; s->right_region = c[off_[D_TL]]; break;
      default: Throw(-4, "boundary: unexpected direction for N4 trace");
    }
  }

  Dir NextDirection(const int32_t* c, Dir prev, int id) const {   // boundary.cpp:355-418
    switch (prev) {
      case D_R:
        if (c[off_[D_T]] != id) return D_T;
        if (c[0] != id) return D_R;
        return D_B;
      case D_T:
        if (c[off_[D_TL]] == id) return c[off_[D_T]] == id ? D_R : D_T;
        return D_L;
      case D_L:
        if (c[off_[D_L]] == id) return c[off_[D_TL]] != id ? D_L : D_T;
        return D_B;
      case D_B:
        if (c[0] == id) return c[off_[D_L]] != id ? D_B : D_L;
        return D_R;
      default:
        Throw(-4, "boundary: unexpected direction for N4 trace");
    }
  }

  void TraceBoundary(int region_id, const Pt& start_pt, Dir dir, Boundary* boundary) const {
    boundary->region = region_id;
    const int32_t* cur = At(start_pt);
    BSegment seg;
    seg.start = start_pt;
    seg.start_order = VertexOrder(cur);
    seg.points.push_back(start_pt);
    Pt cp{start_pt.x + kDx[dir], start_pt.y + kDy[dir]};
    cur += off_[dir];
    seg.points.push_back(cp);
    // A corner of order 4 is passed twice: stop only when the trace would repeat its first step.
    const int32_t* termination = seg.start_order == 4 ? cur : nullptr;
    Dir prev = dir;
    while (cp != start_pt ||
           (termination && cur + off_[NextDirection(cur, prev, region_id)] != termination)) {
      const int order = VertexOrder(cur);
      if (order > 1) {
        seg.end = cp;
        boundary->segments.push_back(seg);
        seg = BSegment();
        seg.start = cp;
        seg.start_order = order;
        seg.points.push_back(cp);
      } else {
        SetSegmentRegions(cur, prev, &seg);
        VSG_REQUIRE(seg.left_region == region_id && seg.right_region != region_id, -4,
                    "boundary: the traced region is not on the left");
      }
      const Dir nd = NextDirection(cur, prev, region_id);
      cp.x += kDx[nd];
      cp.y += kDy[nd];
      cur += off_[nd];
      seg.points.push_back(cp);
      prev = nd;
    }
    seg.end = cp;
    boundary->segments.push_back(seg);
    // The start (top-left pixel of the component) need not be a real vertex: join the last
    // segment with the first one.
    if (boundary->segments.size() > 1 && boundary->segments[0].start_order < 2) {
      BSegment& first = boundary->segments[0];
      const BSegment& last = boundary->segments.back();
      first.start = last.start;
      first.start_order = last.start_order;
      first.points.insert(first.points.begin(), last.points.begin(), last.points.end() - 1);
      boundary->segments.pop_back();
      BSegment& f = boundary->segments[0];
      VSG_REQUIRE(f.points.size() >= 3, -4, "boundary: joined segment too short");
      const Dir d = VectorToDir(f.points[1].x - f.points[0].x, f.points[1].y - f.points[0].y);
      SetSegmentRegions(At(f.points[0]) + off_[d], d, &f);
    }
  }

  int W_, H_, lda_, min_hole_length_;
  std::vector<int32_t> ids_;   // region ids with a one pixel border of -1
  int off_[8];
};

}  // namespace

// Segmentation::RetrieveSegmentation3D, segmentation.cpp:527-532.
void ComputeFrameVectorization(SegDesc* desc) {
  BoundaryComputation bc(desc->frame_width, desc->frame_height, 10);
  std::vector<Boundary> boundaries;
  bc.ComputeBoundary(*desc, &boundaries);
  bc.ComputeVectorization(boundaries, 4, 1.0f, desc);
}

}  // namespace vsg
