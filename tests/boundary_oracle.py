"""CPU oracle (test infrastructure only) for the boundary vectorisation of the dense unit.

Plain-Python restatement of the reference's BoundaryComputation (segmentation/boundary.cpp:
ComputeBoundary :121-244, TraceBoundary :246-352, NextDirection :354-418, VertexOrder :420-452,
SetSegmentRegions :454-482, ComputeVectorization :514-608, BoundarySegmentKey :611-633) called
with min_hole_length 10, min_segment_length 4, max_error 1.0 (segmentation/segmentation.cpp:
527-532), plus OpenCV 2.4's approxPolyDP for integer points (un-vendored third-party code,
restated from the published algorithm: PARITY UNPINNED).

It works on an id image and returns geometry, not bytes: {region id: sorted [(is_hole,
((x, y), ...)), ...]}.  The order of a region's polygons and the numbering of the shared vector
mesh depend on the iteration order of a std::unordered_map in the reference and are deliberately
left out of the comparison; the polygons themselves do not depend on it (a hole's segments are
always copies of polylines an outer boundary has produced before).
"""
import numpy as np

R, TR, T, TL, L, BL, B, BR = range(8)
DX = [1, 1, 0, -1, -1, -1, 0, 1]
DY = [0, -1, -1, -1, 0, 1, 1, 1]


def _dir(dx, dy):
    for d in range(8):
        if DX[d] == dx and DY[d] == dy:
            return d
    raise AssertionError((dx, dy))


def approx_poly_dp(src, eps, closed):
    count = len(src)
    if count == 0:
        return []
    dst = []
    eps = eps * eps
    stack = []
    at = lambda i: src[i % count]  # noqa: E731
    is_closed = closed
    init_iters = 3
    right_start = 0
    le_eps = False
    pos = 0
    start_pt = end_pt = None
    if not is_closed:
        right_start = count
        end_pt = src[0]
        start_pt = src[-1]
        if start_pt != end_pt:
            stack.append((0, count - 1))
        else:
            is_closed = True
            init_iters = 1
    if is_closed:
        right_start = 0
        for _ in range(init_iters):
            pos = (pos + right_start) % count
            start_pt = at(pos)
            pos += 1
            max_dist = 0.0
            for j in range(1, count):
                pt = at(pos)
                pos += 1
                dx, dy = float(pt[0] - start_pt[0]), float(pt[1] - start_pt[1])
                dist = dx * dx + dy * dy
                if dist > max_dist:
                    max_dist = dist
                    right_start = j
            le_eps = max_dist <= eps
            pos %= count
        if not le_eps:
            s_start = pos
            s_end = right_start = right_start + s_start
            if right_start >= count:
                right_start -= count
            r_end = s_start
            if r_end < right_start:
                r_end += count
            stack.append((right_start, r_end))
            stack.append((s_start, s_end))
        else:
            dst.append(start_pt)
    while stack:
        s_start, s_end = stack.pop()
        end_pt = at(s_end)
        start_pt = at(s_start)
        if s_end > s_start + 1:
            dx, dy = float(end_pt[0] - start_pt[0]), float(end_pt[1] - start_pt[1])
            max_dist = 0.0
            split = s_start
            for i in range(s_start + 1, s_end):
                pt = at(i)
                dist = abs((pt[1] - start_pt[1]) * dx - (pt[0] - start_pt[0]) * dy)
                if dist > max_dist:
                    max_dist = dist
                    split = i
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy)
        else:
            le_eps = True
        if le_eps:
            dst.append(start_pt)
        else:
            stack.append((split, s_end))
            stack.append((s_start, split))
    if not closed:
        dst.append(end_pt)
    # clean-up of almost straight joints, in place
    cnt = len(dst)
    new_count = cnt
    rd = cnt - 1 if closed else 0

    def read():
        nonlocal rd
        v = dst[rd]
        rd += 1
        if rd >= cnt:
            rd = 0
        return v

    start_pt = read()
    wr = rd
    pt = read()
    i = 0 if closed else 1
    last = cnt if closed else cnt - 1
    while i < last and new_count > 2:
        end_pt = read()
        dx, dy = float(end_pt[0] - start_pt[0]), float(end_pt[1] - start_pt[1])
        dist = abs((pt[0] - start_pt[0]) * dy - (pt[1] - start_pt[1]) * dx)
        sip = float(pt[0] - start_pt[0]) * (end_pt[0] - pt[0]) + float(pt[1] - start_pt[1]) * (end_pt[1] - pt[1])
        if dist * dist <= 0.5 * eps * (dx * dx + dy * dy) and dx != 0 and dy != 0 and sip >= 0:
            new_count -= 1
            dst[wr] = start_pt = end_pt
            wr = (wr + 1) % cnt
            pt = read()
            i += 2
            continue
        dst[wr] = start_pt = pt
        wr = (wr + 1) % cnt
        pt = end_pt
        i += 1
    if not closed:
        dst[wr] = pt
    return dst[:new_count]


class _Seg:
    __slots__ = ("start", "end", "order", "left", "right", "pts")

    def __init__(self):
        self.start = self.end = None
        self.order = 0
        self.left = self.right = -1
        self.pts = []


def _key(s):
    a, b = s.start, s.end
    if a[0] < b[0] or (a[0] == b[0] and a[1] < b[1]):
        return (a, b, s.left, s.right)
    if a == b:
        return (a, b, min(s.left, s.right), max(s.left, s.right))
    return (b, a, s.right, s.left)


class Tracer:
    def __init__(self, ids):
        self.H, self.W = ids.shape
        self.im = np.full((self.H + 2, self.W + 2), -1, np.int64)
        self.im[1:-1, 1:-1] = ids

    def px(self, x, y, d=None):
        if d is not None:
            x, y = x + DX[d], y + DY[d]
        return int(self.im[y + 1, x + 1])

    def order(self, x, y):
        curr, left, top, tl = self.px(x, y), self.px(x, y, L), self.px(x, y, T), self.px(x, y, TL)
        if curr < 0:
            if left >= 0:
                return 2 if left != tl else 1
            return 2 if tl != top else 1
        if left < 0:
            return 2 if top != curr else 1
        if top < 0:
            return 2 if left != curr else 1
        ch = (curr != left) + (left != tl) + (tl != top) + (top != curr)
        return ch if ch > 2 else 1

    def regions(self, x, y, prev, seg):
        if prev == R:
            seg.left, seg.right = self.px(x, y, TL), self.px(x, y, L)
        elif prev == T:
            seg.left, seg.right = self.px(x, y, L), self.px(x, y)
        elif prev == L:
            seg.left, seg.right = self.px(x, y), self.px(x, y, T)
        elif prev == B:
            seg.left, seg.right = self.px(x, y, T), self.px(x, y, TL)
        else:
            raise AssertionError(prev)

    def nxt(self, x, y, prev, rid):
        if prev == R:
            if self.px(x, y, T) != rid:
                return T
            return R if self.px(x, y) != rid else B
        if prev == T:
            if self.px(x, y, TL) == rid:
                return R if self.px(x, y, T) == rid else T
            return L
        if prev == L:
            if self.px(x, y, L) == rid:
                return L if self.px(x, y, TL) != rid else T
            return B
        if prev == B:
            if self.px(x, y) == rid:
                return B if self.px(x, y, L) != rid else L
            return R
        raise AssertionError(prev)

    def trace(self, rid, start, d):
        segs = []
        seg = _Seg()
        seg.start = start
        seg.order = self.order(*start)
        seg.pts.append(start)
        cx, cy = start[0] + DX[d], start[1] + DY[d]
        seg.pts.append((cx, cy))
        term = (cx, cy) if seg.order == 4 else None
        prev = d

        def more():
            if (cx, cy) != start:
                return True
            if term is None:
                return False
            nd = self.nxt(cx, cy, prev, rid)
            return (cx + DX[nd], cy + DY[nd]) != term

        while more():
            o = self.order(cx, cy)
            if o > 1:
                seg.end = (cx, cy)
                segs.append(seg)
                seg = _Seg()
                seg.start = (cx, cy)
                seg.order = o
                seg.pts.append((cx, cy))
            else:
                self.regions(cx, cy, prev, seg)
                assert seg.left == rid and seg.right != rid
            nd = self.nxt(cx, cy, prev, rid)
            cx, cy = cx + DX[nd], cy + DY[nd]
            seg.pts.append((cx, cy))
            prev = nd
        seg.end = (cx, cy)
        segs.append(seg)
        if len(segs) > 1 and segs[0].order < 2:
            first, last = segs[0], segs[-1]
            first.start, first.order = last.start, last.order
            first.pts = last.pts[:-1] + first.pts
            segs.pop()
            p0, p1 = first.pts[0], first.pts[1]
            dd = _dir(p1[0] - p0[0], p1[1] - p0[1])
            self.regions(p0[0] + DX[dd], p0[1] + DY[dd], dd, first)
        return segs

    def frame_seg(self, s):
        return all(p[0] == 0 or p[1] == 0 or p[0] == self.W or p[1] == self.H for p in s.pts)


def _components_n8(runs):
    """runs: [(y, lx, rx)] sorted; N8 components ordered by first run."""
    n = len(runs)
    parent = list(range(n))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    for i in range(n):
        for k in range(i):
            a, b = runs[i], runs[k]
            if abs(a[0] - b[0]) <= 1 and max(a[1], b[1]) - min(a[2], b[2]) <= 1:
                ra, rb = find(i), find(k)
                if ra != rb:
                    parent[ra] = rb
    comps, where = [], {}
    for i in range(n):
        r = find(i)
        if r not in where:
            where[r] = len(comps)
            comps.append([])
        comps[where[r]].append(runs[i])
    return comps


def runs_of(ids):
    """{region id: [(y, lx, rx)]} in scan order."""
    H, W = ids.shape
    out = {}
    for y in range(H):
        row = ids[y]
        x = 0
        while x < W:
            v = int(row[x])
            x2 = x
            while x2 + 1 < W and int(row[x2 + 1]) == v:
                x2 += 1
            out.setdefault(v, []).append((y, x, x2))
            x = x2 + 1
    return out


def vectorize(ids, min_hole_length=10, min_segment_length=4, max_error=1.0):
    tr = Tracer(np.asarray(ids))
    runs = runs_of(np.asarray(ids))
    boundaries = []   # (region, is_hole, segs)
    for rid in sorted(runs):
        for comp in _components_n8(runs[rid]):
            segs = tr.trace(rid, (comp[0][1], comp[0][0]), B)
            simple = len(segs) == 1 and segs[0].order == 1
            if simple and sum(len(s.pts) - 1 for s in segs) < min_hole_length:
                continue
            boundaries.append((rid, False, segs))
    seen = {}
    for (_, _, segs) in boundaries:
        for s in segs:
            if len(s.pts) < 3 or tr.frame_seg(s):
                continue
            k = _key(s)
            seen[k] = s if k not in seen else None
    pending = [k for k in seen if seen[k] is not None]
    for k in pending:
        s = seen[k]
        if s is None:
            continue
        last, before = s.pts[-1], s.pts[-2]
        segs = tr.trace(s.right, last, _dir(before[0] - last[0], before[1] - last[1]))
        for hs in segs:
            if len(hs.pts) >= 3:
                hk = _key(hs)
                if hk in seen:
                    seen[hk] = None
        boundaries.append((s.right, True, segs))
    min_segment_length = max(3, min_segment_length)
    polylines = {}
    out = {}
    for (rid, hole, segs) in boundaries:
        polygon = []
        for s in segs:
            closed = s.start == s.end
            if not closed and len(s.pts) < min_segment_length:
                polygon.append(s.pts[0])
                continue
            k = _key(s)
            if k not in polylines:
                res = approx_poly_dp(list(s.pts), float(max_error), closed)
                if closed:
                    res = res + [res[0]]
                polygon += res[:-1]
                polylines[k] = res
            else:
                polygon += list(reversed(polylines[k]))[:-1]
        polygon.append(polygon[0])
        if len(polygon) == 3 and polygon[0] == polygon[2]:
            continue
        out.setdefault(rid, []).append((bool(hole), tuple(polygon)))
    return {k: sorted(v) for k, v in out.items()}
