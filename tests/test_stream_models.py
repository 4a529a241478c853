"""Model tests (numpy, no GPU, no library) of two data-parallel restatements of round 4
(video_segment_amd/csrc/merge_spine.hip):

* the STREAMED SPINE: the plain steps of a spine -- R's cluster absorbs one smaller side cluster
  after the other -- evaluated as the device does it (coefficients from an exclusive scan of the
  sizes, the f32 recurrence h = u + c*h in blocks of 16 with the mean after every block kept, every
  merge test re-checked per block from the mean before the block, the first failure cutting the run)
  against the sequential replay with the reference's merge rule: same cut, same mean bit for bit;
* SAMPLED LIST RANKING of the Euler tours: every 64th arc and every tour head walks its piece, the
  list of the splitters is ranked by pointer jumping, an arc's distance to the end of its tour is its
  splitter's minus its own number -- against a plain walk of the lists."""
import numpy as np
import pytest

f32 = np.float32
BLOCK = 16
SPLIT = 64


# ---- streamed spine ---------------------------------------------------------------------------------
def sequential_spine(h, hsz, pm, psz, thr):
    """MergeStates per step (segmentation_graph.h:671-701, pixel_distance.h:495-505) while the
    regular merge test passes and R's cluster is the larger one; returns (steps done, mean, size)."""
    h = h.astype(f32).copy()
    for i in range(len(psz)):
        if not (psz[i] < hsz):
            return i, h, hsz
        d = h - pm[i]
        sd = (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) * f32(1.0 / 3.0)
        if not (sd <= thr):
            return i, h, hsz
        denom = f32(1.0) / f32(psz[i] + hsz)
        a = f32(psz[i]) * denom
        b = f32(hsz) * denom
        h = a * pm[i] + b * h
        hsz += int(psz[i])
    return len(psz), h, hsz


def streamed_spine(h0, hsz0, pm, psz, thr):
    n = len(psz)
    pre = np.concatenate([[0], np.cumsum(psz)[:-1]]).astype(np.int64)     # exclusive scan
    S = hsz0 + pre                                                         # size before every step
    denom = (f32(1.0) / (psz + S).astype(f32)).astype(f32)
    ca = (psz.astype(f32) * denom).astype(f32)
    cb = (S.astype(f32) * denom).astype(f32)
    u = (ca[:, None] * pm).astype(f32)
    # the chain wavefront: sequential over all steps, one checkpoint per block of 16
    h = h0.astype(f32).copy()
    ck = []
    for i in range(n):
        h = (u[i] + cb[i] * h).astype(f32)
        if i % BLOCK == BLOCK - 1 or i == n - 1:
            ck.append(h.copy())

    def replay_block(b, upto=None):
        """One lane of k_spine_verify / k_spine_finish: the block from the mean before it."""
        hb = h0.astype(f32).copy() if b == 0 else ck[b - 1].copy()
        hi = min(BLOCK * b + BLOCK, n) if upto is None else min(upto, n)
        for i in range(BLOCK * b, hi):
            if upto is None:
                if not (psz[i] < S[i]):
                    return i, hb
                d = hb - pm[i]
                sd = (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) * f32(1.0 / 3.0)
                if not (sd <= thr):
                    return i, hb
            hb = (u[i] + cb[i] * hb).astype(f32)
        return -1, hb

    fail = n
    for b in range((n + BLOCK - 1) // BLOCK):     # "all CUs": any order
        bad, _ = replay_block(b)
        if bad >= 0:
            fail = min(fail, bad)
    if fail == 0:
        return 0, h0.astype(f32), hsz0
    _, hcut = replay_block((fail - 1) // BLOCK, upto=fail)
    return fail, hcut, hsz0 + int(pre[fail - 1] + psz[fail - 1])


@pytest.mark.parametrize("seed", range(12))
def test_streamed_spine_equals_sequential_replay(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 400))
    pm = (rng.random((n, 3)) * 0.08 + 0.4).astype(f32)
    psz = rng.integers(1, 9, n).astype(np.int64)
    if seed % 3 == 1:          # a side cluster that does not pass the test, somewhere
        pm[int(rng.integers(0, n))] += f32(0.5)
    if seed % 3 == 2:          # a side cluster larger than R's cluster
        psz[int(rng.integers(0, n))] = 10 ** 6
    h0 = np.array([0.43, 0.44, 0.45], f32)
    thr = f32(0.0025)
    want = sequential_spine(h0, 50, pm, psz, thr)
    got = streamed_spine(h0, 50, pm, psz, thr)
    assert got[0] == want[0] and got[2] == want[2]
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))


# ---- sampled list ranking ---------------------------------------------------------------------------
def random_tours(rng, num_lists, num_arcs):
    """succ over `num_arcs` arcs forming `num_lists` disjoint lists in random arc order; -1 ends a list."""
    perm = rng.permutation(num_arcs)
    cuts = np.sort(rng.choice(np.arange(1, num_arcs), num_lists - 1, replace=False)) if num_lists > 1 else []
    succ = np.full(num_arcs, -1, np.int64)
    heads, start = [], 0
    for end in list(cuts) + [num_arcs]:
        piece = perm[start:end]
        succ[piece[:-1]] = piece[1:]
        heads.append(int(piece[0]))
        start = end
    return succ, heads


def sampled_ranking(succ, heads):
    na = len(succ)
    ns0 = (na + SPLIT - 1) // SPLIT
    owner = np.full(na, -1, np.int64)
    local = np.zeros(na, np.int64)
    r_next = np.full(ns0 + len(heads), -1, np.int64)
    r_len = np.zeros(ns0 + len(heads), np.int64)
    starts = [(t, t * SPLIT) for t in range(ns0)] + \
             [(ns0 + k, a) for k, a in enumerate(heads) if a % SPLIT != 0]
    for t, a in starts:                      # one thread per splitter (k_rank_walk)
        steps, cur = 0, a
        while True:
            owner[cur] = t
            local[cur] = steps
            steps += 1
            nxt = succ[cur]
            if nxt == -1 or nxt % SPLIT == 0:
                break
            cur = nxt
        r_next[t] = -1 if nxt == -1 else nxt // SPLIT
        r_len[t] = steps
    nxt, dist = r_next.copy(), r_len.copy()  # pointer jumping on the reduced list (k_rank_step)
    span = 1
    while span < len(nxt):
        has = nxt >= 0
        dist = np.where(has, dist + dist[np.where(has, nxt, 0)], dist)
        nxt = np.where(has, nxt[np.where(has, nxt, 0)], -1)
        span *= 2
    assert (owner >= 0).all()
    return dist[owner] - local               # k_rank_finish


@pytest.mark.parametrize("seed,num_lists,num_arcs", [(0, 1, 50), (1, 1, 1000), (2, 7, 3000), (3, 40, 641),
                                                      (4, 3, 64), (5, 5, 129)])
def test_sampled_ranking_orders_every_tour(seed, num_lists, num_arcs):
    rng = np.random.default_rng(seed)
    succ, heads = random_tours(rng, num_lists, num_arcs)
    dist = sampled_ranking(succ, heads)
    for hd in heads:                         # distance to the end falls by one along every tour
        cur, want = hd, None
        while cur != -1:
            if want is not None:
                assert dist[cur] == want
            want = dist[cur] - 1
            cur = succ[cur]
        assert want == 0
