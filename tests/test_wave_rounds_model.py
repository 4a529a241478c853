"""Model test of the round rule of the wave worker (video_segment_amd/csrc/merge_wave.hip,
DESIGN.md section 4 item 4): a batch of edges is committed in rounds, not one by one -- an edge
when it is the earliest pending edge on both its regions, and a *kept* edge (DecideEdge keeps it
and changes neither state) ahead of earlier kept edges that are committed in the same round.  The
rounds must leave every region exactly where the sequential replay leaves it, f32 means included.

The model has the reservation words of the kernel (res = earliest pending lane per region,
res2 = earliest blocker per region with the fixpoint over kept lanes that have to wait) and no hot
region / chain: those are covered on the device (self check VSG_WAVE_DBG=16, parity tests).  It
also runs the rule the first implementation of round 3 had -- a kept edge passes EVERY earlier
kept edge -- and shows that it is wrong: a kept edge that waits can turn into a merge once its
other region has changed.  Pure numpy: no GPU, no library."""
import numpy as np

f32 = np.float32
MIN_SIZE = 8
PASS_S = f32(0.0025)    # regular test: squared distance <= pass_s
SPLIT_S = f32(0.0225)   # constrained split: squared distance > split_s


class State:
    def __init__(self, rng, n):
        self.parent = np.arange(n)
        self.mean = (rng.random((n, 3)) * 0.12).astype(f32)
        self.size = rng.choice([1, 2, 3, 9, 12, 40], n).astype(np.int64)
        self.cons = rng.choice([-1, -1, -1, 0, 1, 2], n).astype(np.int64)
        self.fin = rng.random(n) < 0.45

    def copy(self):
        s = State.__new__(State)
        s.parent, s.mean, s.size = self.parent.copy(), self.mean.copy(), self.size.copy()
        s.cons, s.fin = self.cons.copy(), self.fin.copy()
        return s

    def find(self, x):
        while self.parent[x] != x:
            x = self.parent[x]
        return x

    def dist(self, a, b):
        d = self.mean[a] - self.mean[b]
        return (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) * f32(1.0 / 3.0)

    def merge(self, r1, r2):   # MergeRegions + MergeDescriptor: the larger survives, ties keep r2
        first = self.size[r1] > self.size[r2]
        m, o = (r1, r2) if first else (r2, r1)
        denom = f32(1.0) / f32(self.size[o] + self.size[m])
        a, b = f32(self.size[o]) * denom, f32(self.size[m]) * denom
        self.mean[m] = a * self.mean[o] + b * self.mean[m]
        self.size[m] += self.size[o]
        self.cons[m] = max(self.cons[r1], self.cons[r2])
        self.parent[o] = m

    def decide(self, r1, r2):
        """DecideEdge (merge_common.h; segmentation_graph.h:374-440) on two representatives."""
        if self.cons[r1] < 0 or self.cons[r2] < 0:
            if not (self.fin[r1] or self.fin[r2]):
                if self.dist(r1, r2) <= PASS_S:
                    self.merge(r1, r2)
                    return
                self.fin[r1] = self.fin[r2] = True
            if self.size[r1] < MIN_SIZE or self.size[r2] < MIN_SIZE:
                self.merge(r1, r2)
        elif self.cons[r1] == self.cons[r2]:
            if self.dist(r1, r2) > SPLIT_S:
                if self.size[r1] < self.size[r2] * 0.3:
                    self.cons[r1] = -1
                elif self.size[r2] < self.size[r1] * 0.3:
                    self.cons[r2] = -1
                else:
                    self.cons[r1] = self.cons[r2] = -1
            else:
                self.merge(r1, r2)

    def noop(self, r1, r2):   # NoopPair
        if self.cons[r1] >= 0 and self.cons[r2] >= 0:
            return self.cons[r1] != self.cons[r2]
        return bool(self.fin[r1] or self.fin[r2]) and self.size[r1] >= MIN_SIZE and self.size[r2] >= MIN_SIZE

    def signature(self):
        roots = np.array([self.find(i) for i in range(len(self.parent))])
        return (roots.tobytes(), self.mean[roots].tobytes(), self.size[roots].tobytes(),
                self.cons[roots].tobytes(), self.fin[roots].tobytes())


def sequential(state, edges):
    for u, v in edges:
        a, b = state.find(u), state.find(v)
        if a != b:
            state.decide(a, b)


def rounds(state, edges, pass_every_kept_edge=False):
    """Commits the batch in rounds; returns the number of rounds."""
    INF = 1 << 30
    pending = [True] * len(edges)
    n_rounds = 0
    while True:
        lanes = []
        for i, (u, v) in enumerate(edges):
            if pending[i]:
                a, b = state.find(u), state.find(v)
                if a == b:
                    pending[i] = False
                else:
                    lanes.append((i, a, b))
        if not lanes:
            return n_rounds
        n_rounds += 1
        res, noop = {}, {}
        for i, a, b in lanes:
            res[a] = min(res.get(a, INF), i)
            res[b] = min(res.get(b, INF), i)
            noop[i] = state.noop(a, b)
        if all(noop.values()):          # nothing can change a state any more
            for i, a, b in lanes:
                pending[i] = False
            continue
        # blockers: lanes that may change a state, then kept lanes behind a blocker (fixpoint)
        res2 = {}
        for i, a, b in lanes:
            if not noop[i]:
                res2[a] = min(res2.get(a, INF), i)
                res2[b] = min(res2.get(b, INF), i)
        cand = {i for i, a, b in lanes if noop[i]}
        while not pass_every_kept_edge:
            drop = {i for i, a, b in lanes if i in cand and not (res2.get(a, INF) > i and res2.get(b, INF) > i)}
            if not drop:
                break
            cand -= drop
            for i, a, b in lanes:
                if i in drop:
                    res2[a] = min(res2.get(a, INF), i)
                    res2[b] = min(res2.get(b, INF), i)
        if pass_every_kept_edge:        # the unsafe rule: only state-changing lanes block
            cand = {i for i, a, b in lanes if noop[i] and res2.get(a, INF) > i and res2.get(b, INF) > i}
        winners = [(i, a, b) for i, a, b in lanes if (res[a] == i and res[b] == i) or i in cand]
        assert winners, "the earliest pending lane always commits"
        for i, a, b in winners:         # disjoint regions, or kept lanes that change nothing
            state.decide(a, b)
            pending[i] = False


def random_batch(rng):
    n = int(rng.integers(6, 22))
    m = int(rng.integers(8, 64))
    edges = [tuple(int(x) for x in rng.integers(0, n, 2)) for _ in range(m)]
    return State(rng, n), [(a, b) for a, b in edges if a != b]


def test_rounds_equal_the_sequential_replay():
    fewer = 0
    for seed in range(600):
        rng = np.random.default_rng([11, seed])
        st, edges = random_batch(rng)
        a, b = st.copy(), st.copy()
        sequential(a, edges)
        n_rounds = rounds(b, edges)
        assert a.signature() == b.signature(), "seed %d" % seed
        fewer += n_rounds < len(edges)
    assert fewer > 500      # and the rounds do run edges side by side


def test_passing_every_earlier_kept_edge_is_wrong():
    """The first implementation: only edges that may change a state block a kept edge.  Some batch
    has a kept edge that waits, turns into a merge, and changes a region a later kept edge has
    already been decided on."""
    wrong = 0
    for seed in range(600):
        rng = np.random.default_rng([11, seed])
        st, edges = random_batch(rng)
        a, b = st.copy(), st.copy()
        sequential(a, edges)
        rounds(b, edges, pass_every_kept_edge=True)
        wrong += a.signature() != b.signature()
    assert wrong > 0
