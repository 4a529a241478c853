"""Model test of the Kruskal-tree replay (video_segment_amd/csrc/merge_spine.hip):
the decomposition the device uses for the large components of a stage -- spanning tree by rank,
attach times t(x) seen from one vertex R, side clusters, spine -- reproduces the sequential replay
bit for bit, f32 means included, on random component graphs.  Pure numpy: no GPU, no library."""
import numpy as np
import pytest

f32 = np.float32


class Regions:
    """Union-find with the reference's merge rule (MergeRegions / MergeDescriptor,
    segmentation_graph.h:671-701, pixel_distance.h:495-505): the larger region survives (ties keep
    the second), the mean is the size-weighted average with two roundings per channel."""

    def __init__(self, means, sizes):
        self.parent = np.arange(len(sizes))
        self.mean = means.astype(f32).copy()
        self.size = sizes.astype(np.int64).copy()

    def find(self, x):
        while self.parent[x] != x:
            self.parent[x] = self.parent[self.parent[x]]
            x = self.parent[x]
        return x

    def merge(self, r1, r2):
        first_wins = self.size[r1] > self.size[r2]
        m, o = (r1, r2) if first_wins else (r2, r1)
        denom = f32(1.0) / f32(self.size[o] + self.size[m])
        a = f32(self.size[o]) * denom
        b = f32(self.size[m]) * denom
        self.mean[m] = a * self.mean[o] + b * self.mean[m]
        self.size[m] += self.size[o]
        self.parent[o] = m
        return m

    def edge(self, u, v):
        r1, r2 = self.find(u), self.find(v)
        if r1 != r2:
            self.merge(r1, r2)


def random_component(rng, n_vertices, n_extra):
    """A connected graph: a random spanning tree plus extra edges, in a random rank order; one
    vertex is much larger than the others (the region the spine is seen from)."""
    edges = [(int(rng.integers(0, v)), v) for v in range(1, n_vertices)]
    edges += [tuple(int(x) for x in rng.integers(0, n_vertices, 2)) for _ in range(n_extra)]
    edges = [(a, b) if rng.random() < 0.5 else (b, a) for a, b in edges if a != b]
    order = rng.permutation(len(edges))
    edges = [edges[i] for i in order]
    means = rng.random((n_vertices, 3)).astype(f32)
    sizes = rng.integers(1, 6, n_vertices)
    sizes[int(rng.integers(0, n_vertices))] = 1000
    return edges, means, sizes


def decomposition(edges, n_vertices, root):
    """Tree edges (Kruskal by rank), attach time t(x) = largest rank on the tree path root..x, and
    the class of every edge: ('side', t) replayed inside the side cluster attached at rank t,
    'spine', or 'internal'."""
    uf = list(range(n_vertices))

    def find(x):
        while uf[x] != x:
            uf[x] = uf[uf[x]]
            x = uf[x]
        return x

    adj = [[] for _ in range(n_vertices)]
    tree = set()
    for rank, (a, b) in enumerate(edges):
        ra, rb = find(a), find(b)
        if ra != rb:
            uf[ra] = rb
            tree.add(rank)
            adj[a].append((b, rank))
            adj[b].append((a, rank))
    t = [-1] * n_vertices
    stack = [root]
    seen = {root}
    while stack:   # path maxima from the root
        x = stack.pop()
        for y, rank in adj[x]:
            if y not in seen:
                seen.add(y)
                t[y] = max(t[x], rank)
                stack.append(y)
    assert len(seen) == n_vertices
    classes = []
    for rank, (a, b) in enumerate(edges):
        if rank in tree and max(t[a], t[b]) == rank:
            classes.append("spine")
        elif a != root and b != root and t[a] == t[b] and rank < t[a]:
            classes.append(("side", t[a]))
        else:
            classes.append("internal")
    return t, classes


@pytest.mark.parametrize("seed", range(12))
def test_tree_replay_equals_sequential_replay(seed):
    rng = np.random.default_rng(seed)
    n_vertices = int(rng.integers(20, 400))
    edges, means, sizes = random_component(rng, n_vertices, int(rng.integers(0, 3 * n_vertices)))
    root = int(np.argmax(sizes))

    seq = Regions(means, sizes)
    for a, b in edges:
        seq.edge(a, b)

    t, classes = decomposition(edges, n_vertices, root)
    tree = Regions(means, sizes)
    # 1. the side clusters, each on its own (any order between clusters; rank order inside)
    clusters = {}
    for rank, c in enumerate(classes):
        if isinstance(c, tuple):
            clusters.setdefault(c[1], []).append(rank)
    for attach in rng.permutation(sorted(clusters)):
        for rank in clusters[int(attach)]:
            tree.edge(*edges[rank])
    # every side cluster is one region by now
    for x in range(n_vertices):
        if x != root:
            attach_edge = edges[t[x]]
            child = attach_edge[0] if t[attach_edge[0]] == t[x] and attach_edge[0] != root else attach_edge[1]
            assert tree.find(x) == tree.find(child)
    # 2. the spine in rank order; every other edge is internal when it is visited
    for rank, c in enumerate(classes):
        a, b = edges[rank]
        if c == "spine":
            assert tree.find(a) != tree.find(b)
            assert tree.find(root) in (tree.find(a), tree.find(b))   # it attaches to R's cluster
            tree.edge(a, b)
        elif c == "internal":
            assert tree.find(a) == tree.find(b) == tree.find(root)   # both ends joined R's cluster earlier

    # same partition, same representative, same size and bit-identical means
    r_seq, r_tree = seq.find(root), tree.find(root)
    assert r_seq == r_tree
    assert seq.size[r_seq] == tree.size[r_tree] == sizes.sum()
    assert np.array_equal(seq.mean[r_seq].view(np.uint32), tree.mean[r_tree].view(np.uint32))
    for x in range(n_vertices):
        assert seq.find(x) == tree.find(x)
