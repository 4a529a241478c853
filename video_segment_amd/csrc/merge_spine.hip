// merge_spine.hip -- the large components of a stage, replayed along their Kruskal tree.
//
// The ordered replay of one component is Kruskal's algorithm with a merge test: as long as every
// live edge merges (what the force-merge buckets do, and what the giant components of the bench
// input do without exception), the merges are the edges of the component's minimum spanning tree
// by rank (rank = position in the component's edge sequence), and *which* edges merge does not
// depend on any float.  Only the region means do, through the order of the merges.  Seen from one
// vertex R (the largest region of the component), the replay is:
//
//   t(x)  = rank at which vertex x joins R's cluster = largest rank on the tree path R..x;
//   spine = the tree edges that attach something to R's cluster: rank(e) == t(child end of e);
//   side cluster of a spine edge e = the vertices x with t(x) == rank(e): they are connected among
//           themselves by edges of smaller rank, every other edge that leaves them has a larger
//           rank (minimax property), so before rank(e) they only ever interact with each other --
//           a side cluster is an independent sequential sub-problem, exact whatever its outcomes;
//   every other edge has both ends in R's cluster by the time it is visited: internal.
//
// So: the side clusters are replayed by the ordinary workers (all of them at once, they are
// usually a pixel or two), then one wavefront per component walks the spine in rank order and
// absorbs the finished side clusters into R's cluster, 64 at a time, with the same chain
// arithmetic as k_merge_wave (sizes by prefix sum, weights and divisions lane parallel, the mean
// as a systolic recurrence, all merge tests verified afterwards; anything that is not a plain
// chain step runs through the exact DecideEdge).  The region states of the partners are final
// when the spine starts, so a reader wavefront can fetch them arbitrarily far ahead.
//
// The structure is only valid while every live edge merges.  A side cluster that keeps an edge
// or a spine edge that is kept raises the stage's violation flag: the stage is undone from its
// backup and replayed by the ordinary workers (RunBucketStage), so the result is always the
// sequential one.
//
// Tree machinery (all data parallel): Boruvka for the tree edges, Euler tour + list ranking to
// root the tree at R, pointer jumping for the path maxima and for the side cluster of every vertex.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <unordered_map>
#include <vector>

#include "merge_common.h"
#include "scan_device.h"

namespace vsg {

namespace {

constexpr int kNone = -1;

static inline unsigned Blocks(size_t n) { return (unsigned)((n + 255) / 256); }

// Component of edge e of the concatenated list: the largest k with comp_base[k] <= e.
__device__ __forceinline__ int CompOf(const int32_t* __restrict__ comp_base, int K, int e) {
  int lo = 0, hi = K;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (comp_base[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- edges of the large components, concatenated --------------------------------------------------
__global__ __launch_bounds__(256) void k_spine_gather(int mE, int K, const int32_t* __restrict__ comp_base,
                                                       const int32_t* __restrict__ comp_off,
                                                       const int32_t* __restrict__ s_ra,
                                                       const int32_t* __restrict__ s_rb,
                                                       int32_t* __restrict__ eu, int32_t* __restrict__ ev,
                                                       int32_t* __restrict__ estate, int32_t* __restrict__ cc,
                                                       uint32_t* __restrict__ best) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const bool valid = e < mE;
  int u = -1, v = -1;
  if (valid) {
    const int lo = CompOf(comp_base, K, e);
    const int pos = comp_off[lo] + (e - comp_base[lo]);
    u = s_ra[pos];
    v = s_rb[pos];
    eu[e] = u;
    ev[e] = v;
    estate[e] = 0;
  }
  // The per-vertex tables are scattered 4-byte stores (a sector each), and consecutive edges of a
  // large component share a vertex more often than not (the hub): a lane whose vertex the lane
  // before it resets as well leaves the store to that lane.
  const int pu = __shfl_up(u, 1), pv = __shfl_up(v, 1);
  const bool first = (threadIdx.x & 63) == 0;
  if (valid && (first || (u != pu && u != pv))) {
    cc[u] = u;
    best[u] = 0xffffffffu;
  }
  if (valid && v != u && (first || (v != pu && v != pv))) {
    cc[v] = v;
    best[v] = 0xffffffffu;
  }
}

// ---- Boruvka ----------------------------------------------------------------------------------------
// estate: 0 alive, 1 tree edge, 2 internal, 3 tree edge chosen in this round.
// best[c] holds the smallest rank among the edges that leave component c, stamped with the round
// ((31 - round) << 27 | rank, unsigned): a later round's entry is smaller than anything an
// earlier round left behind, so the table never has to be cleared.
constexpr int kAliveSlots = 32;   // counters of the live edges of a round, one cache line apart
__device__ __forceinline__ uint32_t BorKey(int round, int e) { return ((uint32_t)(31 - round) << 27) | (uint32_t)e; }

// The kernels of a round run over all edges (list == null) or over the list of the edges that were
// still alive a round or two ago.
__global__ __launch_bounds__(256) void k_bor_min(int n, const int32_t* __restrict__ list,
                                                  const int32_t* __restrict__ list_len, int round,
                                                  const int32_t* __restrict__ eu,
                                                  const int32_t* __restrict__ ev, int32_t* __restrict__ estate,
                                                  int32_t* __restrict__ cc, uint32_t* __restrict__ best,
                                                  int32_t* __restrict__ ecu, int32_t* __restrict__ ecv,
                                                  int32_t* __restrict__ alive, const int32_t* __restrict__ gate,
                                                  int32_t* __restrict__ next_len) {
  // next_len: the word the compaction after this round counts into (the length of the next list):
  // cleared here, a round ahead of its use -- it has to outlive any number of rounds, which the
  // rotating pool of zeroed counters does not promise.
  if (blockIdx.x == 0 && threadIdx.x == 0) *next_len = 0;
  // gate: the number of live edges the round before found (k_bor_hook).  The host launches a round
  // before it has read that number (the wait for it would leave the GPU idle once per round); a round
  // behind a complete forest is empty.
  if (gate && *gate == 0) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (list_len) n = min(n, *list_len);   // (the list was compacted on the device: n is its capacity)
  const int e = i < n ? (list ? list[i] : i) : -1;
  bool live = false;
  int cu = -1, cv = -1;
  if (e >= 0 && round == 0) {   // every vertex is its own component; the ends of an edge differ
    cu = eu[e];
    cv = ev[e];
    live = true;
  } else if (e >= 0 && estate[e] == 0) {
    cu = CcFind(cc, eu[e]);
    cv = CcFind(cc, ev[e]);
    if (cu == cv) {
      estate[e] = 2;
      cu = cv = -1;
    } else {
      live = true;
      ecu[e] = cu;
      ecv[e] = cv;
    }
  }
  // The hub of a chain-like component is an end of most of its edges.  Ranks grow with the lane,
  // so a lane whose component already appears in the lane before it has nothing to add; and a
  // plain (cached, possibly stale: only ever too large) read keeps all but the first few of the
  // others away from the atomic.
  const int pu = __shfl_up(cu, 1), pv = __shfl_up(cv, 1), pe = __shfl_up(e, 1);
  if (live) {
    const uint32_t key = BorKey(round, e);
    // (a compacted list is in rank order only piecewise: the lane before must hold a smaller rank)
    const bool has_prev = (threadIdx.x & 63) != 0 && pe < e;
    if (!(has_prev && (cu == pu || cu == pv)) && key < best[cu]) atomicMin(&best[cu], key);
    if (!(has_prev && (cv == pu || cv == pv)) && key < best[cv]) atomicMin(&best[cv], key);
  }
  // One addition per workgroup, spread over kAliveSlots cache lines: an atomic per wavefront on ONE
  // word was the kernel's bottleneck (625 K same-address atomics for 40 M edges: 3 ms a round,
  // whatever the tables cost).
  const int cnt = __syncthreads_count(live);
  if (cnt && threadIdx.x == 0) atomicAdd(&alive[(blockIdx.x % kAliveSlots) * 16], cnt);
}

// The edges that are the minimum of one of their two components join them (tree edges); the first
// thread reports the number of edges k_bor_min found alive.
__global__ __launch_bounds__(256) void k_bor_hook(int n, const int32_t* __restrict__ list,
                                                   const int32_t* __restrict__ list_len, int round,
                                                   const int32_t* __restrict__ eu, const int32_t* __restrict__ ev,
                                                   int32_t* __restrict__ estate, int32_t* __restrict__ cc,
                                                   const uint32_t* __restrict__ best,
                                                   const int32_t* __restrict__ ecu, const int32_t* __restrict__ ecv,
                                                   const int32_t* __restrict__ alive,
                                                   unsigned long long* __restrict__ mail, unsigned mail_seq,
                                                   const int32_t* __restrict__ gate, int32_t* __restrict__ total_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool idle = gate && *gate == 0;   // (see k_bor_min)
  if (blockIdx.x == 0) {   // (k_bor_min is complete: its counters are final)
    int v = (!idle && threadIdx.x < kAliveSlots) ? alive[threadIdx.x * 16] : 0;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (threadIdx.x == 0) {
      *total_out = v;   // the next round's gate
      MailPost(mail, mail_seq, 0, v);
    }
  }
  if (idle) return;
  if (list_len) n = min(n, *list_len);
  if (i >= n) return;
  const int e = list ? list[i] : i;
  if (estate[e] != 0) return;
  const uint32_t key = BorKey(round, e);
  const int cu = round == 0 ? eu[e] : ecu[e], cv = round == 0 ? ev[e] : ecv[e];
  if (best[cu] == key || best[cv] == key) {
    CcUnion(cc, eu[e], ev[e]);
    estate[e] = 1;
  }
}

// The edges that are still alive, in any order (the rounds only take minima over them).  One
// reservation per workgroup of 2048 edges (one per wavefront made the single counter the bottleneck:
// 3 ms for 40 M edges).
constexpr int kBorCompactPer = 8;
__global__ __launch_bounds__(256) void k_bor_compact(int n, const int32_t* __restrict__ list,
                                                      const int32_t* __restrict__ list_len,
                                                      const int32_t* __restrict__ estate, int32_t* __restrict__ out,
                                                      int32_t* __restrict__ out_len) {
  __shared__ int wave_tot[4];
  __shared__ int block_base;
  if (list_len) n = min(n, *list_len);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int e[kBorCompactPer];
  unsigned long long m[kBorCompactPer];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < kBorCompactPer; ++k) {
    const int i = (blockIdx.x * kBorCompactPer + k) * 256 + threadIdx.x;
    e[k] = i < n ? (list ? list[i] : i) : -1;
    const bool keep = e[k] >= 0 && estate[e[k]] == 0;
    m[k] = __ballot(keep);
    mine += (int)__popcll(m[k]);
  }
  if (lane == 0) wave_tot[w] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    block_base = total ? atomicAdd(out_len, total) : 0;
  }
  __syncthreads();
  int at = block_base;
  for (int q = 0; q < w; ++q) at += wave_tot[q];
#pragma unroll
  for (int k = 0; k < kBorCompactPer; ++k) {
    if ((m[k] >> lane) & 1ull) out[at + (int)__popcll(m[k] & ((1ull << lane) - 1ull))] = e[k];
    at += (int)__popcll(m[k]);
  }
}

// ---- R: the largest region of every component --------------------------------------------------------
constexpr int kRootStride = 16;
__global__ __launch_bounds__(256) void k_spine_pick_root(int mE, int K, const int32_t* __restrict__ eu,
                                                          const int32_t* __restrict__ ev,
                                                          const int32_t* __restrict__ comp_base, NodeArrays nodes,
                                                          unsigned long long* __restrict__ root_key) {
  __shared__ unsigned long long red[256];
  __shared__ int comp0;
  // Every kRootStride-th edge is looked at (R is an end of most edges of its component, any vertex
  // would do), and the first edge of every component, so that none stays without a root.
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int e = i < K ? comp_base[i] : (i - K) * kRootStride;
  unsigned long long key = 0;
  int k = -1;
  if (e < mE) {
    k = CompOf(comp_base, K, e);
    const int u = eu[e], v = ev[e];
    const unsigned su = (unsigned)__float_as_int(nodes.desc_sz[u].w);
    const unsigned sv = (unsigned)__float_as_int(nodes.desc_sz[v].w);
    const unsigned long long ku = ((unsigned long long)su << 32) | (unsigned)(~u);
    const unsigned long long kv = ((unsigned long long)sv << 32) | (unsigned)(~v);
    key = ku > kv ? ku : kv;
  }
  if (threadIdx.x == 0) comp0 = k;
  __syncthreads();
  const bool same = (k == comp0);
  red[threadIdx.x] = same ? key : 0ull;
  if (!same && k >= 0) atomicMax(&root_key[k], key);   // a block that straddles two components
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && comp0 >= 0) atomicMax(&root_key[comp0], red[0]);
}

__global__ void k_spine_roots(int K, const unsigned long long* __restrict__ root_key,
                              int32_t* __restrict__ root_vertex, int32_t* __restrict__ childidx) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  const int r = (int)(~(unsigned)(root_key[k] & 0xffffffffull));
  root_vertex[k] = r;
  childidx[r] = kNone;
}

// ---- tree edges -> arcs ---------------------------------------------------------------------------------
// Flag -> number -> report, as one fused scan (scan_device.h): flag[e] = (state[e] == want), scan[e] =
// flagged edges before e, the total stored and posted to the host.
struct StateFlagValue {
  const int32_t* state;
  int want;
  __device__ int operator()(int e) const { return state[e] == want ? 1 : 0; }
};
struct FlagScanEmit {
  int32_t* flag;   // (null: not kept)
  int32_t* scan;
  __device__ void operator()(int e, int f, int before) const {
    if (flag) flag[e] = f;
    scan[e] = before;
  }
};
struct TotalPostFinish {
  int32_t* count;
  unsigned long long* mail;
  unsigned mail_seq;
  __device__ void operator()(int total) const {
    *count = total;
    MailPost(mail, mail_seq, 0, total);
  }
};

__global__ __launch_bounds__(256) void k_compact_tree(int mE, const int32_t* __restrict__ flag,
                                                       const int32_t* __restrict__ scan,
                                                       int32_t* __restrict__ te_e) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < mE && flag[e]) te_e[scan[e]] = e;
}

__global__ __launch_bounds__(256) void k_arc_keys(int mt, const int32_t* __restrict__ te_e,
                                                   const int32_t* __restrict__ eu, const int32_t* __restrict__ ev,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= 2 * mt) return;
  const int e = te_e[a >> 1];
  keys[a] = (uint32_t)((a & 1) ? ev[e] : eu[e]);   // source vertex of the arc
  vals[a] = (uint32_t)a;
}

__global__ __launch_bounds__(256) void k_arc_first(int na, const uint32_t* __restrict__ ks,
                                                    const uint32_t* __restrict__ as, int32_t* __restrict__ firstq,
                                                    int32_t* __restrict__ pos_arc) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= na) return;
  if (q == 0 || ks[q] != ks[q - 1]) firstq[ks[q]] = q;
  pos_arc[as[q]] = q;
}

// Euler tour successor: after arc u->v comes the arc that follows v->u in v's list (cyclically).
// The tour of a tree is cut in front of the first arc of its root.
__global__ __launch_bounds__(256) void k_arc_succ(int na, const uint32_t* __restrict__ ks,
                                                   const uint32_t* __restrict__ as,
                                                   const int32_t* __restrict__ firstq,
                                                   const int32_t* __restrict__ pos_arc,
                                                   const int32_t* __restrict__ te_e,
                                                   const int32_t* __restrict__ comp_base, int K,
                                                   const int32_t* __restrict__ root_vertex,
                                                   int32_t* __restrict__ succ, int32_t* __restrict__ dist) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= na) return;
  const int qb = pos_arc[a ^ 1];
  const uint32_t w = ks[qb];   // source of the twin = destination of a
  int q = qb + 1;
  if (q == na || ks[q] != w) q = firstq[w];
  int nx = (int)as[q];
  const int root = root_vertex[CompOf(comp_base, K, te_e[a >> 1])];
  if ((int)w == root && q == firstq[root]) nx = kNone;   // back at the start of the tour
  succ[a] = nx;
  dist[a] = nx == kNone ? 0 : 1;
}

// List ranking by pointer jumping (distance to the end of the tour), double buffered.
__global__ __launch_bounds__(256) void k_rank_step(int na, const int32_t* __restrict__ s_in,
                                                    const int32_t* __restrict__ d_in, int32_t* __restrict__ s_out,
                                                    int32_t* __restrict__ d_out) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= na) return;
  const int s = s_in[a];
  if (s == kNone) {
    s_out[a] = kNone;
    d_out[a] = d_in[a];
  } else {
    s_out[a] = s_in[s];
    d_out[a] = d_in[a] + d_in[s];
  }
}

// List ranking of a long tour by sampling (Helman / JaJa): every kRankSplit-th arc and the first arc
// of every tour is a splitter; one thread per splitter walks its piece of the list (to the next
// splitter), numbering the arcs; the list of the splitters -- weighted with the lengths of their
// pieces, a 64th of the arcs -- is ranked by pointer jumping; an arc's distance to the end of its tour
// is its splitter's minus its number.  One pass over the arcs instead of log2(n) (k_rank_step on
// 14 M arcs: 23 passes, 3 ms).
constexpr int kRankSplit = 64;
__global__ __launch_bounds__(256) void k_rank_walk(int na, int K, const int32_t* __restrict__ succ,
                                                    const int32_t* __restrict__ root_vertex,
                                                    const int32_t* __restrict__ firstq,
                                                    const uint32_t* __restrict__ as, int32_t* __restrict__ owner,
                                                    int32_t* __restrict__ local, int32_t* __restrict__ r_next,
                                                    int32_t* __restrict__ r_len) {
  const int ns0 = (na + kRankSplit - 1) / kRankSplit;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ns0 + K) return;
  int a;
  if (t < ns0) {
    a = t * kRankSplit;
  } else {
    a = (int)as[firstq[root_vertex[t - ns0]]];   // the first arc of the tour of component t - ns0
    if (a % kRankSplit == 0) {                   // a regular splitter as well: that thread walks it
      r_next[t] = kNone;
      r_len[t] = 0;
      return;
    }
  }
  int steps = 0, cur = a, nxt;
  for (;;) {
    owner[cur] = t;
    local[cur] = steps++;
    nxt = succ[cur];
    if (nxt == kNone || nxt % kRankSplit == 0) break;   // (the first arc of a tour is nobody's successor)
    cur = nxt;
  }
  r_next[t] = nxt == kNone ? kNone : nxt / kRankSplit;
  r_len[t] = steps;
}

__global__ __launch_bounds__(256) void k_rank_finish(int na, const int32_t* __restrict__ owner,
                                                      const int32_t* __restrict__ local,
                                                      const int32_t* __restrict__ r_dist, int32_t* __restrict__ dist) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a < na) dist[a] = r_dist[owner[a]] - local[a];
}

// The arc walked first goes down: its destination is the child.
__global__ __launch_bounds__(256) void k_tree_parent(int mt, const int32_t* __restrict__ te_e,
                                                      const int32_t* __restrict__ eu,
                                                      const int32_t* __restrict__ ev,
                                                      const int32_t* __restrict__ dist, int32_t* __restrict__ child,
                                                      int32_t* __restrict__ par, int32_t* __restrict__ childidx) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= mt) return;
  const int e = te_e[j];
  const bool down = dist[2 * j] > dist[2 * j + 1];   // u->v comes first
  const int c = down ? ev[e] : eu[e];
  child[j] = c;
  par[j] = down ? eu[e] : ev[e];
  childidx[c] = j;
}

// jump: tree edge of the parent (kNone: the parent is the root); val: largest rank on the path so far.
__global__ __launch_bounds__(256) void k_jump_init(int mt, const int32_t* __restrict__ te_e,
                                                    const int32_t* __restrict__ par,
                                                    const int32_t* __restrict__ childidx, int32_t* __restrict__ jump,
                                                    int32_t* __restrict__ val) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= mt) return;
  jump[j] = childidx[par[j]];
  val[j] = te_e[j];
}

// more (optional): set when some vertex has not reached the root yet after this step.
__global__ __launch_bounds__(256) void k_jump_max(int mt, const int32_t* __restrict__ j_in,
                                                   const int32_t* __restrict__ v_in, int32_t* __restrict__ j_out,
                                                   int32_t* __restrict__ v_out, int32_t* __restrict__ more) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= mt) return;
  const int p = j_in[j];
  if (p == kNone) {
    j_out[j] = kNone;
    v_out[j] = v_in[j];
  } else {
    const int pp = j_in[p];
    j_out[j] = pp;
    v_out[j] = max(v_in[j], v_in[p]);
    if (more && pp != kNone) *more = 1;
  }
}

// head: the spine edge through which the vertex joins R's cluster -- the tree edge whose rank is
// t(vertex) (ranks are edge indices: unique), found through the scan that compacted the tree edges
// (te_e[scan[e]] == e).
__global__ __launch_bounds__(256) void k_head_lookup(int mt, const int32_t* __restrict__ tval,
                                                      const int32_t* __restrict__ scan, int32_t* __restrict__ head) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < mt) head[j] = scan[tval[j]];
}

// ---- classification of every edge ----------------------------------------------------------------------
// side_key[e]: head (tree edge index) of the side cluster the edge is replayed in, or kNone;
// spine_flag[e]: the edge attaches its side cluster to R's cluster.
__global__ __launch_bounds__(256) void k_classify(int mE, const int32_t* __restrict__ eu,
                                                   const int32_t* __restrict__ ev,
                                                   const int32_t* __restrict__ estate,
                                                   const int32_t* __restrict__ childidx,
                                                   const int32_t* __restrict__ head,
                                                   const int32_t* __restrict__ te_e,
                                                   int32_t* __restrict__ side_flag, int32_t* __restrict__ side_key,
                                                   int32_t* __restrict__ spine_flag) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= mE) return;
  const int ju = childidx[eu[e]], jv = childidx[ev[e]];   // kNone: the root
  const int hu = ju == kNone ? kNone : head[ju];
  const int hv = jv == kNone ? kNone : head[jv];
  const bool side = hu != kNone && hu == hv && e < te_e[hu];
  side_flag[e] = side ? 1 : 0;
  side_key[e] = side ? hu : kNone;
  // a tree edge is a spine edge when it is the head of its own child
  bool spine = false;
  if (estate[e] == 1) {
    const int jc = (ju != kNone && te_e[ju] == e) ? ju : jv;   // the tree edge e itself (child end)
    spine = head[jc] == jc;
  }
  spine_flag[e] = spine ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_compact_side(int mE, const int32_t* __restrict__ flag,
                                                       const int32_t* __restrict__ scan,
                                                       const int32_t* __restrict__ side_key,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= mE) return;
  if (flag[e]) {
    keys[scan[e]] = (uint32_t)side_key[e];
    idx[scan[e]] = (uint32_t)e;
  }
}

__global__ __launch_bounds__(256) void k_gather_side(int n, const uint32_t* __restrict__ sorted_e,
                                                      const int32_t* __restrict__ comp_base,
                                                      const int32_t* __restrict__ comp_off, int K,
                                                      const int32_t* __restrict__ s_ra,
                                                      const int32_t* __restrict__ s_rb,
                                                      const uint32_t* __restrict__ s_gpos,
                                                      int32_t* __restrict__ o_ra, int32_t* __restrict__ o_rb,
                                                      uint32_t* __restrict__ o_gpos) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int e = (int)sorted_e[i];
  const int k = CompOf(comp_base, K, e);
  const int pos = comp_off[k] + (e - comp_base[k]);
  o_ra[i] = s_ra[pos];
  o_rb[i] = s_rb[pos];
  o_gpos[i] = s_gpos[pos];
}

// Spine edges in rank order: the child end (whose side cluster is absorbed) and whether it is the
// edge's first region -- moved together by the fused scan of the spine flags.  comp_spine[k]: start of
// component k's spine (from the scan).
struct SpineEmit {
  const int32_t* eu;
  const int32_t* ev;
  const int32_t* childidx;
  const int32_t* te_e;
  int32_t* scan;
  int32_t* sp_child;
  int32_t* sp_is_a;   // bit 0: the child is the edge's first end; the bits above: where the edge is in the
                      // level's sorted arrays (its kept position is s_gpos[...]: where a stage is cut when
                      // the spine keeps the edge, k_spine)
  const int32_t* comp_base;
  const int32_t* comp_off;
  int K;
  __device__ void operator()(int e, int f, int before) const {
    scan[e] = before;
    if (f) {
      const int ju = childidx[eu[e]];
      const bool child_is_u = ju != kNone && te_e[ju] == e;
      sp_child[before] = child_is_u ? eu[e] : ev[e];
      const int lo = CompOf(comp_base, K, e);
      sp_is_a[before] = ((comp_off[lo] + (e - comp_base[lo])) << 1) | (child_is_u ? 1 : 0);
    }
  }
};
struct SpineFinish {
  int32_t* comp_spine;
  int K;
  __device__ void operator()(int total) const { comp_spine[K] = total; }
};
__global__ __launch_bounds__(256) void k_comp_spine(int K, const int32_t* __restrict__ comp_base,
                                                     const int32_t* __restrict__ scan, int32_t* __restrict__ comp_spine) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < K) comp_spine[k] = scan[comp_base[k]];
}

// ---- the spine ---------------------------------------------------------------------------------------------
constexpr int kSpineFill = 4;        // 64-edge groups per fill
constexpr int kSpineFills = 8;       // fills the ring holds
constexpr int kSpineRing = kSpineFills * kSpineFill * 64;
constexpr int kSpineReaders = 3;     // reader wavefronts: one consumer cannot hide the latency of a
                                     // dependent chain of loads per fill behind one reader

struct SpineRing {
  int32_t p[kSpineRing];       // representative of the side cluster
  float4 ds[kSpineRing];
  int32_t cons[kSpineRing];
  int32_t flags[kSpineRing];
  int32_t is_a[kSpineRing];
  int ready[kSpineFills];   // fill f is complete when ready[f % kSpineFills] == f + 1
  int consumed;
  int abort;   // the consumer gave up (violation): the readers stop
};

__device__ __forceinline__ void WaveSyncS() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

__global__ __launch_bounds__(64 * (1 + kSpineReaders)) void k_spine(int K, const int32_t* __restrict__ comp_spine,
                                               int32_t* __restrict__ root_vertex,
                                               const int32_t* __restrict__ sp_child,
                                               const int32_t* __restrict__ sp_is_a, NodeArrays nodes, StageThr T,
                                               int optimistic, int32_t* __restrict__ violation,
                                               unsigned long long* __restrict__ stats,
                                               int32_t* __restrict__ start_pos, int max_steps,
                                               const uint32_t* __restrict__ s_gpos, int32_t* __restrict__ hub_excl) {
  __shared__ SpineRing ring;
  const int k = blockIdx.x;
  if (k >= K) return;
  const int lane = threadIdx.x & 63;
  // start_pos: what the streamed chain (k_spine_chain and friends, below) has absorbed already;
  // max_steps: this launch only replays that many steps (one batch past the step that stopped the
  // streamed chain) and leaves where it got to -- start_pos, the representative -- behind.
  const int done = start_pos ? start_pos[k] : 0;
  const int beg = comp_spine[k] + done;
  int n = comp_spine[k + 1] - beg;
  if (n <= 0) return;
  if (n > max_steps) n = max_steps;
  if (threadIdx.x < kSpineFills) ring.ready[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    ring.consumed = 0;
    ring.abort = 0;
  }
  __syncthreads();
  if (threadIdx.x >= 64) {
    // ---- readers: representative and state of every side cluster, in spine order ------------------------
    const int reader = (threadIdx.x >> 6) - 1;
    for (int f = reader; f * (kSpineFill * 64) < n; f += kSpineReaders) {
      const int next = f * (kSpineFill * 64);
      bool stop = false;
      while (next + kSpineFill * 64 - __hip_atomic_load(&ring.consumed, __ATOMIC_ACQUIRE,
                                                          __HIP_MEMORY_SCOPE_WORKGROUP) > kSpineRing) {
        if (__hip_atomic_load(&ring.abort, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) {
          stop = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (stop) break;
      int c[kSpineFill], isa[kSpineFill];
      bool vd[kSpineFill];
#pragma unroll
      for (int q = 0; q < kSpineFill; ++q) {
        const int i = next + q * 64 + lane;
        vd[q] = i < n;
        c[q] = vd[q] ? sp_child[beg + i] : root_vertex[k];
        isa[q] = vd[q] ? (sp_is_a[beg + i] & 1) : 0;
      }
      for (bool any = true; any;) {   // all root searches of the fill advance together
        int pa[kSpineFill];
#pragma unroll
        for (int q = 0; q < kSpineFill; ++q) pa[q] = nodes.parent[c[q]];
        any = false;
#pragma unroll
        for (int q = 0; q < kSpineFill; ++q) {
          if (pa[q] != c[q]) {
            c[q] = pa[q];
            any = true;
          }
        }
      }
      RState st[kSpineFill];
#pragma unroll
      for (int q = 0; q < kSpineFill; ++q) st[q] = LoadState(nodes, c[q]);
#pragma unroll
      for (int q = 0; q < kSpineFill; ++q) {
        if (vd[q]) {
          const int slot = (next + q * 64 + lane) & (kSpineRing - 1);
          ring.p[slot] = c[q];
          ring.ds[slot] = make_float4(st[q].d0, st[q].d1, st[q].d2, __int_as_float(st[q].sz));
          ring.cons[slot] = st[q].cons;
          ring.flags[slot] = st[q].flags;
          ring.is_a[slot] = isa[q];
        }
      }
      WaveSyncS();
      if (lane == 0) {
        __hip_atomic_store(&ring.ready[f % kSpineFills], f + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    return;
  }

  // ---- consumer: R's cluster absorbs the side clusters in rank order --------------------------------------
  int rep = root_vertex[k];
  RState H = LoadState(nodes, rep);
  unsigned n_forced = 0, n_regular = 0, n_small = 0;   // per lane
  int violated = 0;   // 2: the structure assumed a merge; 1: a tentatively settled edge is no longer valid
  for (int base = 0; base < n && violated == 0; base += 64) {
    const int cnt = n - base < 64 ? n - base : 64;
    const int fill = base / (kSpineFill * 64);
    while (__hip_atomic_load(&ring.ready[fill % kSpineFills], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) !=
           fill + 1) {
      __builtin_amdgcn_s_sleep(1);
    }
    const bool valid = lane < cnt;
    int p = 0, is_a = 0;
    RState P = {};
    if (valid) {
      const int slot = (base + lane) & (kSpineRing - 1);
      p = ring.p[slot];
      const float4 ds = ring.ds[slot];
      P.d0 = ds.x;
      P.d1 = ds.y;
      P.d2 = ds.z;
      P.sz = __float_as_int(ds.w);
      P.cons = ring.cons[slot];
      P.flags = ring.flags[slot];
      is_a = ring.is_a[slot];
    }
    WaveSyncS();   // the entries are in registers: their slots may be reused
    if (lane == 0) {
      __hip_atomic_store(&ring.consumed, base + cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    int start = 0;       // first lane not replayed yet (wave uniform)
    bool generic_next = false;   // lane `start` failed its chain test: exact replay
    while (start < cnt) {
      // ---- chain: the leading run of plain, smaller partners that merge ---------------------------------------
      const bool fin = (H.flags & kFlagFinalized) != 0;
      const bool mode_ok = !(H.flags & kFlagNoDesc) && (!fin || H.sz >= T.min_size);
      const bool fin_l = fin || (P.flags & kFlagFinalized);   // this lane's (unconstrained) edge is not tested
      const bool part = valid && lane >= start && mode_ok && PlainPartner(P.flags) &&
                        (P.cons < 0 || P.cons == H.cons) && P.sz < H.sz &&
                        (!fin_l || P.cons >= 0 || H.sz >= T.min_size);
      const bool merging = part && (P.cons >= 0 || !fin_l || P.sz < T.min_size);
      const unsigned long long pend = __ballot(valid && lane >= start);
      const unsigned long long elig = __ballot(merging);
      const unsigned long long stop = pend & ~elig;
      int end = stop ? (int)__builtin_ctzll(stop) : cnt;   // chain = [start, end)
      if (generic_next) end = start;
      if (end > start) {
        const bool in_chain = lane >= start && lane < end;
        const bool case_s = in_chain && P.cons >= 0;
        const int v = in_chain ? P.sz : 0;
        const int incl = WaveInclusiveSum(v);
        const int S = H.sz + incl - v;     // size of R's cluster before this lane's merge
        // MergeStates with o = partner, m = R's cluster (symmetric in the two)
        const float denom = 1.0f / (float)(P.sz + S);
        const float ca = (float)P.sz * denom;
        const float cb = (float)S * denom;
        float c = in_chain ? cb : 1.0f;
        float u0 = in_chain ? ca * P.d0 : 0.0f;
        float u1 = in_chain ? ca * P.d1 : 0.0f;
        float u2 = in_chain ? ca * P.d2 : 0.0f;
        if (lane == start) {   // starts from the cluster's mean and ignores what is shifted in
          u0 = u0 + c * H.d0;
          u1 = u1 + c * H.d1;
          u2 = u2 + c * H.d2;
          c = 0.0f;
        }
        float g0 = u0, g1 = u1, g2 = u2;
        const int steps = __builtin_amdgcn_readfirstlane(end - start);   // scalar loop control
        for (int s8 = 1; s8 < steps; s8 += 8) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            g0 = u0 + c * DppWaveShr1Zero(g0);
            g1 = u1 + c * DppWaveShr1Zero(g1);
            g2 = u2 + c * DppWaveShr1Zero(g2);
          }
        }
        float r0 = DppWaveShr1Old(g0, H.d0);   // mean before this lane's merge
        float r1 = DppWaveShr1Old(g1, H.d1);
        float r2 = DppWaveShr1Old(g2, H.d2);
        if (lane == start) {
          r0 = H.d0;
          r1 = H.d1;
          r2 = H.d2;
        }
        const float x = r0 - P.d0, y = r1 - P.d1, z = r2 - P.d2;
        const float sd = (x * x + y * y + z * z) * (1.0f / 3.0f);
        const bool pass = case_s ? !(sd > T.split_s) : (sd <= T.pass_s);
        const bool tested = case_s || (in_chain && !fin_l);
        const unsigned long long fail = __ballot(in_chain && tested && !pass);
        const int fcut = fail ? (int)__builtin_ctzll(fail) : end;
        if (fcut > start) {
          if (fail) {
            H.d0 = ReadLaneF(r0, fcut);
            H.d1 = ReadLaneF(r1, fcut);
            H.d2 = ReadLaneF(r2, fcut);
            H.sz = ReadLaneI(S, fcut);
          } else {
            H.d0 = ReadLaneF(g0, end - 1);
            H.d1 = ReadLaneF(g1, end - 1);
            H.d2 = ReadLaneF(g2, end - 1);
            H.sz = H.sz + ReadLaneI(incl, end - 1);
          }
          if (in_chain && lane < fcut) {
            nodes.parent[p] = rep;
            if (case_s) ++n_forced; else if (fin_l) ++n_small; else ++n_regular;
          }
        }
        start = fcut;
        generic_next = fail != 0;
        if (!generic_next) continue;
      }
      if (start >= cnt) break;
      // ---- lane `start` through the exact edge semantics (uniform) -------------------------------------------
      generic_next = false;
      const RState Ps = ReadLaneState(P, start);
      const int pid = ReadLaneI(p, start);
      const bool child_first = ReadLaneI(is_a, start) != 0;
      RState s1 = SelectState(child_first, Ps, H);
      RState s2 = SelectState(child_first, H, Ps);
      const RState o1 = s1, o2 = s2;
      int stat;
      const int out = DecideEdge(s1, s2, T, stat);
      if (out == kOutKeep) {   // the structure assumed a merge
        violated = 2;
        // (the edge, by its kept position: where the host cuts the stage -- merge_stage.hip)
        if (lane == 0 && hub_excl && s_gpos) {
          HubViolationAt(hub_excl, 2, (int)s_gpos[sp_is_a[beg + base + start] >> 1]);
        }
        break;
      }
      if (optimistic) {
        const bool vio = (out == kOutMerge1) ? TentativeViolated(o1, o2, s1, s1)
                                             : TentativeViolated(o1, o2, s2, s2);
        if (vio) {
          violated = 1;
          break;
        }
      }
      const bool partner_wins = (out == kOutMerge1) == child_first;
      if (lane == 0) {
        n_forced += (stat == 1);
        n_regular += (stat == 2);
        n_small += (stat == 3);
        if (partner_wins) {
          // R's representative is merged away: it keeps its own constraint field (MergeRegions
          // only updates the survivor), which may have changed since it was loaded
          nodes.cons[rep] = H.cons;
          nodes.parent[rep] = pid;
        } else {
          nodes.parent[pid] = rep;
        }
      }
      if (partner_wins) rep = pid;
      H = SelectState(out == kOutMerge1, s1, s2);
      ++start;
    }
  }
  if (violated) {
    if (lane == 0) {
      if (violated == 1) *violation = 1; else atomicOr(violation, 2);
      __hip_atomic_store(&ring.abort, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;   // the stage is undone
  }
  if (lane == 0) {
    StoreState(nodes, rep, H);
    atomicAdd(&stats[24], (unsigned long long)n);   // spine edges absorbed
    if (start_pos) {
      start_pos[k] = done + n;
      root_vertex[k] = rep;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_forced += __shfl_down(n_forced, off);
    n_regular += __shfl_down(n_regular, off);
    n_small += __shfl_down(n_small, off);
  }
  if (lane == 0) {
    if (n_forced) atomicAdd(&stats[0], (unsigned long long)n_forced);
    if (n_regular) atomicAdd(&stats[1], (unsigned long long)n_regular);
    if (n_small) atomicAdd(&stats[2], (unsigned long long)n_small);
  }
}

// ---- the spine, streamed ------------------------------------------------------------------------------------
// k_spine above spends 24 of its ~30 cycles per side cluster in the recurrence  h <- ca*p + cb*h
// (three channels x two dependent VALU operations through DPP), because one wavefront does
// everything.  Nearly every step of a spine is the same plain step, though -- R's cluster absorbs a
// smaller, unflagged side cluster -- and for a run of such steps everything but the recurrence is
// data parallel over the WHOLE spine, because the side clusters are final before the spine starts:
//   k_spine_prep    (all CUs) representative and state of every side cluster; whether the step is
//                   plain, given R's state at the start (the first step that is not ends the run);
//   exclusive scan  of the side cluster sizes: the size of R's cluster before every step;
//   k_spine_chain   one workgroup per component: producer wavefronts turn (state, size prefix)
//                   into the step's coefficients (u = ca*p per channel, c = cb) in an LDS ring; the
//                   chain wavefront replays  h = u + c*h  with one lane per channel -- two dependent
//                   VALU operations per step for all three channels at once (the dependent pair
//                   alone costs 4.2 ns, tools/micro/dep_chain.hip) --, leaving the mean after every
//                   16th step behind;
//   k_spine_verify  (all CUs) one lane per block of 16 steps replays the block from the mean before
//                   it with the same operations and checks every merge test (and that R's cluster is
//                   the larger one) against the mean before the step; the first failure cuts the run;
//   k_spine_commit / k_spine_finish: parent links and statistics of the steps before the cut, R's
//                   state at the cut, and where k_spine has to take over (start_pos).
// k_spine then replays one batch from the cut (the step that is not plain: a side cluster larger
// than R's cluster, a flag, a failed test), and the rest of the spine gets a second streamed pass.
// Same float operations in the same order as MergeStates: the result is bit-identical to k_spine's.
constexpr int kChRing = 4096;   // ring positions (a multiple of the fill)
constexpr int kChFill = 256;    // positions per producer fill
constexpr int kChFills = kChRing / kChFill;
constexpr int kChProducers = 3;    // wavefronts 1-3: the chain wavefront (0) keeps its SIMD to itself
constexpr int kChBlock = 16;    // steps per block = lanes of a DPP row

struct SpineFastArrays {
  int32_t* p;        // [mt] representative of the side cluster
  float4* ds;        // [mt] its mean and size
  int32_t* meta;     // [mt] bit 0: constrained step (forced), bit 1: tested, bit 2: finalized pair (small)
  int32_t* sizes;    // [mt + 1] side cluster sizes (0 outside the run)
  int32_t* pre;      // [mt + 1] exclusive scan of sizes
  float* ck;         // [3][ck_stride] mean after every block of 16 steps, per channel
  int32_t* stop;     // [K] first step of the component that is not plain (its local index; n: none)
  int32_t* fail;     // [K] first step whose merge test fails
  int32_t* start_pos;   // [K] steps absorbed so far (by earlier passes and by k_spine)
  int ck_stride;
};

// Where component k's block means start in ck.
__device__ __forceinline__ int SpineCkOfs(const int32_t* __restrict__ comp_spine, int k) {
  return (comp_spine[k] >> 4) + 2 * k;
}

__global__ void k_spine_fast_init(int K, const int32_t* __restrict__ comp_spine, SpineFastArrays F, int first) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  const int n = comp_spine[k + 1] - comp_spine[k];
  F.stop[k] = n;
  F.fail[k] = n;
  if (first) F.start_pos[k] = 0;
}

// Is the step "R's cluster (state H at the start of the run) absorbs P" a plain chain step?  The
// conditions of k_spine's chain that do not involve the evolving size of R's cluster, except the
// minimum-size ones, for which the size at the start of the run is good enough (it only grows).
__device__ __forceinline__ bool SpinePlainStep(const RState& H, const RState& P, const StageThr& T, int& meta) {
  const bool fin = (H.flags & kFlagFinalized) != 0;
  const bool mode_ok = !(H.flags & kFlagNoDesc) && (!fin || H.sz >= T.min_size);
  const bool fin_l = fin || (P.flags & kFlagFinalized);
  // (P.sz < size of R's cluster: checked with the exact size before the step, by k_spine_verify)
  const bool part = mode_ok && PlainPartner(P.flags) && (P.cons < 0 || P.cons == H.cons) &&
                    (!fin_l || P.cons >= 0 || H.sz >= T.min_size);
  const bool merging = part && (P.cons >= 0 || !fin_l || P.sz < T.min_size);
  const bool case_s = P.cons >= 0;
  meta = (case_s ? 1 : 0) | ((case_s || !fin_l) ? 2 : 0) | (fin_l ? 4 : 0);
  return merging;
}

__global__ __launch_bounds__(256) void k_spine_prep(int mt, int K, const int32_t* __restrict__ comp_spine,
                                                     const int32_t* __restrict__ root_vertex,
                                                     const int32_t* __restrict__ sp_child, NodeArrays nodes,
                                                     StageThr T, SpineFastArrays F) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > mt) return;
  const int mS = comp_spine[K];
  if (i >= mS) {
    F.sizes[i] = 0;
    return;
  }
  const int k = CompOf(comp_spine, K, i);
  const int local = i - comp_spine[k];
  if (local < F.start_pos[k]) {   // absorbed already
    F.sizes[i] = 0;
    return;
  }
  int c = sp_child[i];
  for (int pa = nodes.parent[c]; pa != c; pa = nodes.parent[c]) c = pa;
  const RState P = LoadState(nodes, c);
  const RState H = LoadState(nodes, root_vertex[k]);
  int meta;
  if (!SpinePlainStep(H, P, T, meta)) atomicMin(&F.stop[k], local);
  F.p[i] = c;
  F.ds[i] = make_float4(P.d0, P.d1, P.d2, __int_as_float(P.sz));
  F.meta[i] = meta;
  F.sizes[i] = P.sz;
}

// One step of MergeStates with o = side cluster (mean d, size psz), m = R's cluster (size S before
// the step): coefficients of  h <- u + c*h.
__device__ __forceinline__ void SpineCoefficients(const float4& ds, int S, float& u0, float& u1, float& u2,
                                                  float& c) {
  const int psz = __float_as_int(ds.w);
  const float denom = 1.0f / (float)(psz + S);
  const float ca = (float)psz * denom;
  c = (float)S * denom;
  u0 = ca * ds.x;
  u1 = ca * ds.y;
  u2 = ca * ds.z;
}

struct ChainRing {
  alignas(16) float u[3][kChRing];
  alignas(16) float c[kChRing];
  int ready[kChFills];   // fill f is complete when ready[f % kChFills] == f + 1
  int consumed;          // positions the chain wavefront is done with
};

__global__ __launch_bounds__(64 * (1 + kChProducers)) void k_spine_chain(int K, const int32_t* __restrict__ comp_spine,
                                                      const int32_t* __restrict__ root_vertex, NodeArrays nodes,
                                                      SpineFastArrays F) {
  __shared__ ChainRing ring;
  const int k = blockIdx.x;
  if (k >= K) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int beg = comp_spine[k];
  const int start = F.start_pos[k], stop = F.stop[k];   // the run: steps [start, stop)
  if (stop <= start) return;
  const int f0 = start / kChFill;
  if (threadIdx.x < kChFills) ring.ready[threadIdx.x] = 0;
  if (threadIdx.x == 0) ring.consumed = f0 * kChFill;
  __syncthreads();
  const RState H = LoadState(nodes, root_vertex[k]);
  const int nfill = (stop + kChFill - 1) / kChFill;
  if (wave >= 1) {
    // ---- producers: coefficients of every step -----------------------------------------------------------
    const int pre0 = F.pre[beg];
    for (int f = f0 + wave - 1; f < nfill; f += kChProducers) {
      const int first = f * kChFill;
      while (first + kChFill - __hip_atomic_load(&ring.consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >
             kChRing) {
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int q = 0; q < kChFill / 64; ++q) {
        const int i = first + q * 64 + lane;
        float u0 = 0.0f, u1 = 0.0f, u2 = 0.0f, cc = 1.0f;   // outside the run: the mean stays
        if (i >= start && i < stop) {
          SpineCoefficients(F.ds[beg + i], H.sz + (F.pre[beg + i] - pre0), u0, u1, u2, cc);
        }
        const int slot = i & (kChRing - 1);
        ring.u[0][slot] = u0;
        ring.u[1][slot] = u1;
        ring.u[2][slot] = u2;
        ring.c[slot] = cc;
      }
      WaveSyncS();
      if (lane == 0) {
        __hip_atomic_store(&ring.ready[f % kChFills], f + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    return;
  }
  // ---- the chain: lane ch replays channel ch (the other lanes repeat channel lane % 3, unused) ---------------
  // (Coefficients through DPP row shifts -- one lane per row, the 16 steps of a block in the lanes
  // of the row -- need no LDS reads at all, but a dependent DPP operation has three times the
  // latency of a plain one: 12 ns per step instead of 5.)
  __builtin_amdgcn_s_setprio(3);
  const int ch = lane % 3;
  float h = ch == 0 ? H.d0 : (ch == 1 ? H.d1 : H.d2);
  float* ckout = F.ck + (size_t)ch * F.ck_stride + SpineCkOfs(comp_spine, k);
  const float* urow = ring.u[ch];
  const float* crow = ring.c;
  for (int f = f0; f < nfill; ++f) {
    while (__hip_atomic_load(&ring.ready[f % kChFills], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != f + 1) {
      __builtin_amdgcn_s_sleep(1);
    }
    const int first = f * kChFill;
    const int slot0 = first & (kChRing - 1);   // a fill never wraps
    // 16 steps per block; the coefficients of the next block are read from LDS while this block's
    // dependent chain runs (the chain itself is the only thing the wavefront may wait for).
    float4 un[4], cn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      un[q] = *reinterpret_cast<const float4*>(urow + slot0 + 4 * q);
      cn[q] = *reinterpret_cast<const float4*>(crow + slot0 + 4 * q);
    }
    for (int b = 0; b < kChFill / kChBlock; ++b) {
      float4 u[4], c[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u[q] = un[q];
        c[q] = cn[q];
      }
      if (b + 1 < kChFill / kChBlock) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          un[q] = *reinterpret_cast<const float4*>(urow + slot0 + kChBlock * (b + 1) + 4 * q);
          cn[q] = *reinterpret_cast<const float4*>(crow + slot0 + kChBlock * (b + 1) + 4 * q);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // the reads are issued here, not where the scheduler would sink them
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h = u[q].x + c[q].x * h;
        h = u[q].y + c[q].y * h;
        h = u[q].z + c[q].z * h;
        h = u[q].w + c[q].w * h;
      }
      const int at = first + kChBlock * b;
      if (lane < 3 && at < stop) ckout[at >> 4] = h;
    }
    WaveSyncS();   // the fill has been read: its slots may be reused
    if (lane == 0) {
      __hip_atomic_store(&ring.consumed, first + kChFill, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// The block of 16 steps that holds step `upto` (or the whole block `b` when upto < 0), replayed from
// the mean before it; checks every step when `check` is set.  Returns the mean before step `upto`
// (after the block otherwise) in h; the first step that fails (-1: none) as the result.
__device__ __forceinline__ int SpineReplayBlock(int k, int b, int upto, bool check,
                                                const int32_t* __restrict__ comp_spine, const RState& H,
                                                int start, int stop, const StageThr& T, const SpineFastArrays& F,
                                                float& h0, float& h1, float& h2) {
  const int beg = comp_spine[k];
  const int lo = max(kChBlock * b, start);
  int hi = min(kChBlock * b + kChBlock, stop);
  if (upto >= 0) hi = min(hi, upto);
  if (kChBlock * b <= start) {   // the run starts in this block: R's state
    h0 = H.d0;
    h1 = H.d1;
    h2 = H.d2;
  } else {
    const size_t at = (size_t)SpineCkOfs(comp_spine, k) + b - 1;
    h0 = F.ck[at];
    h1 = F.ck[(size_t)F.ck_stride + at];
    h2 = F.ck[2 * (size_t)F.ck_stride + at];
  }
  const int pre0 = F.pre[beg];
  for (int i = lo; i < hi; ++i) {
    const float4 ds = F.ds[beg + i];
    const int S = H.sz + (F.pre[beg + i] - pre0);
    if (check) {
      // R's cluster has to be the larger one (ties go to the edge's second region: k_spine decides)
      if (!(__float_as_int(ds.w) < S)) return i;
      const int meta = F.meta[beg + i];
      if (meta & 2) {
        const float x = h0 - ds.x, y = h1 - ds.y, z = h2 - ds.z;
        const float sd = (x * x + y * y + z * z) * (1.0f / 3.0f);
        const bool pass = (meta & 1) ? !(sd > T.split_s) : (sd <= T.pass_s);
        if (!pass) return i;
      }
    }
    float u0, u1, u2, c;
    SpineCoefficients(ds, S, u0, u1, u2, c);
    h0 = u0 + c * h0;
    h1 = u1 + c * h1;
    h2 = u2 + c * h2;
  }
  return -1;
}

__global__ __launch_bounds__(256) void k_spine_verify(int K, const int32_t* __restrict__ comp_spine,
                                                       const int32_t* __restrict__ root_vertex, NodeArrays nodes,
                                                       StageThr T, SpineFastArrays F) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= F.ck_stride) return;
  int lo = 0, hi = K;   // the component whose blocks hold slot t of ck
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (SpineCkOfs(comp_spine, mid) <= t) lo = mid; else hi = mid;
  }
  const int k = lo, b = t - SpineCkOfs(comp_spine, k);
  const int start = F.start_pos[k], stop = F.stop[k];
  if (b < 0 || kChBlock * b >= stop || kChBlock * b + kChBlock <= start) return;
  const RState H = LoadState(nodes, root_vertex[k]);
  float h0, h1, h2;
  const int bad = SpineReplayBlock(k, b, -1, true, comp_spine, H, start, stop, T, F, h0, h1, h2);
  if (bad >= 0) atomicMin(&F.fail[k], bad);
}

constexpr int kCommitPer = 4;
__global__ __launch_bounds__(256) void k_spine_commit(int mt, int K, const int32_t* __restrict__ comp_spine,
                                                       const int32_t* __restrict__ root_vertex, NodeArrays nodes,
                                                       SpineFastArrays F, unsigned long long* __restrict__ stats) {
  __shared__ int red[3][4];
  int forced = 0, small = 0, regular = 0;
  const int mS = comp_spine[K];
#pragma unroll
  for (int q = 0; q < kCommitPer; ++q) {
    const int i = (blockIdx.x * kCommitPer + q) * 256 + threadIdx.x;
    if (i < mt && i < mS) {
      const int k = CompOf(comp_spine, K, i);
      const int local = i - comp_spine[k];
      if (local >= F.start_pos[k] && local < min(F.stop[k], F.fail[k])) {
        nodes.parent[F.p[i]] = root_vertex[k];
        const int meta = F.meta[i];
        const bool f = (meta & 1) != 0, sm = !f && (meta & 4);
        forced += f;
        small += sm;
        regular += !f && !sm;
      }
    }
  }
  // (one set of additions per workgroup: the three counters are single words)
  for (int off = 32; off > 0; off >>= 1) {
    forced += __shfl_down(forced, off);
    small += __shfl_down(small, off);
    regular += __shfl_down(regular, off);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = forced;
    red[1][threadIdx.x >> 6] = regular;
    red[2][threadIdx.x >> 6] = small;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (v) atomicAdd(&stats[threadIdx.x], (unsigned long long)v);
  }
}

// R's state after the committed steps, and where k_spine goes on.
__global__ void k_spine_finish(int K, const int32_t* __restrict__ comp_spine,
                               const int32_t* __restrict__ root_vertex, NodeArrays nodes, StageThr T,
                               SpineFastArrays F, unsigned long long* __restrict__ stats) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  const int start = F.start_pos[k], stop = F.stop[k];
  const int a = min(stop, F.fail[k]);
  if (a <= start) return;
  const int rep = root_vertex[k];
  const int beg = comp_spine[k];
  RState H = LoadState(nodes, rep);
  float h0, h1, h2;
  // the mean before step a: the block that holds step a - 1, up to a
  SpineReplayBlock(k, (a - 1) >> 4, a, false, comp_spine, H, start, stop, T, F, h0, h1, h2);
  H.d0 = h0;
  H.d1 = h1;
  H.d2 = h2;
  H.sz += F.pre[beg + a] - F.pre[beg];
  nodes.desc_sz[rep] = make_float4(H.d0, H.d1, H.d2, __int_as_float(H.sz));
  F.start_pos[k] = a;
  atomicAdd(&stats[24], (unsigned long long)(a - start));
}

// Debug self check (VSG_SPINE_CHECK): the classification against a sequential replay of the
// structure on the host -- Kruskal with "is in R's cluster" per cluster.
void SpineSelfCheck(const std::vector<int32_t>& base, int mE, const int32_t* d_eu, const int32_t* d_ev,
                    const int32_t* d_root, const int32_t* d_side_key, const int32_t* d_spine_flag,
                    const int32_t* d_te_e, int mt) {
  const int K = (int)base.size() - 1;
  std::vector<int32_t> eu(mE), ev(mE), side_key(mE), spine_flag(mE), te_e(mt), root(K);
  VSG_HIP(hipMemcpy(eu.data(), d_eu, mE * sizeof(int32_t), hipMemcpyDeviceToHost));
  VSG_HIP(hipMemcpy(ev.data(), d_ev, mE * sizeof(int32_t), hipMemcpyDeviceToHost));
  VSG_HIP(hipMemcpy(side_key.data(), d_side_key, mE * sizeof(int32_t), hipMemcpyDeviceToHost));
  VSG_HIP(hipMemcpy(spine_flag.data(), d_spine_flag, mE * sizeof(int32_t), hipMemcpyDeviceToHost));
  VSG_HIP(hipMemcpy(te_e.data(), d_te_e, mt * sizeof(int32_t), hipMemcpyDeviceToHost));
  VSG_HIP(hipMemcpy(root.data(), d_root, K * sizeof(int32_t), hipMemcpyDeviceToHost));
  long long bad = 0;
  for (int k = 0; k < K; ++k) {
    std::unordered_map<int, int> id;   // vertex -> dense index
    std::vector<int> uf, t;
    std::vector<char> in_r;
    std::vector<std::vector<int>> members;
    auto vid = [&](int x) {
      auto it = id.find(x);
      if (it != id.end()) return it->second;
      const int i = (int)uf.size();
      id.emplace(x, i);
      uf.push_back(i);
      t.push_back(-1);
      in_r.push_back(0);
      members.push_back({i});
      return i;
    };
    auto find = [&](int x) {
      while (uf[x] != x) x = uf[x] = uf[uf[x]];
      return x;
    };
    in_r[vid(root[k])] = 1;
    std::vector<int> cls(base[k + 1] - base[k]);   // 0 dropped, 1 side, 2 spine
    for (int e = base[k]; e < base[k + 1]; ++e) {
      int a = find(vid(eu[e])), b = find(vid(ev[e]));
      int c;
      if (a == b) {
        c = in_r[a] ? 0 : 1;
      } else {
        if (in_r[a] != in_r[b]) {
          c = 2;
          for (int x : members[in_r[a] ? b : a]) t[x] = e;
        } else {
          c = 1;
        }
        if (members[a].size() < members[b].size()) std::swap(a, b);
        uf[b] = a;
        in_r[a] = in_r[a] | in_r[b];
        members[a].insert(members[a].end(), members[b].begin(), members[b].end());
        std::vector<int>().swap(members[b]);
      }
      cls[e - base[k]] = c;
    }
    for (int e = base[k]; e < base[k + 1]; ++e) {
      const int c = cls[e - base[k]];
      const bool side = side_key[e] != kNone;
      if (side != (c == 1)) ++bad;
      else if (side && te_e[side_key[e]] != t[id[eu[e]]]) ++bad;
      if ((spine_flag[e] != 0) != (c == 2)) ++bad;
      if (bad && bad < 8) {
        std::fprintf(stderr, "[vsg] spine check: comp %d edge %d expected class %d t %d, device side_key %d (rank %d) spine %d\n",
                     k, e, c, t[id[eu[e]]], side_key[e], side_key[e] != kNone ? te_e[side_key[e]] : -1,
                     spine_flag[e]);
      }
    }
  }
  if (bad) throw Error(-4 /* VSG_ERR_INTERNAL */, "spine: the structure differs from the sequential replay");
}

// Components of at least min_cnt replayed edges: (offset, count) pairs.  out[0] = number found
// (may exceed the capacity).
// (offset, count) pairs go straight into the mailbox's list in mapped host memory; *found may exceed
// the capacity.
__global__ __launch_bounds__(256) void k_list_large_segments(int max_segs, const int32_t* __restrict__ num_segs,
                                                              const int32_t* __restrict__ seg_off,
                                                              const int32_t* __restrict__ seg_cnt, int min_cnt,
                                                              int32_t* __restrict__ found, int32_t* __restrict__ out) {
  const int seg = blockIdx.x * 256 + threadIdx.x;
  if (seg >= max_segs || seg >= *num_segs) return;
  const int cnt = seg_cnt[seg];
  if (cnt < min_cnt) return;
  const int i = atomicAdd(found, 1);
  if (i < kSpineListCap) {
    out[2 * i] = seg_off[seg];
    out[2 * i + 1] = cnt;
  }
}

struct Pool {   // stack allocator over the spine scratch
  int32_t* base;
  size_t cap, used = 0;
  bool ok = true;
  size_t mark() const { return used; }
  void release(size_t m) { used = m; }
  int32_t* take(size_t n) {
    n = (n + 63) & ~(size_t)63;
    if (used + n > cap) {
      ok = false;
      return base;
    }
    int32_t* p = base + used;
    used += n;
    return p;
  }
};

}  // namespace

// 7 ints per edge, 29 per tree edge while the trees are rooted (a fifth to two thirds of the edges
// are tree edges), nested levels on top: a level that does not fit is left to the ordinary workers.
size_t SpinePoolInts(size_t max_edges) { return 16 * max_edges + 64 * 1024; }

int SelectLargeSegments(int max_segs, const int32_t* num_segs, const int32_t* seg_off, const int32_t* seg_cnt,
                        int min_cnt, long long max_edges, MergeScratch& S, hipStream_t s, SpineInput* out,
                        long long* wanted_edges) {
  out->segs.clear();
  if (wanted_edges) *wanted_edges = 0;
  Mailbox& mb = *S.mail;
  VSG_REQUIRE(mb.list_cap >= 2 * kSpineListCap, -4, "mailbox list too small");
  std::vector<int32_t> large(1 + 2 * kSpineListCap);
  int found = 0;
  for (;; min_cnt *= 4) {   // too many for the list: only the larger ones
    int32_t* d_found = TakeZeroed(S, 1);
    hipLaunchKernelGGL(k_list_large_segments, dim3(Blocks(max_segs)), dim3(256), 0, s, max_segs, num_segs, seg_off,
                       seg_cnt, min_cnt, d_found, mb.list_dev);
    // (the post follows the end of the listing kernel: its stores to the mapped list are complete)
    const MailSlot m = NextMail(mb);
    LaunchMailPost(m, d_found, nullptr, nullptr, nullptr, s);
    MailWait(m, 1, &found, s);
    if (found <= kSpineListCap) break;
  }
  large[0] = found;
  for (int i = 0; i < 2 * found; ++i) large[1 + i] = mb.list_host[i];
  if (getenv("VSG_SPINE_DEBUG") && max_segs > 1000000) {
    long long sum = 0;
    int mx = 0;
    for (int i = 0; i < found; ++i) {
      sum += large[2 + 2 * i];
      mx = std::max(mx, large[2 + 2 * i]);
    }
    std::fprintf(stderr, "[vsg] spine select: max_segs %d min_cnt %d found %d sum %lld max %d (max_edges %lld)\n",
                 max_segs, min_cnt, found, sum, mx, max_edges);
  }
  if (found == 0) return 0x7fffffff;
  // The largest components first, as many as the scratch pool holds (at most 256): the threshold is
  // the size of the smallest one taken -- everything of at least that size has to be taken, so ties
  // at the boundary go together.  (Doubling the threshold until everything fits drops fifteen
  // components of similar size all at once.)
  std::vector<int> sizes((size_t)found);
  long long all = 0;
  for (int i = 0; i < found; ++i) {
    sizes[(size_t)i] = large[2 + 2 * i];
    all += sizes[(size_t)i];
  }
  if (wanted_edges) *wanted_edges = all;
  std::sort(sizes.begin(), sizes.end(), std::greater<int>());
  long long sum = 0;
  int taken = 0, thr = 0x7fffffff;
  for (int i = 0; i < found && i < 256;) {
    int j = i;
    long long group = 0;
    while (j < found && sizes[(size_t)j] == sizes[(size_t)i]) group += sizes[(size_t)j++];
    if (sum + group > max_edges || j > 256) break;
    sum += group;
    taken = j;
    thr = sizes[(size_t)i];
    i = j;
  }
  if (taken == 0) return 0x7fffffff;
  for (int i = 0; i < found; ++i) {
    if (large[2 + 2 * i] >= thr) out->segs.push_back(SpineSeg{large[1 + 2 * i], large[2 + 2 * i]});
  }
  return thr;
}

// Replays the listed large components (segments of the stage's component-sorted edge arrays).
// Returns false when the scratch pool cannot hold them (the caller hands them to the ordinary
// workers instead).  On return the side clusters and the spines are queued on the stream; a
// failed speculation raises *wa.violation.
bool RunSpineComponents(const SpineInput& in, const WorkerArgs& wa, MergeScratch& S, hipStream_t s,
                        const SpineWorkers& run_workers, size_t pool_used, int depth) {
  const int K = (int)in.segs.size();
  if (K == 0) return false;
  std::vector<int32_t> base(K + 1, 0), off(K);
  for (int k = 0; k < K; ++k) {
    base[k + 1] = base[k] + in.segs[k].cnt;
    off[k] = in.segs[k].off;
  }
  const int mE = base[K];
  VSG_REQUIRE(mE < (1 << 27), -4, "spine: too many edges for the stamped ranks");
  double tph[8] = {0};
  auto Mark = [&](int i) {   // debug: phase boundaries (synchronises)
    if (!S.spine_debug) return;
    VSG_HIP(hipStreamSynchronize(s));
    tph[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  Mark(0);
  if (pool_used >= S.spine_pool_ints) return false;
  Pool pool{S.spine_pool + pool_used, S.spine_pool_ints - pool_used};
  int32_t* d_base = pool.take(K + 1);
  int32_t* d_off = pool.take(K);
  int32_t* root_vertex = pool.take(K);
  int32_t* comp_spine = pool.take(K + 1);
  unsigned long long* root_key = reinterpret_cast<unsigned long long*>(TakeZeroed(S, 2 * (size_t)K));
  int32_t* scalars = pool.take(64);
  int32_t* eu = pool.take(mE);
  int32_t* ev = pool.take(mE);
  int32_t* estate = pool.take(mE);
  int32_t* flag = pool.take(mE);
  int32_t* scan = pool.take(mE);
  int32_t* side_key = pool.take(mE);
  int32_t* spine_flag = pool.take(mE);
  if (!pool.ok) return false;
  int32_t* cc = S.cc;
  uint32_t* best = reinterpret_cast<uint32_t*>(S.nmap[0]);
  int32_t* firstq = S.nmap[1];
  int32_t* childidx = S.nmap[2];

  VSG_HIP(hipMemcpyAsync(d_base, base.data(), (K + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
  VSG_HIP(hipMemcpyAsync(d_off, off.data(), K * sizeof(int32_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_spine_gather, dim3(Blocks(mE)), dim3(256), 0, s, mE, K, d_base, d_off, wa.s_ra,
                     wa.s_rb, eu, ev, estate, cc, best);
  hipLaunchKernelGGL(k_spine_pick_root, dim3(Blocks((size_t)K + (mE + kRootStride - 1) / kRootStride)), dim3(256), 0,
                     s, mE, K, eu, ev, d_base, wa.nodes, root_key);
  hipLaunchKernelGGL(k_spine_roots, dim3((K + 63) / 64), dim3(64), 0, s, K, root_key, root_vertex, childidx);

  // ---- tree edges ----------------------------------------------------------------------------------------
  int dbg_rounds = 0;
  auto NowMs = [] {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  const bool dbg_big = S.spine_debug && mE > (8 << 20);
  if (dbg_big) {
    VSG_HIP(hipStreamSynchronize(s));
    std::fprintf(stderr, "[vsg]   forest: gather+root %.2f ms\n", NowMs() - tph[0]);
  }
  {
    const size_t forest_mark = pool.mark();
    const int32_t* list = nullptr;      // the edges the rounds still look at (null: all)
    const int32_t* list_len = nullptr;  // its length, on the device
    int n_list = mE;                    // its capacity
    int32_t* lists[2] = {nullptr, nullptr};
    int which = 0;
    // Rounds over a short list are launched one ahead of the host's knowledge: the number of live
    // edges of round r is read while round r + 1 is queued behind it (gated on the device: empty when
    // round r found nothing alive), so the GPU does not idle through a host round trip per round.
    // Long lists (the giant components) stay synchronous: there the round after next is worth
    // compacting for, and an unneeded pass over 40 M edges is not free.
    const int ahead_max = getenv("VSG_BOR_AHEAD") ? atoi(getenv("VSG_BOR_AHEAD")) : (4 << 20);
    // (lists shorter than this are not worth a compaction pass; the tests lower it so that every
    // small case compacts after every round)
    const int compact_min = getenv("VSG_BOR_COMPACT_MIN") ? atoi(getenv("VSG_BOR_COMPACT_MIN")) : (1 << 16);
    MailSlot waiting[2];
    int n_waiting = 0;
    const int32_t* gate = nullptr;
    int32_t* len_words = scalars + 8;   // [2]: the length of the current list / of the one the next compaction builds
    int32_t* gate_words = scalars + 12;  // [2]: live edges of the last two rounds
    int len_cur = 0;
    for (int round = 0;; ++round) {
      dbg_rounds = round;
      VSG_REQUIRE(round < 32, -4, "spine: the spanning forest did not converge");
      int32_t* d_ctr = TakeZeroed(S, 16 * kAliveSlots + 4);   // edges alive in this round (spread), then the round's total
      int32_t* d_next_len = len_words + (len_cur ^ 1);
      // ecu / ecv (the components of the round) live in the side_key / spine_flag arrays, which are
      // only written once the forest is done
      hipLaunchKernelGGL(k_bor_min, dim3(Blocks(n_list)), dim3(256), 0, s, n_list, list, list_len, round, eu, ev,
                         estate, cc, best, side_key, spine_flag, d_ctr, gate, d_next_len);
      const MailSlot m_alive = NextMail(*S.mail);
      hipLaunchKernelGGL(k_bor_hook, dim3(Blocks(n_list)), dim3(256), 0, s, n_list, list, list_len, round, eu, ev,
                         estate, cc, best, side_key, spine_flag, d_ctr, m_alive.dev, m_alive.seq, gate,
                         gate_words + (round & 1));
      // (the round's total = the next round's gate: one of two words of the pool's scalars, written
      // unconditionally by k_bor_hook -- not a word of the rotating zero pool, which the kernels of
      // round r + 1 would read after the next TakeZeroed)
      gate = gate_words + (round & 1);
      waiting[n_waiting++] = m_alive;
      const int depth = n_list <= ahead_max ? 1 : 0;   // rounds that may stay unanswered
      const double tr0 = dbg_big ? NowMs() : 0;
      int alive = -1;
      bool complete = false;
      while (n_waiting > depth) {
        MailWait(waiting[0], 1, &alive, s);
        waiting[0] = waiting[1];
        --n_waiting;
        if (alive == 0) {
          complete = true;   // (a round launched behind it is empty)
          break;
        }
      }
      if (dbg_big) std::fprintf(stderr, "[vsg]   forest round %d: %d in list, %d alive, %.2f ms\n", round, n_list, alive, NowMs() - tr0);
      if (complete) break;
      // Drop the settled edges from the rounds to come.  Synchronous rounds know how many edges are
      // alive and compact when that is less than half the list.  One round ahead of the answers the
      // host cannot wait for a count before it decides: from the second round on a long list is
      // compacted after every round, and how many edges that leaves is only known on the device
      // (list_len).  n_list -- the capacity the kernels are launched for -- may only shrink together
      // with a compaction: `alive` (the last count answered) bounds what the NEW list holds, not the
      // entries of a list that stays.
      const bool compact = depth == 1 ? (round >= 1 && n_list > compact_min)
                                      : (alive < n_list / 2 && n_list > compact_min);
      if (!compact) continue;
      if (!lists[0]) {
        const size_t m = pool.mark();
        const size_t cap = depth == 1 ? (size_t)n_list : (size_t)alive;
        lists[0] = pool.take(cap);
        lists[1] = pool.take(cap);
        if (!pool.ok) {   // without room the rounds simply keep the longer list
          pool.ok = true;
          pool.release(m);
          lists[0] = lists[1] = nullptr;
        }
      }
      if (!lists[0]) continue;
      hipLaunchKernelGGL(k_bor_compact, dim3(Blocks(((size_t)n_list + kBorCompactPer - 1) / kBorCompactPer)),
                         dim3(256), 0, s, n_list, list, list_len, estate, lists[which], d_next_len);
      list = lists[which];
      list_len = d_next_len;
      len_cur ^= 1;
      which ^= 1;
      // (at most `alive` edges are left: the ones hooked since that count are gone as well)
      if (alive >= 0) n_list = std::min(n_list, alive);
    }
    pool.release(forest_mark);
  }
  Mark(1);
  const MailSlot m_mt = NextMail(*S.mail);
  FusedScan(S.scan, StateFlagValue{estate, 1}, FlagScanEmit{flag, scan}, TotalPostFinish{scalars + 1, m_mt.dev, m_mt.seq},
            mE, s);
  int mt = 0;
  MailWait(m_mt, 1, &mt, s);
  VSG_REQUIRE(mt > 0, -4, "spine: a component without tree edges");
  const int na = 2 * mt;
  int32_t* te_e = pool.take(mt);
  int32_t* child = pool.take(mt);
  int32_t* par = pool.take(mt);
  int32_t* jump[2] = {pool.take(mt), pool.take(mt)};
  int32_t* val[2] = {pool.take(mt), pool.take(mt)};
  int32_t* head = pool.take(mt);
  int32_t* sp_child = pool.take(mt);
  int32_t* sp_is_a = pool.take(mt);
  const size_t arcs_mark = pool.mark();

  // ---- root the trees: Euler tour, list ranking -----------------------------------------------------------------
  uint32_t* ak_in = reinterpret_cast<uint32_t*>(pool.take(na));
  uint32_t* av_in = reinterpret_cast<uint32_t*>(pool.take(na));
  uint32_t* ak = reinterpret_cast<uint32_t*>(pool.take(na));
  uint32_t* av = reinterpret_cast<uint32_t*>(pool.take(na));
  int32_t* pos_arc = pool.take(na);
  int32_t* succ[2] = {pool.take(na), pool.take(na)};
  int32_t* dist[2] = {pool.take(na), pool.take(na)};
  if (!pool.ok) return false;
  hipLaunchKernelGGL(k_compact_tree, dim3(Blocks(mE)), dim3(256), 0, s, mE, flag, scan, te_e);
  hipLaunchKernelGGL(k_arc_keys, dim3(Blocks(na)), dim3(256), 0, s, mt, te_e, eu, ev, ak_in, av_in);
  SortPairsU32(S.cub_temp, S.cub_temp_bytes, ak_in, ak, av_in, av, na, S.node_key_bits, s);
  hipLaunchKernelGGL(k_arc_first, dim3(Blocks(na)), dim3(256), 0, s, na, ak, av, firstq, pos_arc);
  hipLaunchKernelGGL(k_arc_succ, dim3(Blocks(na)), dim3(256), 0, s, na, ak, av, firstq, pos_arc, te_e, d_base, K,
                     root_vertex, succ[0], dist[0]);
  int cur = 0;
  bool ranked = false;
  if (na >= S.rank_split_min) {
    // sampled ranking: the reduced list (every 64th arc + the K tour heads) by pointer jumping
    const int ns = (na + kRankSplit - 1) / kRankSplit + K;
    const size_t rmark = pool.mark();
    int32_t* r_next[2] = {pool.take(ns), pool.take(ns)};
    int32_t* r_len[2] = {pool.take(ns), pool.take(ns)};
    if (pool.ok) {
      int32_t* owner = succ[1];
      int32_t* local = dist[0];   // (k_arc_succ's unit weights are not needed here)
      hipLaunchKernelGGL(k_rank_walk, dim3(Blocks(ns)), dim3(256), 0, s, na, K, succ[0], root_vertex, firstq, av,
                         owner, local, r_next[0], r_len[0]);
      int rc = 0;
      for (int span = 1; span < ns; span *= 2) {
        hipLaunchKernelGGL(k_rank_step, dim3(Blocks(ns)), dim3(256), 0, s, ns, r_next[rc], r_len[rc], r_next[rc ^ 1],
                           r_len[rc ^ 1]);
        rc ^= 1;
      }
      hipLaunchKernelGGL(k_rank_finish, dim3(Blocks(na)), dim3(256), 0, s, na, owner, local, r_len[rc], dist[1]);
      cur = 1;
      ranked = true;
    } else {
      pool.ok = true;
    }
    pool.release(rmark);
  }
  if (!ranked) {
    for (int span = 1; span < na; span *= 2) {
      hipLaunchKernelGGL(k_rank_step, dim3(Blocks(na)), dim3(256), 0, s, na, succ[cur], dist[cur], succ[cur ^ 1],
                         dist[cur ^ 1]);
      cur ^= 1;
    }
  }
  hipLaunchKernelGGL(k_tree_parent, dim3(Blocks(mt)), dim3(256), 0, s, mt, te_e, eu, ev, dist[cur], child, par,
                     childidx);

  Mark(2);
  // ---- t(x), side clusters -------------------------------------------------------------------------------------
  hipLaunchKernelGGL(k_jump_init, dim3(Blocks(mt)), dim3(256), 0, s, mt, te_e, par, childidx, jump[0], val[0]);
  // The jumps double the distance covered; the trees are usually shallow (a hub and what hangs off
  // it), so every few steps the device says whether any vertex is still short of its root.
  int cj = 0;
  {
    // (the answer to a question is read three steps later: a step behind a finished walk only copies,
    // so the steps launched meanwhile change nothing, and the GPU does not wait for the host)
    MailSlot asked = {nullptr, nullptr, 0};
    bool have_asked = false;
    for (int span = 1, step = 0; span < mt; span *= 2, ++step) {
      const bool ask = (step % 3 == 2) && span * 2 < mt;
      int32_t* d_more = ask ? TakeZeroed(S, 1) : nullptr;
      hipLaunchKernelGGL(k_jump_max, dim3(Blocks(mt)), dim3(256), 0, s, mt, jump[cj], val[cj], jump[cj ^ 1],
                         val[cj ^ 1], d_more);
      cj ^= 1;
      if (ask) {
        const MailSlot m = NextMail(*S.mail);
        LaunchMailPost(m, d_more, nullptr, nullptr, nullptr, s);
        if (have_asked) {
          int more = 0;
          MailWait(asked, 1, &more, s);
          if (!more) break;
        }
        asked = m;
        have_asked = true;
      }
    }
  }
  // (scan still maps an edge to its position among the tree edges)
  hipLaunchKernelGGL(k_head_lookup, dim3(Blocks(mt)), dim3(256), 0, s, mt, val[cj], scan, head);
  hipLaunchKernelGGL(k_classify, dim3(Blocks(mE)), dim3(256), 0, s, mE, eu, ev, estate, childidx, head,
                     te_e, flag, side_key, spine_flag);

  Mark(3);
  // ---- side clusters: segments for the ordinary workers ------------------------------------------------------
  pool.release(arcs_mark);   // the tour is done with
  const MailSlot m_side = NextMail(*S.mail);
  FusedScan(S.scan, ScanLoadI32{flag}, FlagScanEmit{nullptr, scan}, TotalPostFinish{scalars + 2, m_side.dev, m_side.seq},
            mE, s);
  int n_side = 0;
  MailWait(m_side, 1, &n_side, s);
  const size_t ns = (size_t)(n_side > 0 ? n_side : 1);
  uint32_t* sk_in = reinterpret_cast<uint32_t*>(pool.take(ns));
  uint32_t* si_in = reinterpret_cast<uint32_t*>(pool.take(ns));
  uint32_t* sk = reinterpret_cast<uint32_t*>(pool.take(ns));
  uint32_t* si = reinterpret_cast<uint32_t*>(pool.take(ns));
  uint32_t* seg_key = reinterpret_cast<uint32_t*>(pool.take(ns));
  int32_t* seg_cnt = pool.take(ns);
  int32_t* seg_off = pool.take(ns);
  int32_t* o_ra = pool.take(ns);
  int32_t* o_rb = pool.take(ns);
  uint32_t* o_gpos = reinterpret_cast<uint32_t*>(pool.take(ns));
  if (!pool.ok) return false;
  hipLaunchKernelGGL(k_compact_side, dim3(Blocks(mE)), dim3(256), 0, s, mE, flag, scan, side_key, sk_in, si_in);
  // spine edges (the scan buffer is reused once the side positions are consumed)
  FusedScan(S.scan, StateFlagValue{spine_flag, 1}, SpineEmit{eu, ev, childidx, te_e, scan, sp_child, sp_is_a, d_base, d_off, K},
            SpineFinish{comp_spine, K}, mE, s);
  hipLaunchKernelGGL(k_comp_spine, dim3(Blocks(K)), dim3(256), 0, s, K, d_base, scan, comp_spine);
  int dbg_spine_edges = 0;
  if (S.spine_debug) {
    std::vector<int32_t> cs(K + 1);
    VSG_HIP(hipMemcpy(cs.data(), comp_spine, (K + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    dbg_spine_edges = cs[K];
  }
  Mark(4);
  if (S.spine_check) SpineSelfCheck(base, mE, eu, ev, root_vertex, side_key, spine_flag, te_e, mt);
  if (n_side > 0) {
    int side_bits = 1;   // the keys are tree-edge indices
    while (side_bits < 32 && (1ll << side_bits) < (long long)mt) ++side_bits;
    SortPairsU32(S.cub_temp, S.cub_temp_bytes, sk_in, sk, si_in, si, n_side, side_bits, s);
    int32_t* d_nseg = scalars + 3;
    RunsOfSortedKeys(S.scan, sk, n_side, seg_off, seg_cnt, d_nseg, s);
    hipLaunchKernelGGL(k_gather_side, dim3(Blocks(n_side)), dim3(256), 0, s, n_side, si, d_base, d_off, K, wa.s_ra,
                       wa.s_rb, wa.s_gpos, o_ra, o_rb, o_gpos);
    WorkerArgs w2 = wa;
    w2.num_segs = d_nseg;
    w2.seg_off = seg_off;
    w2.seg_cnt = seg_cnt;
    w2.s_ra = o_ra;
    w2.s_rb = o_rb;
    w2.s_gpos = o_gpos;
    w2.T.side = 1;
    w2.work_cap = n_side / (w2.small_seg + 1) + 1;
    w2.work_list = (size_t)kWaveClasses * w2.work_cap <= ns ? seg_key : nullptr;
    w2.work_ctl = w2.work_list ? TakeZeroed(S, 2 * kWaveClasses) : nullptr;
    // A large side cluster is a component like any other: one level down.
    SpineInput nested;
    w2.wave_max = 0x7fffffff;
    if (pool.ok && depth < 8 && n_side >= S.spine_min * S.spine_nested_factor) {
      const long long room = (long long)((S.spine_pool_ints - pool_used - pool.used) / 16);
      // (a level costs about a millisecond of launches: only for what the wave worker needs longer for)
      w2.wave_max = SelectLargeSegments(n_side, d_nseg, seg_off, seg_cnt, S.spine_min * S.spine_nested_factor,
                                        room < S.spine_max_edges ? room : S.spine_max_edges, S, s, &nested,
                                        nullptr);
    }
    if (nested.segs.empty()) {
      run_workers(w2, n_side, s);
    } else {
      // the ordinary side clusters on the third stream, beside the level below (disjoint regions)
      const int ef = NextEvent(S), ej = NextEvent(S);
      const bool fork = S.aux2_stream && ef >= 0 && ej >= 0;
      if (fork) {
        VSG_HIP(hipEventRecord((*S.ev_pool)[ef], s));
        VSG_HIP(hipStreamWaitEvent(S.aux2_stream, (*S.ev_pool)[ef], 0));
        run_workers(w2, n_side, S.aux2_stream);
        VSG_HIP(hipEventRecord((*S.ev_pool)[ej], S.aux2_stream));
      } else {
        run_workers(w2, n_side, s);
      }
      if (!RunSpineComponents(nested, w2, S, s, run_workers, pool_used + pool.used, depth + 1)) {
        WorkerArgs w3 = w2;   // no room: the wave worker replays them
        w3.wave_min = w2.wave_max - 1;
        w3.wave_max = 0x7fffffff;
        w3.work_list = nullptr;
        run_workers(w3, n_side, s);
      }
      if (fork) VSG_HIP(hipStreamWaitEvent(s, (*S.ev_pool)[ej], 0));
    }
  }
  Mark(5);
  const int es0 = NextEvent(S);
  if (es0 >= 0) VSG_HIP(hipEventRecord((*S.ev_pool)[es0], s));
  // The streamed chain first (the plain steps up to the first step that is not, or whose test
  // fails), one batch of k_spine from there, a second streamed pass; k_spine takes the rest.
  int32_t* start_pos = nullptr;
  const auto LaunchSpine = [&](int max_steps) {
    hipLaunchKernelGGL(k_spine, dim3(K), dim3(64 * (1 + kSpineReaders)), 0, s, K, comp_spine, root_vertex, sp_child,
                       sp_is_a, wa.nodes, wa.T, wa.optimistic, wa.violation, wa.stats, start_pos, max_steps,
                       wa.s_gpos, wa.hub_excl);
  };
  if (S.spine_fast > 0 && mt >= S.spine_fast_min) {
    SpineFastArrays F;
    F.ck_stride = (mt >> 4) + 2 * K + 4;
    F.p = pool.take(mt);
    F.ds = reinterpret_cast<float4*>(pool.take(4 * (size_t)mt));
    F.meta = pool.take(mt);
    F.sizes = pool.take((size_t)mt + 1);
    F.pre = pool.take((size_t)mt + 1);
    F.ck = reinterpret_cast<float*>(pool.take(3 * (size_t)F.ck_stride));
    F.stop = pool.take(K);
    F.fail = pool.take(K);
    F.start_pos = pool.take(K);
    if (pool.ok) {
      start_pos = F.start_pos;
      for (int pass = 0; pass < S.spine_fast; ++pass) {
        if (pass > 0) LaunchSpine(64);
        hipLaunchKernelGGL(k_spine_fast_init, dim3((K + 63) / 64), dim3(64), 0, s, K, comp_spine, F, pass == 0 ? 1 : 0);
        hipLaunchKernelGGL(k_spine_prep, dim3(Blocks((size_t)mt + 1)), dim3(256), 0, s, mt, K, comp_spine,
                           root_vertex, sp_child, wa.nodes, wa.T, F);
        ExclusiveSum(S.scan, F.sizes, F.pre, mt + 1, s);
        hipLaunchKernelGGL(k_spine_chain, dim3(K), dim3(64 * (1 + kChProducers)), 0, s, K, comp_spine, root_vertex,
                           wa.nodes, F);
        hipLaunchKernelGGL(k_spine_verify, dim3(Blocks((size_t)F.ck_stride)), dim3(256), 0, s, K, comp_spine,
                           root_vertex, wa.nodes, wa.T, F);
        hipLaunchKernelGGL(k_spine_commit, dim3(Blocks(((size_t)mt + kCommitPer - 1) / kCommitPer)), dim3(256), 0, s, mt, K, comp_spine, root_vertex,
                           wa.nodes, F, wa.stats);
        hipLaunchKernelGGL(k_spine_finish, dim3((K + 63) / 64), dim3(64), 0, s, K, comp_spine, root_vertex,
                           wa.nodes, wa.T, F, wa.stats);
        if (S.spine_debug) {
          VSG_HIP(hipStreamSynchronize(s));
          std::vector<int32_t> cs(K + 1), sp(K), st(K), fl(K);
          VSG_HIP(hipMemcpy(cs.data(), comp_spine, (K + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
          VSG_HIP(hipMemcpy(sp.data(), F.start_pos, K * sizeof(int32_t), hipMemcpyDeviceToHost));
          VSG_HIP(hipMemcpy(st.data(), F.stop, K * sizeof(int32_t), hipMemcpyDeviceToHost));
          VSG_HIP(hipMemcpy(fl.data(), F.fail, K * sizeof(int32_t), hipMemcpyDeviceToHost));
          long long total = 0, fast = 0;
          int longest = 0, stopped = 0, failed = 0;
          for (int k = 0; k < K; ++k) {
            const int n = cs[k + 1] - cs[k];
            total += n;
            fast += sp[k];
            longest = std::max(longest, n);
            stopped += st[k] < n;
            failed += fl[k] < n;
          }
          std::fprintf(stderr, "[vsg]   streamed chain, pass %d: %lld of %lld spine steps done (longest spine %d; %d of %d "
                       "components stopped at a step that is not plain, %d at a failed test or a larger side "
                       "cluster)\n", pass, fast, total, longest, stopped, K, failed);
        }
      }
    }
  }
  LaunchSpine(0x7fffffff);
  const int es1 = NextEvent(S);
  if (es1 >= 0) {
    VSG_HIP(hipEventRecord((*S.ev_pool)[es1], s));
    if (S.ev_spine) S.ev_spine->emplace_back(es0, es1);
  }
  VSG_HIP(hipGetLastError());
  Mark(6);
  if (S.spine_debug) {
    std::fprintf(stderr, "[vsg] spine[%d]: %d components, %d edges (largest %d), %d tree edges, %d side edges, %d spine edges, "
                 "%d boruvka rounds | ms: forest %.2f rooting %.2f paths %.2f classify %.2f side %.2f spine %.2f\n",
                 depth, K, mE, in.segs[0].cnt, mt, n_side, dbg_spine_edges, dbg_rounds, tph[1] - tph[0], tph[2] - tph[1],
                 tph[3] - tph[2], tph[4] - tph[3], tph[5] - tph[4], tph[6] - tph[5]);
  }
  return true;
}

}  // namespace vsg
