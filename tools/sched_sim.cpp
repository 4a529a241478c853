// sched_sim.cpp -- offline model of the wave merge worker's round scheduler (analysis aid).
//
//   VSO_TRACE=/tmp/t.bin VSO_TRACE_BUCKETS=2 python <any oracle run>     (oracle/vs_oracle.cpp)
//   g++ -O2 -std=c++17 -o /tmp/sched_sim tools/sched_sim.cpp && /tmp/sched_sim /tmp/t.bin [batch sizes...]
//
// The trace holds, for the first buckets of every SegmentGraph call, every edge that is not
// internal when its bucket starts, with the sequential outcome and the state of its two regions.
// Any valid schedule reproduces those outcomes, so the model only replays the *scheduling* rules
// of k_merge_wave: components of the bucket's active edges, batches of B live
// edges, deterministic reservations, the transitive chain on the batch's hot region -- and counts
// rounds and commits.  B = 64 is the kernel; larger B show how much a bigger window of staged
// edges (an out-of-order window) would shorten the dependency chains per edge.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <vector>

struct TraceRecord {
  int32_t bucket, list;
  int32_t s1, s2;
  int32_t sz1, sz2;
  int32_t cons1, cons2;
  uint8_t flags1, flags2;
  uint8_t outcome;   // 0 internal, 1 regular, 2 small, 3 forced merge, 4 kept, 5 kept after failed test
  uint8_t first_wins;
  uint8_t inert;
  uint8_t pad[3];    // pad[0]: the descriptor test was made and failed
};
static_assert(sizeof(TraceRecord) == 40, "record layout");

static long g_cut_why[7] = {0, 0, 0, 0, 0, 0, 0};   // partner not owned, not plain, not smaller, hot w/o descriptor, two hot ends, other, no cut
static int kAnySize = 0;
static int kPhases = 1;   // VS_TWO_PHASE=1: generic edges first, then the hot edges, in one round
static int kHotMin = 2;   // VS_HOT_MIN: endpoints a region needs to become hot in strategy 2

struct UF {
  std::vector<int> p;
  int find(int x) {
    while (p[x] != x) {
      p[x] = p[p[x]];
      x = p[x];
    }
    return x;
  }
};

struct Stats {
  long batches = 0, rounds = 0, generic = 0, chain = 0, retired_internal = 0, edges = 0, live = 0;
  long comps = 0, max_comp = 0, critical_rounds = 0;   // rounds of the largest component
};

// One component: `ed` = its active edges in order (indices into rec).  Region ids are remapped to
// 0..R-1.
// Strategy 1: chains on every region (upper bound of what a multi-group chain with transitive
// closure can commit per round): lanes are visited in order; a region is *open* while every earlier
// pending lane that touches it commits in this round; a lane commits when both its (effective)
// regions are open and it is either a chain step (one end is a plain, smaller region nobody has
// modified in this round, the other end may be a running chain) or a generic edge on two
// regions nobody has modified in this round.
static void SimComponentMulti(const std::vector<TraceRecord>& rec, const std::vector<int>& ed,
                              std::unordered_map<int, int>& remap_scratch, int B, Stats* st,
                              long* comp_rounds) {
  remap_scratch.clear();
  auto id = [&](int r) {
    auto it = remap_scratch.find(r);
    if (it != remap_scratch.end()) return it->second;
    const int v = (int)remap_scratch.size();
    remap_scratch.emplace(r, v);
    return v;
  };
  std::vector<int> ea(ed.size()), eb(ed.size());
  for (size_t i = 0; i < ed.size(); ++i) {
    ea[i] = id(rec[ed[i]].s1);
    eb[i] = id(rec[ed[i]].s2);
  }
  const int R = (int)remap_scratch.size();
  UF uf;
  uf.p.resize(R);
  std::iota(uf.p.begin(), uf.p.end(), 0);
  std::vector<char> closed(R, 0), modified(R, 0);
  std::vector<int> touched;
  size_t next = 0;
  long rounds_here = 0;
  std::vector<int> lanes;
  std::vector<char> pending, failed;
  while (next < ed.size()) {
    lanes.clear();
    while (next < ed.size() && (int)lanes.size() < B) {
      if (uf.find(ea[next]) != uf.find(eb[next])) lanes.push_back((int)next);
      ++next;
    }
    if (lanes.empty()) break;
    ++st->batches;
    st->live += (long)lanes.size();
    const int n = (int)lanes.size();
    pending.assign(n, 1);
    failed.assign(n, 0);
    for (;;) {
      int npend = 0;
      for (int l = 0; l < n; ++l) npend += pending[l];
      if (!npend) break;
      ++st->rounds;
      ++rounds_here;
      touched.clear();
      int commits = 0;
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        const int x = uf.find(ea[lanes[l]]), y = uf.find(eb[lanes[l]]);   // effective regions
        auto touch = [&](int r) { touched.push_back(r); };
        if (x == y) {
          if (!closed[x]) {
            pending[l] = 0;
            ++st->retired_internal;
            ++commits;
          }
          continue;
        }
        const TraceRecord& r = rec[ed[lanes[l]]];
        bool ok = !closed[x] && !closed[y];
        if (ok) {
          const bool is_merge = r.outcome >= 1 && r.outcome <= 3;
          // chain step: the smaller end is plain and unmodified; generic: both unmodified
          const bool x_small = (r.first_wins ? false : true);   // region 1 loses when first_wins == 0
          const int small_r = is_merge ? (x_small ? x : y) : -1;
          const int pfl = x_small ? r.flags1 : r.flags2, pcons = x_small ? r.cons1 : r.cons2;
          const int hcons = x_small ? r.cons2 : r.cons1, hfl = x_small ? r.flags2 : r.flags1;
          const bool chain_step = is_merge && !failed[l] && !r.pad[0] && !modified[small_r] && pfl == 0 &&
                                  (pcons < 0 || pcons == hcons) && !(hfl & 2) &&
                                  (x_small ? r.sz1 < r.sz2 : r.sz2 < r.sz1);
          const bool generic = !modified[x] && !modified[y];
          ok = chain_step || generic;
          if (ok) {
            if (chain_step) ++st->chain; else ++st->generic;
          }
        }
        if (!ok) {
          closed[x] = closed[y] = 1;
          touch(x);
          touch(y);
          continue;
        }
        pending[l] = 0;
        ++commits;
        if (r.outcome >= 1 && r.outcome <= 3) {
          const int w = r.first_wins ? x : y, lo = r.first_wins ? y : x;
          uf.p[lo] = w;
          modified[w] = 1;
          touch(w);
        } else if (r.outcome == 5 || r.pad[0]) {
          modified[x] = modified[y] = 1;   // finalized by the failed test
          touch(x);
          touch(y);
        }
      }
      for (int t : touched) closed[t] = modified[t] = 0;
      if (!commits) {
        std::fprintf(stderr, "multi model stuck\n");
        std::exit(1);
      }
    }
  }
  *comp_rounds = rounds_here;
}

// Strategy 2: the kernel's transitive chain on the K most contended regions of a batch (K hot
// regions, each with its own closure and cut; an edge with ends in two different hot groups waits
// and ends both chains), everything else by reservations as in the kernel.
static void SimComponentKHot(const std::vector<TraceRecord>& rec, const std::vector<int>& ed,
                             std::unordered_map<int, int>& remap_scratch, int B, int K, Stats* st,
                             long* comp_rounds) {
  remap_scratch.clear();
  auto id = [&](int r) {
    auto it = remap_scratch.find(r);
    if (it != remap_scratch.end()) return it->second;
    const int v = (int)remap_scratch.size();
    remap_scratch.emplace(r, v);
    return v;
  };
  std::vector<int> ea(ed.size()), eb(ed.size());
  for (size_t i = 0; i < ed.size(); ++i) {
    ea[i] = id(rec[ed[i]].s1);
    eb[i] = id(rec[ed[i]].s2);
  }
  const int R = (int)remap_scratch.size();
  UF uf;
  uf.p.resize(R);
  std::iota(uf.p.begin(), uf.p.end(), 0);
  std::vector<int> owner(R), cnt(R, 0), hot_of(R, -1);
  size_t next = 0;
  long rounds_here = 0;
  std::vector<int> lanes;
  std::vector<char> pending, failed;
  while (next < ed.size()) {
    lanes.clear();
    while (next < ed.size() && (int)lanes.size() < B) {
      if (uf.find(ea[next]) != uf.find(eb[next])) lanes.push_back((int)next);
      ++next;
    }
    if (lanes.empty()) break;
    ++st->batches;
    st->live += (long)lanes.size();
    const int n = (int)lanes.size();
    pending.assign(n, 1);
    failed.assign(n, 0);
    std::vector<int> A(n), Bv(n), oa(n), ob(n), ga(n), gb(n);
    std::vector<char> owna(n), ownb(n), elig(n), both(n), em(n), cross(n);
    for (bool batch_done = false; !batch_done;) {
     int committed_in_round = 0;
     for (int phase = 0; phase < kPhases && !batch_done; ++phase) {
      int npend = 0;
      std::vector<int> touched;
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        A[l] = uf.find(ea[lanes[l]]);
        Bv[l] = uf.find(eb[lanes[l]]);
        if (A[l] == Bv[l]) {
          pending[l] = 0;
          ++st->retired_internal;
          continue;
        }
        ++npend;
        for (int r : {A[l], Bv[l]}) {
          if (cnt[r]++ == 0) touched.push_back(r);
        }
      }
      if (!npend) {
        batch_done = true;
        break;
      }
      if (phase == 0) {
        ++st->rounds;
        ++rounds_here;
      }
      // the K most contended regions of this round (>= 2 pending endpoints)
      std::vector<int> hots;
      {
        std::vector<int> order(touched);
        std::sort(order.begin(), order.end(), [&](int x, int y) {
          return cnt[x] != cnt[y] ? cnt[x] > cnt[y] : x > y;
        });
        for (int r : order) {
          if ((int)hots.size() >= K || cnt[r] < kHotMin) break;
          hots.push_back(r);
        }
      }
      if (kHotMin < 0) {
        // variant: the hot region is the larger end of the earliest pending edge (no counting)
        // (K > 1: the next hot region is the larger end of the earliest pending edge that touches
        // none of the hot regions chosen so far)
        hots.clear();
        for (int l = 0; l < n && (int)hots.size() < K; ++l) {
          if (!pending[l]) continue;
          bool touches = false;
          for (int h : hots) touches = touches || A[l] == h || Bv[l] == h;
          if (touches) continue;
          const TraceRecord& r = rec[ed[lanes[l]]];
          hots.push_back(r.sz1 >= r.sz2 ? A[l] : Bv[l]);
        }
      }
      for (int r : touched) cnt[r] = 0;
      for (size_t k = 0; k < hots.size(); ++k) hot_of[hots[k]] = (int)k;
      // reservations (hot regions are not reserved)
      touched.clear();
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        for (int r : {A[l], Bv[l]}) {
          if (hot_of[r] >= 0) continue;
          if (cnt[r]++ == 0) {
            owner[r] = l;
            touched.push_back(r);
          }
        }
      }
      for (int r : touched) cnt[r] = 0;
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        owna[l] = hot_of[A[l]] < 0 && owner[A[l]] == l;
        ownb[l] = hot_of[Bv[l]] < 0 && owner[Bv[l]] == l;
        oa[l] = hot_of[A[l]] < 0 ? owner[A[l]] : -1;
        ob[l] = hot_of[Bv[l]] < 0 ? owner[Bv[l]] : -1;
      }
      // closure: effective group of every end (-1 none), absorbing lanes em
      std::fill(em.begin(), em.end(), 0);
      for (bool changed = true; changed;) {
        changed = false;
        for (int l = 0; l < n; ++l) {
          if (!pending[l]) continue;
          int g1 = hot_of[A[l]], g2 = hot_of[Bv[l]];
          if (g1 < 0 && !owna[l] && oa[l] >= 0 && em[oa[l]]) g1 = ga[oa[l]] >= 0 ? ga[oa[l]] : gb[oa[l]];
          if (g2 < 0 && !ownb[l] && ob[l] >= 0 && em[ob[l]]) g2 = ga[ob[l]] >= 0 ? ga[ob[l]] : gb[ob[l]];
          // (an absorbing lane has exactly one effective end: its group)
          ga[l] = g1;
          gb[l] = g2;
          both[l] = g1 >= 0 && g1 == g2;
          cross[l] = g1 >= 0 && g2 >= 0 && g1 != g2;
          elig[l] = 0;
          if ((g1 >= 0) == (g2 >= 0)) continue;
          const TraceRecord& r = rec[ed[lanes[l]]];
          const bool part_is_2 = g1 >= 0;
          const int psz = part_is_2 ? r.sz2 : r.sz1, hsz = part_is_2 ? r.sz1 : r.sz2;
          const int pcons = part_is_2 ? r.cons2 : r.cons1, hcons = part_is_2 ? r.cons1 : r.cons2;
          const int pfl = part_is_2 ? r.flags2 : r.flags1, hfl = part_is_2 ? r.flags1 : r.flags2;
          const bool own_p = part_is_2 ? ownb[l] : owna[l];
          // VS_ANY_SIZE: a plain hot region may also be absorbed by a larger plain partner (the
          // chain then continues on the partner's representative)
          const bool size_ok = psz < hsz || (kAnySize && hfl == 0 && hcons < 0 && pcons < 0);
          const bool ok = own_p && !failed[l] && pfl == 0 && (pcons < 0 || pcons == hcons) &&
                          size_ok && !(hfl & 2);
          if (!ok) continue;
          elig[l] = 1;
          const bool fin = hfl & 1;
          const bool m = pcons >= 0 || !fin || r.outcome == 2;
          if (m && !em[l]) {
            em[l] = 1;
            changed = true;
          }
        }
      }
      // per group: cut at the first lane of the group that is neither chain lane nor internal
      std::vector<int> cut(hots.size(), n);
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        const bool ok = elig[l] || both[l];
        for (int g : {ga[l], gb[l]}) {
          if (g >= 0 && !ok && l < cut[g]) cut[g] = l;
        }
      }
      if (!hots.empty() && cut[0] < n) {   // why does the chain of the (first) hot region end?
        const int l = cut[0];
        const TraceRecord& r = rec[ed[lanes[l]]];
        const bool part_is_2 = ga[l] >= 0;
        const int psz = part_is_2 ? r.sz2 : r.sz1, hsz = part_is_2 ? r.sz1 : r.sz2;
        const int pcons = part_is_2 ? r.cons2 : r.cons1, hcons = part_is_2 ? r.cons1 : r.cons2;
        const int pfl = part_is_2 ? r.flags2 : r.flags1, hfl = part_is_2 ? r.flags1 : r.flags2;
        const bool own_p = part_is_2 ? ownb[l] : owna[l];
        int why = 5;
        if (cross[l]) why = 4;
        else if (!own_p) why = 0;
        else if (pfl != 0 || !(pcons < 0 || pcons == hcons)) why = 1;
        else if (!(psz < hsz)) why = 2;
        else if (hfl & 2) why = 3;
        ++g_cut_why[why];
      } else if (!hots.empty()) {
        ++g_cut_why[6];
      }
      for (int l = 0; l < n; ++l) {   // failed tests cut their group
        if (!pending[l] || !elig[l]) continue;
        const int g = ga[l] >= 0 ? ga[l] : gb[l];
        if (l >= cut[g]) continue;
        const TraceRecord& r = rec[ed[lanes[l]]];
        const bool is_merge = r.outcome >= 1 && r.outcome <= 3;
        if (r.pad[0] || (em[l] && !is_merge)) {
          failed[l] = 1;
          cut[g] = l;
        }
      }
      std::vector<int> commit;
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        const bool hl = ga[l] >= 0 || gb[l] >= 0;
        // two-phase rounds: generic (non-hot) edges first, then the hot edges on the updated state
        if (kPhases == 2 && (phase == 0) == hl) continue;
        if (hl) {
          if (elig[l] || both[l]) {
            const int g = ga[l] >= 0 ? ga[l] : gb[l];
            // an absorbed end is only valid if its absorber commits: absorbers precede and are in
            // the same group, so "below the cut" covers it
            if (l < cut[g]) {
              commit.push_back(l);
              if (elig[l]) ++st->chain; else ++st->retired_internal;
            }
          } else {
            // a lane that touches hot regions literally, is the first pending lane of each of
            // those groups and owns its other end runs alone (generic code)
            const bool lit_a = hot_of[A[l]] >= 0, lit_b = hot_of[Bv[l]] >= 0;
            const bool eff_only = (ga[l] >= 0 && !lit_a) || (gb[l] >= 0 && !lit_b);
            const bool own = (lit_a || owna[l]) && (lit_b || ownb[l]);
            if ((lit_a || lit_b) && !eff_only && own) {
              bool first = true;
              for (int g : {lit_a ? ga[l] : -1, lit_b ? gb[l] : -1}) {
                if (g < 0) continue;
                for (int e = 0; e < l; ++e) {
                  if (pending[e] && (ga[e] == g || gb[e] == g)) first = false;
                }
              }
              if (first) {
                commit.push_back(l);
                ++st->generic;
              }
            }
          }
        } else if (owna[l] && ownb[l]) {
          commit.push_back(l);
          ++st->generic;
        }
      }
      for (int r : hots) hot_of[r] = -1;
      committed_in_round += (int)commit.size();
      if (phase == kPhases - 1 && committed_in_round == 0) {
        std::fprintf(stderr, "k-hot model stuck (n=%d)\n", n);
        std::exit(1);
      }
      for (int l : commit) {
        pending[l] = 0;
        const TraceRecord& r = rec[ed[lanes[l]]];
        if (r.outcome >= 1 && r.outcome <= 3) {
          const int x = uf.find(ea[lanes[l]]), y = uf.find(eb[lanes[l]]);
          if (x != y) {
            if (r.first_wins) uf.p[y] = x; else uf.p[x] = y;
          }
        }
      }
     }
    }
  }
  *comp_rounds = rounds_here;
}

static void SimComponent(const std::vector<TraceRecord>& rec, const std::vector<int>& ed,
                         std::unordered_map<int, int>& remap_scratch, int B, Stats* st,
                         long* comp_rounds) {
  remap_scratch.clear();
  auto id = [&](int r) {
    auto it = remap_scratch.find(r);
    if (it != remap_scratch.end()) return it->second;
    const int v = (int)remap_scratch.size();
    remap_scratch.emplace(r, v);
    return v;
  };
  std::vector<int> ea(ed.size()), eb(ed.size());
  for (size_t i = 0; i < ed.size(); ++i) {
    ea[i] = id(rec[ed[i]].s1);
    eb[i] = id(rec[ed[i]].s2);
  }
  const int R = (int)remap_scratch.size();
  UF uf;
  uf.p.resize(R);
  std::iota(uf.p.begin(), uf.p.end(), 0);
  std::vector<int> owner(R), cnt(R);
  size_t next = 0;
  long rounds_here = 0;
  std::vector<int> lanes;       // edge positions (into ed) of the batch, in order
  std::vector<char> pending, failed;
  while (next < ed.size()) {
    // ---- stage B live edges
    lanes.clear();
    while (next < ed.size() && (int)lanes.size() < B) {
      if (uf.find(ea[next]) != uf.find(eb[next])) lanes.push_back((int)next);
      ++next;
    }
    if (lanes.empty()) break;
    ++st->batches;
    st->live += (long)lanes.size();
    const int n = (int)lanes.size();
    pending.assign(n, 1);
    failed.assign(n, 0);
    // hot region: most pending endpoints (>= 3)
    int hot = -1;
    {
      std::vector<int> touched;
      for (int l = 0; l < n; ++l) {
        for (int r : {uf.find(ea[lanes[l]]), uf.find(eb[lanes[l]])}) {
          if (cnt[r]++ == 0) touched.push_back(r);
        }
      }
      int best = 0;
      for (int r : touched) {
        if (cnt[r] > best || (cnt[r] == best && r > hot)) {
          best = cnt[r];
          hot = r;
        }
      }
      if (best < 3) hot = -1;
      for (int r : touched) cnt[r] = 0;
    }
    std::vector<int> A(n), Bv(n), oa(n), ob(n);
    std::vector<char> owna(n), ownb(n), efa(n), efb(n), elig(n), both(n), merging(n);
    for (;;) {
      int npend = 0;
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        A[l] = uf.find(ea[lanes[l]]);
        Bv[l] = uf.find(eb[lanes[l]]);
        if (A[l] == Bv[l]) {
          pending[l] = 0;
          ++st->retired_internal;
          continue;
        }
        ++npend;
      }
      if (!npend) break;
      if (hot >= 0) hot = uf.find(hot);
      ++st->rounds;
      ++rounds_here;
      // reservations
      std::vector<int> touched;
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        for (int r : {A[l], Bv[l]}) {
          if (r == hot) continue;
          if (cnt[r]++ == 0) {
            owner[r] = l;
            touched.push_back(r);
          }
        }
      }
      for (int r : touched) cnt[r] = 0;
      int first_hot = -1;
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        owna[l] = A[l] != hot && owner[A[l]] == l;
        ownb[l] = Bv[l] != hot && owner[Bv[l]] == l;
        oa[l] = A[l] != hot ? owner[A[l]] : -1;
        ob[l] = Bv[l] != hot ? owner[Bv[l]] : -1;
        if (first_hot < 0 && (A[l] == hot || Bv[l] == hot)) first_hot = l;
      }
      // chain classification (transitive closure on the absorbing lanes)
      std::fill(elig.begin(), elig.end(), 0);
      std::fill(both.begin(), both.end(), 0);
      std::fill(efa.begin(), efa.end(), 0);
      std::fill(efb.begin(), efb.end(), 0);
      std::vector<char> em(n, 0);
      bool chain_possible = first_hot >= 0 && (owna[first_hot] || ownb[first_hot]);
      if (chain_possible) {
        for (bool changed = true; changed;) {
          changed = false;
          for (int l = 0; l < n; ++l) {
            if (!pending[l]) continue;
            const bool e1 = A[l] == hot || (!owna[l] && oa[l] >= 0 && em[oa[l]]);
            const bool e2 = Bv[l] == hot || (!ownb[l] && ob[l] >= 0 && em[ob[l]]);
            efa[l] = e1;
            efb[l] = e2;
            both[l] = e1 && e2;
            elig[l] = 0;
            if (e1 == e2) continue;
            const TraceRecord& r = rec[ed[lanes[l]]];
            // partner = the end that is not effectively hot
            const bool part_is_2 = e1;
            const int psz = part_is_2 ? r.sz2 : r.sz1, hsz = part_is_2 ? r.sz1 : r.sz2;
            const int pcons = part_is_2 ? r.cons2 : r.cons1, hcons = part_is_2 ? r.cons1 : r.cons2;
            const int pfl = part_is_2 ? r.flags2 : r.flags1, hfl = part_is_2 ? r.flags1 : r.flags2;
            const bool own_p = part_is_2 ? ownb[l] : owna[l];
            const bool ok = own_p && !failed[l] && pfl == 0 && (pcons < 0 || pcons == hcons) &&
                            psz < hsz && !(hfl & 2);
            if (!ok) continue;
            elig[l] = 1;
            const bool fin = hfl & 1;
            const bool m = pcons >= 0 || !fin || r.outcome == 2;
            if (m && !em[l]) {
              em[l] = 1;
              changed = true;
            }
          }
        }
      }
      // cut of the chain: first hot lane that is neither chain lane nor internal
      int cut = n;
      if (chain_possible) {
        for (int l = 0; l < n; ++l) {
          if (!pending[l]) continue;
          const bool hl = efa[l] || efb[l];
          if (hl && !elig[l] && !both[l]) {
            cut = l;
            break;
          }
        }
        // first failing test inside the chain
        for (int l = 0; l < cut; ++l) {
          if (!pending[l] || !elig[l]) continue;
          const TraceRecord& r = rec[ed[lanes[l]]];
          const bool is_merge = r.outcome >= 1 && r.outcome <= 3;
          const bool fails = r.pad[0] || (em[l] && !is_merge);
          if (fails) {
            failed[l] = 1;
            cut = l;
            break;
          }
        }
      }
      // commits
      std::vector<int> commit;
      for (int l = 0; l < n; ++l) {
        if (!pending[l]) continue;
        const bool hl = chain_possible ? (efa[l] || efb[l]) : (A[l] == hot || Bv[l] == hot);
        const bool own = (A[l] == hot || owna[l]) && (Bv[l] == hot || ownb[l]);
        const bool solo = hl && l == first_hot && own && !(chain_possible && (elig[l] || both[l]));
        if (chain_possible && l < cut && (elig[l] || both[l])) {
          commit.push_back(l);
          if (elig[l]) ++st->chain; else ++st->retired_internal;
        } else if (own && (!hl || solo)) {
          commit.push_back(l);
          ++st->generic;
        }
      }
      if (commit.empty()) {
        std::fprintf(stderr, "model stuck (n=%d hot=%d)\n", n, hot);
        std::exit(1);
      }
      for (int l : commit) {
        pending[l] = 0;
        const TraceRecord& r = rec[ed[lanes[l]]];
        if (r.outcome >= 1 && r.outcome <= 3) {
          const int x = uf.find(ea[lanes[l]]), y = uf.find(eb[lanes[l]]);
          if (x != y) {
            if (r.first_wins) uf.p[y] = x; else uf.p[x] = y;
          }
        }
      }
    }
  }
  *comp_rounds = rounds_here;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: sched_sim trace.bin [batch sizes]\n");
    return 2;
  }
  std::vector<int> sizes;
  for (int i = 2; i < argc; ++i) sizes.push_back(std::atoi(argv[i]));
  if (sizes.empty()) sizes = {64, 128, 256, 512};
  if (std::getenv("VS_TWO_PHASE")) kPhases = 2;
  if (std::getenv("VS_ANY_SIZE")) kAnySize = 1;
  if (std::getenv("VS_HOT_MIN")) kHotMin = std::atoi(std::getenv("VS_HOT_MIN"));
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 1;
  std::vector<TraceRecord> rec;
  std::unordered_map<int, int> remap;
  int call = 0, prev_bucket = -1;
  auto flush_stage = [&]() {
    if (rec.empty()) return;
    // components of the active edges
    std::unordered_map<int, int> dense;
    std::vector<int> a(rec.size()), b(rec.size());
    long active = 0;
    for (size_t i = 0; i < rec.size(); ++i) {
      if (rec[i].inert) continue;
      ++active;
      for (int k = 0; k < 2; ++k) {
        const int r = k ? rec[i].s2 : rec[i].s1;
        auto it = dense.find(r);
        const int v = it == dense.end() ? (int)dense.size() : it->second;
        if (it == dense.end()) dense.emplace(r, v);
        (k ? b[i] : a[i]) = v;
      }
    }
    UF cc;
    cc.p.resize(dense.size());
    std::iota(cc.p.begin(), cc.p.end(), 0);
    for (size_t i = 0; i < rec.size(); ++i) {
      if (rec[i].inert) continue;
      const int x = cc.find(a[i]), y = cc.find(b[i]);
      if (x != y) cc.p[std::max(x, y)] = std::min(x, y);
    }
    std::unordered_map<int, std::vector<int>> comps;
    for (size_t i = 0; i < rec.size(); ++i) {
      if (rec[i].inert) continue;
      comps[cc.find(a[i])].push_back((int)i);
    }
    std::printf("call %d bucket %d: %zu traced edges, %ld active, %zu components\n", call,
                rec[0].bucket, rec.size(), active, comps.size());
    for (int Bs : sizes) {
      // negative size: strategy 1 (chains on every region); size + 1000*K: strategy 2 (K hot regions)
      const bool multi = Bs < 0;
      const int khot = Bs >= 1000 ? Bs / 1000 : 0;
      const int B = multi ? -Bs : Bs % 1000;
      Stats st;
      for (auto& kv : comps) {
        if (kv.second.size() <= 24) continue;
        ++st.comps;
        st.edges += (long)kv.second.size();
        long cr = 0;
        if (multi) SimComponentMulti(rec, kv.second, remap, B, &st, &cr);
        else if (khot) SimComponentKHot(rec, kv.second, remap, B, khot, &st, &cr);
        else SimComponent(rec, kv.second, remap, B, &st, &cr);
        if ((long)kv.second.size() > st.max_comp) {
          st.max_comp = (long)kv.second.size();
          st.critical_rounds = cr;
        }
      }
      std::printf("  %s B=%4d: wave components %ld (edges %ld, largest %ld) batches %ld live %ld rounds %ld "
                  "(%.2f per batch, %.2f per 64 live edges) generic %ld chain %ld | largest component: %ld rounds\n",
                  multi ? "multi " : (khot ? "k-hot " : "kernel"), khot ? khot * 1000 + B : B, st.comps, st.edges, st.max_comp, st.batches, st.live, st.rounds,
                  st.batches ? (double)st.rounds / st.batches : 0.0,
                  st.live ? (double)st.rounds * 64.0 / st.live : 0.0, st.generic, st.chain,
                  st.critical_rounds);
    }
    rec.clear();
  };
  TraceRecord r;
  while (std::fread(&r, sizeof(r), 1, f) == 1) {
    if (r.bucket != prev_bucket) {
      flush_stage();
      if (r.bucket < prev_bucket) ++call;
      prev_bucket = r.bucket;
    }
    rec.push_back(r);
  }
  flush_stage();
  std::printf("chain ends: partner not owned %ld, not plain %ld, not smaller %ld, hot without descriptor %ld, two hot ends %ld, other %ld, not cut %ld\n",
              g_cut_why[0], g_cut_why[1], g_cut_why[2], g_cut_why[3], g_cut_why[4], g_cut_why[5], g_cut_why[6]);
  std::fclose(f);
  return 0;
}
