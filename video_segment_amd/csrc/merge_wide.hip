// merge_wide.hip -- the worker of the LARGE ordinary components of the ordered merge: W wavefronts
// replay one component, 64 * W edges per batch, in lock-step rounds.
//
// Why: the wave worker (merge_wave.hip) is bound by instruction issue, not by memory -- a round is
// about 1 500 instructions of one wavefront (5-8 K cycles at one instruction per 4-5 cycles) and a
// batch of 64 edges takes 5-7 rounds, whatever the component looks like: the rounds a batch needs
// follow the LOCAL degree of the regions (an edge waits for the earlier edges on its two regions),
// not the width of the batch.  A percolating component of a noisy input (one component of 40 K
// replayed edges per stage, 300 stages per chunk) therefore kept ONE of the 1 024 SIMDs of the chip
// busy for 13 ms per stage.  Here the batch is 64 * W consecutive live edges and the W wavefronts of
// the workgroup -- one per SIMD of the CU -- run the SAME round on their own 64 lanes at the same
// time: reservations, ownership and the kept-lane rule work on one LDS table with GLOBAL lane
// numbers (deterministic reservations are exact for any number of lanes: a lane that holds the
// minimum on both its regions is the earliest pending edge on both, segmentation_graph.h:374-440
// replayed in order), so the rounds of a batch stay about as many while each commits W times the
// lanes.  What stays inside one wavefront is the chain on the hot region (the larger end of the
// earliest pending edge overall): it is a prefix of the hot lanes in sequence order, and the first
// wavefront that still has pending lanes holds that prefix -- the other wavefronts' hot lanes wait
// for their turn, their generic and kept lanes do not.
//
// No reader wavefront: the W wavefronts stage their own candidates (every global round trip --
// edge records, root searches, region states -- is shared by 64 * W edges instead of 64), and all
// of them meet in plain workgroup barriers.
#include "merge_common.h"

namespace vsg {

namespace {

template <int W>
struct WideTable {
  static constexpr int kSlots = 256 * W;   // >= 2 * 128 W distinct regions of a batch at load factor 1/2
  int32_t key[kSlots];     // region id, -1: empty
  int32_t link[kSlots];    // in-batch union-find over slots
  uint32_t res[kSlots];    // reservation: (kRoundTop - round) << kLaneBits | global lane, smaller wins
  uint32_t res2[kSlots];   // earliest lane a kept lane must not pass
  uint32_t res3[kSlots];   // earliest blocker (merge_wave.hip, ReplayRounds)
  float4 ds[kSlots];
  int32_t cons[kSlots];
  int32_t flags[kSlots];   // region flags | kTabDirty
};

// What the wavefronts tell each other (written by lane 0 of a wavefront, read after a barrier).
template <int W>
struct WideShared {
  unsigned long long pend[W];   // pending lanes of the wavefront
  int first_sa[W], first_sb[W]; // slots of its earliest pending lane
  int kept_kind[W];             // it has a lane with a finalized region or two different constraints
  int not_noop[W];              // it has a pending lane that is not certainly kept
  int has_noop[W];              // it has a certainly kept lane
  unsigned drop_stamp;          // iteration stamp of the kept-lane fixpoint
  int cnt[W];                   // staging: live candidates of the wavefront
  int next_seg;
};

constexpr int kLaneBits = 9;            // global lane numbers up to 512
constexpr unsigned kRoundTop = 0x3fffffu;

template <int W>
__device__ __forceinline__ int WideInsert(WideTable<W>& t, int r, bool& inserted) {
  constexpr int kBits = (W == 8) ? 11 : ((W == 4) ? 10 : ((W == 2) ? 9 : 8));
  static_assert((1 << kBits) == WideTable<W>::kSlots, "table size");
  unsigned h = ((unsigned)r * 2654435761u) >> (32 - kBits);
  for (;;) {
    const int old = atomicCAS(&t.key[h], -1, r);
    if (old == -1) { inserted = true; return (int)h; }
    if (old == r) { inserted = false; return (int)h; }
    h = (h + 1) & (WideTable<W>::kSlots - 1);
  }
}

template <int W>
__device__ __forceinline__ RState WideLoad(const WideTable<W>& t, int s) {
  const float4 ds = t.ds[s];
  RState r;
  r.d0 = ds.x;
  r.d1 = ds.y;
  r.d2 = ds.z;
  r.sz = __float_as_int(ds.w);
  r.cons = t.cons[s];
  r.flags = t.flags[s] & 0xff;
  return r;
}

template <int W>
__device__ __forceinline__ void WideStore(WideTable<W>& t, int s, const RState& r, int dirty) {
  t.ds[s] = make_float4(r.d0, r.d1, r.d2, __int_as_float(r.sz));
  t.cons[s] = r.cons;
  t.flags[s] = r.flags | dirty;
}

// (see CommitLoser, merge_wave.hip)
template <int W>
__device__ __forceinline__ void WideCommitLoser(WideTable<W>& t, const NodeArrays& nodes, int ls, int ws) {
  t.link[ls] = ws;
  const int lid = t.key[ls];
  nodes.parent[lid] = t.key[ws];
  if (t.flags[ls] & kTabDirty) nodes.cons[lid] = t.cons[ls];
}

__device__ __forceinline__ void WaveSyncW() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

struct WideCounters {
  unsigned n_forced = 0, n_regular = 0, n_small = 0;   // per lane
  unsigned rounds = 0, batches = 0;                    // thread 0
  unsigned long long cyc_stage = 0, cyc_rounds = 0, lanes = 0, chain_lanes = 0, fix_iters = 0;   // thread 0 / lane 0
};

// Replays one batch of up to 64 * W edges: thread `tid` holds the edge whose end regions sit in the
// table slots sa / sb.  Returns whether this thread's edge is kept.  Every branch around a
// __syncthreads() depends on workgroup-uniform values only.
template <int W>
__device__ __forceinline__ bool ReplayRoundsWide(WideTable<W>& tab, WideShared<W>& xw, float4* chain_buf,
                                                 const NodeArrays& nodes, const StageThr& T, int optimistic,
                                                 int32_t* violation, unsigned long long* stats, int tid,
                                                 bool valid, int sa, int sb, WideCounters& C) {
  const int lane = tid & 63, w = tid >> 6;
  bool pending = valid;
  bool my_kept = false;
  bool failed = false;    // this lane's chain test failed: replay it with the generic code
  for (unsigned round = 0;; ++round) {
    // ---- current root slots of both ends, their states ---------------------------------------
    for (bool more = pending; more;) {
      const int pa = tab.link[sa], pb = tab.link[sb];
      more = (pa != sa) || (pb != sb);
      sa = pa;
      sb = pb;
    }
    if (pending && sa == sb) pending = false;   // became internal
    RState A = {}, B = {};
    if (pending) {
      A = WideLoad(tab, sa);
      B = WideLoad(tab, sb);
    }
    const unsigned long long pend_wave = __ballot(pending);
    const bool kept_kind_l = pending && (((A.flags | B.flags) & kFlagFinalized) ||
                                         (A.cons >= 0 && B.cons >= 0 && A.cons != B.cons));
    const bool kept_kind_wave = __ballot(kept_kind_l) != 0;
    {
      const int first = pend_wave ? (int)__builtin_ctzll(pend_wave) : 0;
      const int fa = ReadLaneI(sa, first), fb = ReadLaneI(sb, first);
      if (lane == 0) {
        xw.pend[w] = pend_wave;
        xw.first_sa[w] = fa;
        xw.first_sb[w] = fb;
        xw.kept_kind[w] = kept_kind_wave ? 1 : 0;
      }
    }
    __syncthreads();   // B1
    int w0 = -1;       // the first wavefront with a pending lane: it holds the earliest pending edge
    bool any_kept_kind = false;
#pragma unroll
    for (int k = W - 1; k >= 0; --k) {
      if (xw.pend[k]) w0 = k;
      any_kept_kind = any_kept_kind || xw.kept_kind[k] != 0;
    }
    if (w0 < 0) break;   // nothing pending anywhere
    if (round > 150u * W) {   // cannot happen (the earliest pending lane commits): report
      if (tid == 0) atomicAdd(&stats[22], 1ull);
      break;
    }
    if (tid == 0) ++C.rounds;
    // The round's hot region: the larger end of the earliest pending edge.
    int hot;
    {
      const int fa = xw.first_sa[w0], fb = xw.first_sb[w0];
      const int sza = __float_as_int(tab.ds[fa].w), szb = __float_as_int(tab.ds[fb].w);
      hot = (sza >= szb) ? fa : fb;
    }
    const bool a_hot = (sa == hot), b_hot = (sb == hot);
    const bool lit = a_hot || b_hot;
    const uint32_t key = ((kRoundTop - round) << kLaneBits) | (uint32_t)tid;
    if (pending) {
      if (!a_hot) atomicMin(&tab.res[sa], key);
      if (!b_hot) atomicMin(&tab.res[sb], key);
    }
    // certainly kept lanes (NoopPair, see merge_wave.hip): known before the reservations are read
    bool noop_l = false;
    RState Hs0 = {};
    if (any_kept_kind) {
      Hs0 = WideLoad(tab, hot);   // uniform
      noop_l = pending && (a_hot ? NoopPair(B, Hs0, T) : (b_hot ? NoopPair(A, Hs0, T) : NoopPair(A, B, T)));
      const bool nn = __ballot(pending && !noop_l) != 0, an = __ballot(noop_l) != 0;
      if (lane == 0) {
        xw.not_noop[w] = nn ? 1 : 0;
        xw.has_noop[w] = an ? 1 : 0;
      }
    }
    __syncthreads();   // B2: reservations and the noop census are in
    uint32_t res_a = 0, res_b = 0;
    if (pending) {
      res_a = tab.res[sa];
      res_b = tab.res[sb];
    }
    bool kept_now = false;    // certainly kept, away from the hot region, commits in this round
    bool free_p = false;      // certainly kept, on the hot region, nothing open precedes it on its partner
    if (any_kept_kind) {
      bool some_not_noop = false, some_noop = false;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        some_not_noop = some_not_noop || xw.not_noop[k] != 0;
        some_noop = some_noop || xw.has_noop[k] != 0;
      }
      if (!some_not_noop) {
        // every pending lane of the batch is kept whatever happens: all of them, in one round
        if (pending) my_kept = true;
        pending = false;
        break;
      }
      if (some_noop) {
        if (pending && !noop_l) {
          if (!a_hot) {
            atomicMin(&tab.res2[sa], key);
            atomicMin(&tab.res3[sa], key);
          }
          if (!b_hot) {
            atomicMin(&tab.res2[sb], key);
            atomicMin(&tab.res3[sb], key);
          }
        } else if (noop_l && lit) {
          atomicMin(&tab.res2[a_hot ? sb : sa], key);
        }
        __syncthreads();   // B3
        bool cand = noop_l && !lit;
        for (unsigned it = 0;; ++it) {
          // a kept lane behind an earlier open lane becomes a blocker itself; the fixpoint is over
          // all wavefronts: an iteration in which any lane dropped out is followed by another one
          const unsigned stamp = (round << 12) | (it + 1u);
          const bool ok = cand && tab.res2[sa] > key && tab.res2[sb] > key;
          const bool drop = cand && !ok;
          cand = ok;
          __syncthreads();   // every lane has read res2 before the dropped ones lower it
          if (drop) {
            atomicMin(&tab.res2[sa], key);
            atomicMin(&tab.res2[sb], key);
            atomicMin(&tab.res3[sa], key);
            atomicMin(&tab.res3[sb], key);
            xw.drop_stamp = stamp;
          }
          __syncthreads();
          if (tid == 0) ++C.fix_iters;
          if (xw.drop_stamp != stamp) break;
        }
        kept_now = cand;
        free_p = noop_l && lit && tab.res3[a_hot ? sb : sa] > key;
      } else {
        noop_l = false;
      }
    }
    // own_x: this lane is the earliest pending edge on region x (the hot region is not reserved)
    const bool own_a = pending && !a_hot && res_a == key;
    const bool own_b = pending && !b_hot && res_b == key;
    const int oa_g = (int)(res_a & ((1u << kLaneBits) - 1u)), ob_g = (int)(res_b & ((1u << kLaneBits) - 1u));
    bool hot_lane = pending && lit;
    const bool own = pending && (a_hot || own_a) && (b_hot || own_b);
    // ---- the chain on the hot region: the first wavefront with pending lanes only --------------
    bool chain_commit = false, chain_kept = false, chain_internal = false;
    if (w == w0) {
      // (owners of this wavefront's regions are lanes of this wavefront: the earlier ones are done)
      const int oa = oa_g & 63, ob = ob_g & 63;
      const bool oa_here = (oa_g >> 6) == w, ob_here = (ob_g >> 6) == w;
      RState Hs = {}, P = {};
      int ps = 0;
      bool elig = false, both = false, merging = false, case_s = false, fin = false;
      const unsigned long long lit_mask = __ballot(hot_lane);
      const bool chain_possible =
          lit_mask != 0 && ((__ballot(hot_lane && (own_a || own_b || free_p)) >> __builtin_ctzll(lit_mask)) & 1ull);
      if (chain_possible) {
        Hs = WideLoad(tab, hot);   // uniform
        fin = (Hs.flags & kFlagFinalized) != 0;
        const bool mode_ok = !(Hs.flags & kFlagNoDesc) && (!fin || Hs.sz >= T.min_size);
        const bool base = pending && mode_ok && !failed;
        const bool fin_a = fin || (A.flags & kFlagFinalized), fin_b = fin || (B.flags & kFlagFinalized);
        const bool noop_a = any_kept_kind &&
                            ((A.cons >= 0 && Hs.cons >= 0) ? (A.cons != Hs.cons)
                                                           : (fin_a && A.sz >= T.min_size && Hs.sz >= T.min_size));
        const bool noop_b = any_kept_kind &&
                            ((B.cons >= 0 && Hs.cons >= 0) ? (B.cons != Hs.cons)
                                                           : (fin_b && B.sz >= T.min_size && Hs.sz >= T.min_size));
        // (a certainly kept edge changes nothing, whatever marks its partner carries: only a partner
        // that MERGES has to be plain -- no mark to hand on, no missing descriptor)
        const bool plain_a =
            base && ((noop_a && T.relax) ||
                     (PlainPartner(A.flags) &&
                      (noop_a || ((A.cons < 0 || A.cons == Hs.cons) && A.sz < Hs.sz &&
                                  (!fin_a || A.cons >= 0 || Hs.sz >= T.min_size)))));
        const bool plain_b =
            base && ((noop_b && T.relax) ||
                     (PlainPartner(B.flags) &&
                      (noop_b || ((B.cons < 0 || B.cons == Hs.cons) && B.sz < Hs.sz &&
                                  (!fin_b || B.cons >= 0 || Hs.sz >= T.min_size)))));
        const bool part_a = plain_a && (noop_l ? (noop_a && free_p && b_hot) : own_a);
        const bool part_b = plain_b && (noop_l ? (noop_b && free_p && a_hot) : own_b);
        const bool merge_a = part_a && !noop_a && (A.cons >= 0 || !fin_a || A.sz < T.min_size);
        const bool merge_b = part_b && !noop_b && (B.cons >= 0 || !fin_b || B.sz < T.min_size);
        const bool abs_a = pending && !own_a && oa_here, abs_b = pending && !own_b && ob_here;   // may be absorbed
        const bool cand1 = merge_b && !b_hot;   // partner b, hot side a
        const bool cand2 = merge_a && !a_hot;   // partner a, hot side b
        const unsigned long long stat = __ballot((cand1 && a_hot) || (cand2 && b_hot));
        const int dyn = (cand1 && abs_a && !a_hot) ? oa : ((cand2 && abs_b && !b_hot) ? ob : -1);
        unsigned long long em = stat;   // chain lanes that merge
        for (;;) {
          const unsigned long long em2 = stat | __ballot(dyn >= 0 && ((em >> (dyn & 63)) & 1ull));
          if (em2 == em) break;
          em = em2;
        }
        const bool ea = a_hot || (abs_a && ((em >> oa) & 1ull));
        const bool eb = b_hot || (abs_b && ((em >> ob) & 1ull));
        hot_lane = pending && (ea || eb);
        both = hot_lane && ea && eb;
        const bool pb_side = ea;   // the partner is the end that is not effectively hot
        P.d0 = pb_side ? B.d0 : A.d0;
        P.d1 = pb_side ? B.d1 : A.d1;
        P.d2 = pb_side ? B.d2 : A.d2;
        P.sz = pb_side ? B.sz : A.sz;
        P.cons = pb_side ? B.cons : A.cons;
        P.flags = pb_side ? B.flags : A.flags;
        ps = pb_side ? sb : sa;
        elig = hot_lane && !both && (pb_side ? part_b : part_a);
        merging = hot_lane && !both && (pb_side ? merge_b : merge_a);
        case_s = P.cons >= 0;
      }
      const unsigned long long hot_mask = __ballot(hot_lane);
      const unsigned long long elig_mask = __ballot(elig);
      const unsigned long long blocked = hot_mask & ~(elig_mask | __ballot(both));
      const unsigned long long prefix = blocked ? ((1ull << __builtin_ctzll(blocked)) - 1ull) : ~0ull;
      const unsigned long long chain_mask = elig_mask & prefix;
      // The first hot lane, when it is no chain lane, is replayed alone by the generic code.
      const bool solo = hot_lane && own && !elig && !both && lane == (int)__builtin_ctzll(hot_mask | (1ull << 63));
      // (decided here, carried out below together with the other wavefronts' generic lanes)
      chain_internal = false;
      if (chain_mask) {
        const bool in_chain = (chain_mask >> lane) & 1ull;
        merging = in_chain && merging;
        case_s = merging && case_s;
        const bool fin_l = fin || (P.flags & kFlagFinalized);   // this lane's edge is not tested
        const bool tested = merging && (case_s || !fin_l);
        const int v = merging ? P.sz : 0;
        const int incl = WaveInclusiveSum(v);
        const int S = Hs.sz + incl - v;     // size of the hot region before this lane's merge
        const float denom = 1.0f / (float)(P.sz + S);
        const float ca = (float)P.sz * denom;
        const float cb = (float)S * denom;
        const float t0 = ca * P.d0, t1 = ca * P.d1, t2 = ca * P.d2;
        const unsigned long long merging_mask = __ballot(merging);
        float r0 = Hs.d0, r1 = Hs.d1, r2 = Hs.d2;   // hot mean before this lane's merge
        float h0 = Hs.d0, h1 = Hs.d1, h2 = Hs.d2;   // hot mean after the whole chain (uniform)
        if (merging_mask) {
          const int m = (int)__popcll(merging_mask);
          const int idx = (int)__popcll(merging_mask & ((1ull << lane) - 1ull));
          if (merging) chain_buf[idx] = make_float4(t0, t1, t2, cb);
          WaveSyncW();
          float c = 1.0f, u0 = 0.0f, u1 = 0.0f, u2 = 0.0f;   // identity for the lanes >= m
          if (lane < m) {
            const float4 q = chain_buf[lane];
            u0 = q.x;
            u1 = q.y;
            u2 = q.z;
            c = q.w;
          }
          if (lane == 0) {   // lane 0 starts from the hot region's mean and ignores what is shifted in
            u0 = u0 + c * Hs.d0;
            u1 = u1 + c * Hs.d1;
            u2 = u2 + c * Hs.d2;
            c = 0.0f;
          }
          float g0 = u0, g1 = u1, g2 = u2;
          for (int s4 = 1; s4 < m; s4 += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              g0 = u0 + c * DppWaveShr1Zero(g0);
              g1 = u1 + c * DppWaveShr1Zero(g1);
              g2 = u2 + c * DppWaveShr1Zero(g2);
            }
          }
          const float b0 = DppWaveShr1Old(g0, Hs.d0);
          const float b1 = DppWaveShr1Old(g1, Hs.d1);
          const float b2 = DppWaveShr1Old(g2, Hs.d2);
          h0 = ReadLaneF(g0, m - 1);
          h1 = ReadLaneF(g1, m - 1);
          h2 = ReadLaneF(g2, m - 1);
          WaveSyncW();
          if (lane < m) chain_buf[lane] = make_float4(b0, b1, b2, 0.0f);
          WaveSyncW();
          if (merging) {
            const float4 q = chain_buf[idx];
            r0 = q.x;
            r1 = q.y;
            r2 = q.z;
          }
        }
        unsigned long long fail = 0;
        {
          const float x = r0 - P.d0, y = r1 - P.d1, z = r2 - P.d2;
          const float sd = (x * x + y * y + z * z) * (1.0f / 3.0f);
          const bool pass = case_s ? !(sd > T.split_s) : (sd <= T.pass_s);
          fail = __ballot(tested && !pass);
        }
        int fcut = 64;
        RState Hn = Hs;
        if (fail) {
          fcut = (int)__builtin_ctzll(fail);
          if (lane == fcut) failed = true;
          Hn.d0 = ReadLaneF(r0, fcut);
          Hn.d1 = ReadLaneF(r1, fcut);
          Hn.d2 = ReadLaneF(r2, fcut);
          Hn.sz = ReadLaneI(S, fcut);
        } else {
          Hn.d0 = h0;
          Hn.d1 = h1;
          Hn.d2 = h2;
          Hn.sz = Hs.sz + ReadLaneI(incl, 63);
        }
        const unsigned long long below = (fcut < 64) ? ((1ull << fcut) - 1ull) : ~0ull;
        const bool do_commit = in_chain && lane < fcut;
        if (do_commit) {
          if (merging) {
            WideCommitLoser(tab, nodes, ps, hot);
            if (case_s) ++C.n_forced; else if (fin_l) ++C.n_small; else ++C.n_regular;
            chain_commit = true;
          } else {
            chain_kept = true;   // noop_x: nothing changes
          }
        }
        // an edge with both ends (by then) inside the hot region is internal
        if (both && lane < fcut && ((prefix >> lane) & 1ull)) chain_internal = true;
        if (lane == 0 && (merging_mask & below)) WideStore(tab, hot, Hn, kTabDirty);
        if (lane == 0) C.chain_lanes += (unsigned)__popcll(chain_mask & below);
      }
      // this wavefront's generic lanes: the lanes that own both regions and are away from the hot
      // region, the first hot lane alone, the kept lanes
      const bool n_win_here = pending && ((own && (!hot_lane || solo)) || kept_now);
      if (chain_commit) pending = false;
      if (chain_kept) {
        my_kept = true;
        pending = false;
      }
      if (chain_internal) pending = false;
      hot_lane = n_win_here;   // (reused below as "generic lane of the hot wavefront")
    }
    const bool n_win = pending && (w == w0 ? hot_lane : ((own && !lit) || kept_now));
    // ---- lanes that own both regions: generic edge ----------------------------------------------
    if (n_win) {
      RState s1 = A, s2 = B;
      const RState o1 = s1, o2 = s2;
      int stat;
      const int out = DecideEdge(s1, s2, T, stat);
      if (optimistic) {
        const bool v = (out == kOutKeep)     ? TentativeViolated(o1, o2, s1, s2)
                       : (out == kOutMerge1) ? TentativeViolated(o1, o2, s1, s1)
                                             : TentativeViolated(o1, o2, s2, s2);
        if (v) *violation = 1;
      }
      if (stat == 4 && T.rle) *violation = 1;
      C.n_forced += (stat == 1);
      C.n_regular += (stat == 2);
      C.n_small += (stat == 3);
      if (out == kOutKeep) {
        my_kept = true;
        if (!SameState(o1, s1)) WideStore(tab, sa, s1, kTabDirty);
        if (!SameState(o2, s2)) WideStore(tab, sb, s2, kTabDirty);
      } else if (out == kOutMerge1) {
        WideStore(tab, sa, s1, kTabDirty);
        WideCommitLoser(tab, nodes, sb, sa);
      } else {
        WideStore(tab, sb, s2, kTabDirty);
        WideCommitLoser(tab, nodes, sa, sb);
      }
      pending = false;
    }
    __syncthreads();   // the round's commits are in the table
  }
  return my_kept;
}

}  // namespace

template <int W>
__global__ __launch_bounds__(64 * W) void k_merge_wide(const int32_t* __restrict__ seg_off,
                                                       const int32_t* __restrict__ seg_cnt,
                                                       const int32_t* __restrict__ s_ra,
                                                       const int32_t* __restrict__ s_rb,
                                                       const uint32_t* __restrict__ s_gpos,
                                                       NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                       StageThr T, int optimistic,
                                                       int32_t* __restrict__ violation,
                                                       unsigned long long* __restrict__ stats,
                                                       int wide_min, int wave_max,
                                                       const uint32_t* __restrict__ work_list,
                                                       int32_t* __restrict__ work_ctl) {
  constexpr int NT = 64 * W;
  __shared__ WideTable<W> tab;
  __shared__ WideShared<W> xw;
  __shared__ int32_t st_ra[2 * NT], st_rb[2 * NT];
  __shared__ uint32_t st_gpos[2 * NT];
  __shared__ float4 chain_buf[64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int s = tid; s < WideTable<W>::kSlots; s += NT) {
    tab.key[s] = -1;
    tab.res[s] = 0xffffffffu;
    tab.res2[s] = 0xffffffffu;
    tab.res3[s] = 0xffffffffu;
  }
  if (tid == 0) xw.drop_stamp = 0;
  __syncthreads();
  const int c0 = work_ctl[0];   // the components of the largest size class (filed by k_merge_small)
  WideCounters C;
  unsigned long long edges_taken = 0;
  for (;;) {
    if (tid == 0) {
      const int ticket = atomicAdd(&work_ctl[kWaveClasses + 1], 1);
      xw.next_seg = ticket < c0 ? (int)work_list[ticket] : -1;
    }
    __syncthreads();
    const int seg = xw.next_seg;
    __syncthreads();   // everybody has read next_seg before the next ticket overwrites it
    if (seg < 0) break;
    const int cnt = seg_cnt[seg];
    if (cnt < wide_min || cnt >= wave_max) continue;   // the wave worker's, or the tree replay's
    const int beg = seg_off[seg];
    const int end = beg + cnt;
    if (tid == 0) edges_taken += (unsigned long long)cnt;
    int pos = beg;      // next edge of the component to read (uniform)
    int n_raw = 0;      // staged edges left over from the previous batch: roots to be re-validated
    for (;;) {
      // ---- stage 64 * W live edges: internal edges are dropped as they are read --------------------
      const unsigned long long t_b0 = __builtin_readcyclecounter();
      int n_valid = 0;
      while (n_valid < NT && (pos < end || n_raw > 0)) {
        const int room = NT - n_raw;
        const int n_new = (end - pos) < room ? (end - pos) : room;
        const bool cand = tid < n_raw + n_new;
        int ca = 0, cb = 0, xa = -1, xb = -1;
        uint32_t cg = 0;
        if (tid < n_raw) {
          ca = st_ra[tid];
          cb = st_rb[tid];
          cg = st_gpos[tid];
        } else if (cand) {
          const int p = pos + (tid - n_raw);
          ca = xa = s_ra[p];
          cb = xb = s_rb[p];
          cg = s_gpos[p];
        }
        for (bool more = cand; more;) {
          const int pa = nodes.parent[ca], pb = nodes.parent[cb];
          more = (pa != ca) || (pb != cb);
          ca = pa;
          cb = pb;
        }
        // Path compression of the start node only (it is not a root, so nobody else writes it); an
        // optimistic stage must stay undoable and does not compress.
        if (!optimistic && xa >= 0) {
          if (ca != xa) nodes.parent[xa] = ca;
          if (cb != xb) nodes.parent[xb] = cb;
        }
        const bool live = cand && ca != cb;
        const unsigned long long lm = __ballot(live);
        if (lane == 0) xw.cnt[w] = (int)__popcll(lm);
        __syncthreads();   // the counts are in; every left-over has been read from the stage arrays
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
          const int c = xw.cnt[k];
          if (k < w) before += c;
          total += c;
        }
        if (live) {
          const int at = n_valid + before + (int)__popcll(lm & ((1ull << lane) - 1ull));
          st_ra[at] = ca;
          st_rb[at] = cb;
          st_gpos[at] = cg;
        }
        n_valid += total;
        pos += n_new;
        n_raw = 0;
        __syncthreads();
      }
      if (n_valid == 0) break;   // drained
      // ---- the batch: the first 64 * W staged edges (their roots are current) --------------------
      const int take = n_valid < NT ? n_valid : NT;
      const bool valid = tid < take;
      int ra = -1, rb = -1;
      uint32_t gpos = 0;
      if (valid) {
        ra = st_ra[tid];
        rb = st_rb[tid];
        gpos = st_gpos[tid];
      }
      n_raw = n_valid - take;
      if (n_raw > 0) {   // move the rest to the front; it is re-validated by the next pass
        int ya = 0, yb = 0;
        uint32_t yg = 0;
        if (tid < n_raw) {
          ya = st_ra[NT + tid];
          yb = st_rb[NT + tid];
          yg = st_gpos[NT + tid];
        }
        __syncthreads();
        if (tid < n_raw) {
          st_ra[tid] = ya;
          st_rb[tid] = yb;
          st_gpos[tid] = yg;
        }
      }
      int sa = 0, sb = 0;
      int mine_a = -1, mine_b = -1;   // slots this thread inserted (it writes them back and frees them)
      if (valid) {
        const RState A = LoadState(nodes, ra), B = LoadState(nodes, rb);   // both in flight
        bool ins_a, ins_b;
        sa = WideInsert<W>(tab, ra, ins_a);
        sb = WideInsert<W>(tab, rb, ins_b);
        if (ins_a) {
          tab.link[sa] = sa;
          WideStore(tab, sa, A, 0);
          mine_a = sa;
        }
        if (ins_b) {
          tab.link[sb] = sb;
          WideStore(tab, sb, B, 0);
          mine_b = sb;
        }
      }
      if (tid == 0) {
        ++C.batches;
        C.lanes += (unsigned)take;
      }
      __syncthreads();
      const unsigned long long t_r0 = __builtin_readcyclecounter();
      if (tid == 0) C.cyc_stage += t_r0 - t_b0;

      const bool my_kept = ReplayRoundsWide<W>(tab, xw, chain_buf, nodes, T, optimistic, violation, stats, tid,
                                               valid, sa, sb, C);
      __syncthreads();
      if (tid == 0) C.cyc_rounds += __builtin_readcyclecounter() - t_r0;

      if (valid && my_kept) kept_all[gpos] = 1;
      if (T.side && __ballot(my_kept) && lane == 0) atomicOr(violation, 2);
      // ---- write the changed regions back, reset the table ---------------------------------------
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int s = e ? mine_b : mine_a;
        if (s >= 0) {
          if (tab.link[s] == s && (tab.flags[s] & kTabDirty)) {
            const RState r = WideLoad(tab, s);
            StoreState(nodes, tab.key[s], r);
          }
          tab.key[s] = -1;
          tab.res[s] = 0xffffffffu;
          tab.res2[s] = 0xffffffffu;
          tab.res3[s] = 0xffffffffu;
        }
      }
      // Make this batch's stores visible to the next batch's loads (same CU: L1 is shared).
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __syncthreads();
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    C.n_forced += __shfl_down(C.n_forced, off);
    C.n_regular += __shfl_down(C.n_regular, off);
    C.n_small += __shfl_down(C.n_small, off);
  }
  if (lane == 0) {
    if (C.n_forced) atomicAdd(&stats[0], (unsigned long long)C.n_forced);
    if (C.n_regular) atomicAdd(&stats[1], (unsigned long long)C.n_regular);
    if (C.n_small) atomicAdd(&stats[2], (unsigned long long)C.n_small);
  }
  if (tid == 0) {
    if (edges_taken) atomicAdd(&stats[3], edges_taken);
    if (C.rounds) atomicAdd(&stats[5], (unsigned long long)C.rounds);
    if (C.batches) atomicAdd(&stats[7], (unsigned long long)C.batches);
    if (edges_taken) atomicAdd(&stats[31], edges_taken);   // replayed by the wide worker
    if (C.rounds) atomicAdd(&stats[44], (unsigned long long)C.rounds);
    if (C.batches) atomicAdd(&stats[45], (unsigned long long)C.batches);
    if (C.cyc_stage) atomicAdd(&stats[46], C.cyc_stage);
    if (C.cyc_rounds) atomicAdd(&stats[47], C.cyc_rounds);
    if (C.lanes) atomicAdd(&stats[72], C.lanes);
    if (C.fix_iters) atomicAdd(&stats[74], C.fix_iters);
  }
  if (lane == 0 && C.chain_lanes) atomicAdd(&stats[73], C.chain_lanes);
}

void LaunchMergeWide(int grid, const WorkerArgs& a, int waves, hipStream_t s) {
  if (waves >= 4) {
    hipLaunchKernelGGL(k_merge_wide<4>, dim3(grid), dim3(256), 0, s, a.seg_off, a.seg_cnt, a.s_ra, a.s_rb,
                       a.s_gpos, a.nodes, a.kept_all, a.T, a.optimistic, a.violation, a.stats, a.wide_min,
                       a.wave_max, a.work_list, a.work_ctl);
  } else {
    hipLaunchKernelGGL(k_merge_wide<2>, dim3(grid), dim3(128), 0, s, a.seg_off, a.seg_cnt, a.s_ra, a.s_rb,
                       a.s_gpos, a.nodes, a.kept_all, a.T, a.optimistic, a.violation, a.stats, a.wide_min,
                       a.wave_max, a.work_list, a.work_ctl);
  }
  VSG_HIP(hipGetLastError());
}

}  // namespace vsg
