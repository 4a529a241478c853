// merge_wave.hip -- the default worker of the ordered merge: round-based replay of a large
// component by one consumer wavefront fed by a reader wavefront.
#include "merge_common.h"

namespace vsg {

// ------------------------------------------------------------------------------------------
// Worker B: one wavefront replays one large component, 64 edges per batch, in *rounds*.
// ------------------------------------------------------------------------------------------
// A lone wavefront issues roughly one instruction every 4-8 cycles, so replaying the 64 edges of
// a batch one after the other (the worker of round 1a, ~130 instructions per edge) leaves the stage bound
// by its largest component.  This worker keeps the regions of the batch in an LDS table and
// commits as many edges per round as the sequential semantics allow:
//
//   * every pending lane reserves its two regions with its lane number (ds_min); a lane that
//     holds both reservations is the earliest pending edge on both regions, so executing it now
//     is what the sequential replay would do (deterministic reservations).  All such lanes run
//     DecideEdge at once, each on its own pair of regions;
//   * every round has a *hot* region that is not reserved: the larger end of the earliest pending
//     edge (which owns its other end by construction, so the chain can always start).  The
//     leading run of pending hot edges whose partner is a plain region (unflagged; unconstrained
//     or with the hot region's constraint) smaller than the hot region is committed as one
//     *chain*: under the speculation
//     that every merge test passes, the sizes are a prefix sum and the weights ca/cb and the
//     products ca*p are lane-parallel; only  h = ca*p + cb*h  (two flops per channel) is replayed
//     in order, recording the pre-merge mean per lane, and the merge tests are then verified by
//     all lanes at once.  The chain is cut at the first failed test and that lane is replayed
//     by the generic code in the next round, so the result is exactly the sequential one;
//   * the first pending hot edge that does not qualify for the chain runs alone (generic code).
//
// The earliest pending lane always commits, so a batch takes at most 64 rounds.
constexpr int kTabSize = 256;     // >= 2 * 128 distinct regions of a batch at load factor 1/2

struct WaveTable {
  int32_t key[kTabSize];     // region id, -1: empty
  int32_t link[kTabSize];    // in-batch union-find over slots
  uint32_t res[kTabSize];    // reservation: (0xfffff - round) << 6 | lane, smaller wins
  uint32_t res2[kTabSize];   // earliest lane a kept lane must not pass (ReplayRounds; only filled
  uint32_t res3[kTabSize];   // in the rounds that have a kept lane)
  float4 ds[kTabSize];
  int32_t cons[kTabSize];
  int32_t flags[kTabSize];   // region flags | kTabDirty
};

__device__ __forceinline__ int TabInsert(WaveTable& t, int r, bool& inserted) {
  unsigned h = ((unsigned)r * 2654435761u) >> 24;
  for (;;) {
    const int old = atomicCAS(&t.key[h], -1, r);
    if (old == -1) { inserted = true; return (int)h; }
    if (old == r) { inserted = false; return (int)h; }
    h = (h + 1) & (kTabSize - 1);
  }
}

__device__ __forceinline__ int SlotRoot(const WaveTable& t, int s) {
  int p;
  while ((p = t.link[s]) != s) s = p;
  return s;
}

__device__ __forceinline__ RState TabLoad(const WaveTable& t, int s) {
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

__device__ __forceinline__ void TabStore(WaveTable& t, int s, const RState& r, int dirty) {
  t.ds[s] = make_float4(r.d0, r.d1, r.d2, __int_as_float(r.sz));
  t.cons[s] = r.cons;
  t.flags[s] = r.flags | dirty;
}

constexpr int kFill = 4;        // 64-edge chunks read per fill
constexpr int kQueue = 512;     // ring capacity >= kLag + kFill * 64, power of two
// The producer reads on while fewer than kLag edges wait in the ring: enough to keep the consumer
// busy for a batch or two, few enough that the roots it found are mostly still current.
constexpr int kLag = 128;

struct WaveQueue {
  int32_t ra[kQueue];
  int32_t rb[kQueue];
  uint32_t gpos[kQueue];
  uint32_t seq[kQueue];   // sequence number of the edge among the stage's work edges (hubs: WorkerArgs::s_seq)
  int produced;   // entries pushed by the producer wave (monotonic)
  int consumed;   // entries taken by the consumer wave (monotonic)
  int done;       // the producer has read the whole component
};

// Live edges staged by the consumer (<= 63 left over + 64 new).
struct WaveStage {
  int32_t ra[128];
  int32_t rb[128];
  uint32_t gpos[128];
  uint32_t seq[128];
};

// Orders the LDS accesses of the lanes of ONE wavefront (they execute in order in hardware; this
// only keeps the compiler from moving them across the phase boundary).
__device__ __forceinline__ void WaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// Commit of a merge inside the wave worker: `ls` (loser slot) is linked to `ws` (winner slot).
// A region that stops being a representative keeps its own constraint field in the reference
// (MergeRegions only updates the survivor), and MergeConstrainedRegions later reads that field of
// every *node* -- so a constraint the loser acquired or lost earlier in this batch (its table
// entry is dirty and will never be written back) has to reach memory now.
__device__ __forceinline__ void CommitLoser(WaveTable& t, const NodeArrays& nodes, int ls, int ws) {
  t.link[ls] = ws;
  const int lid = t.key[ls];
  nodes.parent[lid] = t.key[ws];
  if (t.flags[ls] & kTabDirty) nodes.cons[lid] = t.cons[ls];
}

// Counters of one worker wavefront (per lane where noted; reduced at the end of the kernel).
struct WaveCounters {
  unsigned n_forced = 0, n_regular = 0, n_small = 0;   // per lane
  unsigned dbg_rounds = 0, dbg_nwin = 0, dbg_chain = 0, dbg_solo = 0, dbg_cut = 0, dbg_kept = 0;
  unsigned long long cyc_ph[5] = {0, 0, 0, 0, 0};   // reserve+load, closure, masks, generic, chain
  unsigned long long dbg_x[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long dbg_r[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // why the first blocked hot lane is no chain lane
};

// Replays one batch: lane `lane` holds the edge whose end regions sit in the table slots sa / sb
// (`valid`: the lane has an edge).  Returns whether this lane's edge is kept.
template <bool kDbg>
__device__ __forceinline__ bool ReplayRounds(WaveTable& tab, float4* chain_buf, const NodeArrays& nodes,
                                             const StageThr& T, int optimistic, int32_t* violation,
                                             unsigned long long* stats, int dbg_flags, int lane,
                                             bool valid, int sa, int sb, WaveCounters& C, uint32_t seq,
                                             int32_t* hub_mark, int32_t* hub_excl) {
  auto Clock = []() -> unsigned long long { return kDbg ? __builtin_readcyclecounter() : 0ull; };
  bool pending = valid;
  int hot = -1;   // wave-uniform slot of the round's hot region
  bool my_kept = false;
  bool failed = false;    // this lane's chain test failed: replay it with the generic code
  for (unsigned round = 0;; ++round) {
    {   // current root slots of both ends
      for (bool more = pending; more;) {
        const int pa = tab.link[sa], pb = tab.link[sb];
        more = (pa != sa) || (pb != sb);
        sa = pa;
        sb = pb;
      }
      if (pending && sa == sb) pending = false;   // became internal
    }
    const unsigned long long pend_mask = __ballot(pending);
    if (!pend_mask) break;
    // The round's hot region: the larger end of the earliest pending edge.  That edge owns
    // its other end by construction, so the chain can always start, and the run of edges that
    // depends on it (the growth front of its cluster) joins the chain in this round (2.6 M instead
    // of 5.5 M rounds for the cluster bucket of the 1080p input, compared with one hot region
    // per batch).
    {
      const int first = (int)__builtin_ctzll(pend_mask);
      const int fa = ReadLaneI(sa, first), fb = ReadLaneI(sb, first);
      const int sza = __float_as_int(tab.ds[fa].w), szb = __float_as_int(tab.ds[fb].w);
      hot = (sza >= szb) ? fa : fb;
      if (kDbg && (dbg_flags & 4)) hot = -1;
      // (an edge on a hub of the stage is decided by its other region alone: no hot region then)
      if (T.hubs && ((tab.flags[fa] | tab.flags[fb]) & kFlagHub)) hot = -1;
    }
    if (round > 140u) {   // cannot happen (the earliest pending lane commits, after at most one failed chain test): report
      if (lane == 0) atomicAdd(&stats[22], 1ull);
      break;
    }
    const unsigned long long ph0 = Clock();
    const bool a_hot = (sa == hot), b_hot = (sb == hot);
    const uint32_t key = ((0xfffffu - round) << 6) | (uint32_t)lane;
    if (pending) {
      if (!a_hot) atomicMin(&tab.res[sa], key);
      if (!b_hot) atomicMin(&tab.res[sb], key);
    }
    WaveSync();
    uint32_t res_a = 0, res_b = 0;
    RState A = {}, B = {};
    if (pending) {
      res_a = tab.res[sa];
      res_b = tab.res[sb];
      A = TabLoad(tab, sa);
      B = TabLoad(tab, sb);
    }
    // Hub ends (kFlagHub): never reserved for real -- every edge on a hub is decided by its other
    // region alone (HubEdge), so edges on the same hub do not wait for each other.
    const bool a_hub = T.hubs && pending && (A.flags & kFlagHub);
    const bool b_hub = T.hubs && pending && (B.flags & kFlagHub);
    if ((a_hub && (A.flags & kFlagHubBroken)) || (b_hub && (B.flags & kFlagHubBroken))) atomicOr(violation, kHubVioBroken);
    // An edge that is certainly *kept without changing anything* (NoopPair: different constraints,
    // or one region finalized and both large) does not have to wait for earlier edges of the same
    // kind *that are committed in the same round*, only for earlier edges whose outcome is open.
    // (A giant region that keeps all its neighbours meets each of them several times within a
    // batch, next to the edges between the neighbours themselves: with one reservation that was
    // 1.4 edges per round on components of half a million.)  Two more reservation words per region,
    // filled only in the rounds that have such a lane:
    //   res3[x]: earliest *blocker* on x -- a lane that may change a state, or a kept lane away from
    //            the hot region that does not commit in this round (found by iteration: a kept lane
    //            behind a blocker becomes a blocker itself; a kept lane that waits may well change
    //            something once its other region has changed);
    //   res2[x]: the same plus the kept lanes on the hot region (they commit with the chain or
    //            not at all, which is not known yet: lanes away from the hot region do not pass
    //            them).
    // A kept lane away from the hot region commits when res2 holds no earlier lane on either
    // end; a kept lane on the hot region may join the chain when res3 holds no earlier lane on its
    // partner (what it passes are committed kept lanes and earlier chain lanes on the same partner).
    // Most rounds of the force-merge buckets have no finalized region and no pair of constraints
    // in the wavefront at all: they skip all of it (any_kept_kind).
    const bool any_kept_kind =
        __ballot(pending && (((A.flags | B.flags) & kFlagFinalized) ||
                             (A.cons >= 0 && B.cons >= 0 && A.cons != B.cons))) != 0 &&
        !(kDbg && (dbg_flags & 512));
    bool noop_l = false;      // this lane is kept and changes nothing (as the states are now)
    bool kept_now = false;    // ... is away from the hot region and commits in this round
    bool free_p = false;      // ... is on the hot region and nothing open precedes it on its partner
    if (any_kept_kind) {
      const bool lit = a_hot || b_hot;
      RState Hs0 = {};
      if (__ballot(pending && lit)) Hs0 = TabLoad(tab, hot);   // uniform
      noop_l = pending && (a_hot ? NoopPair(B, Hs0, T) : (b_hot ? NoopPair(A, Hs0, T) : NoopPair(A, B, T)));
      if (__ballot(pending && !noop_l) == 0) {
        // Every pending lane of the batch is of this kind: nothing can change a state any more, so
        // nothing can change what they are -- all of them are kept, in one round.  (The stages that
        // consist of kept edges only -- 400 K of them around one giant region in the first chunk of
        // a low-contrast video -- took 26 rounds per batch.)
        if (pending) my_kept = true;
        pending = false;
        break;
      }
      if (__ballot(noop_l)) {
        if (pending && !noop_l) {
          if (!a_hot && !a_hub) {
            atomicMin(&tab.res2[sa], key);
            atomicMin(&tab.res3[sa], key);
          }
          if (!b_hot && !b_hub) {
            atomicMin(&tab.res2[sb], key);
            atomicMin(&tab.res3[sb], key);
          }
        } else if (noop_l && lit) {
          atomicMin(&tab.res2[a_hot ? sb : sa], key);
        }
        WaveSync();
        bool cand = noop_l && !lit;
        for (;;) {
          const bool ok = cand && (a_hub || tab.res2[sa] > key) && (b_hub || tab.res2[sb] > key);
          const bool drop = cand && !ok;
          cand = ok;
          if (!__ballot(drop)) break;
          if (drop) {
            if (!a_hub) {
              atomicMin(&tab.res2[sa], key);
              atomicMin(&tab.res3[sa], key);
            }
            if (!b_hub) {
              atomicMin(&tab.res2[sb], key);
              atomicMin(&tab.res3[sb], key);
            }
          }
          WaveSync();
        }
        kept_now = cand;
        free_p = noop_l && lit && tab.res3[a_hot ? sb : sa] > key;
      } else {
        noop_l = false;
      }
    }
    // own_x: this lane is the earliest pending edge on region x (the hot region is not reserved)
    const unsigned long long ph1 = Clock();
    const bool own_a = pending && !a_hot && (res_a == key || a_hub);
    const bool own_b = pending && !b_hot && (res_b == key || b_hub);
    const int oa = (int)(res_a & 63u), ob = (int)(res_b & 63u);   // owners (earlier lanes)
    RState Hs = {}, P = {};
    int ps = 0;            // partner slot of a chain lane
    bool hot_lane = pending && (a_hot || b_hot);
    bool elig = false;     // chain lane
    bool both = false;     // both ends (will) belong to the hot region: internal once committed
    bool merging = false;
    bool case_s = false;
    bool fin = false;
    int why = 0;
    // A chain starts at the first edge that touches the hot region and only if that lane is the
    // earliest pending edge on its other end; otherwise nothing can be absorbed in this round
    // and the classification below is skipped (the other hot edges just wait).
    const unsigned long long lit_mask = __ballot(hot_lane);
    const bool chain_possible =
        lit_mask != 0 && ((__ballot(hot_lane && (own_a || own_b || free_p)) >> __builtin_ctzll(lit_mask)) & 1ull);
    if (chain_possible) {
      Hs = TabLoad(tab, hot);   // uniform
      fin = (Hs.flags & kFlagFinalized) != 0;
      const bool mode_ok = !(Hs.flags & kFlagNoDesc) && (!fin || Hs.sz >= T.min_size) &&
                           !(kDbg && (dbg_flags & 1));
      // A region is *effectively hot* for a lane when it is the hot region or when its owner
      // (an earlier lane) is a chain lane that absorbs it into the hot region: by the time this
      // lane is replayed the region is part of the hot one.  So a run of edges p1-p2, p2-p3, ...
      // hanging off the hot region joins the chain in one round.  The set of absorbing lanes
      // only grows, so the loop ends (no memory access inside).
      // Per end, evaluated once: would this end qualify as the partner of a chain edge
      // (plain, smaller, owned by this lane) and would the edge merge?
      // Case S (partner with the hot region's constraint) merges unless the descriptors are
      // further apart than the split threshold, whatever the sizes and flags; Case U
      // (unconstrained partner): regular test while the hot region is not finalized, a finalized
      // hot region (>= min size) absorbs small partners only.
      const bool base = pending && mode_ok && !failed;
      // (fin_x: the unconstrained edge to end x is not tested -- one of the two is finalized; the
      // hot region has to be large then, or it would be the one that is absorbed as "small")
      const bool fin_a = fin || (A.flags & kFlagFinalized), fin_b = fin || (B.flags & kFlagFinalized);
      // noop_x: the edge to end x is kept and changes nothing, whichever of the two is larger --
      // different constraints, or (unconstrained rule) one of them finalized and both large.  The
      // tail of the first chunk of a video is made of these: the remaining unfinalized regions
      // against their finalized neighbours, one round per edge unless they may join the chain.
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
      // (a lane that took part in the reservations as a kept lane may only be one: on the hot
      // region as a chain lane, away from it through the generic code below -- or it waits)
      const bool part_a = plain_a && (noop_l ? (noop_a && free_p && b_hot) : own_a);
      const bool part_b = plain_b && (noop_l ? (noop_b && free_p && a_hot) : own_b);
      const bool merge_a = part_a && !noop_a && (A.cons >= 0 || !fin_a || A.sz < T.min_size);
      const bool merge_b = part_b && !noop_b && (B.cons >= 0 || !fin_b || B.sz < T.min_size);
      const bool abs_a = pending && !own_a && !a_hub, abs_b = pending && !own_b && !b_hub;   // may be absorbed
      // A lane merges into the chain when one end is effectively hot and the other end is a
      // partner it owns (merge_x implies own_x, so that end can only be hot literally).  Hence a
      // lane depends on at most ONE earlier lane -- the owner of its not-owned end -- and the
      // closure is an OR propagation along those single links:
      //   em = stat | { j : dyn_j in em }.
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
      if constexpr (kDbg) {
        const bool ow = pb_side ? own_b : own_a;
        const int fl = pb_side ? B.flags : A.flags;
        const bool fl_l = fin || (fl & kFlagFinalized);
        why = (!ow ? 1 : 0) | (!PlainPartner(fl) ? 2 : 0) | (!(P.cons < 0 || P.cons == Hs.cons) ? 4 : 0) |
              (!(P.sz < Hs.sz) ? 8 : 0) | (!base ? 16 : 0) |
              (!(!fl_l || P.cons >= 0 || Hs.sz >= T.min_size) ? 32 : 0);
      }
      case_s = P.cons >= 0;
    }
    const unsigned long long ph2 = Clock();
    const unsigned long long hot_mask = __ballot(hot_lane);
    const unsigned long long elig_mask = __ballot(elig);
    // hot lanes that are neither chain lanes nor internal end the chain
    const unsigned long long blocked = hot_mask & ~(elig_mask | __ballot(both));
    const unsigned long long prefix =
        blocked ? ((1ull << __builtin_ctzll(blocked)) - 1ull) : ~0ull;
    unsigned long long chain_mask = elig_mask & prefix;
    if (kDbg && (dbg_flags & 32)) {   // no jumping over earlier pending lanes
      const unsigned long long others = __ballot(pending) & ~chain_mask;
      if (others) chain_mask &= (1ull << __builtin_ctzll(others)) - 1ull;
    }
    if ((kDbg && (dbg_flags & 64)) && chain_mask) chain_mask = 1ull << __builtin_ctzll(chain_mask);
    // The first hot lane, when it is no chain lane, is replayed alone by the generic code (it
    // touches the hot region itself: nothing earlier can have absorbed one of its ends).
    const bool own = pending && (a_hot || own_a) && (b_hot || own_b);
    const bool solo = hot_lane && own && !elig && !both &&
                      lane == (int)__builtin_ctzll(hot_mask | (1ull << 63));
    bool n_win = pending && ((own && (!hot_lane || solo)) || kept_now);
    if (kDbg && (dbg_flags & 8)) n_win = n_win && lane == (int)__builtin_ctzll(__ballot(pending));
    if constexpr (kDbg) {
      const unsigned long long nwin_mask = __ballot(n_win), solo_mask = __ballot(solo);
      const unsigned long long pend_mask = __ballot(pending);
      if (lane == 0) {
        ++C.dbg_rounds;
        C.dbg_nwin += (unsigned)__popcll(nwin_mask);
        C.dbg_solo += (unsigned)__popcll(solo_mask);
        C.dbg_x[0] += (unsigned)__popcll(pend_mask);                 // pending lanes per round
        C.dbg_x[1] += (unsigned)__popcll(hot_mask);                  // (effectively) hot lanes
        C.dbg_x[2] += (unsigned)__popcll(blocked);                   // hot lanes that end the chain
        C.dbg_x[3] += (chain_mask != 0);                             // rounds with a chain
        C.dbg_x[4] += (nwin_mask != 0);                              // rounds with generic commits
        C.dbg_x[5] += (unsigned)__popcll(pend_mask & ~hot_mask & ~nwin_mask);   // waiting non-hot lanes
        if (blocked) {
          const int wb = ReadLaneI(why, (int)__builtin_ctzll(blocked));
          for (int q = 0; q < 6; ++q) C.dbg_r[q] += (wb >> q) & 1;
          C.dbg_r[6] += (wb == 0);
          ++C.dbg_r[7];
        }
      }
    }

    const unsigned long long ph3 = Clock();
    // ---- lanes on a hub of the stage: decided by the other region alone (merge_common.h: HubEdge) --
    if (n_win && (a_hub || b_hub)) {
      const RState& x = a_hub ? B : A;
      const RState& h = a_hub ? A : B;
      const int act = (a_hub && b_hub) ? HubHubEdge(A.cons, A.flags, B.cons, B.flags)
                                       : HubEdge(x, h.cons, h.flags, h.sz, T);
      if (act >= kHubViolation) {
        atomicOr(violation, act);
        HubViolationAt(hub_excl, 1, (int)seq);
        if (a_hub) HubExclude(hub_excl, nodes.flags, tab.key[sa]);
        if (b_hub) HubExclude(hub_excl, nodes.flags, tab.key[sb]);
      } else if (act == kHubKeep) {
        my_kept = true;
      } else {
        const int xs = a_hub ? sb : sa, hs = a_hub ? sa : sb;
        // (the state the hub absorbs is read from memory after the workers: k_hub_apply)
        if (tab.flags[xs] & kTabDirty) StoreState(nodes, tab.key[xs], x);
        CommitLoser(tab, nodes, xs, hs);
        if (x.flags & kFlagTentative) AtomicOrFlags(nodes.flags, tab.key[hs], kFlagTentative);
        hub_mark[seq] = tab.key[xs] | (act == kHubAbsorbTest ? kHubTestBit : 0);
        if (act == kHubAbsorbTest) ++C.n_forced; else ++C.n_small;
      }
      pending = false;
      n_win = false;
    }
    // ---- lanes that own both regions: generic edge ------------------------------------------
    if (n_win) {
      const RState& s1 = A;
      const RState& s2 = B;
      // Fast path, by far the most common generic edge: two plain regions (unconstrained,
      // not finalized, unmarked) that pass the regular test.  Same arithmetic as
      // DecideEdge / MergeStates for this case.
      if (s1.cons < 0 && s2.cons < 0 && (s1.flags | s2.flags) == 0 &&
          SquaredDistance(s1, s2) <= T.pass_s && !(kDbg && (dbg_flags & 256))) {
        const bool first = s1.sz > s2.sz;   // ties keep region 2
        const int ws = first ? sa : sb, ls = first ? sb : sa;
        RState m, o;
        m.d0 = first ? s1.d0 : s2.d0;
        m.d1 = first ? s1.d1 : s2.d1;
        m.d2 = first ? s1.d2 : s2.d2;
        m.sz = first ? s1.sz : s2.sz;
        o.d0 = first ? s2.d0 : s1.d0;
        o.d1 = first ? s2.d1 : s1.d1;
        o.d2 = first ? s2.d2 : s1.d2;
        o.sz = first ? s2.sz : s1.sz;
        const float denom = 1.0f / (float)(o.sz + m.sz);
        const float ca = (float)o.sz * denom;
        const float cb = (float)m.sz * denom;
        m.d0 = ca * o.d0 + cb * m.d0;
        m.d1 = ca * o.d1 + cb * m.d1;
        m.d2 = ca * o.d2 + cb * m.d2;
        m.sz += o.sz;
        m.cons = max(s1.cons, s2.cons);
        m.flags = 0;
        TabStore(tab, ws, m, kTabDirty);
        CommitLoser(tab, nodes, ls, ws);
        ++C.n_regular;
        pending = false;
        n_win = false;
      }
    }
    if (n_win) {
      RState s1 = A, s2 = B;
      const RState o1 = s1, o2 = s2;
      int stat;
      const int out = DecideEdge(s1, s2, T, stat);
      if (optimistic) {
        const bool v = (out == kOutKeep)     ? TentativeViolated(o1, o2, s1, s2)
                       : (out == kOutMerge1) ? TentativeViolated(o1, o2, s1, s1)
                                             : TentativeViolated(o1, o2, s2, s2);
        if (v) {
          atomicOr(violation, kVioCut);
          HubViolationAt(hub_excl, 1, (int)seq);
        }
      }
      if (stat == 4 && T.rle) {
        atomicOr(violation, kVioCut);
        HubViolationAt(hub_excl, 1, (int)seq);
      }
      C.n_forced += (stat == 1);
      C.n_regular += (stat == 2);
      C.n_small += (stat == 3);
      if (out == kOutKeep) {
        my_kept = true;
        if (!SameState(o1, s1)) TabStore(tab, sa, s1, kTabDirty);
        if (!SameState(o2, s2)) TabStore(tab, sb, s2, kTabDirty);
      } else if (out == kOutMerge1) {
        TabStore(tab, sa, s1, kTabDirty);
        CommitLoser(tab, nodes, sb, sa);
      } else {
        TabStore(tab, sb, s2, kTabDirty);
        CommitLoser(tab, nodes, sa, sb);
      }
      pending = false;
    }

    const unsigned long long ph4 = Clock();
    // ---- the chain on the hot region -----------------------------------------------------
    if (chain_mask) {
      const bool in_chain = (chain_mask >> lane) & 1ull;
      merging = in_chain && merging;
      case_s = merging && case_s;
      const bool fin_l = fin || (P.flags & kFlagFinalized);   // this lane's edge is not tested
      const bool tested = merging && (case_s || !fin_l);   // (a kept lane changes nothing: noop_x)
      const int v = merging ? P.sz : 0;
      const int incl = WaveInclusiveSum(v);
      const int S = Hs.sz + incl - v;     // size of the hot region before this lane's merge
      // MergeStates with o = partner, m = hot region
      const float denom = 1.0f / (float)(P.sz + S);
      const float ca = (float)P.sz * denom;
      const float cb = (float)S * denom;
      const float t0 = ca * P.d0, t1 = ca * P.d1, t2 = ca * P.d2;
      // The recurrence  h <- t + cb * h  is replayed on the merging lanes only: they are packed
      // into the lanes 0..m-1 (through LDS, position = rank among the merging lanes) and run as
      // a systolic chain: every step each lane takes the mean its left neighbour holds
      // (v_mul_f32_dpp wave_shr:1) and applies its own merge.  After step k the lanes 0..k hold
      // the mean after their merge, so m steps finish the chain (further steps change nothing).
      // Same two roundings per channel and merge as MergeStates, in the same order.
      const unsigned long long merging_mask = __ballot(merging);
      float r0 = Hs.d0, r1 = Hs.d1, r2 = Hs.d2;   // hot mean before this lane's merge
      float h0 = Hs.d0, h1 = Hs.d1, h2 = Hs.d2;   // hot mean after the whole chain (uniform)
      if (merging_mask) {
        const int m = (int)__popcll(merging_mask);
        const int idx = (int)__popcll(merging_mask & ((1ull << lane) - 1ull));
        if (merging) chain_buf[idx] = make_float4(t0, t1, t2, cb);
        WaveSync();
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
        // packed lane j: mean before its merge = what lane j-1 ends with
        const float b0 = DppWaveShr1Old(g0, Hs.d0);
        const float b1 = DppWaveShr1Old(g1, Hs.d1);
        const float b2 = DppWaveShr1Old(g2, Hs.d2);
        h0 = ReadLaneF(g0, m - 1);
        h1 = ReadLaneF(g1, m - 1);
        h2 = ReadLaneF(g2, m - 1);
        WaveSync();
        if (lane < m) chain_buf[lane] = make_float4(b0, b1, b2, 0.0f);
        WaveSync();
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
        if (kDbg && lane == 0) ++C.dbg_cut;
      } else {
        Hn.d0 = h0;
        Hn.d1 = h1;
        Hn.d2 = h2;
        Hn.sz = Hs.sz + ReadLaneI(incl, 63);
      }
      const unsigned long long below = (fcut < 64) ? ((1ull << fcut) - 1ull) : ~0ull;
      const bool do_commit = in_chain && lane < fcut;
      if constexpr (kDbg) if (dbg_flags & 16) {   // self check: replay the committed chain with DecideEdge
        RState Hc = Hs;
        unsigned bad = 0;
        for (unsigned long long mm = chain_mask & below; mm; mm &= mm - 1) {
          const int k = (int)__builtin_ctzll(mm);
          RState a = Hc, b = ReadLaneState(P, k);
          const RState b0 = b;
          int st;
          const int out = DecideEdge(a, b, T, st);
          const bool km = (merging_mask >> k) & 1ull;
          if (km) {
            if (out != kOutMerge1 ||
                st != (b0.cons >= 0 ? 1 : ((fin || (b0.flags & kFlagFinalized)) ? 3 : 2))) ++bad;
            Hc = a;
          } else {
            if (out != kOutKeep || !SameState(a, Hc) || !SameState(b, b0)) ++bad;
          }
        }
        if (__float_as_int(Hc.d0) != __float_as_int(Hn.d0) || __float_as_int(Hc.d1) != __float_as_int(Hn.d1) ||
            __float_as_int(Hc.d2) != __float_as_int(Hn.d2) || !SameState(Hc, Hn)) ++bad;
        if (fail) {
          RState a = Hc, b = ReadLaneState(P, fcut);
          int st;
          DecideEdge(a, b, T, st);
          if (st == 2 || st == 1) ++bad;
        }
        if (lane == 0 && bad) atomicAdd(&stats[23], (unsigned long long)bad);
      }
      if (do_commit) {
        if (merging) {
          CommitLoser(tab, nodes, ps, hot);
          if (case_s) ++C.n_forced; else if (fin_l) ++C.n_small; else ++C.n_regular;
        } else {
          my_kept = true;   // noop_x: nothing changes
        }
        pending = false;
      }
      // An edge with both ends (by then) inside the hot region is internal: every lane that
      // absorbs one of its ends is an earlier chain lane, committed if this lane is below the
      // cut.
      // (the debug modes that shorten the chain leave these lanes to the next round's internal test)
      if (both && lane < fcut && ((prefix >> lane) & 1ull) && !(kDbg && (dbg_flags & (32 | 64)))) {
        pending = false;
      }
      if (lane == 0) {
        if (kDbg) C.dbg_chain += (unsigned)__popcll(merging_mask & below);
        if (merging_mask & below) TabStore(tab, hot, Hn, kTabDirty);
      }
    }
    if constexpr (kDbg) {
      const unsigned long long ph5 = Clock();
      C.cyc_ph[0] += ph1 - ph0;
      C.cyc_ph[1] += ph2 - ph1;
      C.cyc_ph[2] += ph3 - ph2;
      C.cyc_ph[3] += ph4 - ph3;
      C.cyc_ph[4] += ph5 - ph4;
    }
    if (!__ballot(pending)) break;   // nothing left: skip the next round's root resolution
    WaveSync();
  }
  if (kDbg) C.dbg_kept += (unsigned)__popcll(__ballot(my_kept));
  return my_kept;
}

template <bool kDbg>
__global__ __launch_bounds__(128) void k_merge_wave(const int32_t* __restrict__ num_segs,
                                                    const int32_t* __restrict__ seg_off,
                                                    const int32_t* __restrict__ seg_cnt,
                                                    const int32_t* __restrict__ s_ra,
                                                    const int32_t* __restrict__ s_rb,
                                                    const uint32_t* __restrict__ s_gpos,
                                                    NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                    StageThr T, int optimistic,
                                                    int32_t* __restrict__ violation,
                                                    unsigned long long* __restrict__ stats,
                                                    int dbg_flags, int wave_min, int wave_max,
                                                    const uint32_t* __restrict__ work_list, int work_cap,
                                                    int32_t* __restrict__ work_ctl,
                                                    const uint32_t* __restrict__ s_seq,
                                                    int32_t* __restrict__ hub_mark,
                                                    int32_t* __restrict__ hub_excl) {
  __shared__ WaveTable tab;
  auto Clock = []() -> unsigned long long { return kDbg ? __builtin_readcyclecounter() : 0ull; };
  __shared__ WaveQueue queue;
  __shared__ WaveStage stage;
  __shared__ float4 chain_buf[64];   // the merging lanes of a chain, packed
  const int lane = threadIdx.x & 63;
  const bool producer = threadIdx.x >= 64;   // wave 1 reads ahead, wave 0 replays
  for (int s = threadIdx.x; s < kTabSize; s += 128) {
    tab.key[s] = -1;
    tab.res[s] = 0xffffffffu;
    tab.res2[s] = 0xffffffffu;
    tab.res3[s] = 0xffffffffu;
  }
  __syncthreads();
  const int nseg = *num_segs;
  WaveCounters C;
  unsigned dbg_batches = 0;
  unsigned long long dbg_taken = 0, dbg_live = 0;
  unsigned long long cyc_load = 0, cyc_loop = 0, cyc_wait = 0;
  unsigned long long edges_taken = 0;
  __shared__ int next_seg;
  // With a work list (filed by k_merge_small, three size classes, largest first) the workgroups
  // draw tickets; without one they stride over all segments.
  int total = 0, c0 = 0, c1 = 0;
  if (work_list) {
    c0 = work_ctl[0];
    c1 = work_ctl[1];
    total = c0 + c1 + work_ctl[2];
  }
  for (int it = blockIdx.x;; it += gridDim.x) {
    int seg;
    if (work_list) {
      if (threadIdx.x == 0) {
        const int ticket = atomicAdd(&work_ctl[kWaveClasses], 1);
        int s_ = -1;
        if (ticket < total) {
          const int cls = ticket < c0 ? 0 : (ticket < c0 + c1 ? 1 : 2);
          const int at = ticket - (cls == 0 ? 0 : (cls == 1 ? c0 : c0 + c1));
          s_ = (int)work_list[(size_t)cls * work_cap + at];
        }
        next_seg = s_;
      }
      __syncthreads();
      seg = next_seg;
      __syncthreads();   // everybody has read next_seg before the next ticket overwrites it
      if (seg < 0) break;
    } else {
      seg = it;
      if (seg >= nseg) break;
    }
    const int cnt = seg_cnt[seg];
    if (cnt <= wave_min || cnt >= wave_max) continue;
    const int beg = seg_off[seg];
    const int end = beg + cnt;
    // Edges whose two ends already share a region are dropped when they are read (a large share
    // of a component's edges once its regions have grown): the producer wave reads 256 edges per
    // fill with all root searches in flight together and pushes the surviving edges into an LDS
    // ring; the consumer wave replays them 64 at a time, so the fixed cost of a batch is spent on
    // pending edges only and the global-memory latency of the reads is off the replay's path.
    if (threadIdx.x == 0) {
      queue.produced = 0;
      queue.consumed = 0;
      queue.done = 0;
    }
    __syncthreads();
    if (producer) {
      int produced = 0;
      for (int next = beg; next < end; next += kFill * 64) {
        const unsigned long long pt0 = Clock();
        while (produced - __hip_atomic_load(&queue.consumed, __ATOMIC_ACQUIRE,
                                            __HIP_MEMORY_SCOPE_WORKGROUP) > kLag) {
          __builtin_amdgcn_s_sleep(2);
        }
        const unsigned long long pt1 = Clock();
        cyc_wait += pt1 - pt0;
        int xa[kFill], xb[kFill], ca[kFill], cb[kFill];
        uint32_t gp[kFill], sq[kFill];
        bool vd[kFill];
#pragma unroll
        for (int k = 0; k < kFill; ++k) {
          const int p = next + k * 64 + lane;
          vd[k] = p < end;
          xa[k] = vd[k] ? s_ra[p] : 0;
          xb[k] = vd[k] ? s_rb[p] : 0;
          gp[k] = vd[k] ? s_gpos[p] : 0u;
          sq[k] = (vd[k] && s_seq) ? s_seq[p] : 0u;
          ca[k] = xa[k];
          cb[k] = xb[k];
        }
        for (bool any = true; any;) {   // all root searches of the fill advance together
          int pa[kFill], pb[kFill];
#pragma unroll
          for (int k = 0; k < kFill; ++k) {
            pa[k] = nodes.parent[ca[k]];
            pb[k] = nodes.parent[cb[k]];
          }
          any = false;
#pragma unroll
          for (int k = 0; k < kFill; ++k) {
            if (pa[k] != ca[k]) { ca[k] = pa[k]; any = true; }
            if (pb[k] != cb[k]) { cb[k] = pb[k]; any = true; }
          }
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int k = 0; k < kFill; ++k) {
          // Path compression of the start node only (it is not a root, so the consumer never
          // writes it); an optimistic stage must stay undoable and does not compress.
          if (!optimistic && !(kDbg && (dbg_flags & 1024))) {
            if (vd[k] && ca[k] != xa[k]) nodes.parent[xa[k]] = ca[k];
            if (vd[k] && cb[k] != xb[k]) nodes.parent[xb[k]] = cb[k];
          }
          const bool pend = vd[k] && ca[k] != cb[k];
          const unsigned long long m = __ballot(pend);
          if (pend) {
            const int slot = (produced + (int)__popcll(m & lt)) & (kQueue - 1);
            queue.ra[slot] = ca[k];
            queue.rb[slot] = cb[k];
            queue.gpos[slot] = gp[k];
            queue.seq[slot] = sq[k];
          }
          produced += (int)__popcll(m);
        }
        WaveSync();
        if (lane == 0) {
          __hip_atomic_store(&queue.produced, produced, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        cyc_load += Clock() - pt1;
      }
      if (lane == 0) {
        __hip_atomic_store(&queue.done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();   // end of the segment (matches the consumer's)
      continue;
    }

    // ---- consumer --------------------------------------------------------------------------------
    edges_taken += (unsigned long long)cnt;   // (added to stats[3] once, at the end)
    const unsigned long long seg_t0 = Clock();
    const unsigned seg_cut0 = C.dbg_cut, seg_kept0 = C.dbg_kept;
    const unsigned long long seg_c0[12] = {dbg_batches, C.dbg_rounds, dbg_live, cyc_wait, cyc_load, cyc_loop,
                                           C.dbg_chain, C.dbg_nwin, C.cyc_ph[0], C.cyc_ph[1], C.cyc_ph[2] + C.cyc_ph[3],
                                           C.cyc_ph[4]};
    int consumed = 0;   // wave-uniform
    int n_raw = 0;      // staged edges left over from the previous batch: roots to be re-validated
    for (;;) {
      const unsigned long long bt0 = Clock();
      // ---- stage 64 live edges -----------------------------------------------------------------------
      // The roots the producer found may be stale by now (the ring holds several batches); a large
      // share of the ring's edges is internal by the time it is taken.  A pass re-validates up
      // to 64 candidates (the left-overs of the previous batch first, then ring entries) and packs
      // the live ones in order; passes repeat until 64 live edges are staged or the component is
      // drained, so that the fixed cost of a batch is spent on live edges only.
      int n_valid = 0;
      bool drained = false;
      while (n_valid < 64 && !drained) {
        const int want = 64 - n_raw;
        int avail;
        for (;;) {   // `done` is read before `produced`: once done is set, produced is final
          const int done = __hip_atomic_load(&queue.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
          avail = __hip_atomic_load(&queue.produced, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) -
                  consumed;
          if (avail >= want || done) {
            if (done && avail <= want) drained = true;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        const int t = avail < want ? avail : want;
        if (n_raw + t == 0) break;
        int ca = 0, cb = 0;
        uint32_t cg = 0, cq = 0;
        const bool cand = lane < n_raw + t;
        if (lane < n_raw) {
          ca = stage.ra[lane];
          cb = stage.rb[lane];
          cg = stage.gpos[lane];
          cq = stage.seq[lane];
        } else if (cand) {
          const int slot = (consumed + lane - n_raw) & (kQueue - 1);
          ca = queue.ra[slot];
          cb = queue.rb[slot];
          cg = queue.gpos[slot];
          cq = queue.seq[slot];
        }
        for (bool more = cand; more;) {
          const int pa = nodes.parent[ca], pb = nodes.parent[cb];
          more = (pa != ca) || (pb != cb);
          ca = pa;
          cb = pb;
        }
        consumed += t;
        WaveSync();   // the candidates are in registers: ring slots and stage slots may be reused
        if (lane == 0 && t > 0) {
          __hip_atomic_store(&queue.consumed, consumed, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const bool live = cand && ca != cb;
        const unsigned long long lm = __ballot(live);
        if (live) {
          const int pos = n_valid + (int)__popcll(lm & ((1ull << lane) - 1ull));
          stage.ra[pos] = ca;
          stage.rb[pos] = cb;
          stage.gpos[pos] = cg;
          stage.seq[pos] = cq;
        }
        if (kDbg && lane == 0) {
          dbg_taken += (unsigned)(n_raw + t);
          dbg_live += (unsigned)__popcll(lm);
        }
        n_valid += (int)__popcll(lm);
        n_raw = 0;
        WaveSync();
      }
      if (n_valid == 0) {
        if (drained) break;
        continue;
      }
      const unsigned long long bt0b = Clock();
      cyc_wait += bt0b - bt0;
      // ---- the batch: the first 64 staged edges (their roots are current) ------------------------
      const int take = n_valid < 64 ? n_valid : 64;
      const bool valid = lane < take;
      int ra = -1, rb = -1;
      uint32_t gpos = 0, seq = 0;
      if (valid) {
        ra = stage.ra[lane];
        rb = stage.rb[lane];
        gpos = stage.gpos[lane];
        seq = stage.seq[lane];
      }
      n_raw = n_valid - take;
      if (n_raw > 0) {   // move the rest to the front; it is re-validated by the next pass
        int xa = 0, xb = 0;
        uint32_t xg = 0, xq = 0;
        if (lane < n_raw) {
          xa = stage.ra[64 + lane];
          xb = stage.rb[64 + lane];
          xg = stage.gpos[64 + lane];
          xq = stage.seq[64 + lane];
        }
        WaveSync();
        if (lane < n_raw) {
          stage.ra[lane] = xa;
          stage.rb[lane] = xb;
          stage.gpos[lane] = xg;
          stage.seq[lane] = xq;
        }
      }
      WaveSync();
      const bool pending = valid;
      int sa = 0, sb = 0;     // table slots of the current roots of the two end regions
      int mine_a = -1, mine_b = -1;   // slots this lane inserted (it writes them back and frees them)
      if (pending) {
        const RState A = LoadStateHub(nodes, ra, T.hubs), B = LoadStateHub(nodes, rb, T.hubs);   // both in flight
        bool ins_a, ins_b;
        sa = TabInsert(tab, ra, ins_a);
        sb = TabInsert(tab, rb, ins_b);
        if (ins_a) {
          tab.link[sa] = sa;
          TabStore(tab, sa, A, 0);
        }
        if (ins_b) {
          tab.link[sb] = sb;
          TabStore(tab, sb, B, 0);
        }
        if (ins_a) mine_a = sa;
        if (ins_b) mine_b = sb;
      }
      WaveSync();
      if (kDbg && lane == 0) ++dbg_batches;
      const unsigned long long bt1 = Clock();
      cyc_load += bt1 - bt0b;

      const bool my_kept = ReplayRounds<kDbg>(tab, chain_buf, nodes, T, optimistic, violation, stats,
                                              dbg_flags, lane, valid, sa, sb, C, seq, hub_mark, hub_excl);
      WaveSync();

      if (valid && my_kept) kept_all[gpos] = 1;
      if (T.side) {   // (a side cluster with a kept edge: the earliest one of the batch is where the stage is cut)
        const unsigned long long km = __ballot(valid && my_kept);
        if (km && lane == (int)__builtin_ctzll(km)) {
          atomicOr(violation, 2);
          HubViolationAt(hub_excl, 2, (int)gpos);
        }
      }
      // ---- write the changed regions back, reset the table ---------------------------------------
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int s = e ? mine_b : mine_a;
        if (s >= 0) {
          if (tab.link[s] == s && (tab.flags[s] & kTabDirty)) {
            StoreState(nodes, tab.key[s], TabLoad(tab, s));
          }
          tab.key[s] = -1;
          tab.res[s] = 0xffffffffu;
          tab.res2[s] = 0xffffffffu;
          tab.res3[s] = 0xffffffffu;
        }
      }
      // Make this batch's stores visible to the next batch's loads (same CU: L1 is shared).
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      WaveSync();
      cyc_loop += Clock() - bt1;
    }
    if (kDbg && lane == 0) {
      atomicMax(&stats[16], Clock() - seg_t0);            // slowest component
      if (atomicMax(&stats[17], (unsigned long long)cnt) < (unsigned long long)cnt) {
        // counters of the largest component alone (the stage's critical path)
        const unsigned long long seg_c1[12] = {dbg_batches, C.dbg_rounds, dbg_live, cyc_wait, cyc_load,
                                               cyc_loop, C.dbg_chain, C.dbg_nwin, C.cyc_ph[0], C.cyc_ph[1],
                                               C.cyc_ph[2] + C.cyc_ph[3], C.cyc_ph[4]};
        stats[48] = Clock() - seg_t0;
        for (int k = 0; k < 12; ++k) stats[49 + k] = seg_c1[k] - seg_c0[k];
        stats[61] = C.dbg_cut - seg_cut0;
        stats[62] = C.dbg_kept - seg_kept0;
      }
    }
    __syncthreads();   // end of the segment (matches the producer's)
  }
  if (producer) {
    if (kDbg && lane == 0) {
      atomicAdd(&stats[27], cyc_load);   // producer: reading + root searches
      atomicAdd(&stats[28], cyc_wait);   // producer: ring full
    }
    return;
  }
  for (int off = 32; off > 0; off >>= 1) {
    C.n_forced += __shfl_down(C.n_forced, off);
    C.n_regular += __shfl_down(C.n_regular, off);
    C.n_small += __shfl_down(C.n_small, off);
  }
  if (lane == 0) {
    if (edges_taken) atomicAdd(&stats[3], edges_taken);
    if (C.n_forced) atomicAdd(&stats[0], (unsigned long long)C.n_forced);
    if (C.n_regular) atomicAdd(&stats[1], (unsigned long long)C.n_regular);
    if (C.n_small) atomicAdd(&stats[2], (unsigned long long)C.n_small);
  }
  if (kDbg && lane == 0) {
    atomicAdd(&stats[4], (unsigned long long)C.dbg_nwin);
    atomicAdd(&stats[5], (unsigned long long)C.dbg_rounds);
    atomicAdd(&stats[6], (unsigned long long)C.dbg_solo);
    atomicAdd(&stats[7], (unsigned long long)dbg_batches);
    atomicAdd(&stats[18], cyc_load);
    atomicAdd(&stats[26], cyc_wait);   // consumer: ring empty
    atomicAdd(&stats[19], cyc_loop);
    atomicAdd(&stats[20], (unsigned long long)C.dbg_chain);
    atomicAdd(&stats[21], (unsigned long long)C.dbg_cut);
    atomicAdd(&stats[29], dbg_taken);
    for (int k = 0; k < 5; ++k) atomicAdd(&stats[32 + k], C.cyc_ph[k]);
    for (int k = 0; k < 6; ++k) atomicAdd(&stats[38 + k], C.dbg_x[k]);
    for (int k = 0; k < 8; ++k) atomicAdd(&stats[64 + k], C.dbg_r[k]);
    atomicAdd(&stats[30], dbg_live);
  }
}

void LaunchMergeWave(int grid, const WorkerArgs& a, bool instrumented, int dbg_flags, hipStream_t s) {
  if (instrumented) {
    hipLaunchKernelGGL(k_merge_wave<true>, dim3(grid), dim3(128), 0, s, a.num_segs, a.seg_off,
                       a.seg_cnt, a.s_ra, a.s_rb, a.s_gpos, a.nodes, a.kept_all, a.T, a.optimistic,
                       a.violation, a.stats, dbg_flags, a.wave_min, a.wave_max, a.work_list, a.work_cap,
                       a.work_ctl, a.s_seq, a.hub_mark, a.hub_excl);
  } else {
    hipLaunchKernelGGL(k_merge_wave<false>, dim3(grid), dim3(128), 0, s, a.num_segs, a.seg_off,
                       a.seg_cnt, a.s_ra, a.s_rb, a.s_gpos, a.nodes, a.kept_all, a.T, a.optimistic,
                       a.violation, a.stats, 0, a.wave_min, a.wave_max, a.work_list, a.work_cap,
                       a.work_ctl, a.s_seq, a.hub_mark, a.hub_excl);
  }
  VSG_HIP(hipGetLastError());
}

}  // namespace vsg
