// merge_block.hip -- opt-in worker of the ordered merge: 256-edge batches replayed by a workgroup
// of four wavefronts.
#include "merge_common.h"

namespace vsg {

// ------------------------------------------------------------------------------------------
// Worker C: one WORKGROUP of four wavefronts replays one large component, 256 edges per batch.
// ------------------------------------------------------------------------------------------
// Same algorithm as k_merge_wave (reservations, transitive chain on the batch's hot region, see
// above), with a batch of 256 staged live edges spread over four wavefronts: the fixed cost of a
// batch and of a round is shared by four times the edges, four SIMDs issue the per-lane work, and
// the dependency chains of what would be neighbouring 64-edge batches overlap (tools/sched_sim.cpp:
// 1.9 instead of 4.25 rounds per 64 live edges on the 960x540 trace).  Wave-wide ballots become
// four 64-bit words exchanged through LDS, wave-local ordering becomes workgroup barriers, the
// chain recurrence runs on wavefront 0 with the per-lane inputs handed over through LDS.  There
// is no reader wave (it could not take part in a data-dependent number of barriers): the
// wavefronts read the component's edges themselves, two per thread and pass.
constexpr int kBlk = 256;
constexpr int kBTab = 1024;      // region table slots (<= 512 regions of a batch, load factor 1/2)
constexpr int kBStage = 768;     // staged live edges: < 256 left over + 512 of a pass

struct BlockShared {
  int32_t key[kBTab];
  int32_t link[kBTab];
  uint32_t res[kBTab];
  int32_t cnt[kBTab];
  float4 ds[kBTab];
  int32_t cons[kBTab];
  int32_t flags[kBTab];
  int32_t st_ra[kBStage];
  int32_t st_rb[kBStage];
  uint32_t st_gpos[kBStage];
  unsigned long long bm[4][4];   // ballot exchange (rotating slots)
  unsigned long long bmn[2][4][4];   // multi-predicate exchange (two alternating slots)
  float c_p0[kBlk], c_p1[kBlk], c_p2[kBlk];   // partner means by lane
  int32_t c_v[kBlk], c_incl[kBlk];            // partner size of merging lanes, wave-local prefix
  float c_r0[kBlk], c_r1[kBlk], c_r2[kBlk];               // hot mean before the lane's merge
  int32_t c_S[kBlk];                                      // hot size before the lane's merge
  float h_fin[4];
  int32_t wsum[4];
  int32_t rewind;
  int32_t fl_slot;   // slot of the first hot lane's other end (see the round loop)
};

struct Mask256 {
  unsigned long long w[4];
};
__device__ __forceinline__ bool MaskAny(const Mask256& m) { return (m.w[0] | m.w[1] | m.w[2] | m.w[3]) != 0; }
__device__ __forceinline__ bool MaskEq(const Mask256& a, const Mask256& b) {
  return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && a.w[3] == b.w[3];
}
__device__ __forceinline__ bool MaskBit(const Mask256& m, int i) {   // i in [0, 256)
  const int k = i >> 6;
  const unsigned long long w = k == 0 ? m.w[0] : k == 1 ? m.w[1] : k == 2 ? m.w[2] : m.w[3];
  return (w >> (i & 63)) & 1ull;
}
__device__ __forceinline__ int MaskFirst(const Mask256& m) {   // 256 if empty
  for (int k = 0; k < 4; ++k) {
    if (m.w[k]) return k * 64 + (int)__builtin_ctzll(m.w[k]);
  }
  return 256;
}
__device__ __forceinline__ int MaskCount(const Mask256& m) {
  return (int)(__popcll(m.w[0]) + __popcll(m.w[1]) + __popcll(m.w[2]) + __popcll(m.w[3]));
}
// number of set bits below position g
__device__ __forceinline__ int MaskRank(const Mask256& m, int g) {
  const int k = g >> 6;
  int r = 0;
  for (int j = 0; j < 4; ++j) {
    if (j < k) r += (int)__popcll(m.w[j]);
  }
  const unsigned long long w = k == 0 ? m.w[0] : k == 1 ? m.w[1] : k == 2 ? m.w[2] : m.w[3];
  return r + (int)__popcll(w & ((1ull << (g & 63)) - 1ull));
}

// Workgroup ballot: one barrier.  `slot` is a per-thread counter that advances identically in
// every thread; a slot is rewritten two calls later at the earliest, i.e. behind a barrier every
// wave can only reach after it has read the slot.
__device__ __forceinline__ Mask256 BlockBallot(BlockShared& sh, int& slot, bool pred) {
  const unsigned long long m = __ballot(pred);
  const int s = slot & 3;
  ++slot;
  if ((threadIdx.x & 63) == 0) sh.bm[s][threadIdx.x >> 6] = m;
  __syncthreads();
  Mask256 r;
  r.w[0] = sh.bm[s][0];
  r.w[1] = sh.bm[s][1];
  r.w[2] = sh.bm[s][2];
  r.w[3] = sh.bm[s][3];
  return r;
}

__device__ __forceinline__ int BTabInsert(BlockShared& t, int r, bool& inserted) {
  unsigned h = ((unsigned)r * 2654435761u) >> 22;
  for (;;) {
    const int old = atomicCAS(&t.key[h], -1, r);
    if (old == -1) { inserted = true; return (int)h; }
    if (old == r) { inserted = false; return (int)h; }
    h = (h + 1) & (kBTab - 1);
  }
}
__device__ __forceinline__ RState BTabLoad(const BlockShared& t, int s) {
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
__device__ __forceinline__ void BTabStore(BlockShared& t, int s, const RState& r, int dirty) {
  t.ds[s] = make_float4(r.d0, r.d1, r.d2, __int_as_float(r.sz));
  t.cons[s] = r.cons;
  t.flags[s] = r.flags | dirty;
}
__device__ __forceinline__ void BCommitLoser(BlockShared& t, const NodeArrays& nodes, int ls, int ws) {
  t.link[ls] = ws;
  const int lid = t.key[ls];
  nodes.parent[lid] = t.key[ws];
  if (t.flags[ls] & kTabDirty) nodes.cons[lid] = t.cons[ls];   // see CommitLoser
}

// Workgroup ballot of up to four predicates with ONE barrier.
template <int N>
__device__ __forceinline__ void BlockBallotN(BlockShared& sh, int& slot, const bool (&pred)[N],
                                             Mask256 (&out)[N]) {
  const int s = slot & 1;
  ++slot;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned long long m = __ballot(pred[i]);
    if ((threadIdx.x & 63) == 0) sh.bmn[s][i][threadIdx.x >> 6] = m;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[i].w[k] = sh.bmn[s][i][k];
  }
}

// One staging pass of the block worker: R candidates per thread (the left-overs of the previous
// batch first, then edges read from the component's list), all root searches in flight together;
// the live candidates are packed in order behind the n_valid edges already staged.  When the stage
// is full the pass is cut there and `next` is rewound to the first edge that did not fit.
// Returns the fraction of live candidates in 1/256 units (the caller picks R for the next pass).
template <int R>
__device__ __forceinline__ int StagePass(BlockShared& sh, int& slot, const NodeArrays& nodes,
                                         const int32_t* __restrict__ s_ra,
                                         const int32_t* __restrict__ s_rb,
                                         const uint32_t* __restrict__ s_gpos, int end, int optimistic,
                                         int& next, int& n_raw, int& n_valid) {
  const int g = threadIdx.x;
  const int t_new = min(R * kBlk - n_raw, end - next);
  const int total = n_raw + t_new;
  const int next_base = next;
  int ca[R], cb[R], xa[R], xb[R];
  uint32_t cg[R];
  bool cand[R], fresh[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int i = g + k * kBlk;
    cand[k] = i < total;
    fresh[k] = cand[k] && i >= n_raw;
    ca[k] = cb[k] = 0;
    cg[k] = 0;
    if (cand[k] && !fresh[k]) {
      ca[k] = sh.st_ra[i];
      cb[k] = sh.st_rb[i];
      cg[k] = sh.st_gpos[i];
    } else if (fresh[k]) {
      const int p = next_base + (i - n_raw);
      ca[k] = s_ra[p];
      cb[k] = s_rb[p];
      cg[k] = s_gpos[p];
    }
    xa[k] = ca[k];
    xb[k] = cb[k];
  }
  for (bool more = true; more;) {   // the 2 R root searches of a thread advance together
    int pa[R], pb[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      pa[k] = nodes.parent[ca[k]];
      pb[k] = nodes.parent[cb[k]];
    }
    more = false;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      more = more || (pa[k] != ca[k]) || (pb[k] != cb[k]);
      ca[k] = pa[k];
      cb[k] = pb[k];
    }
  }
  if (!optimistic) {   // path compression of the start nodes (never representatives)
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (fresh[k] && ca[k] != xa[k]) nodes.parent[xa[k]] = ca[k];
      if (fresh[k] && cb[k] != xb[k]) nodes.parent[xb[k]] = cb[k];
    }
  }
  if (g == 0) sh.rewind = -1;
  int base = n_valid, live_total = 0;
#pragma unroll
  for (int k0 = 0; k0 < R; k0 += 2) {
    Mask256 lm[2];
    const bool pr[2] = {cand[k0] && ca[k0] != cb[k0], cand[k0 + 1] && ca[k0 + 1] != cb[k0 + 1]};
    BlockBallotN<2>(sh, slot, pr, lm);   // first barrier: every candidate is in registers
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + j;
      if (pr[j]) {
        const int pos = base + MaskRank(lm[j], g);
        if (pos < kBStage) {
          sh.st_ra[pos] = ca[k];
          sh.st_rb[pos] = cb[k];
          sh.st_gpos[pos] = cg[k];
        } else if (pos == kBStage) {
          // the first edge that does not fit (a fresh one: the left-overs always fit)
          sh.rewind = next_base + (g + k * kBlk - n_raw);
        }
      }
      const int c = MaskCount(lm[j]);
      base += c;
      live_total += c;
    }
  }
  __syncthreads();
  const int rw = sh.rewind;
  next = rw >= 0 ? rw : next_base + t_new;
  n_valid = base < kBStage ? base : kBStage;
  n_raw = 0;
  __syncthreads();   // sh.rewind is rewritten by the next pass
  return total > 0 ? (live_total * 256) / total : 256;
}

__global__ __launch_bounds__(256) void k_merge_block(const int32_t* __restrict__ num_segs,
                                                      const int32_t* __restrict__ seg_off,
                                                      const int32_t* __restrict__ seg_cnt,
                                                      const int32_t* __restrict__ s_ra,
                                                      const int32_t* __restrict__ s_rb,
                                                      const uint32_t* __restrict__ s_gpos,
                                                      NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                      StageThr T, int optimistic,
                                                      int32_t* __restrict__ violation,
                                                      unsigned long long* __restrict__ stats,
                                                      int dbg_flags) {
  __shared__ BlockShared sh;
  const int g = threadIdx.x;          // lane of the batch
  const int lane = g & 63, wave = g >> 6;
  for (int s = g; s < kBTab; s += kBlk) {
    sh.key[s] = -1;
    sh.res[s] = 0xffffffffu;
    sh.cnt[s] = 0;
  }
  __syncthreads();
  int slot = 0;   // BlockBallot slot counter
  const int nseg = *num_segs;
  unsigned n_forced = 0, n_regular = 0, n_small = 0;
  unsigned long long dbg_rounds = 0, dbg_batches = 0, dbg_generic = 0, dbg_chain = 0;
  unsigned long long cyc[6] = {0, 0, 0, 0, 0, 0};   // staging, table, round head, generic, chain, write-back
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int cnt = seg_cnt[seg];
    if (cnt <= kSmallSegment) continue;
    const int beg = seg_off[seg];
    const int end = beg + cnt;
    if (g == 0) atomicAdd(&stats[3], (unsigned long long)cnt);
    int next = beg;     // next edge of the component to be read (uniform)
    int live_frac = 256;   // live candidates of the last staging pass, in 1/256
    int n_raw = 0;      // staged edges left over from the previous batch (roots to be re-validated)
    for (;;) {
      // ---- stage up to 256 live edges ---------------------------------------------------------------
      const unsigned long long c0 = __builtin_readcyclecounter();
      int n_valid = 0;
      while (n_valid < kBlk && (n_raw > 0 || next < end)) {
        // eight candidates per thread when few of them are live (most of a grown component's
        // edges are internal), two otherwise
        live_frac = (live_frac < 77 && n_raw < kBlk)
                        ? StagePass<8>(sh, slot, nodes, s_ra, s_rb, s_gpos, end, optimistic, next, n_raw, n_valid)
                        : StagePass<2>(sh, slot, nodes, s_ra, s_rb, s_gpos, end, optimistic, next, n_raw, n_valid);
      }
      if (n_valid == 0) break;   // the component is drained
      const unsigned long long c1 = __builtin_readcyclecounter();
      cyc[0] += c1 - c0;
      // ---- the batch: the first 256 staged edges (their roots are current) ------------------------
      const int take = n_valid < kBlk ? n_valid : kBlk;
      const bool valid = g < take;
      int ra = -1, rb = -1;
      uint32_t gpos = 0;
      if (valid) {
        ra = sh.st_ra[g];
        rb = sh.st_rb[g];
        gpos = sh.st_gpos[g];
      }
      n_raw = n_valid - take;
      {   // move the rest to the front; it is re-validated by the next pass
        int xa[2] = {0, 0}, xb[2] = {0, 0};
        uint32_t xg[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = g + k * kBlk;
          if (i < n_raw) {
            xa[k] = sh.st_ra[kBlk + i];
            xb[k] = sh.st_rb[kBlk + i];
            xg[k] = sh.st_gpos[kBlk + i];
          }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = g + k * kBlk;
          if (i < n_raw) {
            sh.st_ra[i] = xa[k];
            sh.st_rb[i] = xb[k];
            sh.st_gpos[i] = xg[k];
          }
        }
      }
      bool pending = valid;
      int sa = 0, sb = 0;
      int mine_a = -1, mine_b = -1;
      if (pending) {
        const RState A0 = LoadState(nodes, ra), B0 = LoadState(nodes, rb);
        bool ins_a, ins_b;
        sa = BTabInsert(sh, ra, ins_a);
        sb = BTabInsert(sh, rb, ins_b);
        if (ins_a) {
          sh.link[sa] = sa;
          BTabStore(sh, sa, A0, 0);
          mine_a = sa;
        }
        if (ins_b) {
          sh.link[sb] = sb;
          BTabStore(sh, sb, B0, 0);
          mine_b = sb;
        }
        atomicAdd(&sh.cnt[sa], 1);
        atomicAdd(&sh.cnt[sb], 1);
      }
      __syncthreads();
      int hot = -1;   // block-uniform slot of the hot region
      {
        int best = 0;
        if (pending) best = max((sh.cnt[sa] << 10) | sa, (sh.cnt[sb] << 10) | sb);
        best = WaveMax(best);
        if (lane == 0) sh.wsum[wave] = best;
        __syncthreads();
        best = max(max(sh.wsum[0], sh.wsum[1]), max(sh.wsum[2], sh.wsum[3]));
        __syncthreads();
        if ((best >> 10) >= 3 && !(dbg_flags & 4)) hot = best & (kBTab - 1);
      }
      if (g == 0) ++dbg_batches;
      cyc[1] += __builtin_readcyclecounter() - c1;

      bool my_kept = false;
      bool failed = false;
      for (unsigned round = 0;; ++round) {
        const unsigned long long r0c = __builtin_readcyclecounter();
        {   // current root slots
          int h = hot;
          for (bool more = true; more;) {
            int pa = sa, pb = sb, ph = h;
            if (pending) {
              pa = sh.link[sa];
              pb = sh.link[sb];
            }
            if (h >= 0) ph = sh.link[h];
            more = (pa != sa) || (pb != sb) || (ph != h);
            sa = pa;
            sb = pb;
            h = ph;
          }
          hot = h;
          if (pending && sa == sb) pending = false;   // became internal
        }
        const bool a_hot = (sa == hot), b_hot = (sb == hot);
        bool hot_lane = pending && (a_hot || b_hot);
        // one exchange: who is still pending (loop exit) and which lanes touch the hot region
        Mask256 pl[2];
        {
          const bool pr[2] = {pending, hot_lane};
          BlockBallotN<2>(sh, slot, pr, pl);
        }
        if (!MaskAny(pl[0])) break;
        if (round > 600u) {   // cannot happen (the earliest pending lane commits): report
          if (g == 0) atomicAdd(&stats[22], 1ull);
          break;
        }
        const Mask256& lit = pl[1];
        const int first_lit = MaskFirst(lit);
        // A chain starts at the first edge that touches the hot region and only if that lane is the
        // earliest pending edge on its other end: it publishes that end's slot, everybody checks
        // the reservation word after the barrier.
        if (g == first_lit) sh.fl_slot = a_hot ? sb : sa;
        const uint32_t key = ((0xfffffu - round) << 8) | (uint32_t)g;
        if (pending) {
          if (!a_hot) atomicMin(&sh.res[sa], key);
          if (!b_hot) atomicMin(&sh.res[sb], key);
        }
        __syncthreads();
        uint32_t res_a = 0, res_b = 0;
        RState A = {}, B = {};
        if (pending) {
          res_a = sh.res[sa];
          res_b = sh.res[sb];
          A = BTabLoad(sh, sa);
          B = BTabLoad(sh, sb);
        }
        const bool own_a = pending && !a_hot && res_a == key;
        const bool own_b = pending && !b_hot && res_b == key;
        const int oa = (int)(res_a & 255u), ob = (int)(res_b & 255u);
        RState Hs = {}, P = {};
        int ps = 0;
        bool elig = false, both = false, merging = false, case_s = false, fin = false;
        const bool chain_possible =
            first_lit < 256 && !(dbg_flags & 1) &&
            sh.res[sh.fl_slot] == (((0xfffffu - round) << 8) | (uint32_t)first_lit);
        if (chain_possible) {
          Hs = BTabLoad(sh, hot);
          fin = (Hs.flags & kFlagFinalized) != 0;
          const bool mode_ok = !(Hs.flags & kFlagNoDesc) && (!fin || Hs.sz >= T.min_size);
          const bool base = pending && mode_ok && !failed;
          const bool part_a = base && own_a && A.flags == 0 && (A.cons < 0 || A.cons == Hs.cons) &&
                              A.sz < Hs.sz;
          const bool part_b = base && own_b && B.flags == 0 && (B.cons < 0 || B.cons == Hs.cons) &&
                              B.sz < Hs.sz;
          const bool merge_a = part_a && (A.cons >= 0 || !fin || A.sz < T.min_size);
          const bool merge_b = part_b && (B.cons >= 0 || !fin || B.sz < T.min_size);
          const bool abs_a = pending && !own_a, abs_b = pending && !own_b;
          Mask256 em = {{0, 0, 0, 0}};
          bool ea, eb;
          for (;;) {
            ea = a_hot || (abs_a && MaskBit(em, oa));
            eb = b_hot || (abs_b && MaskBit(em, ob));
            const Mask256 em2 =
                BlockBallot(sh, slot, (ea && !eb && merge_b) || (eb && !ea && merge_a));
            if (MaskEq(em2, em)) break;
            em = em2;
          }
          hot_lane = pending && (ea || eb);
          both = hot_lane && ea && eb;
          const bool pb_side = ea;
          P.d0 = pb_side ? B.d0 : A.d0;
          P.d1 = pb_side ? B.d1 : A.d1;
          P.d2 = pb_side ? B.d2 : A.d2;
          P.sz = pb_side ? B.sz : A.sz;
          P.cons = pb_side ? B.cons : A.cons;
          P.flags = 0;
          ps = pb_side ? sb : sa;
          elig = hot_lane && !both && (pb_side ? part_b : part_a);
          merging = hot_lane && !both && (pb_side ? merge_b : merge_a);
          case_s = P.cons >= 0;
        }
        // Without a chain the hot lanes are the literal ones and all of them wait.
        Mask256 hot_mask = lit, elig_mask = {{0, 0, 0, 0}}, mm_all = {{0, 0, 0, 0}};
        int cut = first_lit;   // first hot lane that ends the chain
        if (chain_possible) {
          Mask256 lb[4];
          const bool pr[4] = {hot_lane, elig, both, merging};
          BlockBallotN<4>(sh, slot, pr, lb);
          hot_mask = lb[0];
          elig_mask = lb[1];
          mm_all = lb[3];
          Mask256 blocked;
          for (int k = 0; k < 4; ++k) blocked.w[k] = lb[0].w[k] & ~(lb[1].w[k] | lb[2].w[k]);
          cut = MaskFirst(blocked);
        }
        const bool in_chain = elig && g < cut;
        const int first_hot = MaskFirst(hot_mask);
        const bool own = pending && (a_hot || own_a) && (b_hot || own_b);
        const bool solo = hot_lane && own && !elig && !both && g == first_hot;
        bool n_win = pending && own && (!hot_lane || solo);
        if (dbg_flags & 8) {
          const Mask256 pm = BlockBallot(sh, slot, pending);
          n_win = n_win && g == MaskFirst(pm);
        }
        if (g == 0) ++dbg_rounds;

        const unsigned long long r1c = __builtin_readcyclecounter();
        cyc[2] += r1c - r0c;
        // ---- lanes that own both regions: generic edge ------------------------------------------
        if (n_win) {
          if (A.cons < 0 && B.cons < 0 && (A.flags | B.flags) == 0 &&
              SquaredDistance(A, B) <= T.pass_s) {
            const bool first = A.sz > B.sz;   // ties keep region 2
            const int ws = first ? sa : sb, ls = first ? sb : sa;
            RState m, o;
            m.d0 = first ? A.d0 : B.d0;
            m.d1 = first ? A.d1 : B.d1;
            m.d2 = first ? A.d2 : B.d2;
            m.sz = first ? A.sz : B.sz;
            o.d0 = first ? B.d0 : A.d0;
            o.d1 = first ? B.d1 : A.d1;
            o.d2 = first ? B.d2 : A.d2;
            o.sz = first ? B.sz : A.sz;
            const float denom = 1.0f / (float)(o.sz + m.sz);
            const float ca = (float)o.sz * denom;
            const float cb = (float)m.sz * denom;
            m.d0 = ca * o.d0 + cb * m.d0;
            m.d1 = ca * o.d1 + cb * m.d1;
            m.d2 = ca * o.d2 + cb * m.d2;
            m.sz += o.sz;
            m.cons = max(A.cons, B.cons);
            m.flags = 0;
            BTabStore(sh, ws, m, kTabDirty);
            BCommitLoser(sh, nodes, ls, ws);
            ++n_regular;
            ++dbg_generic;
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
            if (v) *violation = 1;
          }
          n_forced += (stat == 1);
          n_regular += (stat == 2);
          n_small += (stat == 3);
          ++dbg_generic;
          if (out == kOutKeep) {
            my_kept = true;
            if (!SameState(o1, s1)) BTabStore(sh, sa, s1, kTabDirty);
            if (!SameState(o2, s2)) BTabStore(sh, sb, s2, kTabDirty);
          } else if (out == kOutMerge1) {
            BTabStore(sh, sa, s1, kTabDirty);
            BCommitLoser(sh, nodes, sb, sa);
          } else {
            BTabStore(sh, sb, s2, kTabDirty);
            BCommitLoser(sh, nodes, sa, sb);
          }
          pending = false;
        }

        // ---- the chain on the hot region -----------------------------------------------------
        const unsigned long long r2c = __builtin_readcyclecounter();
        cyc[3] += r2c - r1c;
        // chain lanes = candidates below the cut (no exchange needed: both masks are known)
        Mask256 chain_mask, mm;
        for (int k = 0; k < 4; ++k) {
          const int lo = k * 64;
          const unsigned long long below_cut =
              cut >= lo + 64 ? ~0ull : (cut <= lo ? 0ull : ((1ull << (cut - lo)) - 1ull));
          chain_mask.w[k] = elig_mask.w[k] & below_cut;
          mm.w[k] = mm_all.w[k] & chain_mask.w[k];
        }
        if (MaskAny(chain_mask)) {
          merging = in_chain && merging;
          case_s = in_chain && case_s;
          const bool tested = case_s || (in_chain && !fin);
          // hand the per-lane inputs to wavefront 0: partner size and mean, wave-local size prefix
          const int v = merging ? P.sz : 0;
          const int incl = WaveInclusiveSum(v);
          sh.c_v[g] = v;
          sh.c_incl[g] = incl;
          sh.c_p0[g] = P.d0;
          sh.c_p1[g] = P.d1;
          sh.c_p2[g] = P.d2;
          if (lane == 63) sh.wsum[wave] = incl;
          __syncthreads();
          const int wtot = sh.wsum[0] + sh.wsum[1] + sh.wsum[2] + sh.wsum[3];
          if (wave == 0) {
            // MergeStates with o = partner, m = hot region: sizes are a prefix sum, the weights and
            // ca*p are lane-parallel, only h = ca*p + cb*h is replayed in lane order; the mean and
            // the size of the hot region before each merge are recorded for the verification.
            float h0 = Hs.d0, h1 = Hs.d1, h2 = Hs.d2;
            int woff = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int e = k * 64 + lane;
              const int ev = sh.c_v[e];
              const int S = Hs.sz + woff + sh.c_incl[e] - ev;   // hot size before this lane's merge
              const float denom = 1.0f / (float)(ev + S);
              const float ca = (float)ev * denom;
              const float x_cb = (float)S * denom;
              const float x_t0 = ca * sh.c_p0[e], x_t1 = ca * sh.c_p1[e], x_t2 = ca * sh.c_p2[e];
              sh.c_S[e] = S;
              float r0 = 0.f, r1 = 0.f, r2 = 0.f;
              for (unsigned long long w = mm.w[k]; w; w &= w - 1) {
                const int j = (int)__builtin_ctzll(w);
                if (lane == j) {
                  r0 = h0;
                  r1 = h1;
                  r2 = h2;
                }
                const float cbj = ReadLaneF(x_cb, j);
                h0 = ReadLaneF(x_t0, j) + cbj * h0;
                h1 = ReadLaneF(x_t1, j) + cbj * h1;
                h2 = ReadLaneF(x_t2, j) + cbj * h2;
              }
              sh.c_r0[e] = r0;
              sh.c_r1[e] = r1;
              sh.c_r2[e] = r2;
              woff += sh.wsum[k];
            }
            if (lane == 0) {
              sh.h_fin[0] = h0;
              sh.h_fin[1] = h1;
              sh.h_fin[2] = h2;
            }
          }
          __syncthreads();
          bool pass;
          {
            const float x = sh.c_r0[g] - P.d0, y = sh.c_r1[g] - P.d1, z = sh.c_r2[g] - P.d2;
            const float sd = (x * x + y * y + z * z) * (1.0f / 3.0f);
            pass = case_s ? !(sd > T.split_s) : (sd <= T.pass_s);
          }
          const Mask256 fail = BlockBallot(sh, slot, tested && !pass);
          const int fcut = MaskFirst(fail);
          if (g == fcut) failed = true;
          RState Hn = Hs;
          if (fcut < 256) {
            Hn.d0 = sh.c_r0[fcut];
            Hn.d1 = sh.c_r1[fcut];
            Hn.d2 = sh.c_r2[fcut];
            Hn.sz = sh.c_S[fcut];
          } else {
            Hn.d0 = sh.h_fin[0];
            Hn.d1 = sh.h_fin[1];
            Hn.d2 = sh.h_fin[2];
            Hn.sz = Hs.sz + wtot;
          }
          const bool do_commit = in_chain && g < fcut;
          if (do_commit) {
            if (merging) {
              BCommitLoser(sh, nodes, ps, hot);
              if (case_s) ++n_forced; else if (fin) ++n_small; else ++n_regular;
              ++dbg_chain;
            } else {
              my_kept = true;   // both regions large, the hot one finalized: nothing changes
            }
            pending = false;
          }
          // both ends (by then) inside the hot region: internal once the chain below it is committed
          if (both && g < fcut && g < cut) pending = false;
          bool any_merge = false;   // a merging lane below the failed test, if any
          for (int k = 0; k < 4; ++k) {
            const int lo = k * 64;
            const unsigned long long below_f =
                fcut >= lo + 64 ? ~0ull : (fcut <= lo ? 0ull : ((1ull << (fcut - lo)) - 1ull));
            any_merge = any_merge || (mm.w[k] & below_f) != 0;
          }
          if (g == 0 && any_merge) BTabStore(sh, hot, Hn, kTabDirty);
        }
        __syncthreads();
        cyc[4] += __builtin_readcyclecounter() - r2c;
      }
      const unsigned long long wb0 = __builtin_readcyclecounter();

      if (valid && my_kept) kept_all[gpos] = 1;
      // ---- write the changed regions back, free the table slots ---------------------------------------
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int s = e ? mine_b : mine_a;
        if (s >= 0) {
          if (sh.link[s] == s && (sh.flags[s] & kTabDirty)) StoreState(nodes, sh.key[s], BTabLoad(sh, s));
          sh.key[s] = -1;
          sh.res[s] = 0xffffffffu;
          sh.cnt[s] = 0;
        }
      }
      // Make this batch's stores visible to the next batch's loads (same CU: L1 is shared).
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __syncthreads();
      cyc[5] += __builtin_readcyclecounter() - wb0;
    }
    __syncthreads();   // end of the component
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_forced += __shfl_down(n_forced, off);
    n_regular += __shfl_down(n_regular, off);
    n_small += __shfl_down(n_small, off);
    dbg_generic += __shfl_down(dbg_generic, off);
    dbg_chain += __shfl_down(dbg_chain, off);
  }
  if (lane == 0) {
    if (n_forced) atomicAdd(&stats[0], (unsigned long long)n_forced);
    if (n_regular) atomicAdd(&stats[1], (unsigned long long)n_regular);
    if (n_small) atomicAdd(&stats[2], (unsigned long long)n_small);
    atomicAdd(&stats[4], dbg_generic);
    atomicAdd(&stats[20], dbg_chain);
  }
  if (g == 0) {
    atomicAdd(&stats[5], dbg_rounds);
    atomicAdd(&stats[7], dbg_batches);
    atomicAdd(&stats[18], cyc[0]);
    atomicAdd(&stats[26], cyc[1]);
    for (int k = 0; k < 4; ++k) atomicAdd(&stats[32 + k], cyc[2 + k]);
  }
}

void LaunchMergeBlock(int grid, const WorkerArgs& a, int dbg_flags, hipStream_t s) {
  hipLaunchKernelGGL(k_merge_block, dim3(grid), dim3(256), 0, s, a.num_segs, a.seg_off, a.seg_cnt,
                     a.s_ra, a.s_rb, a.s_gpos, a.nodes, a.kept_all, a.T, a.optimistic, a.violation,
                     a.stats, dbg_flags);
  VSG_HIP(hipGetLastError());
}

}  // namespace vsg
