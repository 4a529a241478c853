// merge_wave_v1.hip -- edge-by-edge replay of a large component by one wavefront (the worker of
// round 1a).  Kept as the simple reference implementation of the worker (VSG_WAVE_V1=1, tested
// against the oracle like the others); see merge_stage.hip for the stage it is part of.
#include "merge_common.h"

namespace vsg {

// ------------------------------------------------------------------------------------------
// Worker B: one wavefront replays one large component, 64 edges per batch.
// ------------------------------------------------------------------------------------------
// The common pattern inside a large component is a chain: one big region absorbs neighbour after
// neighbour.  The winner of the last merge is therefore kept as the "hot" region: its state lives
// in (wave-uniform) registers, is used instead of any lane's cached copy, and is written back
// only when another region becomes hot or the component is finished.  Lane copies of a region are
// refreshed (12 v_cndmask) only when it stops being hot or on the rare flag/constraint change, so
// a chain step costs two id readlanes, six state readlanes, the decision and one parent store.
__global__ __launch_bounds__(64) void k_merge_wave_v1(const int32_t* __restrict__ num_segs,
                                                       const int32_t* __restrict__ seg_off,
                                                       const int32_t* __restrict__ seg_cnt,
                                                       const int32_t* __restrict__ s_ra,
                                                       const int32_t* __restrict__ s_rb,
                                                       const uint32_t* __restrict__ s_gpos,
                                                    NodeArrays nodes, uint8_t* __restrict__ kept_all,
                                                    StageThr T, int optimistic,
                                                    int32_t* __restrict__ violation,
                                                    unsigned long long* __restrict__ stats) {
  const int lane = threadIdx.x;
  const int nseg = *num_segs;
  unsigned n_forced = 0, n_regular = 0, n_small = 0;   // counted on lane 0 (uniform decisions)
  unsigned dbg_iters = 0, dbg_hot = 0, dbg_internal = 0, dbg_batches = 0, dbg_chain = 0;
  unsigned long long cyc_load = 0, cyc_loop = 0;
  unsigned dbg_g[6] = {0, 0, 0, 0, 0, 0};
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int cnt = seg_cnt[seg];
    if (cnt <= kSmallSegment) continue;
    const int beg = seg_off[seg];
    const int end = beg + cnt;
    if (lane == 0) atomicAdd(&stats[3], (unsigned long long)cnt);
    const unsigned long long seg_t0 = __builtin_readcyclecounter();
    int hot = -1;          // wave-uniform
    RState H = {};         // wave-uniform state of region `hot` (authoritative while hot >= 0)
    for (int base = beg; base < end; base += 64) {
      const int p = base + lane;
      const bool valid = p < end;
      int ra = -1, rb = -2;
      uint32_t gpos = 0;
      RState A = {}, B = {};
      const unsigned long long bt0 = __builtin_readcyclecounter();
      if (valid) {
        ra = optimistic ? FindReadOnly(nodes.parent, s_ra[p]) : FindCompress(nodes.parent, s_ra[p]);
        rb = optimistic ? FindReadOnly(nodes.parent, s_rb[p]) : FindCompress(nodes.parent, s_rb[p]);
        gpos = s_gpos[p];
        if (ra != rb) {
          A = LoadState(nodes, ra);
          B = LoadState(nodes, rb);
        }
      }
      bool my_kept = false;
      unsigned long long pending = __ballot(valid && ra != rb);
      // Per-lane flags: the cached region is a plain partner (unconstrained, not finalized, has a
      // descriptor, not tentatively marked).
      bool ok_a = (A.cons < 0 && A.flags == 0);
      bool ok_b = (B.cons < 0 && B.flags == 0);
      if (lane == 0) ++dbg_batches;
      const unsigned long long bt1 = __builtin_readcyclecounter();
      cyc_load += bt1 - bt0;
      while (pending) {
        // ---- tight chain loop: the hot region absorbs plain partners smaller than itself ------
        // Every lane keeps a "partner view" relative to the hot region (which of its two regions
        // is the partner, the partner's state), re-derived with a dozen VALU ops after each
        // merge, so that one chain step is: 5 readlanes, the test, the mean update, one parent
        // store and the id rename -- the same arithmetic as DecideEdge/MergeStates for this case
        // (Case U; the hot region keeps its flags and constraint; the partner carries no mark).
        if (hot >= 0 && !(H.flags & kFlagNoDesc)) {
          for (;;) {
            const bool e1 = (ra == hot), e2 = (rb == hot);
            const unsigned long long internal = __ballot(ra == rb);
            const unsigned long long chain = __ballot((e1 != e2) && (e1 ? ok_b : ok_a));
            // drop leading pending lanes that became internal
            while (pending && ((internal >> __builtin_ctzll(pending)) & 1ull)) {
              pending &= pending - 1;
              if (lane == 0) ++dbg_internal;
            }
            if (!pending) break;
            const int j = __builtin_ctzll(pending);
            if (!((chain >> j) & 1ull)) break;
            const float px0 = e1 ? B.d0 : A.d0;
            const float px1 = e1 ? B.d1 : A.d1;
            const float px2 = e1 ? B.d2 : A.d2;
            const int pszv = e1 ? B.sz : A.sz;
            const int pidv = e1 ? rb : ra;
            const int psz = ReadLaneI(pszv, j);
            if (!(H.sz > psz)) break;
            const float p0 = ReadLaneF(px0, j), p1 = ReadLaneF(px1, j), p2 = ReadLaneF(px2, j);
            bool merge;
            if (!(H.flags & kFlagFinalized)) {
              const float x = H.d0 - p0, y = H.d1 - p1, z = H.d2 - p2;
              if (!((x * x + y * y + z * z) * (1.0f / 3.0f) <= T.pass_s)) break;   // -> generic
              merge = true;
              ++n_regular;
            } else {
              merge = (psz < T.min_size || H.sz < T.min_size);
              n_small += merge;
            }
            pending &= pending - 1;
            if (lane == 0) { ++dbg_iters; ++dbg_hot; ++dbg_chain; }
            if (merge) {
              const float denom = 1.0f / (float)(psz + H.sz);
              const float ca = (float)psz * denom;
              const float cb = (float)H.sz * denom;
              H.d0 = ca * p0 + cb * H.d0;
              H.d1 = ca * p1 + cb * H.d1;
              H.d2 = ca * p2 + cb * H.d2;
              H.sz += psz;
              const int pid = ReadLaneI(pidv, j);
              if (lane == j) nodes.parent[pid] = hot;
              if (ra == pid) ra = hot;
              if (rb == pid) rb = hot;
            } else if (lane == j) {
              my_kept = true;   // both regions finalized / large: kept, nothing changes
            }
          }
          if (!pending) break;
        }
        const int j = __builtin_ctzll(pending);
        pending &= pending - 1;
        const int r1 = ReadLaneI(ra, j);
        const int r2 = ReadLaneI(rb, j);
        if (r1 == r2) { if (lane == 0) ++dbg_internal; continue; }   // became internal
        if (lane == 0) { ++dbg_iters; dbg_hot += (r1 == hot || r2 == hot); }
        RState s1, s2;
        if (r1 == hot) s1 = H; else s1 = ReadLaneState(A, j);
        if (r2 == hot) s2 = H; else s2 = ReadLaneState(B, j);
        if (lane == 0) {   // debug classification of the generic iterations
          const bool h1 = (r1 == hot), h2 = (r2 == hot);
          const bool pl1 = (s1.cons < 0 && s1.flags == 0), pl2 = (s2.cons < 0 && s2.flags == 0);
          if (!h1 && !h2) {
            ++dbg_g[0];
            if (pl1 && pl2) ++dbg_g[4];
            if (s1.sz == 1 && s2.sz == 1) ++dbg_g[5];
          } else {
            const bool ppl = h1 ? pl2 : pl1;
            const int psz = h1 ? s2.sz : s1.sz;
            if (!ppl) ++dbg_g[1];
            else if (!(H.sz > psz)) ++dbg_g[2];
            else ++dbg_g[3];
          }
        }
        const RState o1 = s1, o2 = s2;
        int stat;
        const int out = DecideEdge(s1, s2, T, stat);
        if (optimistic) {
          const bool v = (out == kOutKeep)     ? TentativeViolated(o1, o2, s1, s2)
                         : (out == kOutMerge1) ? TentativeViolated(o1, o2, s1, s1)
                                               : TentativeViolated(o1, o2, s2, s2);
          if (v && lane == 0) *violation = 1;
        }
        if (stat == 4 && T.rle) *violation = 1;
        n_forced += (stat == 1);
        n_regular += (stat == 2);
        n_small += (stat == 3);
        if (out == kOutKeep) {
          if (lane == j) my_kept = true;
          // rare: finalisation or constraint reset changed one or both regions
          if (!SameState(o1, s1)) {
            if (r1 == hot) H = s1;
            if (ra == r1) A = s1;
            if (rb == r1) B = s1;
            if (lane == j) StoreState(nodes, r1, s1);
          }
          if (!SameState(o2, s2)) {
            if (r2 == hot) H = s2;
            if (ra == r2) A = s2;
            if (rb == r2) B = s2;
            if (lane == j) StoreState(nodes, r2, s2);
          }
        } else {
          const int win = (out == kOutMerge1) ? r1 : r2;
          const int lose = (out == kOutMerge1) ? r2 : r1;
          const RState sw = (out == kOutMerge1) ? s1 : s2;
          if (win != hot) {
            if (hot >= 0 && hot != lose) {
              // the previous hot region leaves the registers: refresh lane copies + memory
              if (ra == hot) A = H;
              if (rb == hot) B = H;
              if (lane == 0) StoreState(nodes, hot, H);
            }
            hot = win;
          }
          H = sw;
          if (lane == j) nodes.parent[lose] = win;
          if (ra == lose) ra = win;
          if (rb == lose) rb = win;
        }
        // the generic path may have refreshed lane copies: re-derive the partner flags
        ok_a = (A.cons < 0 && A.flags == 0);
        ok_b = (B.cons < 0 && B.flags == 0);
      }
      if (valid && my_kept) kept_all[gpos] = 1;
      // Make this batch's stores visible to the next batch's loads (same CU: L1 is shared).
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      // Next batch reloads region states from memory; the hot region's memory copy is stale, so
      // write it back here once per batch (one 21-byte store) instead of once per merge.
      if (hot >= 0 && lane == 0) StoreState(nodes, hot, H);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      cyc_loop += __builtin_readcyclecounter() - bt1;
    }
    if (lane == 0) {
      atomicMax(&stats[16], __builtin_readcyclecounter() - seg_t0);   // slowest component
      atomicMax(&stats[17], (unsigned long long)cnt);                 // largest component
    }
  }
  if (lane == 0) {
    if (n_forced) atomicAdd(&stats[0], (unsigned long long)n_forced);
    if (n_regular) atomicAdd(&stats[1], (unsigned long long)n_regular);
    if (n_small) atomicAdd(&stats[2], (unsigned long long)n_small);
    atomicAdd(&stats[4], (unsigned long long)dbg_iters);
    atomicAdd(&stats[5], (unsigned long long)dbg_hot);
    atomicAdd(&stats[6], (unsigned long long)dbg_internal);
    atomicAdd(&stats[7], (unsigned long long)dbg_batches);
    atomicAdd(&stats[18], cyc_load);
    atomicAdd(&stats[19], cyc_loop);
    atomicAdd(&stats[20], (unsigned long long)dbg_chain);
    for (int k = 0; k < 6; ++k) atomicAdd(&stats[24 + k], (unsigned long long)dbg_g[k]);
  }
}

void LaunchMergeWaveV1(int grid, const WorkerArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_merge_wave_v1, dim3(grid), dim3(64), 0, s, a.num_segs, a.seg_off, a.seg_cnt,
                     a.s_ra, a.s_rb, a.s_gpos, a.nodes, a.kept_all, a.T, a.optimistic, a.violation,
                     a.stats);
  VSG_HIP(hipGetLastError());
}

}  // namespace vsg
