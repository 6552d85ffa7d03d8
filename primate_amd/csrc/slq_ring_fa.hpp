// slq_ring_fa.hpp — the ring-fed update pass of step j that ALSO takes step j + 1's alpha dot, W_{j+1} . (A W_{j+1}), a fixed lag
// of tile rounds behind its own write front (r04). The alpha-only pass - a whole sweep of the panel per step,
// a fifth (5-point grid) to a third (7-point grid) of the step's time - disappears: its image lines are re-read from the XCD's
// L2, where the update pass has just written them, instead of from HBM one launch later.
//
// reference: alpha_j = q_c . (A q_c - beta_j q_p), src/primate/include/lanczos.h:127-129. Same formula as the alpha-only pass
// (q_c . (A q_c) here, the -beta q_c . q_p part from the Gram / cross term); only the place where the sum is taken moves, and the
// 1 / nu^2 normalisation is applied to the sum (k_fin_gram) instead of to every term (nu_{j+1} is not known while the pass runs).
//
// How. An XCD's 32 workgroups sweep their chunk's tiles in DESCENDING order, workgroup bl taking tile first + bl + i nbl in
// round rho = imax - i. A workgroup's item sequence is  U(0) A(-L) U(1) A(1-L) ... : U(rho) is the update of its tile of round
// rho (exactly k_ring_pass<PASS_UPDATEG / PASS_UPDATE>), A(rho') the alpha dot of its tile of round rho' over the operator's
// INTERIOR upper-triangle stream - diagonal and in-chunk entries of higher column index - whose image lines are rows of
// W_{j+1} that tiles of rounds <= rho' of the same chunk have written. Entries that cross into the next chunk (another XCD's
// L2) are not in that stream: a small edge kernel adds them after the pass (slq.hip: k_alpha_edges).
//  * the two loader waves land U and A items alternately in the ring's four slots (U: 0, 2; A: 1, 3);
//  * consumer group 0 (7 waves) serves the U items, group 1 (7 waves) the A items;
//  * a U wave learns that its stores of round rho - 1 are complete from the counted wait for its NEXT prefetch (vector-memory
//    operations retire in issue order), adds to an LDS word, and the wave whose add is the round's last adds to the chunk's
//    global counter of that round (an atomic performed in the XCD's L2);
//  * before a loader lands A(rho') it has read that counter (a scalar load past the scalar cache, requested one item earlier):
//    all nbl workgroups of the XCD have completed round rho'. The image DMAs bypass the CU's L1 (sc1); the lines come from
//    the XCD's L2 - the same L2 the writers' stores completed in.
// Correctness rests on "same blockIdx.x % 8 = same XCD" (the dispatcher's round-robin, until now used for speed only): every
// workgroup checks its HW_REG_XCC_ID against the first one seen for its slot and raises the plan's flag (2) on a mismatch.
// Every wait is bounded (kRingSpinMax), and no U item ever waits for an A item or for another workgroup: the grid drains.
//
// STATUS (r04): correct - alpha, beta and the quadrature agree with the separate passes to 4e-15 and with the oracle exactly as they do,
// at 200^2 .. 1000^2 and 40^3 .. 100^3 - and SLOWER than what it replaces, so it is opt-in (SLQ_FUSED_ALPHA=1) and the default keeps
// the alpha-only pass. configs[1], 256 probes, orth 3, ms per step: separate 1.36 + 0.40; fused 5.9; fused without the counter polls
// (racy: timing only) 2.5; fused with the A items empty 1.76; lag 1 / 2 / 3 rounds, three items in flight per loader, nontemporal
// stores: all within 2.48 - 2.59 without polls. The update half alone loses 0.4 ms to its one consumer group and its one U tile in flight
// per loader; the A items add 0.75 ms - twice the standalone pass - whatever the lag, i.e. whether or not their lines are still in
// L2: the ring is paced by what a CU's loaders and request slots carry, not by HBM bytes, so bytes moved from HBM to L2 buy nothing.
#pragma once
#include "slq_ring.hpp"

namespace slq {

#ifndef SLQ_FA_LAG
#define SLQ_FA_LAG 2  // rounds between a tile's update and its alpha dot
#endif
#ifndef SLQ_FA_STORE
#define SLQ_FA_STORE 0  // store flavour of W_{j+1} (0 plain: the line stays in the XCD's L2; 1 nontemporal)
#endif
#ifndef SLQ_FA_INFLIGHT
#define SLQ_FA_INFLIGHT 2  // items a loader keeps in flight (it publishes item k + 1 - this at item k)
#endif
constexpr int kFaLag = SLQ_FA_LAG;
constexpr int kFaXccReg = 20 | (0 << 6) | (3 << 11);  // s_getreg: HW_REG_XCC_ID, bits [3:0]

// RC >= 1: PASS_UPDATEG with RC ring columns (Gram rows in slabs 1 .. RC, alpha in slab 1 + RC);
// RC == 0: PASS_UPDATE at orth = 0 with the cross term W_{j+1} . W_j (slab 1), alpha in slab 2.
template <typename F, int RC>
__global__ __launch_bounds__(1024) void k_ring_fa(
    int n, const int32_t *__restrict__ desc_u, const char *__restrict__ rec_u, const int32_t *__restrict__ desc_a, const char *__restrict__ rec_a, TileRanges xr,
    F *ring, int64_t slot_stride, int S, int j, const double *__restrict__ coefA, const double *__restrict__ coefB, const double *__restrict__ gamma,
    double *__restrict__ part, int bpad, int *cnt, int cnt_rounds, int gen, int *xcc_tab, int *__restrict__ fail, int mode /* timing experiments (SLQ_FA_MODE): 1 no poll, 2 no A items */) {
  using VF = typename VecT<F>::type;
  using RG = RingGeo<64, 16, 0>;
  constexpr int V = Geo<F, 64>::V, PW = Geo<F, 64>::PW;
  constexpr int NX = RC > 2 ? RC - 2 : 1;
  constexpr int NC = RG::NC, MR = RG::MR, NS = RG::kSlots, LAG = SLQ_FA_INFLIGHT, WAVES = 16;
  constexpr int kChunk = kRingChunk;
  constexpr bool kGram = RC > 0;
  constexpr int LR = kFaLag;
  static_assert(RG::kLoaders == 2 && RG::G == 2 && NS == 4 && MR == 2 && RC <= kRingMaxR, "geometry of the fused form");
  static_assert((LAG == 2 || LAG == 3) && LAG + 1 <= NS, "loader read-ahead");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  using lds_int = __attribute__((address_space(3))) int;
  lds_int *flags = (lds_int *)lds_raw;
  unsigned char *slots = lds_raw + RG::kHeadBytes;
  double *red = (double *)slots;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int panel = (int)gridDim.y - 1 - (int)blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + lane * V;
  const int first = (j == 0);
  const F *wcl = ring + (int64_t)(j % S) * slot_stride + poff;
  const F *wp = ring + (int64_t)((j + S - 1) % S) * slot_stride + poff;
  F *wn = ring + (int64_t)((j + 1) % S) * slot_stride + poff;
  const F *ux[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) ux[i] = ring + (int64_t)ring_slot(j - 2 - i, S) * slot_stride + poff;
  if (threadIdx.x < 16) flags[threadIdx.x] = 0;
  __syncthreads();
  lds_int *ready = flags, *done = flags + NS, *abort_f = flags + 2 * NS, *stored = flags + 2 * NS + 1;
  const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int t0 = xr.first[xcd], t_end = xr.first[xcd + 1];
  const int T = t_end - t0;
  const int imax = T > 0 ? (T - 1) / nbl : -1;
  const int nrounds = T > 0 ? imax + 1 + LR : 0, nitems = 2 * nrounds;
  int *cntp = cnt + (int64_t)(panel * 8 + xcd) * cnt_rounds;
  const int target = gen * nbl;  // every workgroup of the XCD adds one per round and pass
  if (threadIdx.x == 0) {
    // the placement this pass's hand-offs rest on: one XCD per value of blockIdx.x % 8
    const int me = (int)(__builtin_amdgcn_s_getreg(kFaXccReg) & 15) + 1;
    const int seen = atomicCAS(xcc_tab + xcd, 0, me);
    if (seen != 0 && seen != me) *fail = 2;
  }
  auto tile_of = [&](int rho) -> int {  // this workgroup's tile of round rho, -1: none
    if (rho < 0 || rho > imax) return -1;
    const int m = t0 + bl + (imax - rho) * nbl;
    return m < t_end ? m : -1;
  };
  VF acc1 = (VF)(F)0, accx = (VF)(F)0, acca = (VF)(F)0;
  VF dacc[RC > 0 ? RC : 1];
#pragma unroll
  for (int i = 0; i < (RC > 0 ? RC : 1); ++i) dacc[i] = (VF)(F)0;
  auto spin = [&](lds_int *w, int want) -> bool {
    for (int it = 0; it < kRingSpinMax; ++it) {
      int have, ab;
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(have), "=&v"(ab)
                   : "v"((unsigned)(uintptr_t)w), "v"((unsigned)(uintptr_t)abort_f)
                   : "memory");
      if (__builtin_amdgcn_readfirstlane(have) >= want) return true;
      if (__builtin_amdgcn_readfirstlane(ab)) return false;
      __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(abort_f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0) *fail = 1;
    return false;
  };
  auto bump = [&](lds_int *w) { asm volatile("ds_add_u32 %0, %1" ::"v"((unsigned)(uintptr_t)w), "v"(1) : "memory"); };
  if (nitems > 0 && wave < RG::kLoaders) {
    // ---------------- loader: U and A items in turn (the R = 1 loader of k_ring_pass: scalar descriptors one item ahead) --------
    constexpr int MAXU = (RG::kLines + RG::kLoaders - 1) / RG::kLoaders;
    const char *wcbase = (const char *)(ring + (int64_t)(j % S) * slot_stride + (int64_t)panel * n * PW);
    const char *wnbase = (const char *)(ring + (int64_t)((j + 1) % S) * slot_stride + (int64_t)panel * n * PW);
    const unsigned lane_off = (unsigned)lane * 16u;
    struct TileDesc {
      int D, recoff, chunks, rho;
      int c[MAXU];
    };
    auto fetch_desc = [&](int it, TileDesc &o) {  // uniform addresses: s_load
      const int rho = (it >> 1) - ((it & 1) ? LR : 0);
      const int m = (it < nitems && !((it & 1) && (mode & 2))) ? tile_of(rho) : -1;
      const int32_t *dsc = ((it & 1) ? desc_a : desc_u) + (int64_t)(m < 0 ? t0 : m) * RG::kDescWords;
      o.rho = m < 0 ? -1 : rho;
      o.D = m < 0 ? 0 : dsc[kDescCols];
      o.recoff = dsc[kDescRecOff];
      o.chunks = m < 0 ? 0 : dsc[kDescRecChunks];
#pragma unroll
      for (int u = 0; u < MAXU; ++u) o.c[u] = dsc[kDescList + wave * kRing1ListHalf + u];
    };
    auto poll_cnt = [&](int rho) -> bool {  // (slow path: the read-ahead came too early)
      for (int it = 0; it < kRingSpinMax; ++it) {
        int v, ab;
        asm volatile("s_load_dword %0, %2, 0x0 glc\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(v), "=&v"(ab)
                     : "s"(cntp + rho), "v"((unsigned)(uintptr_t)abort_f)
                     : "memory");
        if (v >= target) return true;
        if (__builtin_amdgcn_readfirstlane(ab)) return false;
        __builtin_amdgcn_s_sleep(4);
      }
      __hip_atomic_store(abort_f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (lane == 0) *fail = 1;
      return false;
    };
    TileDesc cur;
    fetch_desc(0, cur);
    int dv = 0, ab = 0, cv = 0x7fffffff;
    int hist[LAG - 1];
#pragma unroll
    for (int i = 0; i < LAG - 1; ++i) hist[i] = 0;
    bool ok = true;
    for (int k = 0; k < nitems + LAG - 1 && ok; ++k) {
      int issued = 0;
      if (k < nitems) {
        const int slot = k % NS;
        if (k >= NS) {
          if (__builtin_amdgcn_readfirstlane(ab)) ok = false;
          else if (__builtin_amdgcn_readfirstlane(dv) < NC * (k / NS)) ok = spin(done + slot, NC * (k / NS));
        }
        if (ok && (k & 1) && cur.rho >= 0 && cv < target && !(mode & 1)) ok = poll_cnt(cur.rho);  // the writers of this tile's lines have finished
        if (ok) {
          unsigned char *img = slots + (size_t)slot * RG::kSlotBytes;
          const int nd = cur.D;
          const char *rsrc = ((k & 1) ? rec_a : rec_u) + (int64_t)cur.recoff * 16;
          if (k & 1) {
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
              const int d = wave + u * RG::kLoaders;
              if (d < nd) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wnbase + (int64_t)cur.c[u] * (PW * (int)sizeof(F)) + lane_off),
                                                 (__attribute__((address_space(3))) void *)(img + (size_t)d * 1024), 16, 0, 16 /* sc1: past the L1 */);
                ++issued;
              }
            }
          } else {
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
              const int d = wave + u * RG::kLoaders;
              if (d < nd) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wcbase + (int64_t)cur.c[u] * (PW * (int)sizeof(F)) + lane_off),
                                                 (__attribute__((address_space(3))) void *)(img + (size_t)d * 1024), 16, 0, SLQ_RING_AUX);
                ++issued;
              }
            }
          }
          for (int c = wave; c < cur.chunks; c += RG::kLoaders) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rsrc + c * 1024 + lane_off),
                                             (__attribute__((address_space(3))) void *)(img + RG::kLines * 1024 + c * 1024), 16, 0, 0);
            ++issued;
          }
        }
      }
      // ---- what precedes item k + 1 ----
      TileDesc nxt;
      fetch_desc(k + 1, nxt);
      // (the chunk counter of the next item's round, past the scalar cache; requested for every item - an item without a tile reads
      // round 0's and ignores it - so that no path merges the register before the wait below)
      int dvn, abn, cvn;
      asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(cvn) : "s"(cntp + (nxt.rho < 0 ? 0 : nxt.rho)) : "memory");
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(dvn), "=&v"(abn) : "v"((unsigned)(uintptr_t)(done + (k + 1) % NS)), "v"((unsigned)(uintptr_t)abort_f) : "memory");
#pragma unroll
      for (int i = LAG - 2; i > 0; --i) hist[i] = hist[i - 1];
      hist[0] = issued;
      int since = 0;
#pragma unroll
      for (int i = 0; i < LAG - 1; ++i) since += hist[i];
      wait_vmcnt_at_most(since);
      if (k + 1 >= LAG && lane == 0) bump(ready + (k + 1 - LAG) % NS);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dvn), "+v"(abn), "+s"(cvn)::"memory");
      dv = dvn;
      ab = abn;
      cv = cvn;
      cur = nxt;
    }
  } else if (nitems > 0) {
    const int cwv = wave - RG::kLoaders;
    const int grp = cwv / NC, cw = cwv % NC;
    if (grp == 0) {
      // ---------------- U consumers: the update of round rho's tile (k_ring_pass: do_group, PASS_UPDATEG / PASS_UPDATE) --------
      const int colbase = panel * PW + lane * V;
      VF sc, cp, cb;
      VF gm[RC > 0 ? RC : 1];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        sc[v] = (F)coefA[colbase + v];
        cp[v] = (F)coefA[bpad + colbase + v];
        cb[v] = (F)coefB[colbase + v];
      }
#pragma unroll
      for (int i = 0; i < RC; ++i)
#pragma unroll
        for (int v = 0; v < V; ++v) gm[i][v] = (F)gamma[(int64_t)i * bpad + colbase + v];
      auto load_desc = [&](int rho) -> int {
        const int m = tile_of(rho);
        const int d = desc_u[(int64_t)(m < 0 ? t0 : m) * RG::kDescWords + lane];
        return m < 0 ? 0 : d;  // (no tile: no rows)
      };
      VF xpn[MR], un[MR][NX];
      int nloads = 0;
      auto fetch_group = [&](int i, int r_lo, int nrows) {
        const int lr = cw + i * NC;
        if (lr < nrows) {
          const int64_t ro = (int64_t)(r_lo + lr) * PW;
          if (!first) {
            xpn[i] = stream_load<1>((const VF *)(wp + ro));
            ++nloads;
          }
          if (RC > 2) {
#pragma unroll
            for (int q = 0; q < NX; ++q) un[i][q] = stream_load<1>((const VF *)(ux[q] + ro));
            nloads += NX;
          }
        }
      };
      auto do_group = [&](const unsigned char *img, int i, int r_lo, int nrows, const VF &xp_in, const VF *u_in) {
        const unsigned char *rec = img + RG::kLines * 1024;
        const F *xl = (const F *)img + lane * V;
        const int lr = cw + i * NC;
        const int head = ((const int *)rec)[lane & 31];
        const int valoff = lane_bcast(head, RG::kRecValOffW);
        const int p0 = lane_bcast(head, lr), p1 = lane_bcast(head, lr + 1), si = lane_bcast(head, RG::kRecSelfW + lr);
        const VF xp = first ? (VF)(F)0 : xp_in;
        const VF xc = *(const VF *)(xl + (size_t)si * PW);
        VF acc = (VF)(F)0;
        for (int pb = p0; pb < p1; pb += kChunk) {
          const int cnt_e = p1 - pb;
          const int e = min(pb + (lane & (kChunk - 1)), p1 - 1);
          const int lcv = *(const int *)(rec + RG::kRecHeadB + e * 4);
          const F vav = *(const F *)(rec + valoff + e * (int)sizeof(F));
          VF x[kChunk];
#pragma unroll
          for (int q = 0; q < kChunk; ++q) x[q] = *(const VF *)(xl + (size_t)(q < cnt_e ? lane_bcast(lcv, q) : si) * PW);
#pragma unroll
          for (int q = 0; q < kChunk; ++q) acc += (q < cnt_e ? lane_bcast(vav, q) : (F)0) * x[q];
        }
        const int64_t ro = (int64_t)(r_lo + lr) * PW;
        VF w = sc * acc;
        if (!first) w -= cp * xp;
        w -= cb * xc;
        if constexpr (RC > 0) w -= gm[0] * xc;
        if constexpr (RC > 1) w -= gm[1] * xp;
#pragma unroll
        for (int q = 2; q < RC; ++q) w -= gm[q] * u_in[q - 2];
        stream_store<SLQ_FA_STORE>((VF *)(wn + ro), w);
        acc1 += w * w;
        if constexpr (kGram) {
          dacc[0] += w * xc;
          if constexpr (RC > 1) dacc[1] += w * xp;
#pragma unroll
          for (int q = 2; q < RC; ++q) dacc[q] += w * u_in[q - 2];
        } else {
          accx += w * xc;
        }
      };
      int dcur = load_desc(0), dnext = load_desc(1);
      {
        const int r_lo0 = lane_bcast(dcur, kDescRow0), nrows0 = lane_bcast(dcur, kDescRows);
#pragma unroll
        for (int i = 0; i < MR; ++i) fetch_group(i, r_lo0, nrows0);
      }
      bool ok = true;
      for (int rho = 0; rho < nrounds && ok; ++rho) {
        const int k = 2 * rho, slot = k % NS;
        const int r_lo = lane_bcast(dcur, kDescRow0), nrows = lane_bcast(dcur, kDescRows);
        const int dnext2 = load_desc(rho + 2);
        const int r_lo_n = lane_bcast(dnext, kDescRow0), nrows_n = lane_bcast(dnext, kDescRows);
        const unsigned char *img = slots + (size_t)slot * RG::kSlotBytes;
        VF xpc[MR], uc[MR][NX];
#pragma unroll
        for (int i = 0; i < MR; ++i) {
          xpc[i] = xpn[i];
#pragma unroll
          for (int q = 0; q < NX; ++q) uc[i][q] = un[i][q];
        }
        nloads = 1;  // (dnext2's descriptor word)
#pragma unroll
        for (int i = 0; i < MR; ++i) fetch_group(i, r_lo_n, nrows_n);
        if (rho > 0) {
          // this wave's stores of round rho - 1 are older than the prefetch it has just issued: once at most that prefetch is
          // outstanding they are complete (vector-memory operations retire in issue order). Tell the workgroup; the round's last
          // wave tells the chunk. (BEFORE the wait for this round's tile: the wave would only spin meanwhile, and a signal that waits
          // for the next tile's landing chains the rounds - the next tile lands behind the loader's poll for this very signal)
          wait_vmcnt_at_most(__builtin_amdgcn_readfirstlane(nloads));
          int old = 0;
          if (lane == 0)
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"((unsigned)(uintptr_t)(stored + ((rho - 1) & 3))), "v"(1) : "memory");
          old = __builtin_amdgcn_readfirstlane(old);
          // (an atomic WITHOUT sc1: performed in this XCD's L2 and left there - the only readers are this XCD's loaders, which read
          // past their scalar cache. With sc1 every add is a fabric write: 32 adds to one word took ~1 us each, 36 us per round)
          if (old + 1 == NC * (((rho - 1) >> 2) + 1) && lane == 0)
            asm volatile("global_atomic_add %0, %1, off" ::"v"(cntp + (rho - 1)), "v"(1) : "memory");
        }
        ok = spin(ready + slot, RG::kLoaders * (k / NS + 1));
        if (!ok) break;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
          if (cw + i * NC >= nrows) break;
          do_group(img, i, r_lo, nrows, xpc[i], uc[i]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump(done + slot);
        dcur = dnext;
        dnext = dnext2;
      }
    } else {
      // ---------------- A consumers: W_{j+1} . (A W_{j+1}) over the interior upper-triangle stream (k_ring_pass: do_alpha_padded) ----
      using I4 = int __attribute__((ext_vector_type(4)));
      typedef F F4 __attribute__((ext_vector_type(4), aligned(16)));
      auto load_rows = [&](int rho) -> int {
        const int m = (mode & 2) ? -1 : tile_of(rho);
        return m < 0 ? 0 : desc_a[(int64_t)m * RG::kDescWords + kDescRows];
      };
      int nr_cur = load_rows(-LR), nr_next = load_rows(1 - LR);
      bool ok = true;
      for (int rho = 0; rho < nrounds && ok; ++rho) {
        const int k = 2 * rho + 1, slot = k % NS;
        const int nrows = nr_cur;
        const int nr_next2 = load_rows(rho + 2 - LR);
        const unsigned char *img = slots + (size_t)slot * RG::kSlotBytes;
        ok = spin(ready + slot, RG::kLoaders * (k / NS + 1));
        if (!ok) break;
        if (nrows > 0) {
          const unsigned char *rec = img + RG::kLines * 1024;
          const F *xl = (const F *)img + lane * V;
          const int *rw = (const int *)rec;
          const int valoff = rw[RG::kRecValOffW];
          int p0[MR], p1[MR], si[MR];
          bool live[MR];
#pragma unroll
          for (int i = 0; i < MR; ++i) {
            const int lr = cw + i * NC;
            live[i] = lr < nrows;
            const int lrc = live[i] ? lr : 0;
            p0[i] = rw[lrc];
            p1[i] = live[i] ? rw[lrc + 1] : 0;
            si[i] = rw[RG::kRecSelfW + lrc];
          }
          I4 lc[MR];
          F4 va[MR];
          VF xc[MR];
#pragma unroll
          for (int i = 0; i < MR; ++i) {
            lc[i] = *(const I4 *)(rec + RG::kRecHeadB + p0[i] * 4);
            va[i] = *(const F4 *)(rec + valoff + p0[i] * (int)sizeof(F));
            xc[i] = *(const VF *)(xl + (size_t)si[i] * PW);
          }
          VF x[MR][4];
#pragma unroll
          for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) x[i][q] = *(const VF *)(xl + (size_t)lc[i][q] * PW);
#pragma unroll
          for (int i = 0; i < MR; ++i) {
            VF acc = va[i][0] * x[i][0];
#pragma unroll
            for (int q = 1; q < 4; ++q) acc += va[i][q] * x[i][q];
            for (int pb = p0[i] + 4; pb < p1[i]; pb += 4) {  // rows of more than four (padded) entries
              const I4 lw = *(const I4 *)(rec + RG::kRecHeadB + pb * 4);
              const F4 vw = *(const F4 *)(rec + valoff + pb * (int)sizeof(F));
#pragma unroll
              for (int q = 0; q < 4; ++q) acc += vw[q] * *(const VF *)(xl + (size_t)lw[q] * PW);
            }
            if (live[i]) acca += xc[i] * acc;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) bump(done + slot);
        nr_cur = nr_next;
        nr_next = nr_next2;
      }
    }
  }
  __syncthreads();
  const int64_t nblk = gridDim.x;
  auto reduce_out = [&](const VF &a, double *out) {
#pragma unroll
    for (int v = 0; v < V; ++v) red[(wave * 64 + lane) * V + v] = (double)a[v];
    __syncthreads();
    if ((int)threadIdx.x < PW) {
      const int t = threadIdx.x, c = t / V, v = t % V;
      double sum = 0.0;
      for (int w = 0; w < WAVES; ++w) sum += red[(w * 64 + c) * V + v];
      out[t] = sum;
    }
    __syncthreads();
  };
  reduce_out(acc1, part + (int64_t)blockIdx.x * bpad + panel * PW);
  if constexpr (kGram) {
#pragma unroll
    for (int i = 0; i < RC; ++i) reduce_out(dacc[i], part + ((int64_t)(1 + i) * nblk + blockIdx.x) * bpad + panel * PW);
  } else {
    reduce_out(accx, part + (nblk + blockIdx.x) * bpad + panel * PW);
  }
  reduce_out(acca, part + ((int64_t)(kGram ? 1 + RC : 2) * nblk + blockIdx.x) * bpad + panel * PW);
}

}  // namespace slq
