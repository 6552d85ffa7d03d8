// slq_kernels.hpp — hand-written CDNA4 (gfx950) kernels of the SLQ engine.
//
// Data layout. P probes advance in lock-step. Vectors are stored as PANELS:
// panel p of Lanczos vector t is a row-major n x PW array (PW probes per row, PW*sizeof(F) bytes
// = 256 B..1 KiB contiguous), so that
//   * the SpMM gather of x[col, :] is one coalesced 16 B/lane wave load serving PW probes,
//   * every per-probe reduction (alpha = q.w, ||w||^2, Q^T w) is a COLUMN sum: a lane owns its
//     V = 16/sizeof(F) probe columns for the whole kernel and accumulates in registers; there is
//     no cross-lane traffic in the inner loops. Cross-wave/cross-block sums go through LDS and a
//     fixed-order two-stage reduction (bitwise reproducible; no float atomics).
// A wave covers RPW = 64/LPR rows per instruction, LPR = PW/V lanes per row.
//
// Lanczos vectors are kept UNNORMALISED: slot t holds w_t with q_t = w_t / nu_t
// (nu_0 = ||v||, nu_t = beta_t). The reference's normalisation pass (lanczos.h:143) is folded
// into per-probe scalar coefficients, which removes one read and one write of n x P per step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slq_common.hpp"

namespace slq {

#ifdef SLQ_DEBUG_TIMES
// diagnostic build only (scripts/wave_drift.py): per-wave progress stamps of the merged dots pass
__device__ unsigned long long *g_dbg_times = nullptr;
#endif

// ---- CSR row gather for wave-uniform rows (one row per wave: RPW == 1) -------------------------------
// acc = sum_k vals[p] * X[colind[p], lane's columns] over the row's nonzeros p0 <= p < p1.
// The row is wave-uniform, so colind/vals are SCALAR loads; what bounds the gather passes is the chain of
// dependent memory round trips per row (rowptr -> colind -> gather), not bytes. Nonzeros are therefore
// taken in batches of 8 under a wave-uniform count mask: the whole batch's indices and values come in
// one s_load each, and all of its gathers are in flight together. (The plain loop takes 4 at a time and
// the remainder ONE AT A TIME, each behind an s_waitcnt vmcnt(0): 9 serialised round trips for a 7-point
// row, 5 for a 5-point row.) The index/value loads of a batch may run up to 7 entries past the row's
// end; they are never dereferenced. The CSR arrays carry kCsrPad spare entries for that
// (slq.hip: slq_csr_create pads every array it uploads).
// Summation order is p0, p0+1, ...: bitwise the same as a scalar loop over the row.
constexpr int kCsrPad = 8;
// 8 consecutive CSR entries as ONE scalar load each (s_load_dwordx8 / x16 need dword alignment only)
typedef int32_t csr_i8 __attribute__((ext_vector_type(8), aligned(4)));
typedef float csr_f8 __attribute__((ext_vector_type(8), aligned(4)));
typedef double csr_d8 __attribute__((ext_vector_type(8), aligned(8)));
template <typename F> struct CsrVals;
template <> struct CsrVals<float> { typedef csr_f8 type; };
template <> struct CsrVals<double> { typedef csr_d8 type; };

template <typename F, int PW>
__device__ __forceinline__ typename VecT<F>::type gather_row_uniform(const int32_t *__restrict__ colind,
                                                                      const F *__restrict__ vals, int p0, int p1,
                                                                      const F *xbase /* wave-uniform */,
                                                                      unsigned lane_off /* elements */) {
  using VF = typename VecT<F>::type;
  VF acc = (VF)(F)0;
  for (int p = p0; p < p1; p += 8) {
    const int cnt = p1 - p;
    const csr_i8 c = *(const csr_i8 *)(colind + p);
    const typename CsrVals<F>::type a = *(const typename CsrVals<F>::type *)(vals + p);
    VF x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < cnt) x[k] = *(const VF *)((xbase + (int64_t)c[k] * PW) + lane_off);  // scalar row base + lane offset
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < cnt) acc += a[k] * x[k];
  }
  return acc;
}

// ---- block-level column reduction -------------------------------------------------------------
// Every lane holds V partial sums for its V columns (column = (lane % LPR) * V + v). Sum them over
// the RPW row groups of each wave and over the waves of the block in a fixed order and let thread
// t < PW write column t of out[] (stride 1).
template <typename F, int LPR>
__device__ __forceinline__ void block_reduce_columns(const typename VecT<F>::type &acc,
                                                     double *red /* kWaves*64*V doubles */,
                                                     double *out) {
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < V; ++v) red[(wave * 64 + lane) * V + v] = (double)acc[v];
  __syncthreads();
  if ((int)threadIdx.x < PW) {
    const int t = threadIdx.x, cl = t / V, v = t % V;
    double s = 0.0;
    for (int w = 0; w < kWaves; ++w)
#pragma unroll
      for (int g = 0; g < RPW; ++g) s += red[(w * 64 + g * LPR + cl) * V + v];
    out[t] = s;
  }
  __syncthreads();
}

// ---- sweep A: panel SpMM + three-term update + alpha partials ----------------------------------
// For every row i and probe column c of the panel:
//   w[i,c]      = sc[c] * sum_k A[i,k] * Wc[k,c]  -  cp[c] * Wp[i,c]
//   partA[c]   += (sc[c] * Wc[i,c]) * w[i,c]
// i.e. w = A q_c - beta_j q_p and alpha_j = q_c . w   (reference: lanczos.h:127-129), with
// q_c = sc * Wc (sc = 1/nu_j) and beta_j q_p = cp * Wp (cp = beta_j / nu_{j-1}).
// Wn may alias Wp (orth = 0 ring of two slots): each element is read then written by the same
// lane. Row -> wave mapping is XCD-aware: blocks with equal (blockIdx.x % 8) share an XCD's L2
// (observed round-robin dispatch; speed only), so XCD x sweeps the contiguous row range
// [x*n/8, (x+1)*n/8) with all of its waves interleaved row by row: the set of rows in flight is a
// narrow band whose stencil neighbours stay L2-resident, and each panel row is fetched from HBM
// about once per XCD.
template <typename F, int LPR, int LP, int SP>
__global__ __launch_bounds__(kBlock) void k_spmm_3term(
    int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
    const F *__restrict__ vals, const F *Wc, const F *Wp, F *Wn, const double *__restrict__ coefA,
    double *__restrict__ partA, int bpad, int first) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  __shared__ double red[kWaves * 64 * V];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + cl * V;
  const F *wc = Wc + poff;
  const F *wp = Wp + poff;
  F *wn = Wn + poff;
  const int colbase = panel * PW + cl * V;
  VF sc, cp;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    sc[v] = (F)coefA[colbase + v];
    cp[v] = (F)coefA[bpad + colbase + v];
  }
  const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int chunk = (n + 7) / 8;
  const int r_begin = xcd * chunk;
  const int r_end = min(n, r_begin + chunk);
  const int stride = nbl * kWaves * RPW;
  VF aacc = (VF)(F)0;
  constexpr bool kUniform = RPW == 1;
  // (the wave index is wave-uniform by construction; readfirstlane tells the compiler, so that the row loop's
  // control flow and the CSR loads become scalar)
  const int rfirst = r_begin + (bl * kWaves + (kUniform ? __builtin_amdgcn_readfirstlane(wave) : wave)) * RPW;
  // wave-uniform rows: the NEXT row's rowptr pair is requested while this row's gathers are in flight
  int pn0 = 0, pn1 = 0;
  if (kUniform && rfirst < r_end) {
    const int rr = __builtin_amdgcn_readfirstlane(rfirst);
    pn0 = rowptr[rr];
    pn1 = rowptr[rr + 1];
  }
  for (int r0 = rfirst; r0 < r_end; r0 += stride) {
    int row = r0 + g;
    if (RPW == 1) row = __builtin_amdgcn_readfirstlane(row);
    if (row < r_end) {
      int p0, p1;
      if (kUniform) {
        p0 = __builtin_amdgcn_readfirstlane(pn0);
        p1 = __builtin_amdgcn_readfirstlane(pn1);
        const int rn = __builtin_amdgcn_readfirstlane(r0 + stride);
        if (rn < r_end) {
          pn0 = rowptr[rn];
          pn1 = rowptr[rn + 1];
        }
      } else {
        p0 = rowptr[row];
        p1 = rowptr[row + 1];
      }
      VF acc = (VF)(F)0;
      int p = p0;
      if (kUniform) {
        acc = gather_row_uniform<F, PW>(colind, vals, p0, p1, Wc + (int64_t)panel * n * PW, (unsigned)(cl * V));
        p = p1;
      }
      for (; p + 4 <= p1; p += 4) {
        const int c0 = colind[p], c1 = colind[p + 1], c2 = colind[p + 2], c3 = colind[p + 3];
        const F a0 = vals[p], a1 = vals[p + 1], a2 = vals[p + 2], a3 = vals[p + 3];
        const VF x0 = *(const VF *)(wc + (int64_t)c0 * PW);
        const VF x1 = *(const VF *)(wc + (int64_t)c1 * PW);
        const VF x2 = *(const VF *)(wc + (int64_t)c2 * PW);
        const VF x3 = *(const VF *)(wc + (int64_t)c3 * PW);
        acc += a0 * x0;
        acc += a1 * x1;
        acc += a2 * x2;
        acc += a3 * x3;
      }
      for (; p < p1; ++p) {
        const int c = colind[p];
        const F a = vals[p];
        acc += a * *(const VF *)(wc + (int64_t)c * PW);
      }
      const int64_t ro = (int64_t)row * PW;
      const VF xc = *(const VF *)(wc + ro);
      VF w = sc * acc;
      if (!first) w -= cp * stream_load<LP>((const VF *)(wp + ro));
      aacc += (sc * xc) * w;
      stream_store<SP>((VF *)(wn + ro), w);
    }
  }
  block_reduce_columns<F, LPR>(aacc, red, partA + (int64_t)blockIdx.x * bpad + panel * PW);
}

// ---- fused CSR passes: the SpMM is RECOMPUTED row-locally in every pass, never stored ----------
// Writes cost ~1.6x reads on MI355X HBM, and the intermediate vector
// u = A q_c - beta_j q_p of a Lanczos step is needed only row-locally by everything that follows
// (alpha = q_c.u, w' = u - alpha q_c, the reorthogonalisation dots and the final update). So
// instead of storing u (sweep A) and re-reading/re-writing it (sweeps B, C), each pass re-derives
// u[row] from the gather — bitwise the same value every time — and only the last pass writes:
//   PASS_ALPHA   partA += (sc*Wc[row]) * u[row]                               reads: Wc (gather), Wp
//   PASS_DOTS    w' = u - cB*Wc[row]; partD[i] += W_{t_i}[row] * w'           reads: + r ring columns
//   PASS_UPDATE  w'' = u - cB*Wc[row] - sum_i gamma_i W_{t_i}[row]; store; partN += w''^2
// Per step that is (2 + 2 max(r,2)) panel reads + 1 write (r > 0) or 4 reads + 1 write (r = 0)
// against (2r + 4) reads + 3 writes / 4 reads + 2 writes for the store-and-revisit sweeps.
// Ring columns t_0 = j (W_c) and t_1 = j-1 (W_p) are the rows the three-term part loads anyway.
// RC = compile-time number of ring columns (registers); the host uses these passes
// for r <= kFusedMaxR and the store-and-revisit sweeps (k_reorth_dot / k_reorth_update) for deeper
// reorthogonalisation.
// PASS_ADOTS merges the alpha pass into the dots pass (r >= 1). With u = A q_c - beta q_p formed row by row,
//   alpha = q_c.u                 and       W_t.(u - alpha q_c) = W_t.u - (alpha/nu_c) (W_t.W_c),
// so one sweep accumulates a = (sc W_c).u, d_i = W_{t_i}.u and g_i = W_{t_i}.W_c for the ring columns
// i = 1..RC-1 and k_fin_adots forms alpha and the projections exactly as lanczos.h:59-63,127-135 would, up to
// rounding. Column i = 0 (W_c itself) needs no sums: q_c.(u - alpha q_c) = alpha (1 - |q_c|^2) is pure
// rounding of alpha - in the reference too, where it stays below the 2 eps sqrt(n) threshold - so gamma_0 = 0.

template <typename F, int LPR, int PASS, int NTP, int RC, int PIPE>
__global__ __launch_bounds__(kBlock) void k_csr_pass(
    int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
    const F *__restrict__ vals, F *ring, int64_t slot_stride, int S, int j,
    const double *__restrict__ coefA, const double *__restrict__ coefB,
    const double *__restrict__ gamma /* PASS_UPDATE: [RC][bpad] */, double *__restrict__ part,
    int bpad, int xt) {
  // RC = number of ring columns of this step (r_j, compile time: the loops below carry no runtime guards
  // and every row's loads are issued back to back); column 0 is W_c, column 1 is W_p - the rows the
  // three-term part holds anyway - columns i >= 2 are W_{j-i}. NTP: nontemporal policy for streamed rows.
  // xt bit 1 (stored u, for operators whose gathers are expensive): PASS_ADOTS also WRITES u into slot j+1 and
  // PASS_UPDATE reads it back instead of gathering again (one gather pass per step, one extra write + read).
  // xt bit 0 (cross term): PASS_UPDATE also reduces X = sum_i w_{j+1}[i] w_j[i] into the second partial slab,
  // and the NEXT step's PASS_ALPHA then leaves W_p unread: alpha = q_c.(A q_c) - beta (q_c.q_p) with the
  // second dot taken from X (k_fin_alpha) - one panel sweep less per step, same formula as lanczos.h.
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  constexpr int NX = RC > 2 ? RC - 2 : 1;  // ring columns beyond W_c, W_p
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double *red = (double *)lds_raw;  // kWaves*64*V doubles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + cl * V;
  const int stored = (xt & 2) != 0;
  const int nostore = (xt >> 4) & 1;  // update passes: the new vector is not stored (a run's last step without a kept basis, r04)
  xt &= 1;
  const int first = (j == 0) || (PASS == PASS_ALPHA && xt);
  const F *wc = ring + (int64_t)(j % S) * slot_stride + poff;
  const F *wcu = ring + (int64_t)(j % S) * slot_stride + (int64_t)panel * n * PW;  // wave-uniform part of wc
  const unsigned loff = (unsigned)(cl * V);                                        // ... and the lane's part
  const F *wp = ring + (int64_t)((j + S - 1) % S) * slot_stride + poff;
  F *wn = ring + (int64_t)((j + 1) % S) * slot_stride + poff;
  const F *ux[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) ux[i] = ring + (int64_t)ring_slot(j - 2 - i, S) * slot_stride + poff;
  const int colbase = panel * PW + cl * V;
  VF sc, cp, cb = (VF)(F)0;
  VF gm[RC > 0 ? RC : 1];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    sc[v] = (F)coefA[colbase + v];
    cp[v] = (F)coefA[bpad + colbase + v];
    if (PASS != PASS_ALPHA) cb[v] = (F)coefB[colbase + v];
  }
  if (PASS == PASS_UPDATE || PASS == PASS_UPDATEG) {
#pragma unroll
    for (int i = 0; i < RC; ++i)
#pragma unroll
      for (int v = 0; v < V; ++v) gm[i][v] = (F)gamma[(int64_t)i * bpad + colbase + v];
  }
  const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int chunk = (n + 7) / 8;
  const int r_begin = xcd * chunk;
  const int r_end = min(n, r_begin + chunk);
  const int stride = nbl * kWaves * RPW;
  VF acc1 = (VF)(F)0;  // alpha or norm partial
  VF accx = (VF)(F)0;  // cross term
  VF dacc[RC > 0 ? RC : 1], gacc[RC > 0 ? RC : 1];
#pragma unroll
  for (int i = 0; i < (RC > 0 ? RC : 1); ++i) dacc[i] = gacc[i] = (VF)(F)0;
  // ---- row loop. Two forms of the CSR side of a row (template PIPE; slq_plan_create decides per operator):
  //  * plain (PIPE = 0): row pointers, then 4 column indices / gathers at a time, the remainder one by one;
  //  * pipelined (PIPE = 1, one row per wave only): the row is wave-uniform, so its CSR entries are
  //    scalar loads; the row pointers are requested two rows ahead and the first 8 column indices of the NEXT row
  //    one row ahead, both while this row's gathers are in flight; the row's (up to) 8 gathers are issued back to
  //    back under a count mask, and the diagonal entry is served by the row-local operand instead of a gather.
  // Neither form changes the order in which a row's products are summed.
  static_assert(PIPE == 0 || RPW == 1, "the pipelined row loop needs wave-uniform rows");
  constexpr bool pipeU = PIPE != 0;
  const bool reuse = PASS == PASS_UPDATE && stored;  // u was stored by the merged pass: no gather
  // (the wave index is wave-uniform by construction; readfirstlane tells the compiler, so that the row loop's
  // control flow and, with one row per wave, the CSR loads become scalar)
  const int rfirst = r_begin + (bl * kWaves + __builtin_amdgcn_readfirstlane(wave)) * RPW;
  using A8 = typename CsrVals<F>::type;
  int pn0 = 0, pn1 = 0, pq0 = 0, pq1 = 0;  // row pointers of the next row and of the one after it
  csr_i8 cn = (csr_i8)0;                   // first 8 column indices of the next row
  if (pipeU && !reuse) {
    const int rr = rfirst, rq = rr + stride;
    if (rr < r_end) {
      pn0 = rowptr[rr];
      pn1 = rowptr[rr + 1];
    }
    if (rq < r_end) {
      pq0 = rowptr[rq];
      pq1 = rowptr[rq + 1];
    }
    cn = *(const csr_i8 *)(colind + pn0);
  }
#ifdef SLQ_DEBUG_TIMES
  int dbg_q = 0, dbg_rows = 0;
  unsigned long long *dbg = (PASS == PASS_ADOTS && g_dbg_times)
                                ? g_dbg_times + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWaves + wave) * 8
                                : nullptr;
#endif
  for (int r0 = rfirst; r0 < r_end; r0 += stride) {
#ifdef SLQ_DEBUG_TIMES
    if (dbg && lane == 0) {
      const int q = (int)(((int64_t)(r0 - r_begin) * 4) / (r_end - r_begin));
      if (dbg_rows == 0 || q > dbg_q) dbg[q < 4 ? q : 3] = __builtin_amdgcn_s_memrealtime();
      dbg_q = q;
    }
    ++dbg_rows;
#endif
    int row = r0 + g;
    if (RPW == 1) row = __builtin_amdgcn_readfirstlane(row);
    if (row < r_end) {
      const int64_t ro = (int64_t)row * PW;
      int p0 = 0, p1 = 0;
      const csr_i8 c8 = cn;
      if (pipeU) {
        p0 = __builtin_amdgcn_readfirstlane(pn0);
        p1 = __builtin_amdgcn_readfirstlane(pn1);
      } else if (!reuse) {
        p0 = rowptr[row];
        p1 = rowptr[row + 1];
      }
      // row-local operands first: their latency overlaps the gathers
      const VF xc = *(const VF *)(wc + ro);
      VF xp = (VF)(F)0;
      if (!first) xp = stream_load<NTP>((const VF *)(wp + ro));
      // ring columns beyond W_c, W_p: up to two are requested before the gather chain (their latency hides behind
      // it); with more columns the rest are loaded after it, so that their registers are not live across it
      constexpr int NE = (PASS != PASS_ALPHA && RC > 2) ? (NX < 2 ? NX : 2) : 0;
      VF u[NX];
#pragma unroll
      for (int i = 0; i < NE; ++i) u[i] = stream_load<NTP>((const VF *)(ux[i] + ro));
      VF acc = (VF)(F)0;
      int p = p0;
      if (pipeU) {
        // batches of 8 nonzeros; the first batch's column indices are already in SGPRs (requested one row ago)
        csr_i8 cb = c8;
        for (int pb = p0; pb < p1; pb += 8) {
          const int cnt = p1 - pb;
          // live = entries to gather; the diagonal entry's panel row is xc (substituted at the multiply, not at
          // the load, so that no gather waits for it)
          unsigned live = cnt >= 8 ? 0xffu : ((1u << cnt) - 1u);
          unsigned diag = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) diag |= (cb[k] == row ? 1u : 0u) << k;
          diag &= live;
          live &= ~diag;
          VF x[8];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (live & (1u << k)) x[k] = *(const VF *)((wcu + (int64_t)cb[k] * PW) + loff);
          // with the gathers in flight: this batch's values (needed only when the gathers are back) and, once
          // per row, the next row's indices and the row pointers after it
          const A8 a8 = *(const A8 *)(vals + pb);
          if (pb == p0) {
            pn0 = pq0;
            pn1 = pq1;
            cn = *(const csr_i8 *)(colind + pq0);
            const int rn2 = r0 + 2 * stride;
            if (!reuse && rn2 < r_end) {
              pq0 = rowptr[rn2];
              pq1 = rowptr[rn2 + 1];
            }
            // pin the request HERE, behind the gathers' issue and ahead of their wait (the optimiser otherwise
            // sinks it to the loop latch, i.e. behind the wait, and the next row starts with an exposed s_load)
            asm volatile("" : "+s"(cn));
          }
          if (cnt > 8) cb = *(const csr_i8 *)(colind + pb + 8);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (live & (1u << k)) acc += a8[k] * x[k];
            else if (diag & (1u << k)) acc += a8[k] * xc;
          }
        }
        if (p0 == p1) {
          // a row without stored entries never enters the batch loop: the prefetch state still advances ONE row per row
          // (otherwise every later row of this wave would reuse the empty row's pointers and come out as -cp * W_p)
          pn0 = pq0;
          pn1 = pq1;
          cn = *(const csr_i8 *)(colind + pq0);
          const int rn2 = r0 + 2 * stride;
          if (!reuse && rn2 < r_end) {
            pq0 = rowptr[rn2];
            pq1 = rowptr[rn2 + 1];
          }
        }
        p = p1;
      }
      if constexpr (RPW > 1) {
        // several rows per wave (narrow panels): a row's last two or three entries go as ONE batch - indices clamped to the row's last
        // entry, coefficients of the surplus zero - instead of one dependent colind -> gather round trip each (r04: the upper triangle
        // of a 5-point grid is three entries per row, i.e. three round trips in the loops below; same products in the same order, plus zeros)
        for (; p + 4 <= p1; p += 4) {
          const int c0 = colind[p], c1 = colind[p + 1], c2 = colind[p + 2], c3 = colind[p + 3];
          const F a0 = vals[p], a1 = vals[p + 1], a2 = vals[p + 2], a3 = vals[p + 3];
          const VF x0 = *(const VF *)(wc + (int64_t)c0 * PW);
          const VF x1 = *(const VF *)(wc + (int64_t)c1 * PW);
          const VF x2 = *(const VF *)(wc + (int64_t)c2 * PW);
          const VF x3 = *(const VF *)(wc + (int64_t)c3 * PW);
          acc += a0 * x0;
          acc += a1 * x1;
          acc += a2 * x2;
          acc += a3 * x3;
        }
        if (p + 2 <= p1) {  // two or three entries left: one masked batch (a single one goes through the loop below: one gather, not four)
          const int last = p1 - 1;
          const int q1 = min(p + 1, last), q2 = min(p + 2, last), q3 = min(p + 3, last);
          const int c0 = colind[p], c1 = colind[q1], c2 = colind[q2], c3 = colind[q3];
          const F a0 = vals[p], a1 = p + 1 <= last ? vals[q1] : (F)0, a2 = p + 2 <= last ? vals[q2] : (F)0, a3 = p + 3 <= last ? vals[q3] : (F)0;
          const VF x0 = *(const VF *)(wc + (int64_t)c0 * PW);
          const VF x1 = *(const VF *)(wc + (int64_t)c1 * PW);
          const VF x2 = *(const VF *)(wc + (int64_t)c2 * PW);
          const VF x3 = *(const VF *)(wc + (int64_t)c3 * PW);
          acc += a0 * x0;
          acc += a1 * x1;
          acc += a2 * x2;
          acc += a3 * x3;
          p = p1;
        }
      }
      for (; p + 4 <= p1; p += 4) {
        const int c0 = colind[p], c1 = colind[p + 1], c2 = colind[p + 2], c3 = colind[p + 3];
        const F a0 = vals[p], a1 = vals[p + 1], a2 = vals[p + 2], a3 = vals[p + 3];
        const VF x0 = *(const VF *)(wc + (int64_t)c0 * PW);
        const VF x1 = *(const VF *)(wc + (int64_t)c1 * PW);
        const VF x2 = *(const VF *)(wc + (int64_t)c2 * PW);
        const VF x3 = *(const VF *)(wc + (int64_t)c3 * PW);
        acc += a0 * x0;
        acc += a1 * x1;
        acc += a2 * x2;
        acc += a3 * x3;
      }
      for (; p < p1; ++p) {
        const int c = colind[p];
        const F a = vals[p];
        acc += a * *(const VF *)(wc + (int64_t)c * PW);
      }
      if (PASS != PASS_ALPHA && RC > 2) {
#pragma unroll
        for (int i = NE; i < NX; ++i) u[i] = stream_load<NTP>((const VF *)(ux[i] + ro));
      }
      VF w = sc * acc;
      if (!first) w -= cp * xp;
      if (reuse) w = *(const VF *)(wn + ro);
      if (PASS == PASS_ADOTS && stored) stream_store<NTP>((VF *)(wn + ro), w);
      if (PASS == PASS_ALPHA) {
        acc1 += (sc * xc) * w;
      } else if (PASS == PASS_ADOTS) {
        acc1 += (sc * xc) * w;
        if constexpr (RC > 1) {
          dacc[1] += xp * w;
          gacc[1] += xp * xc;
        }
#pragma unroll
        for (int i = 2; i < RC; ++i) {
          dacc[i] += u[i - 2] * w;
          gacc[i] += u[i - 2] * xc;
        }
      } else {
        w -= cb * xc;
        if (PASS == PASS_DOTS) {
          if constexpr (RC > 0) dacc[0] += xc * w;
          if constexpr (RC > 1) dacc[1] += xp * w;
#pragma unroll
          for (int i = 2; i < RC; ++i) dacc[i] += u[i - 2] * w;
        } else {
          if constexpr (RC > 0) w -= gm[0] * xc;
          if constexpr (RC > 1) w -= gm[1] * xp;
#pragma unroll
          for (int i = 2; i < RC; ++i) w -= gm[i] * u[i - 2];
          if (!nostore) stream_store<NTP>((VF *)(wn + ro), w);
          acc1 += w * w;
          if constexpr (PASS == PASS_UPDATEG) {
            // Gram sequence outside the ring-fed tiles (r04): the new vector against every ring column the pass
            // holds in registers anyway - the NEXT step's projections are assembled from these rows (k_fin_gram), no dots pass
            if constexpr (RC > 0) dacc[0] += w * xc;
            if constexpr (RC > 1) dacc[1] += w * xp;
#pragma unroll
            for (int i = 2; i < RC; ++i) dacc[i] += w * u[i - 2];
          } else {
            accx += w * xc;
          }
        }
      }
    }
  }
#ifdef SLQ_DEBUG_TIMES
  if (dbg && lane == 0) {
    dbg[4] = __builtin_amdgcn_s_memrealtime();
    dbg[5] = (unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf);  // HW_REG_XCC_ID
    dbg[6] = (unsigned long long)dbg_rows;
    dbg[7] = (unsigned long long)blockIdx.x | ((unsigned long long)blockIdx.y << 32);
  }
#endif
  const int64_t nblk = gridDim.x;
  if (PASS == PASS_DOTS) {
#pragma unroll
    for (int i = 0; i < RC; ++i)
      block_reduce_columns<F, LPR>(dacc[i], red, part + ((int64_t)i * nblk + blockIdx.x) * bpad + panel * PW);
  } else if (PASS == PASS_ADOTS) {
    // slab 0: alpha partials; slabs 1..RC-1: d_i; slabs RC..2RC-2: g_i
    block_reduce_columns<F, LPR>(acc1, red, part + (int64_t)blockIdx.x * bpad + panel * PW);
#pragma unroll
    for (int i = 1; i < RC; ++i) {
      block_reduce_columns<F, LPR>(dacc[i], red, part + ((int64_t)i * nblk + blockIdx.x) * bpad + panel * PW);
      block_reduce_columns<F, LPR>(gacc[i], red, part + ((int64_t)(RC - 1 + i) * nblk + blockIdx.x) * bpad + panel * PW);
    }
  } else if (PASS == PASS_UPDATEG) {
    // slab 0: ||w||^2; slab 1 + q: w . W_{j-q} (the layout k_ring_pass<PASS_UPDATEG> writes: k_fin_beta_gram reads either)
    block_reduce_columns<F, LPR>(acc1, red, part + (int64_t)blockIdx.x * bpad + panel * PW);
#pragma unroll
    for (int i = 0; i < RC; ++i) block_reduce_columns<F, LPR>(dacc[i], red, part + ((int64_t)(1 + i) * nblk + blockIdx.x) * bpad + panel * PW);
  } else {
    block_reduce_columns<F, LPR>(acc1, red, part + (int64_t)blockIdx.x * bpad + panel * PW);
    if (PASS == PASS_UPDATE && xt)
      block_reduce_columns<F, LPR>(accx, red, part + (nblk + blockIdx.x) * bpad + panel * PW);
  }
}

// ---- tiled fused passes: a workgroup tile's distinct panel rows staged ONCE in LDS (opt-in, SLQ_TILES) -----------
// Every gather pass is bound by the bytes its CUs' vector-memory pipes carry per row: 6-7 gathered
// panel rows of 1 KiB per output row on the 7-point grid, 4-5 on the 5-point one, next to 2-3 KiB of streams. Rows are
// therefore grouped into compact CLUSTERS (slq.hip: build_clusters - greedy breadth-first blobs of <= kTileRows rows
// whose rows and columns together span <= kTileCols distinct panel rows: 2.8-3.0 per row on the 7-point grid, 1.7 on
// the 5-point one), a cluster is one workgroup TILE, and the tile's distinct rows are fetched once, by LDS-DMA
// (global_load_lds_dwordx4: one wave instruction lands one 1-KiB panel row in the tile image, no registers), after
// which every nonzero - and the row-local operand W_c[row] - is an LDS read. `lcol` holds, per nonzero, the position
// of its column in its tile's list (it replaces colind in the CSR stream); `self_idx` the position of each row itself.
// One panel row per wave instruction, i.e. LPR = 64 panels only. One workgroup per CU with TWO images (2 x 72 KiB):
// the next tile is staged while the current one is computed. The arithmetic per row is that of k_csr_pass, products summed in CSR order.
#ifndef SLQ_TILE_DB
#define SLQ_TILE_DB 0
#endif
constexpr int kTileRows = 24;  // rows per tile at most (3 per wave)
constexpr int kTileCols = 72;  // distinct panel rows per tile at most: 72 KiB of LDS image

template <typename F, int PASS, int NTP, int RC>
__global__ __launch_bounds__(kBlock) void k_csr_tile_pass(
    int n, const int32_t *__restrict__ rowptr, const F *__restrict__ vals,
    // (the lists are separate __restrict__ parameters, not a struct of pointers: only then are they provably read-only
    // and wave-uniform accesses to them scalar loads - as vector loads their waits would also drain the LDS-DMAs)
    const int32_t *__restrict__ tile_row, const int32_t *__restrict__ tile_ptr, const int32_t *__restrict__ tile_cols,
    const int32_t *__restrict__ lcol, const int32_t *__restrict__ self_idx, TileRanges xr, int max_cols, F *ring, int64_t slot_stride,
    int S, int j, const double *__restrict__ coefA, const double *__restrict__ coefB,
    const double *__restrict__ gamma /* PASS_UPDATE: [RC][bpad] */, double *__restrict__ part, int bpad, int xt) {
  using VF = typename VecT<F>::type;
  constexpr int LPR = 64;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW;
  constexpr int NX = RC > 2 ? RC - 2 : 1;
  static_assert(PASS == PASS_ALPHA || PASS == PASS_ADOTS || PASS == PASS_UPDATE, "tiled passes: alpha, merged dots, update");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double *red = (double *)lds_raw;                             // kWaves*64*V doubles
  F *tbuf = (F *)(lds_raw + sizeof(double) * kWaves * 64 * V);  // tile image: max_cols rows of PW
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + lane * V;
  xt &= 1;
  const int first = (j == 0) || (PASS == PASS_ALPHA && xt);
  const F *wcl = ring + (int64_t)(j % S) * slot_stride + poff;  // this lane's part of W_c's panel rows
  const F *wp = ring + (int64_t)((j + S - 1) % S) * slot_stride + poff;
  F *wn = ring + (int64_t)((j + 1) % S) * slot_stride + poff;
  const F *ux[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) ux[i] = ring + (int64_t)ring_slot(j - 2 - i, S) * slot_stride + poff;
  const int colbase = panel * PW + lane * V;
  VF sc, cp, cb = (VF)(F)0;
  VF gm[RC > 0 ? RC : 1];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    sc[v] = (F)coefA[colbase + v];
    cp[v] = (F)coefA[bpad + colbase + v];
    if (PASS != PASS_ALPHA) cb[v] = (F)coefB[colbase + v];
  }
  if (PASS == PASS_UPDATE) {
#pragma unroll
    for (int i = 0; i < RC; ++i)
#pragma unroll
      for (int v = 0; v < V; ++v) gm[i][v] = (F)gamma[(int64_t)i * bpad + colbase + v];
  }
  VF acc1 = (VF)(F)0, accx = (VF)(F)0;
  VF dacc[RC > 0 ? RC : 1], gacc[RC > 0 ? RC : 1];
#pragma unroll
  for (int i = 0; i < (RC > 0 ? RC : 1); ++i) dacc[i] = gacc[i] = (VF)(F)0;
  const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  constexpr int MR = (kTileRows + kWaves - 1) / kWaves;  // rows of a tile per wave at most
  const size_t img = (size_t)max_cols * PW;               // elements of one tile image; two images: double buffering
  // Software pipeline over this workgroup's tiles t, t + nbl, ...: while tile t is computed out of image A, the LDS-DMAs
  // of tile t + nbl fill image B and its row-local streams (W_p, ring columns) are on their way into registers. One
  // barrier per tile: it publishes image t AND says that every wave is done reading the image the next DMAs overwrite.
  auto stage = [&](int t, F *buf) {
    const int c0 = tile_ptr[t], D = tile_ptr[t + 1] - c0;
    const int dper = (D + kWaves - 1) / kWaves;
    const int d_lo = wave * dper, d_hi = min(D, d_lo + dper);
    for (int d = d_lo; d < d_hi; d += 8) {
      const csr_i8 cols = *(const csr_i8 *)(tile_cols + c0 + d);  // (reads past d_hi are spare or later entries: unused)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (d + k < d_hi)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wcl + (int64_t)cols[k] * PW),
                                           (__attribute__((address_space(3))) void *)(buf + (size_t)(d + k) * PW), 16, 0, 0);
    }
  };
  VF xpn[MR], un[MR][NX];  // row-local operands of the NEXT tile's rows of this wave
  auto fetch_rows = [&](int t) {
    const int r_lo = tile_row[t], r_hi = tile_row[t + 1];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int row = r_lo + wave + i * kWaves;
      if (row < r_hi) {
        const int64_t ro = (int64_t)row * PW;
        if (!first) xpn[i] = stream_load<NTP>((const VF *)(wp + ro));
        if (PASS != PASS_ALPHA && RC > 2) {
#pragma unroll
          for (int q = 0; q < NX; ++q) un[i][q] = stream_load<NTP>((const VF *)(ux[q] + ro));
        }
      }
    }
  };
  int t = xr.first[xcd] + bl;
  const int t_end = xr.first[xcd + 1];
  int cur = 0;
  constexpr bool kDB = SLQ_TILE_DB != 0;  // two images per workgroup (1 workgroup per CU) or one (2-4 per CU)
  if (kDB && t < t_end) {
    stage(t, tbuf);
    fetch_rows(t);
  }
  for (; t < t_end; t += nbl) {
    if (!kDB) {
      stage(t, tbuf);
      fetch_rows(t);
    }
    __syncthreads();  // (drains this wave's DMAs and loads first: vmcnt(0))
    const F *xl = tbuf + (size_t)cur * img + lane * V;  // this lane's part of the current image's rows
    VF xpc[MR], uc[MR][NX];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      xpc[i] = xpn[i];
#pragma unroll
      for (int q = 0; q < NX; ++q) uc[i][q] = un[i][q];
    }
    const int tn = t + nbl;
    if (kDB && tn < t_end) {
      stage(tn, tbuf + (size_t)(cur ^ 1) * img);
      fetch_rows(tn);
    }
    const int r_lo = tile_row[t], r_hi = tile_row[t + 1];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int row = r_lo + wave + i * kWaves;
      if (row >= r_hi) break;
      const int64_t ro = (int64_t)row * PW;
      const int p0 = rowptr[row], p1 = rowptr[row + 1];
      const int si = self_idx[row];
      const VF xp = first ? (VF)(F)0 : xpc[i];
      const VF xc = *(const VF *)(xl + (size_t)si * PW);
      VF acc = (VF)(F)0;
      for (int pb = p0; pb < p1; pb += 8) {
        const int cnt = p1 - pb;
        const csr_i8 lc = *(const csr_i8 *)(lcol + pb);
        const typename CsrVals<F>::type a8 = *(const typename CsrVals<F>::type *)(vals + pb);
        VF x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < cnt) x[k] = *(const VF *)(xl + (size_t)lc[k] * PW);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < cnt) acc += a8[k] * x[k];
      }
      VF w = sc * acc;
      if (!first) w -= cp * xp;
      if (PASS == PASS_ALPHA) {
        acc1 += (sc * xc) * w;
      } else if (PASS == PASS_ADOTS) {
        acc1 += (sc * xc) * w;
        if constexpr (RC > 1) {
          dacc[1] += xp * w;
          gacc[1] += xp * xc;
        }
#pragma unroll
        for (int q = 2; q < RC; ++q) {
          dacc[q] += uc[i][q - 2] * w;
          gacc[q] += uc[i][q - 2] * xc;
        }
      } else {
        w -= cb * xc;
        if constexpr (RC > 0) w -= gm[0] * xc;
        if constexpr (RC > 1) w -= gm[1] * xp;
#pragma unroll
        for (int q = 2; q < RC; ++q) w -= gm[q] * uc[i][q - 2];
        stream_store<NTP>((VF *)(wn + ro), w);
        acc1 += w * w;
        accx += w * xc;
      }
    }
    if (kDB) cur ^= 1;
    else __syncthreads();  // single image: every wave is done with it before the next tile's DMAs land
  }
  __syncthreads();  // the reductions below reuse LDS
  const int64_t nblk = gridDim.x;
  if (PASS == PASS_ADOTS) {
    // slab 0: alpha partials; slabs 1..RC-1: d_i; slabs RC..2RC-2: g_i
    block_reduce_columns<F, LPR>(acc1, red, part + (int64_t)blockIdx.x * bpad + panel * PW);
#pragma unroll
    for (int i = 1; i < RC; ++i) {
      block_reduce_columns<F, LPR>(dacc[i], red, part + ((int64_t)i * nblk + blockIdx.x) * bpad + panel * PW);
      block_reduce_columns<F, LPR>(gacc[i], red, part + ((int64_t)(RC - 1 + i) * nblk + blockIdx.x) * bpad + panel * PW);
    }
  } else {
    block_reduce_columns<F, LPR>(acc1, red, part + (int64_t)blockIdx.x * bpad + panel * PW);
    if (PASS == PASS_UPDATE && xt)
      block_reduce_columns<F, LPR>(accx, red, part + (nblk + blockIdx.x) * bpad + panel * PW);
  }
}

// ---- the tiled passes with a continuously fed ring of tile images (SLQ_TILES=2, the default for stencil-like operators;
// DESIGN.md §4.1a) ---------------------------------------------------------------------------------------------------
// k_csr_tile_pass alternates "land a tile, barrier, compute it, barrier": a CU's memory pipe is idle half the time, and
// every wave walks the CSR with dependent scalar loads. Here the workgroup (one per CU, kRingWaves = 16 waves) splits into
// kRingLoaders LOADER waves and the rest CONSUMER waves around a ring of kRingSlots slots in LDS, and EVERYTHING a tile
// needs arrives in its slot by LDS-DMA: the image (its distinct panel rows, 1 KiB each) and a RECORD of its CSR (row
// offsets, the line of every row's own panel row, per nonzero the line of its column and the value; layout below).
//  * every tile has a 256-byte DESCRIPTOR (tile_desc, 64 ints: counts, where its record and rows start, its distinct
//    panel rows) that a wave gets with ONE coalesced request, tiles ahead of its use, so no scalar-memory round trip sits
//    between tiles (scalar loads share lgkmcnt with LDS and return out of order: one in flight turns every LDS wait into
//    a wait for memory);
//  * a loader, per tile k: requests descriptor k + kRingLag - 1; waits with ONE counted s_waitcnt vmcnt(N) (N = its DMAs
//    of the kRingLag - 1 tiles before k plus the descriptor requests since descriptor k's) - which says both "descriptor
//    k is here" and "tile k - kRingLag has landed"; publishes that tile (ready[slot] += 1); waits until the consumers
//    have released slot k % kRingSlots (done[slot]); issues its share of tile k's DMAs. kRingLag tiles of DMAs per loader
//    stay in flight throughout;
//  * the consumer waves form kRingGroups groups that take the tiles in turn. A wave prefetches the row-local streams
//    (W_p, ring columns) of its rows of its NEXT tile - kRingGroups tiles ahead - into registers, polls ready[slot(k)]
//    in LDS, computes its rows of tile k out of the slot (CSR entries fetched kRingChunk at a time by the first lanes and
//    broadcast with v_readlane), and releases the slot (done[slot] += 1).
// Nothing but these LDS counters synchronises the waves inside the loop; every poll is bounded (kRingSpinMax, then the
// workgroup raises *fail and leaves: the host reports the run as failed instead of hanging the GPU).
template <typename F, int PASS, int NTP, int RC>
__global__ __launch_bounds__(kRingBlock) void k_csr_ring_pass(
    int n, const int32_t *__restrict__ tile_desc, const char *__restrict__ tile_rec, TileRanges xr, F *ring, int64_t slot_stride, int S, int j,
    const double *__restrict__ coefA, const double *__restrict__ coefB, const double *__restrict__ gamma, double *__restrict__ part,
    int bpad, int xt, int *__restrict__ fail) {
  using VF = typename VecT<F>::type;
  constexpr int LPR = 64;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW;
  constexpr int NX = RC > 2 ? RC - 2 : 1;
  constexpr int NCW = kRingWaves - kRingLoaders;     // consumer waves, in kRingGroups groups that take the tiles in turn
  constexpr int NC = NCW / kRingGroups;              // consumer waves of one tile
  constexpr int MR = (kRingTileRows + NC - 1) / NC;  // rows of a tile per consumer wave
  constexpr int kSlotBytes = kRingTileCols * 1024 + kRingMetaBytes;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // ready[kRingSlots], done[kRingSlots], abort. Explicit LDS pointers: a generic one would make every poll a flat load
  // that waits on vmcnt too - on the loader's own DMAs
  using lds_int = __attribute__((address_space(3))) int;
  lds_int *flags = (lds_int *)lds_raw;
  unsigned char *slots = lds_raw + kRingHeadBytes;
  double *red = (double *)slots;  // the final reduction (kRingWaves*64*V doubles) runs when the slots are dead
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // xt bit 2: sweep the panels and the tiles of every chunk in REVERSE order. The update pass does, so that each pass begins
  // where the one before it ended: the rows touched last are the ones still in the Infinity Cache / L2.
  const int rev = (xt >> 2) & 1;
  const int panel = rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + lane * V;
  xt &= 1;
  const int first = (j == 0) || (PASS == PASS_ALPHA && xt);
  const F *wcl = ring + (int64_t)(j % S) * slot_stride + poff;
  const F *wp = ring + (int64_t)((j + S - 1) % S) * slot_stride + poff;
  F *wn = ring + (int64_t)((j + 1) % S) * slot_stride + poff;
  const F *ux[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) ux[i] = ring + (int64_t)ring_slot(j - 2 - i, S) * slot_stride + poff;
  if (threadIdx.x < 2 * kRingSlots + 1) flags[threadIdx.x] = 0;  // (ordinary store, before the barrier)
  __syncthreads();
  lds_int *ready = flags, *done = flags + kRingSlots, *abort_f = flags + 2 * kRingSlots;
  const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int t_first = xr.first[xcd] + bl, t_end = xr.first[xcd + 1];
  const int ntiles = t_first < t_end ? (t_end - t_first + nbl - 1) / nbl : 0;
  VF acc1 = (VF)(F)0, accx = (VF)(F)0;
  VF dacc[RC > 0 ? RC : 1], gacc[RC > 0 ? RC : 1];
#pragma unroll
  for (int i = 0; i < (RC > 0 ? RC : 1); ++i) dacc[i] = gacc[i] = (VF)(F)0;
  // Bounded LDS poll: until *w >= want; false when the workgroup is aborting. Polls and counter updates are written in
  // assembly: a compiler-visible LDS access in the loader would be preceded by s_waitcnt vmcnt(0) (the compiler orders LDS
  // accesses after every outstanding LDS-DMA), draining exactly the DMAs meant to stay in flight. The orderings that
  // matter are explicit instead: the loader's counted vmcnt wait before it publishes, the consumer's lgkmcnt(0) before it
  // releases.
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
  // descriptor of the k-th tile of this workgroup, one word per lane (tiles past the end: the last one again, unused)
  auto tile_at = [&](int k) -> int64_t {  // this workgroup's k-th tile (past the end: its last one again)
    const int kk = min(k, ntiles - 1);
    return (int64_t)(t_first + (rev ? ntiles - 1 - kk : kk) * nbl);
  };
  auto load_desc = [&](int k) -> int { return tile_desc[tile_at(k) * 64 + lane]; };
  if (ntiles > 0 && wave < kRingLoaders) {
    // ---------------- loader ----------------
    // Descriptors reach the loader through LDS as well (a 256-byte DMA into a small staging ring, read back with an
    // assembly ds_read after the counted wait): a descriptor loaded into a register across the DMA loop would make the
    // compiler drain vmcnt to 0 at every use, and with it the DMAs meant to stay in flight.
    unsigned char *stage = lds_raw + 256 + (size_t)wave * 4 * 256;  // 4 descriptors per loader, after the flag words
    auto stage_desc = [&](int k) {
      const int32_t *src = tile_desc + tile_at(k) * 64 + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(stage + (k & 3) * 256), 4, 0, 0);
    };
    // descriptor k is requested kRingLag - 1 iterations before it is used, i.e. BEFORE the DMAs of tile k - kRingLag + 1:
    // everything issued after it is then exactly what the counted wait below leaves outstanding
    static_assert(kRingLag >= 2 && kRingLag <= 4, "the descriptor staging ring holds 4");
#pragma unroll
    for (int i = 0; i < kRingLag - 1; ++i) stage_desc(i);
    int hist[kRingLag - 1];  // DMAs issued for tiles k - 1, k - 2, ... (the ones that may still be in flight)
#pragma unroll
    for (int i = 0; i < kRingLag - 1; ++i) hist[i] = 0;
    bool ok = true;
    for (int k = 0; k < ntiles + kRingLag && ok; ++k) {
#ifdef SLQ_DEBUG_TIMES
      // diagnostic build (scripts/ring_timeline.py): workgroup 0 of panel 0, loader 0 and consumer 0, the first 256 tiles
      unsigned long long *dbg = (PASS == PASS_ADOTS && g_dbg_times && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && k < 256) ? g_dbg_times + (size_t)k * 8 : nullptr;
      if (dbg && lane == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
#endif
      stage_desc(k + kRingLag - 1);
      // descriptor k is here and tile k - kRingLag has landed once only what was issued after descriptor k is outstanding:
      // the DMAs of tiles k - kRingLag + 1 .. k - 1 and the kRingLag - 1 descriptor requests since
      int since = kRingLag - 1;
#pragma unroll
      for (int i = 0; i < kRingLag - 1; ++i) since += hist[i];
      wait_vmcnt_at_most(__builtin_amdgcn_readfirstlane(since));
      if (k >= kRingLag && lane == 0) bump(ready + (k - kRingLag) % kRingSlots);
#ifdef SLQ_DEBUG_TIMES
      if (dbg && lane == 0) dbg[1] = __builtin_amdgcn_s_memrealtime();
#endif
      int issued = 0;
      if (k < ntiles) {
        const int slot = k % kRingSlots;
        int dcur;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(dcur) : "v"((unsigned)(uintptr_t)(lds_int *)(stage + (k & 3) * 256) + lane * 4) : "memory");
        if (k >= kRingSlots) ok = spin(done + slot, NC * (k / kRingSlots));  // the slot's previous tile has been consumed
#ifdef SLQ_DEBUG_TIMES
        if (dbg && lane == 0) dbg[2] = __builtin_amdgcn_s_memrealtime();
#endif
        if (ok) {
          unsigned char *img = slots + (size_t)slot * kSlotBytes;
          const int D = lane_bcast(dcur, kDescCols);
          for (int d = wave; d < D; d += kRingLoaders) {
            const int col = lane_bcast(dcur, kDescList + ring1_list_pos(d));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wcl + (int64_t)col * PW),
                                             (__attribute__((address_space(3))) void *)(img + (size_t)d * 1024), 16, 0, SLQ_RING_AUX);
            ++issued;
          }
          // the record, 1 KiB per DMA, by the loaders in turn
          const int chunks = lane_bcast(dcur, kDescRecChunks);
          const char *rsrc = tile_rec + (int64_t)lane_bcast(dcur, kDescRecOff) * 16 + lane * 16;
          for (int c = wave; c < chunks; c += kRingLoaders) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rsrc + c * 1024),
                                             (__attribute__((address_space(3))) void *)(img + kRingTileCols * 1024 + c * 1024), 16, 0, 0);
            ++issued;
          }
        }
      }
#ifdef SLQ_DEBUG_TIMES
      if (dbg && lane == 0) {
        dbg[3] = __builtin_amdgcn_s_memrealtime();
        dbg[7] = (unsigned long long)issued;
      }
#endif
#pragma unroll
      for (int i = kRingLag - 2; i > 0; --i) hist[i] = hist[i - 1];
      hist[0] = issued;
    }
  } else if (ntiles > 0) {
    // ---------------- consumer ----------------
    const int grp = (wave - kRingLoaders) / NC, cw = (wave - kRingLoaders) % NC;  // this wave serves tiles grp, grp + G, ...
    const int colbase = panel * PW + lane * V;
    VF sc, cp, cb = (VF)(F)0;
    VF gm[RC > 0 ? RC : 1];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      sc[v] = (F)coefA[colbase + v];
      cp[v] = (F)coefA[bpad + colbase + v];
      if (PASS != PASS_ALPHA) cb[v] = (F)coefB[colbase + v];
    }
    if (PASS == PASS_UPDATE) {
#pragma unroll
      for (int i = 0; i < RC; ++i)
#pragma unroll
        for (int v = 0; v < V; ++v) gm[i][v] = (F)gamma[(int64_t)i * bpad + colbase + v];
    }
    VF xpn[MR], un[MR][NX];
    auto fetch_rows = [&](int r_lo, int nrows) {
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const int lr = cw + i * NC;
        if (lr < nrows) {
          const int64_t ro = (int64_t)(r_lo + lr) * PW;
          if (!first) xpn[i] = stream_load<NTP>((const VF *)(wp + ro));
          if (PASS != PASS_ALPHA && RC > 2) {
#pragma unroll
            for (int q = 0; q < NX; ++q) un[i][q] = stream_load<NTP>((const VF *)(ux[q] + ro));
          }
        }
      }
    };
    constexpr int G = kRingGroups;
    int dcur = load_desc(grp), dnext = load_desc(grp + G);
    fetch_rows(lane_bcast(dcur, kDescRow0), lane_bcast(dcur, kDescRows));
    bool ok = true;
    if (grp >= ntiles) ok = false;  // (fewer tiles than groups: nothing to prefetch either; rows stay undefined, unused)
    for (int k = grp; k < ntiles && ok; k += G) {
      const int slot = k % kRingSlots;
      const int r_lo = lane_bcast(dcur, kDescRow0), nrows = lane_bcast(dcur, kDescRows);
      const int dnext2 = load_desc(k + 2 * G);
      VF xpc[MR], uc[MR][NX];
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        xpc[i] = xpn[i];
#pragma unroll
        for (int q = 0; q < NX; ++q) uc[i][q] = un[i][q];
      }
      if (k + G < ntiles) fetch_rows(lane_bcast(dnext, kDescRow0), lane_bcast(dnext, kDescRows));
#ifdef SLQ_DEBUG_TIMES
      unsigned long long *dbg = (PASS == PASS_ADOTS && g_dbg_times && blockIdx.x == 0 && blockIdx.y == 0 && cw == 0 && k < 256) ? g_dbg_times + (size_t)k * 8 : nullptr;
      if (dbg && lane == 0) dbg[4] = __builtin_amdgcn_s_memrealtime();
#endif
      ok = spin(ready + slot, kRingLoaders * (k / kRingSlots + 1));
      if (!ok) break;
#ifdef SLQ_DEBUG_TIMES
      if (dbg && lane == 0) dbg[5] = __builtin_amdgcn_s_memrealtime();
#endif
      const unsigned char *img = slots + (size_t)slot * kSlotBytes;
      const F *xl = (const F *)img + lane * V;
      const unsigned char *rec = img + kRingTileCols * 1024;
      const int head = ((const int *)rec)[lane & 31];  // row offsets and own-line positions, one word per lane
      const int valoff = lane_bcast(head, kRecValOff);
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const int lr = cw + i * NC;
        if (lr >= nrows) break;
        const int64_t ro = (int64_t)(r_lo + lr) * PW;
        const int p0 = lane_bcast(head, lr), p1 = lane_bcast(head, lr + 1), si = lane_bcast(head, kRecSelf + lr);
        const VF xp = first ? (VF)(F)0 : xpc[i];
        const VF xc = *(const VF *)(xl + (size_t)si * PW);
        VF acc = (VF)(F)0;
        // the row's entries are fetched kRingChunk at a time by the first lanes and broadcast with v_readlane; entries past
        // the row's end read the row's own image line with a zero coefficient, so a chunk's image reads go out back to back
        for (int pb = p0; pb < p1; pb += kRingChunk) {
          const int cnt = p1 - pb;
          const int e = min(pb + (lane & (kRingChunk - 1)), p1 - 1);
          const int lcv = *(const int *)(rec + kRecHeadBytes + e * 4);
          const F vav = *(const F *)(rec + valoff + e * (int)sizeof(F));
          VF x[kRingChunk];
#pragma unroll
          for (int q = 0; q < kRingChunk; ++q) x[q] = *(const VF *)(xl + (size_t)(q < cnt ? lane_bcast(lcv, q) : si) * PW);
#pragma unroll
          for (int q = 0; q < kRingChunk; ++q) acc += (q < cnt ? lane_bcast(vav, q) : (F)0) * x[q];
        }
        VF w = sc * acc;
        if (!first) w -= cp * xp;
        if (PASS == PASS_ALPHA) {
          acc1 += (sc * xc) * w;
        } else if (PASS == PASS_SPMM) {  // the store-and-revisit sequence's first sweep: w = A q_c - beta q_p stored, alpha partials
          stream_store<NTP>((VF *)(wn + ro), w);
          acc1 += (sc * xc) * w;
        } else if (PASS == PASS_ADOTS) {
          acc1 += (sc * xc) * w;
          if constexpr (RC > 1) {
            dacc[1] += xp * w;
            gacc[1] += xp * xc;
          }
#pragma unroll
          for (int q = 2; q < RC; ++q) {
            dacc[q] += uc[i][q - 2] * w;
            gacc[q] += uc[i][q - 2] * xc;
          }
        } else {
          w -= cb * xc;
          if constexpr (RC > 0) w -= gm[0] * xc;
          if constexpr (RC > 1) w -= gm[1] * xp;
#pragma unroll
          for (int q = 2; q < RC; ++q) w -= gm[q] * uc[i][q - 2];
          stream_store<NTP>((VF *)(wn + ro), w);
          acc1 += w * w;
          accx += w * xc;
        }
      }
      // every LDS read of this wave from the slot has returned before the release
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) bump(done + slot);
#ifdef SLQ_DEBUG_TIMES
      if (dbg && lane == 0) dbg[6] = __builtin_amdgcn_s_memrealtime();
#endif
      dcur = dnext;
      dnext = dnext2;
    }
  }
  __syncthreads();
  // partial sums of the consumer waves, column by column in wave order (loaders hold zeros)
  const int64_t nblk = gridDim.x;
  auto reduce_out = [&](const VF &a, double *out) {
#pragma unroll
    for (int v = 0; v < V; ++v) red[(wave * 64 + lane) * V + v] = (double)a[v];
    __syncthreads();
    if ((int)threadIdx.x < PW) {
      const int t = threadIdx.x, cl = t / V, v = t % V;
      double sum = 0.0;
      for (int w = 0; w < kRingWaves; ++w) sum += red[(w * 64 + cl) * V + v];
      out[t] = sum;
    }
    __syncthreads();
  };
  if (PASS == PASS_ADOTS) {
    reduce_out(acc1, part + (int64_t)blockIdx.x * bpad + panel * PW);
#pragma unroll
    for (int i = 1; i < RC; ++i) {
      reduce_out(dacc[i], part + ((int64_t)i * nblk + blockIdx.x) * bpad + panel * PW);
      reduce_out(gacc[i], part + ((int64_t)(RC - 1 + i) * nblk + blockIdx.x) * bpad + panel * PW);
    }
  } else {
    reduce_out(acc1, part + (int64_t)blockIdx.x * bpad + panel * PW);
    if (PASS == PASS_UPDATE && xt) reduce_out(accx, part + (nblk + blockIdx.x) * bpad + panel * PW);
  }
}

// ---- plain panel SpMM: Y = A X (operator plugin surface; also the dense/CSR matmat entry) -------
// (rectangular matrices too - the Gram operator's two passes: n output rows, ncols input rows per panel)
template <typename F, int LPR>
__global__ __launch_bounds__(kBlock) void k_spmm_plain(int n, const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ colind,
                                                       const F *__restrict__ vals,
                                                       const F *__restrict__ X, F *__restrict__ Y, int ncols) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const F *x = X + (int64_t)blockIdx.y * ncols * PW + cl * V;
  F *y = Y + (int64_t)blockIdx.y * n * PW + cl * V;
  const int stride = gridDim.x * kWaves * RPW;
  for (int r0 = (blockIdx.x * kWaves + wave) * RPW; r0 < n; r0 += stride) {
    const int row = r0 + g;
    if (row < n) {
      VF acc = (VF)(F)0;
      for (int p = rowptr[row]; p < rowptr[row + 1]; ++p)
        acc += vals[p] * *(const VF *)(x + (int64_t)colind[p] * PW);
      *(VF *)(y + (int64_t)row * PW) = acc;
    }
  }
}

// Affine sparse operator A + t B on the union pattern: vals = va + t * vb (slq_operator_set_parameter)
template <typename F> __global__ void k_affine_vals(int64_t nnz, const F *__restrict__ va, const F *__restrict__ vb, double t, F *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (F)((double)va[i] + t * (double)vb[i]);
}

// Dense symmetric operator, column-major with leading dimension lda: Y[i,:] = sum_k A[k,i] X[k,:]
// (A symmetric, so row i is read as the contiguous column i). One wave per RPW rows.
template <typename F, int LPR>
__global__ __launch_bounds__(kBlock) void k_dense_panel(int n, const F *__restrict__ A,
                                                        int64_t lda, const F *__restrict__ X,
                                                        F *__restrict__ Y) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int64_t poff = (int64_t)blockIdx.y * n * PW + cl * V;
  const F *x = X + poff;
  F *y = Y + poff;
  const int stride = gridDim.x * kWaves * RPW;
  for (int r0 = (blockIdx.x * kWaves + wave) * RPW; r0 < n; r0 += stride) {
    const int row = r0 + g;
    if (row < n) {
      const F *a = A + (int64_t)row * lda;
      VF acc = (VF)(F)0;
      int k = 0;
      for (; k + 4 <= n; k += 4) {
        const F a0 = a[k], a1 = a[k + 1], a2 = a[k + 2], a3 = a[k + 3];
        const VF x0 = *(const VF *)(x + (int64_t)k * PW);
        const VF x1 = *(const VF *)(x + (int64_t)(k + 1) * PW);
        const VF x2 = *(const VF *)(x + (int64_t)(k + 2) * PW);
        const VF x3 = *(const VF *)(x + (int64_t)(k + 3) * PW);
        acc += a0 * x0;
        acc += a1 * x1;
        acc += a2 * x2;
        acc += a3 * x3;
      }
      for (; k < n; ++k) acc += a[k] * *(const VF *)(x + (int64_t)k * PW);
      *(VF *)(y + (int64_t)row * PW) = acc;
    }
  }
}

// ---- dense operator on the matrix cores: Y = A W_c as an fp64 MFMA panel GEMM, fused three-term ----
// The one GEMM-shaped product on this path: a dense symmetric A (n x n, column-major) times a probe
// panel (n x PW): 2 n^2 PW flops on n^2 matrix elements — 16..32 flop per byte of A, so the kernel
// should stream A at the HBM rate with v_mfma_f64_16x16x4_f64 doing the arithmetic. (The VALU kernel
// k_dense_panel re-reads the whole panel from L2 for every output row: 0.97 ms per product at
// n = 5000, PW = 64, i.e. 0.2 TB/s on A.)
// Workgroup = 16 output rows x all PW columns; its 8 waves split K = n eight ways (split-K, so that
// n/16 workgroups x 8 waves fill the chip at n = 5000) and are summed through LDS in wave order.
// Fragments (cdna_hip_programming.md §3, f64 layout): lane l holds A[m = l&15][k = l>>4] and
// B[k = l>>4][n = l&15]; the accumulator's 4 f64 are rows (l>>4) + 4*reg of column l&15.
// A is symmetric, so A[m][k] is read as element (k, m) of the column-major array: 16 contiguous
// doubles per k. Epilogue as k_3term: w = sc*acc - cp*W_p; alpha partial += (sc*W_c)*w.
typedef double d4_t __attribute__((ext_vector_type(4)));

// TW = columns handled per launch (<= 64 keeps the split-K LDS image at 64 KiB: 2 workgroups per CU);
// ldw = panel row stride (PW), col0 = first column of this launch inside the panel.
template <int TW>
__global__ __launch_bounds__(kBlock) void k_dense_mfma_3term(
    int n, const double *__restrict__ A, int64_t lda, const double *Wc, const double *Wp, double *Wn,
    const double *__restrict__ coefA, double *__restrict__ partA, int bpad, int first, int plain, int ldw,
    int col0) {
  constexpr int NT = TW / 16;  // 16-column accumulator tiles per wave
  const int PW = ldw;
  extern __shared__ __attribute__((aligned(16))) double lds_acc[];  // [kWaves][NT][4][64] split-K partial tiles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int panel = blockIdx.y;
  const int r0 = blockIdx.x * 16;
  const int64_t poff = (int64_t)panel * n * PW;
  const double *wc = Wc + poff;
  d4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4_t)0.0;
  // this wave's K slice, in multiples of 4
  const int kq = (n + 3) / 4;                       // number of k-quads
  const int per = (kq + kWaves - 1) / kWaves;
  const int q_begin = wave * per, q_end = min(kq, q_begin + per);
  const bool row_ok = (r0 + lr) < n;
  for (int q = q_begin; q < q_end; ++q) {
    const int k = q * 4 + lk;
    const bool k_ok = k < n;
    const double a = (row_ok && k_ok) ? A[(int64_t)k * lda + r0 + lr] : 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const double b = k_ok ? wc[(int64_t)k * PW + col0 + t * 16 + lr] : 0.0;
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
  // split-K reduction through LDS, wave order fixed
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_acc[((wave * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  // wave w finishes tiles t = w, w + 8, ...: sum the 8 partials, apply the epilogue
  double *alpha_red = lds_acc + kWaves * NT * 4 * 64;  // [TW][4 row groups]
  for (int t = wave; t < NT; t += kWaves) {
    const int col = col0 + t * 16 + lr;
    const int gc = panel * PW + col;
    const double sc = plain ? 1.0 : coefA[gc], cp = plain ? 0.0 : coefA[bpad + gc];
    double apart = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double u = 0.0;
      for (int w = 0; w < kWaves; ++w) u += lds_acc[((w * NT + t) * 4 + r) * 64 + lane];
      const int row = r0 + lk + 4 * r;
      if (row < n) {
        const int64_t off = poff + (int64_t)row * PW + col;
        double wv = sc * u;
        if (!first && !plain) wv -= cp * Wp[off];
        if (!plain) apart += (sc * Wc[off]) * wv;
        Wn[off] = wv;
      }
    }
    alpha_red[(col - col0) * 4 + lk] = apart;
  }
  __syncthreads();
  if (!plain && (int)threadIdx.x < TW) {
    const int c = threadIdx.x;
    partA[(int64_t)blockIdx.x * bpad + panel * PW + col0 + c] =
        (alpha_red[c * 4] + alpha_red[c * 4 + 1]) + (alpha_red[c * 4 + 2] + alpha_red[c * 4 + 3]);
  }
}

// ---- the same product for panels of 32+ columns: big workgroup tiles, wide loads, K split over workgroups ----------
// k_dense_mfma_3term above re-reads the whole probe panel for every 16 output rows (n/16 workgroups x n x TW x 8 B:
// 4x the bytes of A at TW = 64) with one 8-byte load per fragment element and nothing in flight while the matrix
// cores work: 0.196 ms per 5000^2 x 64 product, 1.0 TB/s on A, bound by what a CU's vector-memory pipe can have
// in flight. Here:
//  * a WAVE owns 32 output rows x 32 columns (2 x 2 accumulator tiles). The assignment of matrix rows / columns to
//    MFMA fragment indices is free, so both are INTERLEAVED: fragment row m of row tile rt is matrix row
//    rbase + 2m + rt, fragment column c of column tile h is matrix column cbase + 2c + h. A lane then fetches both
//    row tiles' A elements with ONE 16-byte load (rows 2m, 2m+1 of one k are adjacent in the column-major A) and both
//    column tiles' panel elements with one 16-byte load: 2 loads of 1 KiB feed 4 MFMAs;
//  * a WORKGROUP is (8 / NCG) row groups x NCG column groups of such waves (128 x 64 or 256 x 32): the waves of a row
//    group issue the same A loads, those of a column group the same panel loads, at the same time - the CU's L1
//    serves the repeats, so per k-quad the CU pulls 4 + 2 KiB from L2 for 32 MFMAs;
//  * the fragments of quad q + kDensePrefetch are requested when quad q's MFMAs issue (register ring): A streams
//    from HBM / the Infinity Cache, 1-2 us away;
//  * n/128 row tiles do not fill 256 CUs, so K is split over gridDim.z workgroups; every workgroup writes its raw
//    partial product to slab z and k_3term_slabs sums the slabs in slab order and applies the three-term epilogue
//    (bitwise reproducible; the slabs are a few MB).
typedef double d2u_t __attribute__((ext_vector_type(2), aligned(8)));
#ifndef SLQ_DENSE_PREFETCH
#define SLQ_DENSE_PREFETCH 6
#endif
constexpr int kDensePrefetch = SLQ_DENSE_PREFETCH;

template <int NCG>
__global__ __launch_bounds__(kBlock) void k_dense_mfma_tile(int n, const double *__restrict__ A, int64_t lda,
                                                            const double *__restrict__ X, int ldw, int col0,
                                                            double *__restrict__ raw, int64_t raw_stride) {
  constexpr int RG = kWaves / NCG;  // row groups of 32 rows per workgroup
  const int PW = ldw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int rg = wave / NCG, cg = wave % NCG;
  const int panel = blockIdx.y;
  const int rbase = (blockIdx.x * RG + rg) * 32;
  const int cbase = col0 + 32 * cg;
  const int64_t poff = (int64_t)panel * n * PW;
  const double *xw = X + poff + cbase + 2 * lr;
  d4_t acc[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[rt][h] = (d4_t)0.0;
  const int kq_all = (n + 3) / 4;  // k-quads, split over the gridDim.z workgroups of this row tile
  const int ks = gridDim.z, kz = blockIdx.z;
  const int perz = (kq_all + ks - 1) / ks;
  const int q_begin = min(kq_all, kz * perz), q_end = min(kq_all, q_begin + perz);
  const int rowA = rbase + 2 * lr;  // this lane's two rows: rowA (row tile 0), rowA + 1 (row tile 1)
  const bool pair_ok = rowA + 1 < n, one_ok = rowA < n;
  auto loadA = [&](int q) -> d2u_t {
    const int k = q * 4 + lk;
    d2u_t a = (d2u_t)0.0;
    if (q < q_end && k < n) {
      const double *ap = A + (int64_t)k * lda + rowA;
      if (pair_ok) a = *(const d2u_t *)ap;
      else if (one_ok) a[0] = ap[0];
    }
    return a;
  };
  auto loadX = [&](int q) -> d2u_t {
    const int k = q * 4 + lk;
    d2u_t b = (d2u_t)0.0;
    if (q < q_end && k < n) b = *(const d2u_t *)(xw + (int64_t)k * PW);
    return b;
  };
  constexpr int D = kDensePrefetch;
  d2u_t ar[D], br[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    ar[i] = loadA(q_begin + i);
    br[i] = loadX(q_begin + i);
  }
  for (int q0 = q_begin; q0 < q_end; q0 += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const d2u_t a = ar[i], b = br[i];
      ar[i] = loadA(q0 + i + D);  // refill the slot; quads past the range contribute zeros
      br[i] = loadX(q0 + i + D);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        acc[0][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[h], acc[0][h], 0, 0, 0);
        acc[1][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[h], acc[1][h], 0, 0, 0);
      }
    }
  }
  double *out = raw + (int64_t)kz * raw_stride + poff;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + 2 * (lk + 4 * r) + rt;
        if (row < n) out[(int64_t)row * PW + cbase + 2 * lr + h] = acc[rt][h][r];
      }
}

// k_dense_mfma_lds (r03): the same product with the operands staged ONCE per workgroup in LDS. k_dense_mfma_tile lets every wave
// fetch its own fragments and leaves the repeats to the CU's L1: 16 KiB of L1 traffic per 6 KiB of unique bytes per k-quad, all of
// it competing for the ~106 requests a CU keeps in flight - 45 % of the fp64 MFMA rate. Here a workgroup
// (RG x NCG waves of 32 x 32 outputs, as there) walks K in stages of kDenseBK: all 512 threads bring the stage's A block
// (kDenseBK x 32 RG, column-major like A: a wave = one column = 1 KiB) and panel block (kDenseBK x 32 NCG) into registers with
// 16-byte loads - zero beyond n, so no size is special - while the MFMAs of the stage before run out of LDS, then store them
// (two LDS buffers, two barriers per stage). Fragment reads are the old kernel's: one ds_read_b128 per operand gives a lane both
// interleaved row (column) tiles; with k-strides of 1 KiB / 512 B every 16-lane group of a ds_read_b128 hits 64 distinct banks.
// Requires lda even (16-byte aligned row pairs); slq.hip falls back to k_dense_mfma_tile otherwise.
constexpr int kDenseBK = 16;
#ifndef SLQ_DENSE_DEPTH
#define SLQ_DENSE_DEPTH 4
#endif
constexpr int kDenseDepth = SLQ_DENSE_DEPTH;  // stages of operands in registers ahead of the one in LDS (k_dense_mfma_lds, k_dense_mfma32_lds)
template <int NCG>
__global__ __launch_bounds__(kBlock) void k_dense_mfma_lds(int n, const double *__restrict__ A, int64_t lda, const double *__restrict__ X, int ldw,
                                                           int col0, double *__restrict__ raw, int64_t raw_stride) {
  constexpr int RG = kWaves / NCG, BM = 32 * RG, BN = 32 * NCG, BK = kDenseBK;
  constexpr int APT = (BK * BM / 2) / kBlock;  // 16-byte pieces of the A block per thread
  constexpr int XPT = (BK * BN / 2 + kBlock - 1) / kBlock;
  static_assert((BK * BM / 2) % kBlock == 0, "A block divides over the threads");
  __shared__ __attribute__((aligned(16))) double As[2][BK][BM];
  __shared__ __attribute__((aligned(16))) double Xs[2][BK][BN];
  const int PW = ldw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int rg = wave / NCG, cg = wave % NCG;
  const int panel = blockIdx.y;
  const int rbase = blockIdx.x * BM;
  const int64_t poff = (int64_t)panel * n * PW;
  const double *xp = X + poff + col0;
  d4_t acc[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[rt][h] = (d4_t)0.0;
  // K range of this workgroup (gridDim.z splits K as in k_dense_mfma_tile), in stages of BK
  const int kq_all = (n + 3) / 4;
  const int ks = gridDim.z, kz = blockIdx.z;
  const int perz = (kq_all + ks - 1) / ks;
  const int k_begin = min(n, kz * perz * 4), k_end = min(n, k_begin + perz * 4);
  const int nstage = (k_end - k_begin + BK - 1) / BK;
  // kDenseDepth register sets hold the stages st + 1 .. st + kDenseDepth while stage st's MFMAs run out of LDS: a stage is
  // 0.4 us of MFMAs, a load from the Infinity Cache or HBM 1-2 us away (one set, r03 first form: every stage waited for its
  // successor's loads - 87 us per 5000^2 x 64 product). A set is 12 VGPRs.
  constexpr int D = kDenseDepth;
  d2u_t ra[D][APT], rx[D][XPT];
  auto fetch = [&](int st, d2u_t (&fa)[APT], d2u_t (&fx)[XPT]) {
    const int k0 = k_begin + st * BK;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int pc = threadIdx.x + i * kBlock, kk = pc / (BM / 2), rp = pc % (BM / 2);
      const int k = k0 + kk, row = rbase + 2 * rp;
      d2u_t v = (d2u_t)0.0;
      if (k < k_end) {
        const double *ap = A + (int64_t)k * lda + row;
        if (row + 1 < n) v = *(const d2u_t *)ap;
        else if (row < n) v[0] = ap[0];
      }
      fa[i] = v;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int pc = threadIdx.x + i * kBlock, kk = pc / (BN / 2), cp = pc % (BN / 2);
      const int k = k0 + kk;
      d2u_t v = (d2u_t)0.0;
      if (pc < BK * BN / 2 && k < k_end) v = *(const d2u_t *)(xp + (int64_t)k * PW + 2 * cp);
      fx[i] = v;
    }
  };
  auto stash = [&](int buf, const d2u_t (&fa)[APT], const d2u_t (&fx)[XPT]) {
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int pc = threadIdx.x + i * kBlock, kk = pc / (BM / 2), rp = pc % (BM / 2);
      *(d2u_t *)&As[buf][kk][2 * rp] = fa[i];
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int pc = threadIdx.x + i * kBlock, kk = pc / (BN / 2), cp = pc % (BN / 2);
      if (pc < BK * BN / 2) *(d2u_t *)&Xs[buf][kk][2 * cp] = fx[i];
    }
  };
  // set (s % D) holds stage s; stage 0 goes straight into LDS, sets 1 .. D - 1 and then 0 are loaded (stages past the end: zeros)
  fetch(0, ra[0], rx[0]);
  stash(0, ra[0], rx[0]);
#pragma unroll
  for (int d = 1; d < D; ++d) fetch(d, ra[d], rx[d]);
  __syncthreads();
  for (int st0 = 0; st0 < nstage; st0 += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const int st = st0 + i;
      if (st < nstage) {  // (uniform)
        const int buf = st & 1;
        fetch(st + D, ra[i], rx[i]);  // the set stage st came from is free: D stages ahead
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) {
          const d2u_t a = *(const d2u_t *)&As[buf][q * 4 + lk][rg * 32 + 2 * lr];
          const d2u_t b = *(const d2u_t *)&Xs[buf][q * 4 + lk][cg * 32 + 2 * lr];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            acc[0][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[h], acc[0][h], 0, 0, 0);
            acc[1][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[h], acc[1][h], 0, 0, 0);
          }
        }
        // stage st + 1 into the other buffer (last read in stage st - 1: every wave is past the barrier that ended it)
        if (st + 1 < nstage) stash(buf ^ 1, ra[(i + 1) % D], rx[(i + 1) % D]);
        __syncthreads();
      }
    }
  }
  double *out = raw + (int64_t)kz * raw_stride + poff;
  const int rb = rbase + rg * 32, cbase = col0 + 32 * cg;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb + 2 * (lk + 4 * r) + rt;
        if (row < n) out[(int64_t)row * PW + cbase + 2 * lr + h] = acc[rt][h][r];
      }
}

// k_dense_mfma32_lds (r03): the fp32 dense operator on the matrix cores (reference: eigen_operators.h:24-30 with F = float,
// _lanczos.cpp:104). v_mfma_f32_16x16x4_f32 is an exact f32 fma chain in k order (cdna_hip_programming.md §3), so the result is
// what the VALU kernel k_dense_panel computes up to the K split; it runs at the f32 vector peak (157 TFLOP/s) with ONE register
// per operand, where the VALU kernel re-reads the probe panel from L2 for every output row.
// Same machine as k_dense_mfma_lds with the f32 shapes: a wave owns 32 rows x 64 columns (2 x 4 accumulator tiles; rows
// interleaved by 2, columns by 4: fragment row m of row tile rt is row rb + 2m + rt, fragment column c of column tile h is column
// 4c + h), so ONE ds_read_b64 gives a lane both row tiles' A elements and ONE ds_read_b128 all four column tiles' panel elements:
// 2 LDS reads feed 8 MFMAs. A workgroup = 8 waves = 256 rows x 64 columns; K walks in stages of 16 through two LDS buffers (A
// block 16 x 256, panel block 16 x 64), the next stage's 20 KiB in registers while this stage's 64 MFMAs per wave run. The rows
// of the A buffer are padded by 32 words: a ds_read_b64 serves 32 lanes per pass - two k values of the fragment - and the pad
// puts them on the two halves of the banks. Output as k_dense_mfma_lds: raw partial products in K-split slabs, summed in slab
// order by k_3term_slabs. a_vec = 1 when A's columns are 16-byte aligned (lda % 4 == 0): 16-byte loads; else element loads.
typedef float f4v_t __attribute__((ext_vector_type(4)));
typedef float f2v_t __attribute__((ext_vector_type(2)));
constexpr int kDense32BM = 256, kDense32BN = 64, kDense32BK = 16, kDense32Pad = 32;
__global__ __launch_bounds__(kBlock) void k_dense_mfma32_lds(int n, const float *__restrict__ A, int64_t lda, int a_vec, const float *__restrict__ X, int ldw,
                                                             int col0, float *__restrict__ raw, int64_t raw_stride) {
  constexpr int BM = kDense32BM, BN = kDense32BN, BK = kDense32BK, LDS_A = BM + kDense32Pad;
  constexpr int APT = (BK * BM / 4) / kBlock;  // 16-byte pieces of the A block per thread (2); the panel block is 256 pieces: threads 0..255
  static_assert((BK * BM / 4) % kBlock == 0 && BK * BN / 4 <= kBlock && kWaves * 32 == BM, "block shapes divide over the threads");
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDS_A];
  __shared__ __attribute__((aligned(16))) float Xs[2][BK][BN];
  const int PW = ldw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int panel = blockIdx.y;
  const int rbase = blockIdx.x * BM;
  const int64_t poff = (int64_t)panel * n * PW;
  const float *xp = X + poff + col0;
  const int ncols = min(BN, PW - col0);  // panels of 16 / 32 columns: the columns beyond are fed zeros and not written
  f4v_t acc[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int h = 0; h < 4; ++h) acc[rt][h] = (f4v_t)0.0f;
  const int kq_all = (n + 3) / 4;
  const int ks = gridDim.z, kz = blockIdx.z;
  const int perz = (kq_all + ks - 1) / ks;
  const int k_begin = min(n, kz * perz * 4), k_end = min(n, k_begin + perz * 4);
  const int nstage = (k_end - k_begin + BK - 1) / BK;
  constexpr int D = kDenseDepth;  // register sets ahead of the stage in LDS, as in k_dense_mfma_lds (12 VGPRs each)
  f4v_t ra[D][APT], rx[D];
  auto fetch = [&](int st, f4v_t (&fa)[APT], f4v_t &fx) {
    const int k0 = k_begin + st * BK;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int pc = threadIdx.x + i * kBlock, kk = pc / (BM / 4), r4 = pc % (BM / 4);
      const int k = k0 + kk, row = rbase + 4 * r4;
      f4v_t v = (f4v_t)0.0f;
      if (k < k_end && row < n) {
        const float *ap = A + (int64_t)k * lda + row;
        if (a_vec && row + 3 < n) v = *(const f4v_t *)ap;
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (row + e < n) v[e] = ap[e];
        }
      }
      fa[i] = v;
    }
    {
      const int pc = threadIdx.x, kk = pc / (BN / 4), c4 = pc % (BN / 4);
      const int k = k0 + kk;
      fx = (f4v_t)0.0f;
      if (pc < BK * BN / 4 && k < k_end && 4 * c4 < ncols) fx = *(const f4v_t *)(xp + (int64_t)k * PW + 4 * c4);
    }
  };
  auto stash = [&](int buf, const f4v_t (&fa)[APT], const f4v_t &fx) {
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int pc = threadIdx.x + i * kBlock, kk = pc / (BM / 4), r4 = pc % (BM / 4);
      *(f4v_t *)&As[buf][kk][4 * r4] = fa[i];
    }
    const int pc = threadIdx.x, kk = pc / (BN / 4), c4 = pc % (BN / 4);
    if (pc < BK * BN / 4) *(f4v_t *)&Xs[buf][kk][4 * c4] = fx;
  };
  fetch(0, ra[0], rx[0]);
  stash(0, ra[0], rx[0]);
#pragma unroll
  for (int d = 1; d < D; ++d) fetch(d, ra[d], rx[d]);
  __syncthreads();
  for (int st0 = 0; st0 < nstage; st0 += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const int st = st0 + i;
      if (st < nstage) {  // (uniform)
        const int buf = st & 1;
        fetch(st + D, ra[i], rx[i]);
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) {
          const f2v_t a = *(const f2v_t *)&As[buf][q * 4 + lk][wave * 32 + 2 * lr];
          const f4v_t b = *(const f4v_t *)&Xs[buf][q * 4 + lk][4 * lr];
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            acc[0][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[h], acc[0][h], 0, 0, 0);
            acc[1][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[h], acc[1][h], 0, 0, 0);
          }
        }
        if (st + 1 < nstage) stash(buf ^ 1, ra[(i + 1) % D], rx[(i + 1) % D]);
        __syncthreads();
      }
    }
  }
  // C/D of the f32 16x16x4 form: column lane & 15, row 4 (lane >> 4) + reg (the f64 form differs: cdna_hip_programming.md §3)
  float *out = raw + (int64_t)kz * raw_stride + poff + col0 + 4 * lr;
  const int rb = rbase + wave * 32;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rb + 2 * (4 * lk + r) + rt;
      if (row < n && 4 * lr < ncols) *(f4v_t *)(out + (int64_t)row * PW) = (f4v_t){acc[rt][0][r], acc[rt][1][r], acc[rt][2][r], acc[rt][3][r]};
    }
}

// Three-term epilogue for operators whose product is computed by a separate kernel (dense,
// host callback): in: T = A (Wc) unscaled. w = sc*T - cp*Wp ; partA += (sc*Wc) * w ; Wn = w.
template <typename F, int LPR>
__global__ __launch_bounds__(kBlock) void k_3term(int n, const F *T, const F *Wc, const F *Wp,
                                                  F *Wn, const double *__restrict__ coefA,
                                                  double *__restrict__ partA, int bpad,
                                                  int first) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  __shared__ double red[kWaves * 64 * V];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + cl * V;
  const int colbase = panel * PW + cl * V;
  VF sc, cp;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    sc[v] = (F)coefA[colbase + v];
    cp[v] = (F)coefA[bpad + colbase + v];
  }
  VF aacc = (VF)(F)0;
  const int stride = gridDim.x * kWaves * RPW;
  for (int r0 = (blockIdx.x * kWaves + wave) * RPW; r0 < n; r0 += stride) {
    const int row = r0 + g;
    if (row < n) {
      const int64_t ro = poff + (int64_t)row * PW;
      VF w = sc * *(const VF *)(T + ro);
      if (!first) w -= cp * *(const VF *)(Wp + ro);
      aacc += (sc * *(const VF *)(Wc + ro)) * w;
      *(VF *)(Wn + ro) = w;
    }
  }
  block_reduce_columns<F, LPR>(aacc, red, partA + (int64_t)blockIdx.x * bpad + panel * PW);
}

// The same epilogue over a product that arrives as `nslab` partial slabs (K split over workgroups, k_dense_mfma32):
// T = sum of the slabs in slab order; plain != 0: Wn = T only.
template <typename F, int LPR>
__global__ __launch_bounds__(kBlock) void k_3term_slabs(int n, const F *T, int nslab, int64_t slab_stride, const F *Wc, const F *Wp,
                                                        F *Wn, const double *__restrict__ coefA, double *__restrict__ partA,
                                                        int bpad, int first, int plain) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  __shared__ double red[kWaves * 64 * V];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + cl * V;
  const int colbase = panel * PW + cl * V;
  VF sc = (VF)(F)1, cp = (VF)(F)0;
  if (!plain) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      sc[v] = (F)coefA[colbase + v];
      cp[v] = (F)coefA[bpad + colbase + v];
    }
  }
  VF aacc = (VF)(F)0;
  const int stride = gridDim.x * kWaves * RPW;
  for (int r0 = (blockIdx.x * kWaves + wave) * RPW; r0 < n; r0 += stride) {
    const int row = r0 + g;
    if (row < n) {
      const int64_t ro = poff + (int64_t)row * PW;
      VF t = *(const VF *)(T + ro);
      for (int z = 1; z < nslab; ++z) t += *(const VF *)(T + (int64_t)z * slab_stride + ro);
      VF w = sc * t;
      if (!plain) {
        if (!first) w -= cp * *(const VF *)(Wp + ro);
        aacc += (sc * *(const VF *)(Wc + ro)) * w;
      }
      *(VF *)(Wn + ro) = w;
    }
  }
  if (!plain) block_reduce_columns<F, LPR>(aacc, red, partA + (int64_t)blockIdx.x * bpad + panel * PW);
}

// ---- sweep B (orth = 0): w -= cB * Wc ; partN += w^2 --------------------------------------------
// (reference: lanczos.h:130,139 — v -= alpha q_c ; beta = ||v||, with cB = alpha / nu_j.)
// mode 0: as above.  mode 1: norm only (no update; used for ||v||^2 of the probes).
template <typename F, int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void k_axpy_norm(int n, F *W, const F *Wc,
                                                      const double *__restrict__ coefB,
                                                      double *__restrict__ partN, int bpad) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  __shared__ double red[kWaves * 64 * V];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + cl * V;
  const int colbase = panel * PW + cl * V;
  VF cb = (VF)(F)0;
  if (MODE == 0) {
#pragma unroll
    for (int v = 0; v < V; ++v) cb[v] = (F)coefB[colbase + v];
  }
  VF nacc = (VF)(F)0;
  // One row per wave per iteration, consecutive waves on consecutive rows: the grid sweeps one
  // contiguous window. (Unrolling over far-apart rows, i.e. several windows a power of two apart,
  // measured 25 % SLOWER on MI355X: scripts/microbench_stream.hip.)
  const int stride = gridDim.x * kWaves * RPW;
  for (int row = (blockIdx.x * kWaves + wave) * RPW + g; row < n; row += stride) {
    const int64_t ro = poff + (int64_t)row * PW;
    VF x = *(const VF *)(W + ro);
    if (MODE == 0) {
      x -= cb * *(const VF *)(Wc + ro);
      *(VF *)(W + ro) = x;
    }
    nacc += x * x;
  }
  block_reduce_columns<F, LPR>(nacc, red, partN + (int64_t)blockIdx.x * bpad + panel * PW);
}

// ---- sweep B (orth > 0): w -= cB*Wc ; partD[i] += W_{t_i} . w ------------------------------------
// apply_axpy: 0 - w is final as stored; 1 - the three-term step's `w -= cB W_c` (lanczos.h:130) is applied AND stored (first
// chunk of the MGS-order and fp32-archive sequences, which update w in place between dots); 2 (r04) - it is applied in
// registers only: the sweep stays read-only, and k_reorth_update applies the same term (same fma, same order: bitwise the
// same w) together with its projections and makes the step's one store. A store among 17 read streams costs this sweep a
// fifth of its rate whatever its flavour (scripts/microbench_reorth.hip: 17 reads 6.5-6.9 TB/s, 17 reads + 1 write 5.2-5.4),
// a second read of W_c in a later chunk costs 1/16.
// The reorthogonalisation set is the last r ring vectors t_i = j - i, i = 0..r-1 (reference:
// orth_vector(v, Q, c, orth, reverse=true), lanczos.h:135 with :58). The reference runs modified
// Gram-Schmidt (r sequentially dependent dot+axpy passes, lanczos.h:58-65); here all r dots are
// taken against the same w in ONE pass (block classical Gram-Schmidt as the second
// orthogonalisation pass after the three-term step), r_chunk <= kReorthChunk columns per launch
// so the accumulators stay in registers.
// ring: base of slot 0; vector t lives in slot t % S; slot stride = NP*n*PW elements.
template <typename F, int LPR>
__global__ __launch_bounds__(kBlock) void k_reorth_dot(
    int n, F *ring, int64_t slot_stride, int S, int j, int i0, int rc, int apply_axpy,
    const double *__restrict__ coefB, double *__restrict__ partD /* [rc][nblk][bpad] */, int bpad) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  __shared__ double red[kWaves * 64 * V];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + cl * V;
  const int colbase = panel * PW + cl * V;
  F *W = ring + (int64_t)((j + 1) % S) * slot_stride + poff;
  const F *Wc = ring + (int64_t)(j % S) * slot_stride + poff;
  const F *U0 = ring + poff;
  VF cb = (VF)(F)0;
  if (apply_axpy) {
#pragma unroll
    for (int v = 0; v < V; ++v) cb[v] = (F)coefB[colbase + v];
  }
  VF dacc[kReorthChunk];
#pragma unroll
  for (int i = 0; i < kReorthChunk; ++i) dacc[i] = (VF)(F)0;
  const int stride = gridDim.x * kWaves * RPW;
  for (int row = (blockIdx.x * kWaves + wave) * RPW + g; row < n; row += stride) {
    const int64_t ro = (int64_t)row * PW;
    VF w = stream_load<SLQ_SWEEP_LDW>((const VF *)(W + ro));
    if (apply_axpy) {
      w -= cb * *(const VF *)(Wc + ro);
      if (apply_axpy == 1) stream_store<SLQ_SWEEP_ST>((VF *)(W + ro), w);  // (2: in registers only - k_reorth_update applies it again and stores)
    }
    VF u[kReorthChunk];
#pragma unroll
    for (int i = 0; i < kReorthChunk; ++i)
      if (i < rc) u[i] = *(const VF *)(U0 + (int64_t)ring_slot(j - i0 - i, S) * slot_stride + ro);
#pragma unroll
    for (int i = 0; i < kReorthChunk; ++i)
      if (i < rc) dacc[i] += u[i] * w;
  }
  const int64_t nblk = gridDim.x;
#pragma unroll
  for (int i = 0; i < kReorthChunk; ++i)
    if (i < rc)
      block_reduce_columns<F, LPR>(dacc[i], red,
                                   partD + ((int64_t)i * nblk + blockIdx.x) * bpad + panel * PW);
}

// ---- sweep C: w -= sum_i gamma_i * W_{t_i} ; partN += w^2 ----------------------------------------
// gamma[i][col] (coefficient on the UNNORMALISED ring vector; zero where the reference's skip
// thresholds apply, lanczos.h:62) is staged in LDS once per block: r * PW doubles.
template <typename F, int LPR>
__global__ __launch_bounds__(kBlock) void k_reorth_update(
    int n, F *ring, int64_t slot_stride, int S, int j, int i0, int r,
    const double *__restrict__ gamma /* [r][bpad], already offset to column i0 */,
    double *__restrict__ partN, int bpad, const double *__restrict__ coefB /* non-null: w -= cB W_c first (k_reorth_dot mode 2) */,
    unsigned long long *__restrict__ cols_stat /* null, or {columns read, columns offered} summed over launches and panels (byte accounting) */,
    int skip_zero_cols /* 0: every column is read (SLQ_SWEEP_SKIP=0: A/B and the bitwise check) */) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  constexpr int UR = SLQ_UPD_UR;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double *red = (double *)lds_raw;                       // kWaves*64*V doubles
  F *gl = (F *)(lds_raw + sizeof(double) * kWaves * 64 * V); // r * PW
  int *fl = (int *)(gl + (size_t)r * PW);                    // r flags: column i has a non-zero coefficient for SOME probe of this panel
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + cl * V;
  for (int t = threadIdx.x; t < r; t += kBlock) fl[t] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < r * PW; t += kBlock) {
    const F gv = (F)gamma[(int64_t)(t / PW) * bpad + panel * PW + (t % PW)];
    gl[t] = gv;
    if (gv != (F)0) fl[t / PW] = 1;  // (every writer writes 1)
  }
  __syncthreads();
  // Columns whose coefficient is zero for EVERY probe of the panel are not read at all (r04). The reference skips a projection per probe when it is below its
  // threshold (lanczos.h:62) - in a well-conditioned run that is nearly every column of a deep window - and `w -= 0 * x` is w for finite x: bitwise the same result,
  // minus the panel read. The flags as wave-uniform bit masks (r <= 192: update_chunk_cols).
  unsigned long long live_cols[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) live_cols[q] = __builtin_amdgcn_ballot_w64(q * 64 + lane < r && fl[min(q * 64 + lane, r - 1)] != 0);
  if (coefB) live_cols[0] |= 1ull;  // (column 0 carries the deferred three-term axpy)
  if (!skip_zero_cols) live_cols[0] = live_cols[1] = live_cols[2] = ~0ull;
  if (cols_stat && blockIdx.x == 0 && threadIdx.x == 0) {
    const int nlive = __builtin_popcountll(live_cols[0]) + __builtin_popcountll(live_cols[1]) + __builtin_popcountll(live_cols[2]);
    atomicAdd(cols_stat, (unsigned long long)(skip_zero_cols ? nlive : r));
    atomicAdd(cols_stat + 1, (unsigned long long)r);
  }
  F *W = ring + (int64_t)((j + 1) % S) * slot_stride + poff;
  const F *U0 = ring + poff;
  VF cb = (VF)(F)0;
  if (coefB) {
#pragma unroll
    for (int v = 0; v < V; ++v) cb[v] = (F)coefB[panel * PW + cl * V + v];
  }
  VF nacc = (VF)(F)0;
  // UR CONSECUTIVE row groups per wave and iteration (one contiguous UR*RPW*PW*sizeof(F) block):
  // the gamma read from LDS is amortised over UR rows and UR loads per column are in flight.
  const int stride = gridDim.x * kWaves * RPW * UR;
  for (int r0 = (blockIdx.x * kWaves + wave) * RPW * UR + g; r0 < n; r0 += stride) {
    VF w[UR];
    int64_t ro[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int row = r0 + u * RPW;
      ro[u] = (int64_t)(row < n ? row : 0) * PW;
      w[u] = stream_load<SLQ_SWEEP_LDW>((const VF *)(W + ro[u]));
    }
    for (int i = 0; i < r; ++i) {
      if (!((live_cols[i >> 6] >> (i & 63)) & 1ull)) continue;
      const F *U = U0 + (int64_t)ring_slot(j - i0 - i, S) * slot_stride;
      const VF gm = *(const VF *)(gl + i * PW + cl * V);
      VF x[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) x[u] = *(const VF *)(U + ro[u]);
      if (coefB && i == 0) {
        // the deferred three-term axpy on the row the sweep holds anyway: column 0 of the first chunk IS W_c (the host passes coefB with
        // i0 = 0 only). First cB, then gamma_0, as the stored form did it: bitwise the same w. (A load of its own for this term - the first
        // version - made the sweep 4-6 % slower: two requests for one row.)
#pragma unroll
        for (int u = 0; u < UR; ++u) w[u] -= cb * x[u];
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) w[u] -= gm * x[u];
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int row = r0 + u * RPW;
      if (row < n) {
        stream_store<SLQ_SWEEP_ST>((VF *)(W + ro[u]), w[u]);
        nacc += w[u] * w[u];
      }
    }
  }
  block_reduce_columns<F, LPR>(nacc, red, partN + (int64_t)blockIdx.x * bpad + panel * PW);
}

// ---- opt-in fp32 archive of finished Lanczos vectors (SLQ_RING32, fp64 plans) ---------------------
// The fp64 ring holds only the three live vectors (slots t % 3); every finished vector is also kept as fp32 in an archive
// ring (slot t % S32, same panel layout). Reorthogonalisation columns i >= 2 (vectors j-2 and older) are read from the
// archive - half the bytes - and accumulated in fp64. For the archive rows to be read at full width (16 bytes per lane)
// these sweeps use their own lane layout: a lane owns FOUR probe columns (4 floats of the archive = one 16-byte load, 4
// doubles of an fp64 vector = two), LR = PW/4 lanes per row, RW = 64/LR rows per wave instruction.
typedef double d2a_t __attribute__((ext_vector_type(2)));
typedef float f4a_t __attribute__((ext_vector_type(4)));
#ifndef SLQ_CHUNK32
#define SLQ_CHUNK32 8
#endif
constexpr int kReorthChunk32 = SLQ_CHUNK32;  // reorth columns per dots launch (4 accumulators per column and lane)

template <int LPR> struct Geo32 {
  static constexpr int PW = LPR * 2, LR = PW / 4, RW = 64 / LR;
};
// column sums of 4 partials per lane over the RW row groups of a wave and the waves of the block, fixed order
template <int LPR>
__device__ __forceinline__ void block_reduce_columns4(const double (&acc)[4], double *red /* kWaves*64*4 */, double *out) {
  constexpr int PW = Geo32<LPR>::PW, LR = Geo32<LPR>::LR, RW = Geo32<LPR>::RW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < 4; ++v) red[(wave * 64 + lane) * 4 + v] = acc[v];
  __syncthreads();
  if ((int)threadIdx.x < PW) {
    const int t = threadIdx.x, c4 = t / 4, v = t % 4;
    double s = 0.0;
    for (int w = 0; w < kWaves; ++w)
#pragma unroll
      for (int g = 0; g < RW; ++g) s += red[(w * 64 + g * LR + c4) * 4 + v];
    out[t] = s;
  }
  __syncthreads();
}
// ring column gi of step j as 4 doubles: W_c / W_p from the fp64 ring, older vectors from the archive
__device__ __forceinline__ void ring_column4(const double *ring, int64_t slot_stride, int S, const float *ring32, int64_t stride32, int S32,
                                             int64_t off, int t, int gi, double (&u)[4]) {
  if (gi >= 2) {
    const f4a_t x = *(const f4a_t *)(ring32 + (int64_t)ring_slot(t, S32) * stride32 + off);
#pragma unroll
    for (int v = 0; v < 4; ++v) u[v] = (double)x[v];
  } else {
    const double *p = ring + (int64_t)ring_slot(t, S) * slot_stride + off;
    const d2a_t a = *(const d2a_t *)p, b = *(const d2a_t *)(p + 2);
    u[0] = a[0]; u[1] = a[1]; u[2] = b[0]; u[3] = b[1];
  }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void k_reorth_dot32(
    int n, double *ring, int64_t slot_stride, int S, int j, int i0, int rc, int apply_axpy, const double *__restrict__ coefB,
    double *__restrict__ partD /* [rc][nblk][bpad] */, int bpad, const float *__restrict__ ring32, int64_t stride32, int S32) {
  constexpr int PW = Geo32<LPR>::PW, LR = Geo32<LPR>::LR, RW = Geo32<LPR>::RW;
  __shared__ double red[kWaves * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LR, c4 = lane % LR;
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + c4 * 4;
  double *W = ring + (int64_t)((j + 1) % S) * slot_stride + poff;
  const double *Wc = ring + (int64_t)(j % S) * slot_stride + poff;
  double cb[4] = {0.0, 0.0, 0.0, 0.0};
  if (apply_axpy) {
#pragma unroll
    for (int v = 0; v < 4; ++v) cb[v] = coefB[panel * PW + c4 * 4 + v];
  }
  double dacc[kReorthChunk32][4];
#pragma unroll
  for (int i = 0; i < kReorthChunk32; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) dacc[i][v] = 0.0;
  const int stride = gridDim.x * kWaves * RW;
  for (int row = (blockIdx.x * kWaves + wave) * RW + g; row < n; row += stride) {
    const int64_t ro = (int64_t)row * PW;
    d2a_t wa = *(const d2a_t *)(W + ro), wb = *(const d2a_t *)(W + ro + 2);
    if (apply_axpy) {
      const d2a_t ca = *(const d2a_t *)(Wc + ro), cc = *(const d2a_t *)(Wc + ro + 2);
      wa[0] -= cb[0] * ca[0]; wa[1] -= cb[1] * ca[1]; wb[0] -= cb[2] * cc[0]; wb[1] -= cb[3] * cc[1];
      *(d2a_t *)(W + ro) = wa;
      *(d2a_t *)(W + ro + 2) = wb;
    }
    const double w[4] = {wa[0], wa[1], wb[0], wb[1]};
    double u[kReorthChunk32][4];
#pragma unroll
    for (int i = 0; i < kReorthChunk32; ++i)
      if (i < rc) ring_column4(ring, slot_stride, S, ring32, stride32, S32, poff + ro, j - i0 - i, i0 + i, u[i]);
#pragma unroll
    for (int i = 0; i < kReorthChunk32; ++i)
      if (i < rc) {
#pragma unroll
        for (int v = 0; v < 4; ++v) dacc[i][v] += u[i][v] * w[v];
      }
  }
  const int64_t nblk = gridDim.x;
#pragma unroll
  for (int i = 0; i < kReorthChunk32; ++i)
    if (i < rc) block_reduce_columns4<LPR>(dacc[i], red, partD + ((int64_t)i * nblk + blockIdx.x) * bpad + panel * PW);
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void k_reorth_update32(
    int n, double *ring, int64_t slot_stride, int S, int j, int i0, int r, const double *__restrict__ gamma /* [r][bpad], offset to i0 */,
    double *__restrict__ partN, int bpad, float *ring32, int64_t stride32, int S32, int archive /* w is final: store it as fp32 too */,
    unsigned long long *__restrict__ cols_stat, int skip_zero_cols /* as k_reorth_update */) {
  constexpr int PW = Geo32<LPR>::PW, LR = Geo32<LPR>::LR, RW = Geo32<LPR>::RW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double *red = (double *)lds_raw;               // kWaves*64*4 doubles
  double *gl = red + kWaves * 64 * 4;            // r * PW
  int *fl = (int *)(gl + (size_t)r * PW);        // r flags, then the list of the columns to read and its length (r + 1 ints)
  int *lst = fl + r;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LR, c4 = lane % LR;
  const int panel = blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + c4 * 4;
  for (int t = threadIdx.x; t < r; t += kBlock) fl[t] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < r * PW; t += kBlock) {
    const double gv = gamma[(int64_t)(t / PW) * bpad + panel * PW + (t % PW)];
    gl[t] = gv;
    if (gv != 0.0) fl[t / PW] = 1;
  }
  __syncthreads();
  // columns nobody of this panel projects on are not read (k_reorth_update, r04): the list of the others, ascending - the order of the subtractions is kept
  if (threadIdx.x == 0) {
    int m = 0;
    for (int i = 0; i < r; ++i)
      if (fl[i] || !skip_zero_cols) lst[m++] = i;
    lst[r] = m;
    if (cols_stat && blockIdx.x == 0) {
      atomicAdd(cols_stat, (unsigned long long)m);
      atomicAdd(cols_stat + 1, (unsigned long long)r);
    }
  }
  __syncthreads();
  const int nlive = lst[r];
  double *W = ring + (int64_t)((j + 1) % S) * slot_stride + poff;
  float *W32 = ring32 + (int64_t)ring_slot(j + 1, S32) * stride32 + poff;
  double nacc[4] = {0.0, 0.0, 0.0, 0.0};
  const int stride = gridDim.x * kWaves * RW;
  for (int row = (blockIdx.x * kWaves + wave) * RW + g; row < n; row += stride) {
    const int64_t ro = (int64_t)row * PW;
    const d2a_t wa = *(const d2a_t *)(W + ro), wb = *(const d2a_t *)(W + ro + 2);
    double w[4] = {wa[0], wa[1], wb[0], wb[1]};
    int ii = 0;
    for (; ii + 4 <= nlive; ii += 4) {  // four columns' loads in flight
      double u[4][4];
      int ci[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ci[q] = lst[ii + q];
        ring_column4(ring, slot_stride, S, ring32, stride32, S32, poff + ro, j - i0 - ci[q], i0 + ci[q], u[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const d2a_t ga = *(const d2a_t *)(gl + ci[q] * PW + c4 * 4), gb = *(const d2a_t *)(gl + ci[q] * PW + c4 * 4 + 2);
        w[0] -= ga[0] * u[q][0]; w[1] -= ga[1] * u[q][1]; w[2] -= gb[0] * u[q][2]; w[3] -= gb[1] * u[q][3];
      }
    }
    for (; ii < nlive; ++ii) {
      const int i = lst[ii];
      double u[4];
      ring_column4(ring, slot_stride, S, ring32, stride32, S32, poff + ro, j - i0 - i, i0 + i, u);
      const d2a_t ga = *(const d2a_t *)(gl + i * PW + c4 * 4), gb = *(const d2a_t *)(gl + i * PW + c4 * 4 + 2);
      w[0] -= ga[0] * u[0]; w[1] -= ga[1] * u[1]; w[2] -= gb[0] * u[2]; w[3] -= gb[1] * u[3];
    }
    d2a_t oa, ob;
    oa[0] = w[0]; oa[1] = w[1]; ob[0] = w[2]; ob[1] = w[3];
    *(d2a_t *)(W + ro) = oa;
    *(d2a_t *)(W + ro + 2) = ob;
    if (archive) {
      f4a_t y;
#pragma unroll
      for (int v = 0; v < 4; ++v) y[v] = (float)w[v];
      *(f4a_t *)(W32 + ro) = y;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) nacc[v] += w[v] * w[v];
  }
  block_reduce_columns4<LPR>(nacc, red, partN + (int64_t)blockIdx.x * bpad + panel * PW);
}

// fp64 panel slot -> fp32 archive slot (the probes, vector 0 of a run)
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_archive32(int n, const double *__restrict__ W, float *__restrict__ W32) {
  constexpr int PW = Geo<double, LPR>::PW, RPW = Geo<double, LPR>::RPW;
  typedef double d2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int64_t poff = (int64_t)blockIdx.y * n * PW + cl * 2;
  const int stride = gridDim.x * kWaves * RPW;
  for (int row = (blockIdx.x * kWaves + wave) * RPW + g; row < n; row += stride) {
    const d2_t x = *(const d2_t *)(W + poff + (int64_t)row * PW);
    f2_t y;
    y[0] = (float)x[0];
    y[1] = (float)x[1];
    *(f2_t *)(W32 + poff + (int64_t)row * PW) = y;
  }
}

// ---- per-step scalar kernels ("finalize"): partials -> alpha / beta / next coefficients ---------
struct StepState {
  double *alpha;   // [deg+1][bpad]
  double *nu;      // [deg+1][bpad]  nu[0] = ||v||, nu[t] = beta_t
  double *vnorm2;  // [bpad]        ||v||^2 used in the final scaling
  double *coefA;   // [2][bpad]     sc, cp for the next sweep A
  double *coefB;   // [bpad]        cB for sweep B
  double *gamma;   // [rmax][bpad]
  double *cross;   // [bpad]        W_{j+1} . W_j from the fused update pass (k_csr_pass xt)
  double *gram;    // [2][kFusedMaxR + 1][bpad]  Gram rows of the last two vectors written (gram sequence): set t & 1, entry q = W_t . W_{t-q}
  int *active;     // [bpad]
  int *steps;      // [bpad]
  int bpad, nprobes, deg;
};

// Sum part[blk][col] over blk in a fixed order. Block = kFinThreads = 64 columns x kFinSlices
// slices; each thread adds every kFinSlices-th partial (4 independent loads in flight), then the
// slices are folded through LDS in slice order: bitwise reproducible.
constexpr int kFinSlices = 16;
constexpr int kFinThreads = 64 * kFinSlices;
__device__ __forceinline__ double sum_partials(const double *__restrict__ part, int nblk, int bpad,
                                               int col, double *red /* kFinThreads doubles */) {
  const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (col < bpad) {
    int b = s;
    for (; b + 3 * kFinSlices < nblk; b += 4 * kFinSlices) {
      a0 += part[(int64_t)b * bpad + col];
      a1 += part[(int64_t)(b + kFinSlices) * bpad + col];
      a2 += part[(int64_t)(b + 2 * kFinSlices) * bpad + col];
      a3 += part[(int64_t)(b + 3 * kFinSlices) * bpad + col];
    }
    for (; b < nblk; b += kFinSlices) a0 += part[(int64_t)b * bpad + col];
  }
  red[s * 64 + c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int k = 0; k < kFinSlices; ++k) tot += red[k * 64 + c];
  __syncthreads();
  return tot;
}

// After the probe-norm sweep: nu_0 = ||v||, first coefficients, activity flags.
__global__ __launch_bounds__(kFinThreads) void k_fin_init(StepState st, const double *__restrict__ partN,
                                                  int nblk, int sphere, double n_as_double) {
  __shared__ double red4[kFinThreads];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  // nblk = 0: Rademacher probes drawn on the device - ||v||^2 = n exactly (what the norm sweep sums to as well: integers are exact), no sweep
  const double s = nblk > 0 ? sum_partials(partN, nblk, st.bpad, col, red4) : (col < st.nprobes ? n_as_double : 0.0);
  if ((threadIdx.x >> 6) == 0 && col < st.bpad) {
    const double nu0 = sqrt(s);
    const int act = (col < st.nprobes) && (nu0 > 0.0);
    st.nu[col] = nu0;
    // sphere probes are sqrt(n) g/||g|| (src/primate/random.py:36-41): same direction, norm^2 = n
    st.vnorm2[col] = sphere ? n_as_double : s;
    st.active[col] = act;
    st.steps[col] = 0;
    st.coefA[col] = act ? 1.0 / nu0 : 0.0;
    st.coefA[st.bpad + col] = 0.0;
  }
}

// After sweep A of step j: alpha_j, and the sweep-B coefficient cB = alpha_j / nu_j.
__global__ __launch_bounds__(kFinThreads) void k_fin_alpha(StepState st, const double *__restrict__ partA,
                                                   int nblk, int j, int xt) {
  __shared__ double red4[kFinThreads];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  double a = sum_partials(partA, nblk, st.bpad, col, red4);
  if ((threadIdx.x >> 6) == 0 && col < st.bpad) {
    const int act = st.active[col];
    // the alpha pass left out -beta q_c.q_p: q_c = sc W_c, beta q_p = cp W_p, W_c.W_p = cross
    if (xt) a -= st.coefA[col] * st.coefA[st.bpad + col] * st.cross[col];
    if (act) st.alpha[(int64_t)j * st.bpad + col] = a;
    st.coefB[col] = act ? a / st.nu[(int64_t)j * st.bpad + col] : 0.0;
  }
}

// After the last sweep of step j: beta_{j+1} = ||w||, stop rule (lanczos.h:139-142), next sc/cp.
__global__ __launch_bounds__(kFinThreads) void k_fin_beta(StepState st, const double *__restrict__ partN,
                                                  int nblk, int j, double residual_tol, int xt) {
  __shared__ double red4[kFinThreads];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const double s = sum_partials(partN, nblk, st.bpad, col, red4);
  if (xt) {
    const double x = sum_partials(partN + (int64_t)nblk * st.bpad, nblk, st.bpad, col, red4);
    if ((threadIdx.x >> 6) == 0 && col < st.bpad) st.cross[col] = x;
  }
  if ((threadIdx.x >> 6) == 0 && col < st.bpad) {
    double sc = 0.0, cp = 0.0;
    if (st.active[col]) {
      const double beta = sqrt(s);
      st.nu[(int64_t)(j + 1) * st.bpad + col] = beta;
      st.steps[col] = j + 1;
      if (beta < residual_tol || (j + 1) == st.deg || !(beta == beta)) {
        st.active[col] = 0;
      } else {
        sc = 1.0 / beta;
        cp = beta / st.nu[(int64_t)j * st.bpad + col];
      }
    }
    st.coefA[col] = sc;
    st.coefA[st.bpad + col] = cp;
  }
}

// After the merged alpha+dots pass of step j (PASS_ADOTS, RC ring columns): blockIdx.y = i.
//   i = 0: alpha_j, cB = alpha_j / nu_j, gamma_0 = 0;
//   i >= 1: W_t.(u - alpha q_c) = d_i - cB g_i, then the threshold and scaling of k_fin_gamma.
__global__ __launch_bounds__(kFinThreads) void k_fin_adots(StepState st, const double *__restrict__ part, int nblk,
                                                   int j, int RC, double orth_tol) {
  __shared__ double red4[kFinThreads];
  const int i = blockIdx.y;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t slab = (int64_t)nblk * st.bpad;
  const double a = sum_partials(part, nblk, st.bpad, col, red4);
  double d = 0.0, g = 0.0;
  if (i > 0) {
    d = sum_partials(part + i * slab, nblk, st.bpad, col, red4);
    g = sum_partials(part + (RC - 1 + i) * slab, nblk, st.bpad, col, red4);
  }
  if ((threadIdx.x >> 6) == 0 && col < st.bpad) {
    const int act = st.active[col];
    const double nuj = st.nu[(int64_t)j * st.bpad + col];
    const double cb = (act && nuj > 0.0) ? a / nuj : 0.0;
    if (i == 0) {
      if (act) st.alpha[(int64_t)j * st.bpad + col] = a;
      st.coefB[col] = cb;
      st.gamma[col] = 0.0;
    } else {
      const double nu = st.nu[(int64_t)(j - i) * st.bpad + col];
      double gm = 0.0;
      if (act && nu > 0.0) {
        const double sproj = (d - cb * g) / nu;
        if (fabs(sproj) > orth_tol) gm = sproj / nu;
      }
      st.gamma[(int64_t)i * st.bpad + col] = gm;
    }
  }
}

// ---- the Gram sequence (slq.hip: enqueue_run, ring-fed plans, 1 <= r <= kFusedMaxR) -----------------------------------------
// The merged alpha+dots pass reads every ring column once more only to take d_i = W_t . u and g_i = W_t . W_j. Both follow
// from inner products of ring vectors that the UPDATE pass of the step before can take on the fly, while it holds those
// rows in registers anyway (k_ring_pass<PASS_UPDATEG>: the new vector against every column it reads):
//   g_i = W_t . W_j                                                  is such an entry itself, and with A symmetric
//   d_i = W_t . (sc A W_j - cp W_{j-1}) = sc (A W_t) . W_j - cp (W_t . W_{j-1}),
//   A W_t = nu_t [ W_{t+1} + (alpha_t / nu_t) W_t + (nu_t / nu_{t-1}) W_{t-1} + (projections of step t) ]
// (step t's own update, solved for A W_t). The projections of step t are O(eps) coefficients on vectors whose inner
// product with W_j is O(eps) again: dropped. What is left needs G_j(s) = W_s . W_j for s = t+1, t, t-1 - inside the
// window the update pass of step j-1 read - and W_t . W_{j-1} from the row before. The step then is an alpha-only pass
// (one panel read: alpha_j needs the gather) + the update pass: (r + 2) panel reads + 1 write instead of (2r + 1) + 1.
// Numerically d_i is now a difference of large terms where the merged pass took a small quantity directly; measured
// (DESIGN.md §4.6): alpha / beta and the quadrature agree with the oracle exactly as well as before - a window of
// 2-8 columns behind a three-term step only ever removes rounding-level components, most of them below the
// reference's own threshold (lanczos.h:53,62).
// k_fin_gram: blockIdx.y = i as in k_fin_adots. part: alpha partials of the alpha-only pass.
// part2 / nblk2 (may be 0) / raw: the alpha dot taken by the previous step's fused update pass (slq_ring_fa.hpp) and the edge kernel's share of it
// (k_alpha_edges) - raw sums W_j . (A W_j), normalised here by 1 / nu_j^2 (= coefA^2) instead of term by term.
__global__ __launch_bounds__(kFinThreads) void k_fin_gram(StepState st, const double *__restrict__ part, int nblk, int j, int RC, double orth_tol,
                                                          const double *__restrict__ part2, int nblk2, int raw) {
  __shared__ double red4[kFinThreads];
  const int i = blockIdx.y;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  double a_raw = sum_partials(part, nblk, st.bpad, col, red4);
  if (nblk2 > 0) a_raw += sum_partials(part2, nblk2, st.bpad, col, red4);
  if (raw && col < st.bpad) a_raw *= st.coefA[col] * st.coefA[col];
  if ((threadIdx.x >> 6) == 0 && col < st.bpad) {
    constexpr int R1 = kFusedMaxR + 1;
    const int64_t bp = st.bpad;
    const double *Gc = st.gram + (int64_t)(j & 1) * R1 * bp + col;        // Gc[q * bp] = W_j . W_{j-q}
    const double *Gp = st.gram + (int64_t)((j + 1) & 1) * R1 * bp + col;  // Gp[q * bp] = W_{j-1} . W_{j-1-q}
    const int act = st.active[col];
    const double nuj = st.nu[(int64_t)j * bp + col];
    const double sc = st.coefA[col], cp = st.coefA[bp + col];
    // alpha_j = q_j . (A q_j) - beta_j (q_j . q_{j-1}): the pass took the first dot, the second is a Gram entry
    const double a = a_raw - (j > 0 ? sc * cp * Gc[bp] : 0.0);
    const double cb = (act && nuj > 0.0) ? a / nuj : 0.0;
    if (i == 0) {
      if (act) st.alpha[(int64_t)j * bp + col] = a;
      st.coefB[col] = cb;
      st.gamma[col] = 0.0;
    } else {
      const int t = j - i;
      const double nut = st.nu[(int64_t)t * bp + col];
      double gm = 0.0;
      if (act && nut > 0.0) {
        const double g1 = Gc[(int64_t)(i - 1) * bp], g0 = Gc[(int64_t)i * bp];
        const double alt = st.alpha[(int64_t)t * bp + col];
        double s = g1 + (alt / nut) * g0;
        if (t >= 1 && i + 1 <= kFusedMaxR) {
          const double nutm = st.nu[(int64_t)(t - 1) * bp + col];
          if (nutm > 0.0) s += (nut / nutm) * Gc[(int64_t)(i + 1) * bp];  // (zero where the window did not reach W_{t-1}: never in range, see slq.hip)
        }
        // W_t . W_{j-1}: t = j - 1 is the squared norm nu_{j-1}^2 itself (W_0's row is never written: take nu everywhere)
        const double nujm = st.nu[(int64_t)(j - 1) * bp + col];
        const double wt_wjm = i == 1 ? nujm * nujm : Gp[(int64_t)(i - 1) * bp];
        const double d = sc * (nut * s) - cp * wt_wjm;
        const double sproj = (d - cb * g0) / nut;
        if (fabs(sproj) > orth_tol) gm = sproj / nut;
      }
      st.gamma[(int64_t)i * bp + col] = gm;
    }
  }
}

// After the update pass of step j in the Gram sequence: blockIdx.y = 0 does k_fin_beta's job (beta_{j+1}, stop rule, next
// sc / cp) and stores ||W_{j+1}||^2 as entry 0 of the new Gram row; blockIdx.y = q >= 1 stores W_{j+1} . W_{j+1-q}
// (slab q of the pass's partials; the grid has RC + 1 rows of blocks). The next step reads entries 0 .. RC only.
__global__ __launch_bounds__(kFinThreads) void k_fin_beta_gram(StepState st, const double *__restrict__ part, int nblk, int j, int RC,
                                                       double residual_tol) {
  __shared__ double red4[kFinThreads];
  constexpr int R1 = kFusedMaxR + 1;
  const int q = blockIdx.y;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  double s = 0.0;
  if (q <= RC) s = sum_partials(part + (int64_t)q * nblk * st.bpad, nblk, st.bpad, col, red4);
  if ((threadIdx.x >> 6) == 0 && col < st.bpad) {
    st.gram[((int64_t)((j + 1) & 1) * R1 + q) * st.bpad + col] = s;
    if (q == 0) {
      double sc = 0.0, cp = 0.0;
      if (st.active[col]) {
        const double beta = sqrt(s);
        st.nu[(int64_t)(j + 1) * st.bpad + col] = beta;
        st.steps[col] = j + 1;
        if (beta < residual_tol || (j + 1) == st.deg || !(beta == beta)) {
          st.active[col] = 0;
        } else {
          sc = 1.0 / beta;
          cp = beta / st.nu[(int64_t)j * st.bpad + col];
        }
      }
      st.coefA[col] = sc;
      st.coefA[st.bpad + col] = cp;
    }
  }
}

// After a sweep-B chunk: gamma_i = (W_t . w) / nu_t^2 unless |q_t . w| <= 2 eps sqrt(n)
// (lanczos.h:53,62: the projection is skipped when it is below the orthogonality tolerance).
__global__ __launch_bounds__(kFinThreads) void k_fin_gamma(StepState st, const double *__restrict__ partD,
                                                   int nblk, int j, int i0, double orth_tol) {
  __shared__ double red4[kFinThreads];
  const int i = blockIdx.y;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const double d = sum_partials(partD + (int64_t)i * nblk * st.bpad, nblk, st.bpad, col, red4);
  if ((threadIdx.x >> 6) == 0 && col < st.bpad) {
    const int t = j - i0 - i;
    const double nu = st.nu[(int64_t)t * st.bpad + col];
    double gm = 0.0;
    if (st.active[col] && nu > 0.0) {
      const double sproj = d / nu;
      if (fabs(sproj) > orth_tol) gm = sproj / nu;
    }
    st.gamma[(int64_t)(i0 + i) * st.bpad + col] = gm;
  }
}

// ---- probes ------------------------------------------------------------------------------------------
// Column-major host-order staging chunk Xs (n x nc, ld = n, columns c0..c0+nc of the batch)
// -> panel layout slot. 64x64 tiles through LDS so both sides are coalesced.
// perm (optional): panel row i holds caller row perm[i] (operator stored in a permuted row order).
template <typename F>
__global__ __launch_bounds__(256) void k_cols_to_panel(int n, const F *__restrict__ Xs, int c0,
                                                       int nc, F *W, int PW,
                                                       const int32_t *__restrict__ perm) {
  __shared__ F tile[64][65];
  const int r0 = blockIdx.x * 64, cb = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int srow = (r0 + tx < n) ? (perm ? perm[r0 + tx] : r0 + tx) : 0;
  for (int c = ty; c < 64; c += 4) {
    const int col = cb + c, row = r0 + tx;
    tile[c][tx] = (col < nc && row < n) ? Xs[(int64_t)col * n + srow] : (F)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int row = r0 + r, col = cb + tx;
    if (row < n && col < nc) {
      const int gc = c0 + col;
      W[((int64_t)(gc / PW) * n + row) * PW + (gc % PW)] = tile[tx][r];
    }
  }
}

// panel layout -> column-major chunk, optionally scaled per column (scale[gc], or 1 if null)
template <typename F>
__global__ __launch_bounds__(256) void k_panel_to_cols(int n, const F *__restrict__ W, int c0,
                                                       int nc, F *Xs, int PW,
                                                       const double *__restrict__ scale,
                                                       const int32_t *__restrict__ perm) {
  __shared__ F tile[64][65];
  const int r0 = blockIdx.x * 64, cb = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int row = r0 + r, col = cb + tx;
    F x = (F)0;
    if (row < n && col < nc) {
      const int gc = c0 + col;
      x = W[((int64_t)(gc / PW) * n + row) * PW + (gc % PW)];
      if (scale) x = (F)((double)x * scale[gc]);
    }
    tile[r][tx] = x;
  }
  __syncthreads();
  const int drow = (r0 + tx < n) ? (perm ? perm[r0 + tx] : r0 + tx) : 0;
  for (int c = ty; c < 64; c += 4) {
    const int col = cb + c, row = r0 + tx;
    if (col < nc && row < n) Xs[(int64_t)col * n + drow] = tile[tx][c];
  }
}

template <typename F> __global__ void k_fill_zero(F *p, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = (F)0;
}

// Philox4x32-10 (Salmon et al., SC'11), counter-based: the stream of probe id depends only on
// (seed, id), never on the launch geometry or on how probes are sharded over GPUs.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Probe generation straight into the panel layout. A lane owns its V probe columns; one Philox
// block serves 128 consecutive rows (Rademacher: 1 bit each) or 2 rows (normal: Box-Muller on two
// 53-bit uniforms). Padding columns (id >= nprobes) are zero.
//   Rademacher support is exactly {-1,+1} (reference: floor(2u)*2-1, src/primate/random.py:22-29).
// One element of the same stream: probe `id`, caller row `row`. Used when the operator is stored in a
// permuted row order, so that the probes - seen in the caller's order - do not depend on that choice.
template <typename F>
__device__ __forceinline__ F probe_element(int pdf, uint32_t k0, uint32_t k1, uint64_t id, int row) {
  uint32_t r[4];
  if (pdf == 0) {
    philox4x32_10((uint32_t)(row >> 7), 0u, (uint32_t)id, (uint32_t)(id >> 32), k0, k1, r);
    const int b = row & 127;
    return ((r[b >> 5] >> (b & 31)) & 1u) ? (F)1 : (F)-1;
  }
  philox4x32_10((uint32_t)(row >> 1), 1u, (uint32_t)id, (uint32_t)(id >> 32), k0, k1, r);
  const double u1 = ((double)(((uint64_t)r[0] << 21) ^ (r[1] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(((uint64_t)r[2] << 21) ^ (r[3] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincos(6.283185307179586476925 * u2, &sn, &cs);
  return (F)(rad * ((row & 1) ? sn : cs));
}

// Column-major generator (stand-alone entry slq_dmat_generate): X[row, c] = element (seed, probe id0 + c, row)
// of the same stream as the panel generators. One thread per row, columns walked in the loop.
__global__ __launch_bounds__(256) void k_gen_cols(int64_t n, double *X, int nc, int pdf01, uint64_t seed, uint64_t id0) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int c = 0; c < nc; ++c) X[(int64_t)c * n + row] = probe_element<double>(pdf01, k0, k1, id0 + (uint64_t)c, (int)row);
}

// sphere probes: scale every column to norm sqrt(n) (src/primate/random.py:36-41). One workgroup per column.
__global__ __launch_bounds__(256) void k_scale_cols_sphere(int64_t n, double *X) {
  __shared__ double red[256];
  double *x = X + (int64_t)blockIdx.x * n;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i] * x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const double scale = red[0] > 0.0 ? sqrt((double)n) / sqrt(red[0]) : 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) x[i] *= scale;
}

template <typename F, int LPR>
__global__ __launch_bounds__(256) void k_gen_probes(int n, F *W, int pdf, uint64_t seed,
                                                    uint64_t probe_offset, int nprobes,
                                                    const int32_t *__restrict__ inv /* caller row -> panel row; null: identity */) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW, RPW = Geo<F, LPR>::RPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int panel = blockIdx.y;
  const int rows_per_item = (pdf == 0) ? 128 : 2;
  const int nitems = (n + rows_per_item - 1) / rows_per_item;
  const int stride = gridDim.x * 4 * RPW;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const int col0 = panel * PW + cl * V;
  F *dst = W + ((int64_t)panel * n) * PW + cl * V;
  VF mask;
#pragma unroll
  for (int v = 0; v < V; ++v) mask[v] = (col0 + v < nprobes) ? (F)1 : (F)0;
  for (int it = (blockIdx.x * 4 + wave) * RPW + g; it < nitems; it += stride) {
    uint32_t r[V][4];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const uint64_t id = probe_offset + (uint64_t)(col0 + v);
      philox4x32_10((uint32_t)it, (uint32_t)(pdf != 0), (uint32_t)id, (uint32_t)(id >> 32), k0, k1,
                    r[v]);
    }
    if (pdf == 0) {
      const int rbase = it * 128;
#pragma unroll 4
      for (int b = 0; b < 128; ++b) {
        const int row = rbase + b;
        if (row < n) {
          VF x;
#pragma unroll
          for (int v = 0; v < V; ++v) x[v] = ((r[v][b >> 5] >> (b & 31)) & 1u) ? (F)1 : (F)-1;
          *(VF *)(dst + (int64_t)(inv ? inv[row] : row) * PW) = x * mask;
        }
      }
    } else {
      VF xa, xb;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const double u1 = ((double)(((uint64_t)r[v][0] << 21) ^ (r[v][1] >> 11)) + 0.5) *
                          (1.0 / 9007199254740992.0);
        const double u2 = ((double)(((uint64_t)r[v][2] << 21) ^ (r[v][3] >> 11)) + 0.5) *
                          (1.0 / 9007199254740992.0);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincos(6.283185307179586476925 * u2, &sn, &cs);
        xa[v] = (F)(rad * cs);
        xb[v] = (F)(rad * sn);
      }
      const int row = it * 2;
      if (row < n) *(VF *)(dst + (int64_t)(inv ? inv[row] : row) * PW) = xa * mask;
      if (row + 1 < n) *(VF *)(dst + (int64_t)(inv ? inv[row + 1] : row + 1) * PW) = xb * mask;
    }
  }
}

// ---- Gauss quadrature on the device -----------------------------------------------------------------
__device__ __forceinline__ double apply_fun(int fun_id, double p0, double p1, double x) {
  switch (fun_id) {
    case 0: return x;
    case 1: return fabs(x);
    case 2: return sqrt(x);
    case 3: return log(x > 2.220446049250313e-16 ? x : 2.220446049250313e-16);
    case 4: return 1.0 / x;
    case 5: return exp(p0 * x);
    case 6: {
      const double d = (p0 != p1) ? (p1 - p0) : 1.0;
      double y = (x - p0) / d;
      y = y < 0.0 ? 0.0 : (y > 1.0 ? 1.0 : y);
      return 3.0 * y * y - 2.0 * y * y * y;
    }
    case 7: {
      const double xx = (p1 != 0.0) ? fabs(x) : x;
      return xx < p0 ? 0.0 : 1.0;
    }
    case 8: {
      const int q = (int)p0;
      const double xc = x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x);
      double J = 1.0, pw = 1.0, s = 0.0;
      for (int i = 0; i <= q; ++i) {
        if (i > 0) {
          J *= (2.0 * i - 1.0) / (2.0 * i);
          pw *= (1.0 - xc * xc);
        }
        s += xc * pw * J;
      }
      return s;
    }
    default: return 0.0;
  }
}

// One probe per lane; d, e, z live in LDS as [k][lanes] so a wave's accesses are conflict-free.
// Implicit-shift QL (Wilkinson shift) carrying only the first row of the eigenvector matrix:
// nodes = eigenvalues of T_k, weights = z^2 (Golub-Welsch; reference: integrate.py:61-64 via
// LAPACK stemr, tridiag.py:10-11). Nodes are sorted ascending like LAPACK's output.
// T_k is the full deg x deg Jacobi matrix with a zero tail after an early stop — what the
// reference sees with freshly allocated alpha/beta (lanczos.py:101-102).
__global__ __launch_bounds__(64) void k_quadrature(StepState st, int lanes, int fun_id, double p0,
                                                   double p1, double *__restrict__ quad,
                                                   double *__restrict__ nodes,
                                                   double *__restrict__ weights,
                                                   int *__restrict__ fail) {
  extern __shared__ double lds[];
  const int k = st.deg;
  const int lane = threadIdx.x;
  // `lanes` <= 64 probes per workgroup: the LDS stride, chosen by the host so 3*k*lanes doubles fit
  const int col = blockIdx.x * lanes + lane;
  const int ll = lane < lanes ? lane : 0;
  double *d = lds + ll, *e = lds + k * lanes + ll, *z = lds + 2 * k * lanes + ll;
#define D(i) d[(i) * lanes]
#define E(i) e[(i) * lanes]
#define Z(i) z[(i) * lanes]
  const bool live = lane < lanes && col < st.nprobes;
  if (!live) return;
  for (int i = 0; i < k; ++i) {
    D(i) = live ? st.alpha[(int64_t)i * st.bpad + col] : 0.0;
    // E(i) couples i and i+1: beta_{i+1} = nu[i+1]
    E(i) = (live && i + 1 < k) ? st.nu[(int64_t)(i + 1) * st.bpad + col] : 0.0;
    Z(i) = (i == 0) ? 1.0 : 0.0;
  }
  int bad = 0;
  if (live) {
    for (int l = 0; l < k; ++l) {
      int iter = 0;
      for (;;) {
        int m = l;
        for (; m < k - 1; ++m) {
          const double dd = fabs(D(m)) + fabs(D(m + 1));
          if (fabs(E(m)) <= 2.220446049250313e-16 * dd) break;
        }
        if (m == l) break;
        if (iter++ >= 60) {
          bad = 1;
          break;
        }
        double g = (D(l + 1) - D(l)) / (2.0 * E(l));
        double r = hypot(g, 1.0);
        g = D(m) - D(l) + E(l) / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i = m - 1;
        bool underflow = false;
        for (; i >= l; --i) {
          double f = s * E(i);
          const double b = c * E(i);
          r = hypot(f, g);
          E(i + 1) = r;
          if (r == 0.0) {
            D(i + 1) -= p;
            E(m) = 0.0;
            underflow = true;
            break;
          }
          s = f / r;
          c = g / r;
          g = D(i + 1) - p;
          r = (D(i) - g) * s + 2.0 * c * b;
          p = s * r;
          D(i + 1) = g + p;
          g = c * r - b;
          f = Z(i + 1);
          Z(i + 1) = s * Z(i) + c * f;
          Z(i) = c * Z(i) - s * f;
        }
        if (underflow) continue;
        D(l) -= p;
        E(l) = g;
        E(m) = 0.0;
      }
    }
    // insertion sort by node
    for (int i = 1; i < k; ++i) {
      const double dv = D(i), zv = Z(i);
      int jj = i - 1;
      for (; jj >= 0 && D(jj) > dv; --jj) {
        D(jj + 1) = D(jj);
        Z(jj + 1) = Z(jj);
      }
      D(jj + 1) = dv;
      Z(jj + 1) = zv;
    }
    double s = 0.0;
    for (int i = 0; i < k; ++i) {
      const double th = D(i), tau = Z(i) * Z(i);
      if (nodes) nodes[(int64_t)col * k + i] = th;
      if (weights) weights[(int64_t)col * k + i] = tau;
      if (fun_id >= 0) s += apply_fun(fun_id, p0, p1, th) * tau;
    }
    if (quad) {
      const double vn2 = st.vnorm2[col];
      // an all-zero probe is 0/0 in the reference (lanczos.h:120): surface it as NaN
      quad[col] = (vn2 > 0.0) ? s * vn2 : __builtin_nan("");
    }
    if (bad) atomicOr(fail, 1);
  }
#undef D
#undef E
#undef Z
}


// Implicit QL with Wilkinson shifts on a symmetric tridiagonal (d, e: e[i] couples i and i+1) held in LDS,
// accumulating the full eigenvector matrix Z (k x k, leading dimension ldz, identity on entry). Run by one
// wave: every lane repeats the scalar recurrences (identical values, so the LDS writes to d/e agree) and the
// plane rotations are applied lane-parallel, lane r owning rows r, r+64, ... of Z. Returns 1 if some
// eigenvalue did not converge in 60 sweeps.
__device__ __forceinline__ int ql_implicit_full(double *d, double *e, double *Z, int k, int ldz, int lane) {
  int bad = 0;
  for (int l = 0; l < k; ++l) {
    int iter = 0;
    for (;;) {
      int m = l;
      for (; m < k - 1; ++m) {
        const double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= 2.220446049250313e-16 * dd) break;
      }
      if (m == l) break;
      if (iter++ >= 60) {
        bad = 1;
        break;
      }
      double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
      double r = hypot(g, 1.0);
      g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
      double s = 1.0, c = 1.0, p = 0.0;
      int i = m - 1;
      bool underflow = false;
      for (; i >= l; --i) {
        double f = s * e[i];
        const double b = c * e[i];
        r = hypot(f, g);
        e[i + 1] = r;
        if (r == 0.0) {
          d[i + 1] -= p;
          e[m] = 0.0;
          underflow = true;
          break;
        }
        s = f / r;
        c = g / r;
        g = d[i + 1] - p;
        r = (d[i] - g) * s + 2.0 * c * b;
        p = s * r;
        d[i + 1] = g + p;
        g = c * r - b;
        for (int row = lane; row < k; row += 64) {
          const double z1 = Z[row * ldz + i + 1], z0 = Z[row * ldz + i];
          Z[row * ldz + i + 1] = s * z0 + c * z1;
          Z[row * ldz + i] = c * z0 - s * z1;
        }
      }
      if (underflow) continue;
      d[l] -= p;
      e[l] = g;
      e[m] = 0.0;
    }
  }
  return bad;
}

// ---- f(A)v coefficients: one wave per probe, full eigenvector matrix in LDS -----------------------
// f(A) x ~= ||x|| Q Y (f(theta) * Y[0,:])   (reference: MatrixFunction._matvec,
// src/primate/operators.py:118-124). With unnormalised ring vectors W_t = nu_t q_t this is
// sum_t g_t W_t with g_t = ||x|| c_t / nu_t, c = Y (f(theta) * Y[0,:]).
// All 64 lanes run the scalar QL recurrences redundantly (identical values, so the LDS writes to
// d/e are the same from every lane); the plane rotations are applied lane-parallel, lane r owning
// rows r, r+64, ... of Y. coef[t][col] = sign * g_t (0 where nu_t = 0, i.e. past an early stop);
// reverse_rows stores g_t in row k-1-t, the order in which k_reorth_update walks the ring.
// ZG: the k x k eigenvector matrix lives in LDS (false, k <= 141) or in a per-probe global scratch (true).
template <bool ZG>
__global__ __launch_bounds__(64) void k_fun_coeffs(StepState st, int fun_id, double p0, double p1,
                                                   double sign, int reverse_rows,
                                                   double *__restrict__ coef, double *zscratch,
                                                   int *__restrict__ fail) {
  extern __shared__ double lds[];
  const int k = st.deg, ldz = k + 1;
  const int lane = threadIdx.x, col = blockIdx.x;
  double *d = lds, *e = lds + k;
  double *Z = ZG ? zscratch + (int64_t)col * k * ldz : lds + 2 * k;
  for (int i = lane; i < k; i += 64) {
    d[i] = st.alpha[(int64_t)i * st.bpad + col];
    e[i] = (i + 1 < k) ? st.nu[(int64_t)(i + 1) * st.bpad + col] : 0.0;
  }
  for (int row = lane; row < k; row += 64)  // a lane initialises (and later rotates) the rows it owns
    for (int c = 0; c < k; ++c) Z[row * ldz + c] = (c == row) ? 1.0 : 0.0;
  __threadfence_block();
  __syncthreads();
  const int bad = ql_implicit_full(d, e, Z, k, ldz, lane);
  __threadfence_block();
  __syncthreads();
  // e[i] <- f(theta_i) * Y[0,i]
  for (int i = lane; i < k; i += 64) e[i] = apply_fun(fun_id, p0, p1, d[i]) * Z[i];
  __syncthreads();
  const double xnorm = sqrt(st.vnorm2[col]);
  for (int t = lane; t < k; t += 64) {
    double c = 0.0;
    for (int i = 0; i < k; ++i) c += Z[t * ldz + i] * e[i];
    const double nu = st.nu[(int64_t)t * st.bpad + col];
    const int row = reverse_rows ? (k - 1 - t) : t;
    coef[(int64_t)row * st.bpad + col] = (nu > 0.0) ? sign * xnorm * c / nu : 0.0;
  }
  if (bad && lane == 0) atomicOr(fail, 1);
}

// Full eigendecomposition of a batch of symmetric tridiagonals (stand-alone entry slq_eigh_tridiag_batch;
// reference: eigh_tridiag, src/primate/tridiag.py:25-44). One wave per matrix; din/ein are nb x k row-major with
// ein[:, i] coupling i-1 and i (ein[:, 0] ignored); eigenvalues ascending into w (nb x k), eigenvectors as the
// COLUMNS of Zout (nb x k x k, row-major per matrix) when Zout != null.
template <bool ZG>
__global__ __launch_bounds__(64) void k_eigh_tridiag(int k, const double *__restrict__ din, const double *__restrict__ ein,
                                                     double *__restrict__ w, double *__restrict__ Zout, double *zscratch,
                                                     int *__restrict__ fail) {
  // ZG = false: the k x k eigenvector matrix lives in LDS (k <= 141); ZG = true: in a per-matrix global scratch
  // (k x (k+1) doubles, L2-resident), for larger k. Each lane only ever touches its own rows of Z until the end.
  extern __shared__ double lds[];
  const int ldz = k + 1;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  double *d = lds, *e = lds + k;
  double *Z = ZG ? zscratch + b * (int64_t)k * ldz : lds + 2 * k;
  int *ord = (int *)(ZG ? lds + 2 * k : lds + 2 * k + (size_t)k * ldz);
  for (int i = lane; i < k; i += 64) {
    d[i] = din[b * k + i];
    e[i] = (i + 1 < k) ? ein[b * k + i + 1] : 0.0;
  }
  for (int row = lane; row < k; row += 64)
    for (int c = 0; c < k; ++c) Z[row * ldz + c] = (c == row) ? 1.0 : 0.0;
  __threadfence_block();
  __syncthreads();
  const int bad = ql_implicit_full(d, e, Z, k, ldz, lane);
  __threadfence_block();
  __syncthreads();
  // ascending order by rank counting (ties broken by index): ord[rank] = source column
  for (int i = lane; i < k; i += 64) {
    int rank = 0;
    const double di = d[i];
    for (int t = 0; t < k; ++t) rank += (d[t] < di) || (d[t] == di && t < i);
    ord[rank] = i;
  }
  __syncthreads();
  for (int i = lane; i < k; i += 64) w[b * k + i] = d[ord[i]];
  if (Zout)
    for (int row = lane; row < k; row += 64)  // a lane writes out the rows it owns
      for (int col = 0; col < k; ++col) Zout[b * k * k + (int64_t)row * k + col] = Z[row * ldz + ord[col]];
  if (bad && lane == 0) atomicOr(fail, 1);
}

// transpose host-order (nb x deg row-major) tridiagonal coefficients into the [deg+1][bpad] device
// layout of StepState (stand-alone quadrature entry)
__global__ void k_load_tridiag(StepState st, const double *__restrict__ d, const double *__restrict__ e,
                               int nb) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= st.bpad) return;
  for (int i = 0; i < st.deg; ++i) {
    st.alpha[(int64_t)i * st.bpad + col] = (col < nb) ? d[(int64_t)col * st.deg + i] : 0.0;
    st.nu[(int64_t)i * st.bpad + col] = (col < nb && i > 0) ? e[(int64_t)col * st.deg + i] : 0.0;
  }
  st.nu[(int64_t)st.deg * st.bpad + col] = 0.0;
  st.vnorm2[col] = 1.0;
}


// ---- diagonal estimator accumulation -------------------------------------------
// Per probe p (in order): numer += u_p * v_p ; denom += v_p^2 ; msum += numer/denom, with
// u_p = f(A) v_p — the loop body of the reference's diag() (src/primate/diagonal.py:74-79,86-91),
// whose estimate is the running MEAN of the successive ratios numer/denom. Rows are independent;
// the dependence is sequential over probes, so a wave takes 64 rows, stages 64 x 64 tiles of V and U
// through LDS (coalesced loads, padded rows) and each lane scans its own row.
// vscale[col] rescales the stored probe (sphere probes are kept as g with v = sqrt(n) g/||g||).
template <typename F>
__global__ __launch_bounds__(64) void k_diag_accumulate(int n, const F *__restrict__ Vp,
                                                        const F *__restrict__ Up, int PW, int nprobes,
                                                        const double *__restrict__ vscale,
                                                        double *__restrict__ numer,
                                                        double *__restrict__ denom,
                                                        double *__restrict__ msum) {
  __shared__ F tv[64][65], tu[64][65];
  const int lane = threadIdx.x;
  const int r0 = blockIdx.x * 64;
  const int row = r0 + lane;
  double nu = 0.0, de = 0.0, ms = 0.0;
  if (row < n) {
    nu = numer[row];
    de = denom[row];
    ms = msum[row];
  }
  for (int c0 = 0; c0 < nprobes; c0 += 64) {
    const int col = c0 + lane;  // this lane loads column `col` of every row of the tile
    const int panel = col / PW, pc = col % PW;
    for (int r = 0; r < 64; ++r) {
      const int rr = r0 + r;
      F a = (F)0, b = (F)0;
      if (rr < n && col < nprobes) {
        const int64_t off = ((int64_t)panel * n + rr) * PW + pc;
        a = Vp[off];
        b = Up[off];
      }
      tv[r][lane] = a;
      tu[r][lane] = b;
    }
    __syncthreads();
    if (row < n) {
      const int cmax = min(64, nprobes - c0);
      for (int c = 0; c < cmax; ++c) {
        const double v = (double)tv[lane][c] * vscale[c0 + c];
        const double u = (double)tu[lane][c];
        nu += u * v;
        de += v * v;
        ms += nu / de;
      }
    }
    __syncthreads();
  }
  if (row < n) {
    numer[row] = nu;
    denom[row] = de;
    msum[row] = ms;
  }
}

// per-probe scale of the stored probe panel: sqrt(vnorm2)/nu_0 (1 unless sphere)
__global__ void k_probe_scale(StepState st, double *__restrict__ vscale) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col < st.bpad) {
    const double nu0 = st.nu[col];
    vscale[col] = nu0 > 0.0 ? sqrt(st.vnorm2[col]) / nu0 : 0.0;
  }
}

// ---- FTTR weights ------------------------------------------------------------------
// Forward three-term recurrence of the orthonormal polynomials of the Jacobi matrix (alpha, beta)
// at each node: w_i = 1 / (mu_0 * sum_t p_t(theta_i)^2), mu_0 = sum |theta[:k]|, p_0 = mu_0^{-1/2}
// (reference: src/primate/fttr.py:5-29, algorithm of Laudadio, Mastronardi & Van Dooren 2023).
// One thread per (rule, node); theta/weights are [nb][k], alpha/beta [nb][n] (beta[:,0] unused).
__global__ void k_fttr(int nb, int n, int k, const double *__restrict__ theta,
                       const double *__restrict__ alpha, const double *__restrict__ beta,
                       double *__restrict__ weights) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nb * k) return;
  const int rule = idx / k;
  const double *th = theta + (int64_t)rule * k;
  const double *a = alpha + (int64_t)rule * n;
  const double *b = beta + (int64_t)rule * n;
  double mu0 = 0.0;
  for (int i = 0; i < k; ++i) mu0 += fabs(th[i]);
  const double x = th[idx % k];
  double p0 = 1.0 / sqrt(mu0), ss = p0 * p0, p1 = 0.0;
  if (n > 1) {
    p1 = (x - a[0]) * p0 / b[1];
    ss += p1 * p1;
  }
  for (int t = 2; t < n; ++t) {
    const double pt = ((x - a[t - 1]) / b[t]) * p1 + (-b[t - 1] / b[t]) * p0;
    ss += pt * pt;
    p0 = p1;
    p1 = pt;
  }
  weights[idx] = (1.0 / ss) / mu0;
}


// ---- device bandwidth probe (bench.py: "fraction of the measured triad") ----------
// mode 0: read-only dot of two streams (2R); 1: in-place triad w -= c*q (2R + 1W); 2: copy (1R + 1W).
// 16 B per lane, one contiguous window swept by the whole grid — the access shape of the sweeps.
__global__ __launch_bounds__(kBlock) void k_stream_probe(double *w, const double *q, int64_t nvec,
                                                         int mode, double c, double *sink) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 *W = (d2 *)w;
  const d2 *Q = (const d2 *)q;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  d2 acc = (d2)0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    d2 x;
    if (mode == 0) x = W[i] * Q[i];
    else if (mode == 1) { x = W[i] - c * Q[i]; W[i] = x; }
    else { x = Q[i]; W[i] = x; }
    acc += x;
  }
  if (acc[0] + acc[1] == 1.234567e-300) sink[0] = acc[0];
}


// ---- tall-skinny dense algebra on the matrix cores (xtrace / hutch++ sketches) ----
// Column-major n x m device matrices (n ~ 1e5..1e7, m <= a few hundred). Two products cover the
// blocked Gram-Schmidt / CholeskyQR2 and the m x m summaries of XTrace:
//   TN:  C (ma x mb) = A[:, a-range]^T B[:, b-range]          K = n: split into row slabs
//   NN:  OUT[:, o-range] = beta*OUT + alpha * A[:, a-range] C  (C: ma x mb, row-major, small)
// Both use v_mfma_f64_16x16x4_f64. In the TN kernel a lane loads 4 consecutive k for its column with
// one 32-byte load and MFMA #s consumes element s: lane group g = lane>>4 then contributes
// k = k0 + 4 g + s, the same for both operands, so every k in [k0, k0+16) is used exactly once and
// each column contributes a full 128-byte line per load.
typedef double d4v __attribute__((ext_vector_type(4)));

constexpr int kTnA = 2;  // 16-column A fragments per wave (32 output rows)
constexpr int kTnB = 4;  // 16-column B fragments per wave (64 output columns)

__global__ __launch_bounds__(64) void k_gemm_tn(int n, const double *__restrict__ A, int64_t lda,
                                                const double *__restrict__ B, int64_t ldb, int ma, int mb,
                                                int slab_rows, double *__restrict__ partial /* [nslab][ma][mb] */) {
  const int lane = threadIdx.x, lc = lane & 15, g = lane >> 4;
  const int tiles_b = (mb + 16 * kTnB - 1) / (16 * kTnB);
  const int ta = blockIdx.x / tiles_b, tb = blockIdx.x % tiles_b;
  const int i0 = ta * 16 * kTnA, j0 = tb * 16 * kTnB;
  const int slab = blockIdx.y;
  const int k_begin = slab * slab_rows, k_end = min(n, k_begin + slab_rows);
  d4v acc[kTnA][kTnB];
#pragma unroll
  for (int x = 0; x < kTnA; ++x)
#pragma unroll
    for (int y = 0; y < kTnB; ++y) acc[x][y] = (d4v)0.0;
  for (int k0 = k_begin; k0 < k_end; k0 += 16) {
    const int k = k0 + 4 * g;
    d4v a[kTnA], b[kTnB];
#pragma unroll
    for (int x = 0; x < kTnA; ++x) {
      const int col = i0 + x * 16 + lc;
      a[x] = (d4v)0.0;
      if (col < ma) {
        const double *p = A + (int64_t)col * lda + k;
        if (k + 3 < k_end) a[x] = *(const d4v *)p;
        else
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < k_end) a[x][e] = p[e];
      }
    }
#pragma unroll
    for (int y = 0; y < kTnB; ++y) {
      const int col = j0 + y * 16 + lc;
      b[y] = (d4v)0.0;
      if (col < mb) {
        const double *p = B + (int64_t)col * ldb + k;
        if (k + 3 < k_end) b[y] = *(const d4v *)p;
        else
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < k_end) b[y][e] = p[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int x = 0; x < kTnA; ++x)
#pragma unroll
        for (int y = 0; y < kTnB; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x][e], b[y][e], acc[x][y], 0, 0, 0);
  }
  double *P = partial + (int64_t)slab * ma * mb;
#pragma unroll
  for (int x = 0; x < kTnA; ++x)
#pragma unroll
    for (int y = 0; y < kTnB; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + x * 16 + g + 4 * r, jj = j0 + y * 16 + lc;
        if (i < ma && jj < mb) P[(int64_t)i * mb + jj] = acc[x][y][r];
      }
}

__global__ void k_sum_slabs(const double *__restrict__ partial, int nslab, int64_t count, double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (int k = 0; k < nslab; ++k) s += partial[(int64_t)k * count + i];
  out[i] = s;
}

constexpr int kNnB = 4;  // 16-column output fragments per wave

__global__ __launch_bounds__(64) void k_gemm_nn(int n, double *OUT, int64_t ldo, const double *__restrict__ A,
                                                int64_t lda, int ma, const double *__restrict__ C, int mb,
                                                double alpha, double beta) {
  const int lane = threadIdx.x, lc = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const int j0 = blockIdx.y * 16 * kNnB;
  d4v acc[kNnB];
#pragma unroll
  for (int y = 0; y < kNnB; ++y) acc[y] = (d4v)0.0;
  const bool row_ok = row0 + lc < n;
  for (int k0 = 0; k0 < ma; k0 += 4) {
    const int k = k0 + g;
    const double a = (row_ok && k < ma) ? A[(int64_t)k * lda + row0 + lc] : 0.0;
#pragma unroll
    for (int y = 0; y < kNnB; ++y) {
      const int col = j0 + y * 16 + lc;
      const double b = (k < ma && col < mb) ? C[(int64_t)k * mb + col] : 0.0;
      acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[y], 0, 0, 0);
    }
  }
#pragma unroll
  for (int y = 0; y < kNnB; ++y) {
    const int col = j0 + y * 16 + lc;
    if (col >= mb) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + g + 4 * r;
      if (row < n) {
        double *o = OUT + (int64_t)col * ldo + row;
        *o = (beta == 0.0 ? 0.0 : beta * *o) + alpha * acc[y][r];
      }
    }
  }
}

// The upper-triangle entries that cross from one XCD chunk into the next: alpha's share of them, after the pass (a kernel boundary
// makes every chunk's rows visible everywhere). One wave per edge and iteration: acc += val * W[r] * W[c] (val = the doubled
// off-diagonal entry, as in the upper-triangle streams); per-block column sums into out[blockIdx.x][bpad].
template <typename F, int LPR>
__global__ __launch_bounds__(kBlock) void k_alpha_edges(int n, int nedges, const int32_t *__restrict__ er, const int32_t *__restrict__ ec, const F *__restrict__ ev,
                                                        const F *__restrict__ W /* slot of W_{j+1} */, double *__restrict__ out, int bpad) {
  using VF = typename VecT<F>::type;
  constexpr int V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW;
  static_assert(LPR == 64, "whole-row panels");
  __shared__ double red[kWaves * 64 * V];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int panel = blockIdx.y;
  const F *Wp = W + (int64_t)panel * n * PW + lane * V;
  VF acc = (VF)(F)0;
  for (int e = blockIdx.x * kWaves + wave; e < nedges; e += gridDim.x * kWaves) {
    const VF a = *(const VF *)(Wp + (int64_t)er[e] * PW), b = *(const VF *)(Wp + (int64_t)ec[e] * PW);
    acc += (ev[e] * a) * b;
  }
  block_reduce_columns<F, LPR>(acc, red, out + (int64_t)blockIdx.x * bpad + panel * PW);
}

}  // namespace slq
