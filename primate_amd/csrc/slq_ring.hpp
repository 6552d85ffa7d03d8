// slq_ring.hpp — the ring-fed LDS-tile passes for every panel width and ring-column count.
//
// k_csr_ring_pass (slq_kernels.hpp) serves ONE shape: 1-KiB panel rows (128 fp64 / 256 fp32 probes per panel) and at most
// three ring columns per step. What the reference's drivers submit is mostly something else: hutch() draws 32 probes per
// batch (src/primate/trace.py:36,104-116), an 8-GPU shard of 256 probes is 32-64 per GPU, and orth is any number up to
// deg (src/primate/include/lanczos.h:58-65,133-136; src/primate/lanczos.py:88-89). k_ring_pass is the same machine -
// loader waves that land everything a tile needs in a ring of LDS slots by LDS-DMA, consumer waves behind LDS counters -
// generalised along two axes:
//
//  * LPR lanes per panel row (64, 32, 16: panel rows of 1 KiB, 512 B, 256 B). R = 64 / LPR panel rows share one wave
//    instruction: one 1-KiB DMA lands R consecutive lines of the tile image (every lane gives its own source address),
//    and a consumer wave works on R adjacent rows of the tile at a time (lanes split rows x probe columns), so the row-
//    local streams of a wave stay 1 KiB contiguous and an instruction moves as many bytes as in the wide kernel. The tiles
//    of this form are R consecutive base tiles merged (one line list: what two neighbours share is landed once; slq.hip:
//    build_ring_stream), so a slot still holds 36 KiB of image and a tile period still moves the same bytes.
//  * WAVES = 16 (two consumer groups of 7 taking the tiles in turn, 128 VGPRs per wave: up to 3 ring columns, the
//    row-local streams double-buffered one tile of the group ahead) or 8 (one group of 6 consumers, 256 VGPRs per wave:
//    4..8 ring columns; a row's stream registers are refilled for the next tile as soon as the row has been consumed).
//
// Arithmetic per row and the order of every sum are those of k_csr_pass / k_csr_ring_pass: products of a row in CSR
// order, rows -> waves and the final reduction over waves static. Results are reproducible bit for bit and equal to
// the other launch sequences' up to the order of the cross-row sums.
#pragma once
#include <type_traits>

#include "slq_common.hpp"

namespace slq {

#ifndef SLQ_RINGN_LOADERS
#define SLQ_RINGN_LOADERS 2  // loader waves of k_ring_pass (A/B builds: -DSLQ_RINGN_LOADERS=4)
#endif
#ifndef SLQ_RINGN_PRIO
#define SLQ_RINGN_PRIO 0     // s_setprio of the loader waves (0: none)
#endif
constexpr int kRingNLoaders = SLQ_RINGN_LOADERS;
constexpr int kRingRecStride = 1536;  // bytes of record per base tile: 128 B of header + kRingTileNnz x (4 + 8)

// GEO 0: the tile's lines and record land by LDS-DMA (loader waves issue global_load_lds and count them with vmcnt).
// GEO 1 ("staged"): the loader waves bring them through their REGISTERS instead (global_load_dwordx4, ds_write_b128) - for the
// alpha-only pass. Measured (profiles/r03_tcp_counters.json): LDS-DMA lands ~25-27 GB/s per CU whatever the
// number of loaders, slots or tiles in flight (the cadence MI355X_MICROARCH.md gives for one loader wave), and the alpha-only
// pass - nothing but its tile images to move, 49 of the ~106 requests a CU can keep in flight - sits on that; plain vector
// loads are not held to it (the generic gather passes move 50-58 GB/s per CU). The passes with row-local streams beside the
// images are paced by the memory system (106 requests in flight) and keep the DMA form, which costs no registers.
template <int LPR, int WAVES, int GEO = 0> struct RingGeo {
  static constexpr int R = 64 / LPR;                  // panel rows per wave instruction and per 1-KiB DMA
  static constexpr bool kStaged = GEO == 1;
  static constexpr int kLines = kRingTileCols;        // KiB of image per slot (lines of 1 KiB / R)
  static constexpr int kRecBytes = (kRingRecStride * R + 1023) / 1024 * 1024;  // landed in whole KiB
  static constexpr int kSlotBytes = kLines * 1024 + kRecBytes;
  static constexpr int kFlagBytes = 64;               // ready[], done[], abort
  static constexpr int kSlots = (160 * 1024 - kFlagBytes - 2 * kRingLag * R * 256) / kSlotBytes;  // R = 1, 2: 4; R = 4: 3
  static constexpr int kLag = kRingLag;               // tiles of DMAs a loader keeps in flight (it publishes tile k - kLag at tile k)
  // loader waves: two (the A/B macro's number where their descriptor staging fits); staged: four, no staging area
  static constexpr int kLoaders = kStaged ? (WAVES == 16 ? 4 : 2)
                                          : ((WAVES == 16 && kFlagBytes + kRingNLoaders * kLag * R * 256 + kSlots * kSlotBytes <= 160 * 1024) ? kRingNLoaders : 2);
  static constexpr int NCW = WAVES - kLoaders;        // consumer waves
  static constexpr int G = WAVES >= 16 ? 2 : 1;       // consumer groups taking the tiles in turn
  static constexpr int NC = NCW / G;                  // consumer waves of one tile
  static constexpr int MR = (kRingTileRows + NC - 1) / NC;  // row groups (R rows each) of a tile per consumer wave
  static constexpr int kStageBytes = kStaged ? 0 : kLoaders * kLag * R * 256;  // the loaders' descriptor staging
  static constexpr int kHeadBytes = kFlagBytes + kStageBytes;
  static constexpr int kLdsBytes = kHeadBytes + kSlots * kSlotBytes;
  static constexpr int kDescWords = 64 * R;           // descriptor: R blocks of 64 words (block b, word 8 + d: line d * R + b)
  static constexpr int kRecValOffW = 16 * R - 1;      // record header: [0 .. rows] row offsets, [16R - 1] byte offset of the values,
  static constexpr int kRecSelfW = 16 * R;            //   [16R .. 16R + rows) line of each row's own panel row
  static constexpr int kRecHeadB = 128 * R;           //   then the column lines (int32) and the values (F)
  static_assert(WAVES == 16 || WAVES == 8, "16 waves (<= 3 ring columns) or 8 (more)");
  static_assert(NCW % G == 0 && kLag + 1 <= kSlots && (2 * kSlots + 1) * 4 <= kFlagBytes, "ring geometry");  // (a slot's counters count tiles, whichever group consumed them)
  static_assert(kLdsBytes <= 160 * 1024, "the ring must fit the LDS");
  static_assert(kRecHeadB + kRingTileNnz * R * (4 + 8) <= kRecBytes, "a tile's record must fit its slot");
  static_assert(32 * R >= 16 * R + kRingTileRows * R && kRingTileRows * R < 16 * R - 1, "record header layout");
  static_assert((size_t)WAVES * 64 * 4 * 8 <= (size_t)kSlots * kSlotBytes, "the final reduction reuses the slots");
  static_assert(((kLines + kLoaders - 1) / kLoaders + (kRecBytes + 1023) / 1024) * (kLag - 1) + R * kLag <= 56,
                "a loader's DMAs in flight are counted by vmcnt");
};

template <typename F, int PASS, int NTP, int RC, int LPR, int WAVES, int GEO = 0>
__global__ __launch_bounds__(WAVES * 64) void k_ring_pass(
    int n, const int32_t *__restrict__ tile_desc, const char *__restrict__ tile_rec, TileRanges xr, F *ring, int64_t slot_stride, int S, int j,
    const double *__restrict__ coefA, const double *__restrict__ coefB, const double *__restrict__ gamma, double *__restrict__ part,
    int bpad, int xt, int *__restrict__ fail, unsigned long long *dbg_base /* diagnostic builds (-DSLQ_DEBUG_TIMES) only */) {
  using VF = typename VecT<F>::type;
  using RG = RingGeo<LPR, WAVES, GEO>;
  constexpr int R = RG::R, V = Geo<F, LPR>::V, PW = Geo<F, LPR>::PW;
  constexpr int NX = RC > 2 ? RC - 2 : 1;
  constexpr int NC = RG::NC, MR = RG::MR, G = RG::G, NS = RG::kSlots;
  constexpr bool kRefill = G == 1;  // 8 waves: stream registers refilled in place (no second set)
  constexpr int kChunk = (R == 1 && WAVES == 16) ? kRingChunk : 2;  // (8 waves, R > 1: registers)
  constexpr int LAG = RG::kLag;
  static_assert(LAG == 2 || LAG == 3, "descriptor staging ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  using lds_int = __attribute__((address_space(3))) int;
  lds_int *flags = (lds_int *)lds_raw;
  unsigned char *slots = lds_raw + RG::kHeadBytes;
  double *red = (double *)slots;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = R == 1 ? 0 : lane / LPR, cl = R == 1 ? lane : lane % LPR;
  const int rev = (xt >> 2) & 1;
  const int panel = rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int64_t poff = (int64_t)panel * n * PW + cl * V;
  const int padded_rows = (xt >> 3) & 1;  // the stream's rows are padded to whole chunks of four entries (slq.hip: build_ring_stream)
  const int nostore = (xt >> 4) & 1;      // update passes: the new vector is not stored (a run's LAST step without a kept basis: only its norm is ever used, r04)
  xt &= 1;
  const int first = (j == 0) || (PASS == PASS_ALPHA && xt);
  const F *wcl = ring + (int64_t)(j % S) * slot_stride + poff;
  const F *wp = ring + (int64_t)((j + S - 1) % S) * slot_stride + poff;
  F *wn = ring + (int64_t)((j + 1) % S) * slot_stride + poff;
  const F *ux[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) ux[i] = ring + (int64_t)ring_slot(j - 2 - i, S) * slot_stride + poff;
  if (threadIdx.x < 2 * NS + 1) flags[threadIdx.x] = 0;
  __syncthreads();
  lds_int *ready = flags, *done = flags + NS, *abort_f = flags + 2 * NS;
  const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int t_first = xr.first[xcd] + bl, t_end = xr.first[xcd + 1];
  const int ntiles = t_first < t_end ? (t_end - t_first + nbl - 1) / nbl : 0;
  VF acc1 = (VF)(F)0, accx = (VF)(F)0;
  VF dacc[RC > 0 ? RC : 1], gacc[RC > 0 ? RC : 1];
#pragma unroll
  for (int i = 0; i < (RC > 0 ? RC : 1); ++i) dacc[i] = gacc[i] = (VF)(F)0;
  // (polls and counter updates in assembly, for the reasons given at k_csr_ring_pass: a compiler-visible LDS access in the
  // loader would drain the DMAs that are meant to stay in flight)
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
  auto tile_at = [&](int k) -> int64_t {  // this workgroup's k-th tile (past the end: its last one again)
    const int kk = min(k, ntiles - 1);
    return (int64_t)(t_first + (rev ? ntiles - 1 - kk : kk) * nbl);
  };
  // slot of the k-th tile and how many times that slot has been used before: the counters' targets. (NS may be 3: no
  // power-of-two mask; the divisions are by a constant)
  if (ntiles > 0 && wave < RG::kLoaders && RG::kStaged) {
    // ---------------- loader, through registers (GEO 1) ----------------
    // Every loader takes every kLoaders-th line (and record chunk) of a tile: 16 bytes per lane into a register, then one
    // ds_write_b128 into the slot. Two register sets: the loads of tile k + 1 are issued BEFORE tile k's registers are written
    // out, so a loader always has a tile's share in flight; descriptors are plain loads two tiles ahead. No counted waits: the
    // compiler's own vmcnt bookkeeping (loads return in order) covers register data.
    constexpr int NL = RG::kLoaders;
    constexpr int MAXI = (RG::kLines + NL - 1) / NL, MAXC = (RG::kRecBytes / 1024 + NL - 1) / NL;
    typedef int v4i __attribute__((ext_vector_type(4)));
    auto fetch_desc = [&](int k, int (&d)[R]) {
#pragma unroll
      for (int b = 0; b < R; ++b) d[b] = tile_desc[tile_at(k) * RG::kDescWords + b * 64 + lane];
    };
    // (every loader issues the SAME number of loads for every tile - a share past the tile's last line or chunk loads that last
    // one again, an L1 hit - so that the compiler can count them: with data-dependent counts it waits for vmcnt(0) before the
    // first write-out, i.e. for the next tile's loads as well, and nothing overlaps)
    auto issue = [&](const int (&d)[R], v4i (&buf)[MAXI + MAXC]) {
      const int D = lane_bcast(d[0], kDescCols), nd = (D + R - 1) / R;
#pragma unroll
      for (int i = 0; i < MAXI; ++i) {
        const int dd = min(wave + i * NL, nd - 1);
        int col = lane_bcast(d[0], kDescList + (R == 1 ? ring1_list_pos(dd) : dd));
#pragma unroll
        for (int b = 1; b < R; ++b) {
          const int cb = lane_bcast(d[b], kDescList + dd);
          col = g == b ? cb : col;
        }
        buf[i] = *(const v4i *)(wcl + (int64_t)col * PW);
      }
      const int chunks = lane_bcast(d[0], kDescRecChunks);
      const char *rsrc = tile_rec + (int64_t)lane_bcast(d[0], kDescRecOff) * 16 + lane * 16;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) buf[MAXI + i] = *(const v4i *)(rsrc + min(wave + i * NL, chunks - 1) * 1024);
    };
    auto put = [&](int k, const int (&d)[R], const v4i (&buf)[MAXI + MAXC]) -> bool {
      const int slot = k % NS;
      if (k >= NS && !spin(done + slot, NC * (k / NS))) return false;  // the slot's previous tile has been consumed
      unsigned char *img = slots + (size_t)slot * RG::kSlotBytes;
      const int D = lane_bcast(d[0], kDescCols), nd = (D + R - 1) / R, chunks = lane_bcast(d[0], kDescRecChunks);
#pragma unroll
      for (int i = 0; i < MAXI; ++i) {
        const int dd = wave + i * NL;
        if (dd < nd) *(v4i *)(img + (size_t)dd * 1024 + lane * 16) = buf[i];
      }
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = wave + i * NL;
        if (c < chunks) *(v4i *)(img + RG::kLines * 1024 + c * 1024 + lane * 16) = buf[MAXI + i];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this loader's share is in the slot
      if (lane == 0) bump(ready + slot);
      return true;
    };
    // Order of the requests: a tile's descriptor is asked for BEFORE the loads of the tile in front of it, so that using it
    // never waits for those (loads complete in order); and every iteration issues the same loads whether or not a next tile
    // exists (tile_at repeats the last one), so that no count depends on control flow.
    int da[R], db[R], dc[R], dd[R];
    v4i ba[MAXI + MAXC], bb[MAXI + MAXC];
    fetch_desc(0, da);
    fetch_desc(1, db);
    issue(da, ba);
    for (int k = 0; k < ntiles; k += 2) {
      fetch_desc(k + 2, dc);
      issue(db, bb);  // tile k + 1 on its way while tile k is written out
      if (!put(k, da, ba)) break;
      fetch_desc(k + 3, dd);
      issue(dc, ba);
      if (k + 1 >= ntiles || !put(k + 1, db, bb)) break;
#pragma unroll
      for (int b = 0; b < R; ++b) da[b] = dc[b], db[b] = dd[b];
    }
  } else if (ntiles > 0 && wave < RG::kLoaders && R == 1) {
    // ---------------- loader, whole-row panels (R = 1) ----------------
    // A loader's own instruction stream was a tile period of the alpha-only pass (r03, scripts/ring_timeline.py: ~50 ns per DMA - a
    // v_readlane, two scalar shifts, a 64-bit VALU add and the M0 write per line, behind an LDS-staged descriptor - plus two LDS
    // round trips and a 100-instruction vmcnt switch per tile). Here the descriptor comes by SCALAR loads (one tile ahead, straight
    // into SGPRs: no staging DMAs to count, no LDS read, no readlane), a line's source is an SGPR base + the lane's constant
    // 16-byte offset, and the `done` counter of the next tile's slot is read behind this tile's DMAs and waited for after the
    // counted wait that follows. Order of a loader's requests: T0 w0 T1 w1 T2 ...; w_k leaves the DMAs of tiles k + 2 - LAG .. k
    // outstanding, so tile k + 1 - LAG has landed and is published. (The loop is rotated - tile k's DMAs first, then everything that
    // precedes tile k + 1's - so that the read-ahead registers are waited for in the iteration that loads them.)
    if (SLQ_RINGN_PRIO) __builtin_amdgcn_s_setprio(SLQ_RINGN_PRIO);
    constexpr int MAXU = (RG::kLines + RG::kLoaders - 1) / RG::kLoaders;  // lines of a tile per loader at most
    const char *wbase = (const char *)(ring + (int64_t)(j % S) * slot_stride + (int64_t)panel * n * PW);  // (wave-uniform)
    const unsigned lane_off = (unsigned)lane * 16u;
    struct TileDesc {
      int D, recoff, chunks;
      int c[MAXU];
    };
    auto fetch_desc = [&](int k, TileDesc &o) {  // uniform addresses: s_load
      const int32_t *dsc = tile_desc + tile_at(k) * RG::kDescWords;
      o.D = dsc[kDescCols];
      o.recoff = dsc[kDescRecOff];
      o.chunks = dsc[kDescRecChunks];
#pragma unroll
      for (int u = 0; u < MAXU; ++u)  // line wave + u kLoaders; two loaders: consecutive words (ring1_list_pos), fetched as wide loads
        o.c[u] = RG::kLoaders == 2 ? dsc[kDescList + wave * kRing1ListHalf + u] : dsc[kDescList + ring1_list_pos(wave + u * RG::kLoaders)];
    };
    TileDesc cur;
    fetch_desc(0, cur);
    int dv = 0, ab = 0;
    int hist[LAG - 1];  // DMAs issued for tiles k, k - 1, ... (the ones that may still be in flight before tile k + 1)
#pragma unroll
    for (int i = 0; i < LAG - 1; ++i) hist[i] = 0;
    bool ok = true;
#ifdef SLQ_DEBUG_TIMES
    auto dbg_of = [&](int k) -> unsigned long long * {
      return (dbg_base && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && k < 256 && lane == 0) ? dbg_base + (size_t)k * 8 : nullptr;
    };
    if (auto *q = dbg_of(0)) q[0] = q[1] = __builtin_amdgcn_s_memrealtime();
#endif
    for (int k = 0; k < ntiles + LAG - 1 && ok; ++k) {
      int issued = 0;
      if (k < ntiles) {
        const int slot = k % NS;
        if (k >= NS) {  // the slot's previous tile has been consumed? (usually known from the read-ahead; else poll)
          if (__builtin_amdgcn_readfirstlane(ab)) ok = false;
          else if (__builtin_amdgcn_readfirstlane(dv) < NC * (k / NS)) ok = spin(done + slot, NC * (k / NS));
        }
#ifdef SLQ_DEBUG_TIMES
        if (auto *q = dbg_of(k)) q[2] = __builtin_amdgcn_s_memrealtime();
#endif
        if (ok) {
          unsigned char *img = slots + (size_t)slot * RG::kSlotBytes;
          const int nd = cur.D;
#pragma unroll
          for (int u = 0; u < MAXU; ++u) {
            const int d = wave + u * RG::kLoaders;
            if (d < nd) {
              const int col = cur.c[u];
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wbase + (int64_t)col * (PW * (int)sizeof(F)) + lane_off),
                                               (__attribute__((address_space(3))) void *)(img + (size_t)d * 1024), 16, 0, SLQ_RING_AUX);
              ++issued;
            }
          }
          const char *rsrc = tile_rec + (int64_t)cur.recoff * 16;
          for (int c = wave; c < cur.chunks; c += RG::kLoaders) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rsrc + c * 1024 + lane_off),
                                             (__attribute__((address_space(3))) void *)(img + RG::kLines * 1024 + c * 1024), 16, 0, 0);
            ++issued;
          }
        }
      }
#ifdef SLQ_DEBUG_TIMES
      if (auto *q = dbg_of(k)) {
        q[3] = __builtin_amdgcn_s_memrealtime();
        q[7] = (unsigned long long)issued;
      }
      if (auto *q = dbg_of(k + 1)) q[0] = __builtin_amdgcn_s_memrealtime();
#endif
      // ---- what precedes tile k + 1 ----
      TileDesc nxt;
      fetch_desc(k + 1, nxt);
      int dvn, abn;
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(dvn), "=&v"(abn) : "v"((unsigned)(uintptr_t)(done + (k + 1) % NS)), "v"((unsigned)(uintptr_t)abort_f) : "memory");
#pragma unroll
      for (int i = LAG - 2; i > 0; --i) hist[i] = hist[i - 1];
      hist[0] = issued;
      int since = 0;
#pragma unroll
      for (int i = 0; i < LAG - 1; ++i) since += hist[i];
      wait_vmcnt_at_most(since);
      if (k + 1 >= LAG && lane == 0) bump(ready + (k + 1 - LAG) % NS);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dvn), "+v"(abn)::"memory");
      dv = dvn;
      ab = abn;
      cur = nxt;
#ifdef SLQ_DEBUG_TIMES
      if (auto *q = dbg_of(k + 1)) q[1] = __builtin_amdgcn_s_memrealtime();
#endif
    }
  } else if (ntiles > 0 && wave < RG::kLoaders) {
    // ---------------- loader, merged tiles (R > 1): descriptors staged in LDS, per-lane-group sources ----------------
    if (SLQ_RINGN_PRIO) __builtin_amdgcn_s_setprio(SLQ_RINGN_PRIO);
    unsigned char *stage = lds_raw + RG::kFlagBytes + (size_t)wave * (LAG * R * 256);
    auto stage_desc = [&](int k) {
      const int32_t *src = tile_desc + tile_at(k) * RG::kDescWords + lane;
#pragma unroll
      for (int b = 0; b < R; ++b)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + b * 64),
                                         (__attribute__((address_space(3))) void *)(stage + ((k % LAG) * R + b) * 256), 4, 0, 0);
    };
    // The loader's iteration is the tile period of the alpha-only pass (scripts/ring_timeline.py, r03: 0.24 us counted wait +
    // 0.23 us slot poll + 0.65 us for 12 DMAs = the pass's 1.0-1.2 us per tile; consumers idle half the time), so nothing in it
    // waits for an LDS round trip any more: descriptor k + 1 and the `done` counter of tile k + 1's slot are READ right behind
    // tile k's DMAs and waited for after the counted wait that follows - by then they are long there. (The loop is rotated -
    // tile k's DMAs first, then everything that precedes tile k + 1's - so that the registers of the read-ahead are waited
    // for in the iteration that loads them: a value carried over the back edge may be copied by the compiler at any time.)
    // Order of a loader's requests: D0 D1 | D2 .. D_LAG T0 w0 D_LAG+1 T1 w1 D_LAG+2 T2 w2 ... (D_j: descriptor j, R requests; T_k:
    // tile k's DMAs; w_k: the counted wait). w_k leaves outstanding exactly what was issued after D_k+2 - the DMAs of tiles
    // k + 2 - LAG .. k and the LAG - 2 descriptors k + 3 .. k + LAG - so tile k + 1 - LAG has landed (it is published) and so has
    // descriptor k + 2, which iteration k + 1 reads ahead.
    // A lane's source of line d is word kDescList + d of ITS lane group's descriptor block: read straight out of the staged
    // descriptor, one ds_read_b32 per line with a per-lane base (r03, late: the loader took R v_readlanes and R - 1 selects per
    // line, 11-15 instructions per DMA - the narrow panels' alpha pass is bound by its loaders). All of a tile's reads go out
    // together in the read-ahead and are waited for once.
    constexpr int MAXU = (RG::kLines + RG::kLoaders - 1) / RG::kLoaders;  // lines of a tile per loader at most
    static_assert(R == 1 || MAXU <= 18, "the read-ahead's operand list");
    auto block0_addr = [&](int k) { return (unsigned)(uintptr_t)(lds_int *)(stage + ((k % LAG) * R) * 256) + lane * 4; };
    auto lines_addr = [&](int k) { return (unsigned)(uintptr_t)(lds_int *)(stage + ((k % LAG) * R + g) * 256) + (kDescList + wave) * 4; };
    int d0cur = 0, dv = 0, ab = 0;
    int cc[18], cn[18];
#pragma unroll
    for (int u = 0; u < 18; ++u) cc[u] = cn[u] = 0;
#define SLQ_RD_LINE(U)                                                                                                              \
  if constexpr (U < MAXU) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(cn[U]) : "v"(la), "n"(U * RG::kLoaders * 4) : "memory");
#define SLQ_RD_LINES                                                                                                                \
  SLQ_RD_LINE(0) SLQ_RD_LINE(1) SLQ_RD_LINE(2) SLQ_RD_LINE(3) SLQ_RD_LINE(4) SLQ_RD_LINE(5) SLQ_RD_LINE(6) SLQ_RD_LINE(7) SLQ_RD_LINE(8)        \
  SLQ_RD_LINE(9) SLQ_RD_LINE(10) SLQ_RD_LINE(11) SLQ_RD_LINE(12) SLQ_RD_LINE(13) SLQ_RD_LINE(14) SLQ_RD_LINE(15) SLQ_RD_LINE(16) SLQ_RD_LINE(17)
#define SLQ_WAIT_LINES(D0, DV, AB)                                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                               \
               : "+v"(D0), "+v"(DV), "+v"(AB), "+v"(cn[0]), "+v"(cn[1]), "+v"(cn[2]), "+v"(cn[3]), "+v"(cn[4]), "+v"(cn[5]), "+v"(cn[6]), "+v"(cn[7]),  \
                 "+v"(cn[8]), "+v"(cn[9]), "+v"(cn[10]), "+v"(cn[11]), "+v"(cn[12]), "+v"(cn[13]), "+v"(cn[14]), "+v"(cn[15]), "+v"(cn[16]), "+v"(cn[17])    \
               :                                                                                                                   \
               : "memory")
    stage_desc(0);
    stage_desc(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
      const unsigned la = lines_addr(0);
      asm volatile("ds_read_b32 %0, %1" : "=&v"(d0cur) : "v"(block0_addr(0)) : "memory");
      SLQ_RD_LINES
      SLQ_WAIT_LINES(d0cur, dv, ab);
#pragma unroll
      for (int u = 0; u < MAXU; ++u) cc[u] = cn[u];
    }
#pragma unroll
    for (int i = 2; i <= LAG; ++i) stage_desc(i);  // (descriptor LAG goes where descriptor 0 was: read just above)
    int hist[LAG - 1];  // DMAs issued for tiles k, k - 1, ... (the ones that may still be in flight before tile k + 1)
#pragma unroll
    for (int i = 0; i < LAG - 1; ++i) hist[i] = 0;
    bool ok = true;
#ifdef SLQ_DEBUG_TIMES
    // (scripts/ring_timeline.py: workgroup 0 of panel 0, loader 0 and consumer 0, the first 256 tiles)
    auto dbg_of = [&](int k) -> unsigned long long * {
      return (dbg_base && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && k < 256 && lane == 0) ? dbg_base + (size_t)k * 8 : nullptr;
    };
    if (auto *q = dbg_of(0)) q[0] = q[1] = __builtin_amdgcn_s_memrealtime();
#endif
    for (int k = 0; k < ntiles + LAG - 1 && ok; ++k) {
      int issued = 0;
      if (k < ntiles) {
        const int slot = k % NS;
        if (k >= NS) {  // the slot's previous tile has been consumed? (usually known from the read-ahead; else poll)
          if (__builtin_amdgcn_readfirstlane(ab)) ok = false;
          else if (__builtin_amdgcn_readfirstlane(dv) < NC * (k / NS)) ok = spin(done + slot, NC * (k / NS));
        }
#ifdef SLQ_DEBUG_TIMES
        if (auto *q = dbg_of(k)) q[2] = __builtin_amdgcn_s_memrealtime();
#endif
        if (ok) {
          unsigned char *img = slots + (size_t)slot * RG::kSlotBytes;
          const int D = lane_bcast(d0cur, kDescCols);
          const int nd = (D + R - 1) / R;
          // lane group b lands line d * R + b (block b of the descriptor); past the tile's last line the descriptor repeats
          // that line: it lands once more, in a line nobody reads (a lane's destination is fixed by its number)
#pragma unroll
          for (int u = 0; u < MAXU; ++u) {
            const int d = wave + u * RG::kLoaders;
            if (d < nd) {
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wcl + (int64_t)cc[u] * PW),
                                               (__attribute__((address_space(3))) void *)(img + (size_t)d * 1024), 16, 0, SLQ_RING_AUX);
              ++issued;
            }
          }
          const int chunks = lane_bcast(d0cur, kDescRecChunks);
          const char *rsrc = tile_rec + (int64_t)lane_bcast(d0cur, kDescRecOff) * 16 + lane * 16;
          for (int c = wave; c < chunks; c += RG::kLoaders) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rsrc + c * 1024),
                                             (__attribute__((address_space(3))) void *)(img + RG::kLines * 1024 + c * 1024), 16, 0, 0);
            ++issued;
          }
        }
      }
#ifdef SLQ_DEBUG_TIMES
      if (auto *q = dbg_of(k)) {
        q[3] = __builtin_amdgcn_s_memrealtime();
        q[7] = (unsigned long long)issued;
      }
      if (auto *q = dbg_of(k + 1)) q[0] = __builtin_amdgcn_s_memrealtime();
#endif
      // ---- what precedes tile k + 1 ----
      // read ahead: descriptor k + 1 (landed: w_k-1, or the prologue's wait) - its head block and this lane's lines -, its
      // slot's counter, the abort flag
      int d0new, dvn, abn;
      {
        const unsigned la = lines_addr(k + 1);
        asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5"
                     : "=&v"(d0new), "=&v"(dvn), "=&v"(abn)
                     : "v"(block0_addr(k + 1)), "v"((unsigned)(uintptr_t)(done + (k + 1) % NS)), "v"((unsigned)(uintptr_t)abort_f)
                     : "memory");
        SLQ_RD_LINES
      }
#pragma unroll
      for (int i = LAG - 2; i > 0; --i) hist[i] = hist[i - 1];
      hist[0] = issued;
      int since = (LAG - 2) * R;
#pragma unroll
      for (int i = 0; i < LAG - 1; ++i) since += hist[i];
      wait_vmcnt_at_most(__builtin_amdgcn_readfirstlane(since));
      if (k + 1 >= LAG && lane == 0) bump(ready + (k + 1 - LAG) % NS);
      SLQ_WAIT_LINES(d0new, dvn, abn);
      d0cur = d0new;
      dv = dvn;
      ab = abn;
#pragma unroll
      for (int u = 0; u < MAXU; ++u) cc[u] = cn[u];
      stage_desc(k + 1 + LAG);  // (into descriptor k + 1's place: read and waited for just above)
#ifdef SLQ_DEBUG_TIMES
      if (auto *q = dbg_of(k + 1)) q[1] = __builtin_amdgcn_s_memrealtime();
#endif
    }
#undef SLQ_RD_LINE
#undef SLQ_RD_LINES
#undef SLQ_WAIT_LINES
  } else if (ntiles > 0) {
    // ---------------- consumer ----------------
    const int cwv = wave - RG::kLoaders;
    const int grp = cwv / NC, cw = cwv % NC;  // this wave serves tiles grp, grp + G, ...
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
    auto load_desc = [&](int k) -> int { return tile_desc[tile_at(k) * RG::kDescWords + lane]; };
    VF xpn[MR], un[MR][NX];
    // the row-local streams (W_p, ring columns beyond W_c, W_p) of row group i of a tile, into registers
    auto fetch_group = [&](int i, int r_lo, int nrows) {
      const int lr = (cw + i * NC) * R + g;
      if (lr < nrows) {
        const int64_t ro = (int64_t)(r_lo + lr) * PW;
        if (!first) xpn[i] = stream_load<NTP>((const VF *)(wp + ro));
        if (PASS != PASS_ALPHA && RC > 2) {
#pragma unroll
          for (int q = 0; q < NX; ++q) un[i][q] = stream_load<NTP>((const VF *)(ux[q] + ro));
        }
      }
    };
    // one row group of the tile in `img`: the SpMM of the lane's row out of the image, then the pass's own arithmetic
    auto do_group = [&](const unsigned char *img, int i, int r_lo, int nrows, const VF &xp_in, const VF *u_in) {
      const unsigned char *rec = img + RG::kLines * 1024;
      const F *xl = (const F *)img + cl * V;
      const int lr = (cw + i * NC) * R + g;
      const bool live = lr < nrows;
      int p0, p1, si, valoff;
      if constexpr (R == 1) {
        const int head = ((const int *)rec)[lane & 31];  // row offsets and own-line positions, one word per lane
        valoff = lane_bcast(head, RG::kRecValOffW);
        p0 = lane_bcast(head, lr);
        p1 = lane_bcast(head, lr + 1);
        si = lane_bcast(head, RG::kRecSelfW + lr);
      } else {
        const int lrc = live ? lr : nrows - 1;  // (a lane past the tile's last row walks no entries and stores nothing)
        const int *rw = (const int *)rec;
        p0 = rw[lrc];
        p1 = live ? rw[lrc + 1] : p0;
        si = rw[RG::kRecSelfW + lrc];
        valoff = rw[RG::kRecValOffW];
      }
      const VF xp = first ? (VF)(F)0 : xp_in;
      const VF xc = *(const VF *)(xl + (size_t)si * PW);
      VF acc = (VF)(F)0;
      if constexpr (R == 1) {
        // entries fetched kChunk at a time by the first lanes and broadcast with v_readlane; entries past the row's end read
        // the row's own image line with a zero coefficient, so a chunk's image reads go out back to back
        for (int pb = p0; pb < p1; pb += kChunk) {
          const int cnt = p1 - pb;
          const int e = min(pb + (lane & (kChunk - 1)), p1 - 1);
          const int lcv = *(const int *)(rec + RG::kRecHeadB + e * 4);
          const F vav = *(const F *)(rec + valoff + e * (int)sizeof(F));
          VF x[kChunk];
#pragma unroll
          for (int q = 0; q < kChunk; ++q) x[q] = *(const VF *)(xl + (size_t)(q < cnt ? lane_bcast(lcv, q) : si) * PW);
#pragma unroll
          for (int q = 0; q < kChunk; ++q) acc += (q < cnt ? lane_bcast(vav, q) : (F)0) * x[q];
        }
      } else {
        // R rows at once: every lane reads its own row's entries (the lanes of a row read the same words: LDS broadcast)
        for (int pb = p0; pb < p1; pb += kChunk) {
          int lc[kChunk];
          F va[kChunk];
#pragma unroll
          for (int q = 0; q < kChunk; ++q) {
            const int e = min(pb + q, p1 - 1);
            lc[q] = *(const int *)(rec + RG::kRecHeadB + e * 4);
            va[q] = *(const F *)(rec + valoff + e * (int)sizeof(F));
          }
          VF x[kChunk];
#pragma unroll
          for (int q = 0; q < kChunk; ++q) x[q] = *(const VF *)(xl + (size_t)lc[q] * PW);
#pragma unroll
          for (int q = 0; q < kChunk; ++q) acc += (pb + q < p1 ? va[q] : (F)0) * x[q];
        }
      }
      if (R == 1 || live) {
        const int64_t ro = (int64_t)(r_lo + lr) * PW;
        VF w = sc * acc;
        if (!first) w -= cp * xp;
        if (PASS == PASS_ALPHA) {
          acc1 += (sc * xc) * w;
        } else if (PASS == PASS_SPMM) {
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
            dacc[q] += u_in[q - 2] * w;
            gacc[q] += u_in[q - 2] * xc;
          }
        } else {
          w -= cb * xc;
          if constexpr (RC > 0) w -= gm[0] * xc;
          if constexpr (RC > 1) w -= gm[1] * xp;
#pragma unroll
          for (int q = 2; q < RC; ++q) w -= gm[q] * u_in[q - 2];
          if (!nostore) stream_store<NTP>((VF *)(wn + ro), w);
          acc1 += w * w;
          if constexpr (PASS == PASS_UPDATEG) {
            // the new vector against every ring column the pass has in registers: the Gram row W_{j+1} . W_{j-q} from which
            // the NEXT step's projections are assembled without reading those columns again (slq.hip: gram sequence)
            if constexpr (RC > 0) dacc[0] += w * xc;
            if constexpr (RC > 1) dacc[1] += w * xp;
#pragma unroll
            for (int q = 2; q < RC; ++q) dacc[q] += w * u_in[q - 2];
          } else {
            accx += w * xc;
          }
        }
      }
    };
    // The alpha-only pass has nothing but this to do per tile, and a row is a chain of three LDS round trips (record header ->
    // its entries -> their image lines): with the loaders no longer in the way (r03: four of them land a 5-point grid's upper
    // tiles in 0.35 ms) the consumers' chains were the pass's 0.54 ms. Here a wave walks ALL its rows of the tile together -
    // every row's entries requested before any is used, then every row's lines - so a tile costs three round trips, not 3 MR.
    auto do_alpha_joint = [&](const unsigned char *img, int nrows, const VF *xp_in) {
      const unsigned char *rec = img + RG::kLines * 1024;
      const F *xl = (const F *)img + cl * V;
      const int head = ((const int *)rec)[lane & 31];
      const int valoff = lane_bcast(head, RG::kRecValOffW);
      int p0[MR], p1[MR], si[MR], lcv[MR];
      F vav[MR];
      VF xc[MR];
      bool live[MR];
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const int lr = cw + i * NC;
        live[i] = lr < nrows;
        const int lrc = live[i] ? lr : 0;  // (a row group past the tile's end walks no entries and adds nothing)
        p0[i] = lane_bcast(head, lrc);
        p1[i] = live[i] ? lane_bcast(head, lrc + 1) : p0[i];
        si[i] = lane_bcast(head, RG::kRecSelfW + lrc);
        const int e = max(min(p0[i] + (lane & (kChunk - 1)), p1[i] - 1), 0);
        lcv[i] = *(const int *)(rec + RG::kRecHeadB + e * 4);
        vav[i] = *(const F *)(rec + valoff + e * (int)sizeof(F));
        xc[i] = *(const VF *)(xl + (size_t)si[i] * PW);
      }
      VF x[MR][kChunk];
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int q = 0; q < kChunk; ++q) x[i][q] = *(const VF *)(xl + (size_t)(q < p1[i] - p0[i] ? lane_bcast(lcv[i], q) : si[i]) * PW);
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const int cnt0 = p1[i] - p0[i];
        VF acc = (VF)(F)0;
#pragma unroll
        for (int q = 0; q < kChunk; ++q) acc += (q < cnt0 ? lane_bcast(vav[i], q) : (F)0) * x[i][q];
        for (int pb = p0[i] + kChunk; pb < p1[i]; pb += kChunk) {  // rows of more than kChunk entries: the rest as do_group walks them
          const int cnt = p1[i] - pb;
          const int e = min(pb + (lane & (kChunk - 1)), p1[i] - 1);
          const int lcw = *(const int *)(rec + RG::kRecHeadB + e * 4);
          const F vaw = *(const F *)(rec + valoff + e * (int)sizeof(F));
          VF y[kChunk];
#pragma unroll
          for (int q = 0; q < kChunk; ++q) y[q] = *(const VF *)(xl + (size_t)(q < cnt ? lane_bcast(lcw, q) : si[i]) * PW);
#pragma unroll
          for (int q = 0; q < kChunk; ++q) acc += (q < cnt ? lane_bcast(vaw, q) : (F)0) * y[q];
        }
        if (live[i]) {
          VF w = sc * acc;
          if (!first) w -= cp * xp_in[i];
          acc1 += (sc * xc[i]) * w;
        }
      }
    };
    // ... and on a stream with padded rows (the upper-triangle streams; slq.hip: build_ring_stream) nothing per entry is conditional
    // and nothing goes through v_readlane: a row's four line numbers and four coefficients are aligned 16-/32-byte LDS reads that
    // every lane makes for itself (the lanes of a row read one address: a broadcast), a line's address is one VALU add, an entry
    // two FMAs. ~80 instructions per tile and wave instead of ~350 with a branch per entry - the consumers' instruction stream,
    // not the memory system, was this pass's limit. Row groups are walked two at a time (registers: 16 waves leave 128 each).
    auto do_alpha_padded = [&](const unsigned char *img, int nrows, const VF *xp_in) {
      using I4 = int __attribute__((ext_vector_type(4)));
      typedef F F4 __attribute__((ext_vector_type(4), aligned(16)));  // (four doubles: two 16-byte reads)
      const unsigned char *rec = img + RG::kLines * 1024;
      const F *xl = (const F *)img + cl * V;
      const int *rw = (const int *)rec;
      const int valoff = rw[RG::kRecValOffW];
      auto batch = [&](auto i0_c, auto cnt_c) {
        constexpr int I0 = decltype(i0_c)::value, CNT = decltype(cnt_c)::value;
        int p0[CNT], p1[CNT], si[CNT];
        bool live[CNT];
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
          const int lr = (cw + (I0 + i) * NC) * R + g;
          live[i] = lr < nrows;
          const int lrc = live[i] ? lr : 0;  // (a lane past the tile's last row walks row 0's first chunk and adds nothing)
          p0[i] = rw[lrc];
          p1[i] = live[i] ? rw[lrc + 1] : 0;
          si[i] = rw[RG::kRecSelfW + lrc];
        }
        I4 lc[CNT];
        F4 va[CNT];
        VF xc[CNT];
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
          lc[i] = *(const I4 *)(rec + RG::kRecHeadB + p0[i] * 4);
          va[i] = *(const F4 *)(rec + valoff + p0[i] * (int)sizeof(F));
          xc[i] = *(const VF *)(xl + (size_t)si[i] * PW);
        }
        VF x[CNT][4];
#pragma unroll
        for (int i = 0; i < CNT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) x[i][q] = *(const VF *)(xl + (size_t)lc[i][q] * PW);
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
          VF acc = va[i][0] * x[i][0];
#pragma unroll
          for (int q = 1; q < 4; ++q) acc += va[i][q] * x[i][q];
          // rows of more than four entries (R > 1: as long as some lane group's row has more; the others add zeros)
          for (int pb = p0[i] + 4; __builtin_amdgcn_ballot_w64(pb < p1[i]) != 0; pb += 4) {
            const bool on = pb < p1[i];
            const int pc = on ? pb : p0[i];
            const I4 lw = *(const I4 *)(rec + RG::kRecHeadB + pc * 4);
            const F4 vw = *(const F4 *)(rec + valoff + pc * (int)sizeof(F));
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += (on ? vw[q] : (F)0) * *(const VF *)(xl + (size_t)lw[q] * PW);
          }
          if (live[i]) {
            VF w = sc * acc;
            if (!first) w -= cp * xp_in[I0 + i];
            acc1 += (sc * xc[i]) * w;
          }
        }
      };
      if constexpr (MR == 1) batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      else {
        batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        if constexpr (MR == 3) batch(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
        else if constexpr (MR >= 4) batch(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
        static_assert(MR <= 4, "row groups of a tile per consumer wave");
      }
    };
    int dcur = load_desc(grp), dnext = load_desc(grp + G);
    {
      const int r_lo0 = lane_bcast(dcur, kDescRow0), nrows0 = lane_bcast(dcur, kDescRows);
#pragma unroll
      for (int i = 0; i < MR; ++i) fetch_group(i, r_lo0, nrows0);
    }
    bool ok = grp < ntiles;
    for (int k = grp; k < ntiles && ok; k += G) {
      const int slot = k % NS;
      const int r_lo = lane_bcast(dcur, kDescRow0), nrows = lane_bcast(dcur, kDescRows);
      const int dnext2 = load_desc(k + 2 * G);
      const bool more = k + G < ntiles;
      const int r_lo_n = lane_bcast(dnext, kDescRow0), nrows_n = more ? lane_bcast(dnext, kDescRows) : 0;
      const unsigned char *img = slots + (size_t)slot * RG::kSlotBytes;
#ifdef SLQ_DEBUG_TIMES
      unsigned long long *dbg = (dbg_base && blockIdx.x == 0 && blockIdx.y == 0 && cw == 0 && k < 256) ? dbg_base + (size_t)k * 8 : nullptr;
#endif
      if constexpr (!kRefill) {
        VF xpc[MR], uc[MR][NX];
#pragma unroll
        for (int i = 0; i < MR; ++i) {
          xpc[i] = xpn[i];
#pragma unroll
          for (int q = 0; q < NX; ++q) uc[i][q] = un[i][q];
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) fetch_group(i, r_lo_n, nrows_n);
#ifdef SLQ_DEBUG_TIMES
        if (dbg && lane == 0) dbg[4] = __builtin_amdgcn_s_memrealtime();
#endif
        ok = spin(ready + slot, RG::kLoaders * (k / NS + 1));
        if (!ok) break;
#ifdef SLQ_DEBUG_TIMES
        if (dbg && lane == 0) dbg[5] = __builtin_amdgcn_s_memrealtime();
#endif
        if constexpr (PASS == PASS_ALPHA && R == 1) {
          if (padded_rows) do_alpha_padded(img, nrows, xpc);
          else do_alpha_joint(img, nrows, xpc);
        } else if (PASS == PASS_ALPHA && padded_rows) {
          do_alpha_padded(img, nrows, xpc);
        } else {
#pragma unroll
          for (int i = 0; i < MR; ++i) {
            if ((cw + i * NC) * R >= nrows) break;
            do_group(img, i, r_lo, nrows, xpc[i], uc[i]);
          }
        }
      } else {
#ifdef SLQ_DEBUG_TIMES
        if (dbg && lane == 0) dbg[4] = __builtin_amdgcn_s_memrealtime();
#endif
        ok = spin(ready + slot, RG::kLoaders * (k / NS + 1));
        if (!ok) break;
#ifdef SLQ_DEBUG_TIMES
        if (dbg && lane == 0) dbg[5] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int i = 0; i < MR; ++i) {
          if ((cw + i * NC) * R < nrows) do_group(img, i, r_lo, nrows, xpn[i], un[i]);
          fetch_group(i, r_lo_n, nrows_n);  // the same registers, for this wave's next tile
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
  // partial sums of the consumer waves, column by column in wave order (loaders hold zeros), the R row groups of a wave in turn
  const int64_t nblk = gridDim.x;
  auto reduce_out = [&](const VF &a, double *out) {
#pragma unroll
    for (int v = 0; v < V; ++v) red[(wave * 64 + lane) * V + v] = (double)a[v];
    __syncthreads();
    if ((int)threadIdx.x < PW) {
      const int t = threadIdx.x, c = t / V, v = t % V;
      double sum = 0.0;
      for (int w = 0; w < WAVES; ++w)
#pragma unroll
        for (int gg = 0; gg < R; ++gg) sum += red[(w * 64 + gg * LPR + c) * V + v];
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
  } else if (PASS == PASS_UPDATEG) {
    // slab 0: ||w||^2; slab 1 + q: w . W_{j-q}
    reduce_out(acc1, part + (int64_t)blockIdx.x * bpad + panel * PW);
#pragma unroll
    for (int i = 0; i < RC; ++i) reduce_out(dacc[i], part + ((int64_t)(1 + i) * nblk + blockIdx.x) * bpad + panel * PW);
  } else {
    reduce_out(acc1, part + (int64_t)blockIdx.x * bpad + panel * PW);
    if (PASS == PASS_UPDATE && xt) reduce_out(accx, part + (nblk + blockIdx.x) * bpad + panel * PW);
  }
}

}  // namespace slq
