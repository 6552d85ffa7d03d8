// slq_common.hpp — what every translation unit of libslq shares: vector types, cache-policy helpers, panel geometry, the
// pass codes, the tile descriptors and the constants of the ring-fed tile passes. Templates, constants and inline functions
// only (slq_kernels.hpp also defines non-template kernels and is included by slq.hip alone).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slq {



template <typename F> struct VecT;
template <> struct VecT<double> {
  typedef double type __attribute__((ext_vector_type(2)));
  static constexpr int V = 2;
};
template <> struct VecT<float> {
  typedef float type __attribute__((ext_vector_type(4)));
  static constexpr int V = 4;
};

// ---- cache-policy helpers ---------------------------------------------------------------------
// Streamed-once panel rows should not evict the gather window from the XCD's 4 MiB L2.
// POLICY 0: plain; 1: nontemporal hint (global_load/store ... nt);
// stores only, 2: write-through-and-drop (global_store ... sc1; MI355X_MICROARCH.md 'stores of each
// flavour': sc1 stores do not keep the line in L2).
// (Round 2 tried the other flavours on the streamed rows of the fused passes - sc1, sc0 sc1 and sc1 nt loads, sc1
// and sc0 sc1 stores: none fetched less than nt, the sc1 loads 19 % more)
template <int POLICY, typename VF> __device__ __forceinline__ VF stream_load(const VF *p) {
  if (POLICY == 1) return __builtin_nontemporal_load(p);
  return *p;
}
template <int POLICY, typename VF> __device__ __forceinline__ void stream_store_impl(VF *p, VF v) {
  if (POLICY == 1) __builtin_nontemporal_store(v, p);
  else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
  else *p = v;
}
template <int POLICY> __device__ __forceinline__ void stream_store(
    double __attribute__((ext_vector_type(2))) * p, double __attribute__((ext_vector_type(2))) v) {
  stream_store_impl<POLICY>(p, v);
}
template <int POLICY> __device__ __forceinline__ void stream_store(
    float __attribute__((ext_vector_type(4))) * p, float __attribute__((ext_vector_type(4))) v) {
  stream_store_impl<POLICY>(p, v);
}

#ifndef SLQ_SWEEP_ST
#define SLQ_SWEEP_ST 0  // store flavour of the in-place sweeps (k_reorth_dot's first chunk, k_reorth_update): A/B builds
#endif
#ifndef SLQ_SWEEP_LDW
#define SLQ_SWEEP_LDW 0  // load flavour of the row that is stored again
#endif
constexpr int kBlock = 512;          // threads per workgroup for the sweep kernels (8 waves)
constexpr int kWaves = kBlock / 64;
constexpr int kReorthChunk = 16;     // reorth columns whose dot accumulators live in registers
constexpr int kMaxDeg = 512;
constexpr int kFusedMaxR = 8;        // fused recompute passes handle up to this many reorth columns
#ifndef SLQ_UPD_UR
#define SLQ_UPD_UR 2
#endif         // upper bound on the Krylov degree (LDS sizing of the QL kernel)

// Slot of Lanczos vector t in a ring of S slots. t may be negative: vectors "before the run" are the
// caller's stale ring columns of the single-vector drop-in entry (slq.hip:lanczos_single), stored
// at the far end of the ring.
__device__ __forceinline__ int ring_slot(int t, int S) {
  const int m = t % S;
  return m < 0 ? m + S : m;
}

template <typename F, int LPR> struct Geo {
  static constexpr int V = VecT<F>::V;
  static constexpr int PW = LPR * V;   // probes per panel row
  static constexpr int RPW = 64 / LPR; // rows per wave instruction
};

enum { PASS_ALPHA = 0, PASS_DOTS = 1, PASS_UPDATE = 2, PASS_ADOTS = 3, PASS_SPMM = 4 /* ring kernels only: k_spmm_3term's job */,
       PASS_UPDATEG = 5 /* k_ring_pass only: PASS_UPDATE that also takes w against every ring column it reads (slq_ring.hpp) */ };

struct TileMeta {
  const int32_t *tile_row;   // [ntiles + 1] first (stored) row of each tile
  const int32_t *tile_ptr;   // [ntiles + 1] offsets into tile_cols
  const int32_t *tile_cols;  // distinct stored row indices each tile reads, ascending (+ kCsrPad spare entries)
  const int32_t *lcol;       // [nnz + kCsrPad] position of every nonzero's column in its tile's list
  const int32_t *self_idx;   // [n] position of the row itself
  int32_t xcd_tile[9];       // tiles of XCD chunk x: [xcd_tile[x], xcd_tile[x + 1])
  int max_cols;              // longest list (LDS sizing)
};
struct TileRanges {
  int32_t first[9];  // TileMeta::xcd_tile, by value in the kernel arguments
};

#ifndef SLQ_RING_LOADERS
#define SLQ_RING_LOADERS 2
#endif
#ifndef SLQ_RING_SLOTS
#define SLQ_RING_SLOTS 4
#endif
#ifndef SLQ_RING_AUX
#define SLQ_RING_AUX 0  // cache policy of the image DMAs (1 sc0, 2 nt, 16 sc1)
#endif
#ifndef SLQ_RING_LAG
#define SLQ_RING_LAG 2
#endif
#ifndef SLQ_RING_GROUPS
#define SLQ_RING_GROUPS 2
#endif
#ifndef SLQ_RING_CHUNK
#define SLQ_RING_CHUNK 4
#endif
#ifndef SLQ_RING_WAVES
#define SLQ_RING_WAVES 16
#endif
#ifndef SLQ_RING_ROWS
#define SLQ_RING_ROWS 14
#endif
#ifndef SLQ_RING_COLS
#define SLQ_RING_COLS 36
#endif
constexpr int kRingWaves = SLQ_RING_WAVES;  // a consumer's work per row is a chain of LDS latencies: many consumer waves hide it
constexpr int kRingBlock = kRingWaves * 64;
constexpr int kRingChunk = SLQ_RING_CHUNK;  // nonzeros of a row gathered per batch (power of two)
constexpr int kRingLoaders = SLQ_RING_LOADERS;
constexpr int kRingGroups = SLQ_RING_GROUPS;  // consumer groups taking the tiles in turn: a wave's prefetch runs kRingGroups tiles ahead
constexpr int kRingSlots = SLQ_RING_SLOTS;
constexpr int kRingLag = SLQ_RING_LAG;        // a loader's tiles in flight
constexpr int kRingTileRows = SLQ_RING_ROWS;  // two rows per consumer wave of a group
constexpr int kRingTileCols = SLQ_RING_COLS;  // distinct panel rows per tile at most
constexpr int kRingTileNnz = 112;             // nonzeros per tile at most: 128 B of header + 112 x (4 + 8) B fit 1.5 KiB (slq_ring.hpp: R merged tiles per slot)
constexpr int kRingMetaBytes = 2048;          // 4 slots x (36 + 2) KiB + kRingHeadBytes = 156 KiB
constexpr int kRingHeadBytes = 4096;          // flag words + the loaders' descriptor staging
constexpr int kRingSpinMax = 1 << 20;         // ~0.1 s of polling
constexpr int kRingMaxR = 3;                  // ring columns per step served (more: 128 VGPRs at 16 waves do not hold the sums)
// descriptor words (tile_desc[t * 64 + ...])
constexpr int kDescCols = 0, kDescRecOff = 1, kDescRecChunks = 2, kDescRow0 = 3, kDescRows = 4, kDescList = 8;
// The line list of an R = 1 descriptor is stored de-interleaved (r03): line d at word kDescList + ring1_list_pos(d), i.e. the
// even lines first, then the odd ones - the lines of each of TWO loader waves are then consecutive words, which a loader
// fetches with two wide scalar loads instead of eighteen single ones (slq_ring.hpp). Merged tiles (R > 1) keep line d at d.
constexpr int kRing1ListHalf = (kRingTileCols + 1) / 2;
__host__ __device__ inline int ring1_list_pos(int d) { return (d & 1) * kRing1ListHalf + (d >> 1); }
// record words: [0 .. rows] row offsets into the record's own nonzeros, [15] byte offset of the values,
// [16 .. 16 + rows) line of each row's own panel row, then from byte 128 the column lines (int32) and the values (F)
constexpr int kRecValOff = 15, kRecSelf = 16, kRecHeadBytes = 128;
static_assert((kRingWaves - kRingLoaders) % kRingGroups == 0 && kRingSlots % kRingGroups == 0, "every slot is served by one consumer group");
static_assert(kRingLag < kRingSlots && kRingLoaders < kRingWaves && kRingLoaders <= 3 && kRingTileRows < kRecValOff && kRecSelf + kRingTileRows <= 32 && kRingTileCols <= 64 - kDescList, "ring geometry");
static_assert(kRingWaves <= 16 && (size_t)kRingSlots * (kRingTileCols * 1024 + kRingMetaBytes) + kRingHeadBytes <= 160 * 1024, "ring slots must fit the LDS");
static_assert((size_t)kRingWaves * 64 * 4 * 8 <= (size_t)kRingSlots * (kRingTileCols * 1024 + kRingMetaBytes), "the final reduction reuses the slots");
static_assert(kRecHeadBytes + ((kRingTileNnz + 3) / 4 * 4) * (4 + 8) <= kRingMetaBytes, "a tile's record must fit its slot");
static_assert(((kRingTileCols + kRingLoaders - 1) / kRingLoaders + 2) * kRingLag + 1 <= 63, "a loader's DMAs in flight are counted by vmcnt");

// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n: the instruction takes an immediate, so n indexes a table of 57
// {s_waitcnt, s_branch} pairs (8 bytes each) entered by a computed s_setpc - seven scalar instructions whatever n is. (A `switch`
// over the cases, and a tree of ifs as well, compiled into chains of compares and branches, up to ~100 instructions per call:
// a quarter of a ring loader's time per tile, r03 scripts/ring_timeline.py.) s[98:99] and SCC are clobbered.
__device__ __forceinline__ void wait_vmcnt_at_most(int n) {
  int t = __builtin_amdgcn_readfirstlane(n < 0 ? 0 : (n > 56 ? 56 : n));  // (a stricter wait than asked for is always safe)
  asm volatile(
      "s_lshl_b32 %0, %0, 3\n\t"
      "s_getpc_b64 s[98:99]\n\t"      // the address of the next instruction; the table starts 20 bytes (five instructions) further
      "s_add_u32 s98, s98, %0\n\t"
      "s_addc_u32 s99, s99, 0\n\t"
      "s_add_u32 s98, s98, 20\n\t"
      "s_addc_u32 s99, s99, 0\n\t"
      "s_setpc_b64 s[98:99]\n\t"
      "s_waitcnt vmcnt(0)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(1)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(2)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(3)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(4)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(5)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(6)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(7)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(8)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(9)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(10)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(11)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(12)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(13)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(14)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(15)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(16)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(17)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(18)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(19)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(20)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(21)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(22)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(23)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(24)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(25)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(26)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(27)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(28)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(29)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(30)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(31)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(32)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(33)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(34)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(35)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(36)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(37)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(38)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(39)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(40)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(41)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(42)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(43)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(44)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(45)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(46)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(47)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(48)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(49)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(50)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(51)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(52)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(53)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(54)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(55)\n\ts_branch .Lslq_vmw%=\n\t"
      "s_waitcnt vmcnt(56)\n"
      ".Lslq_vmw%=:"
      : "+s"(t)
      :
      : "memory", "s98", "s99", "scc");
}

// broadcast lane `l` (wave-uniform) of a value held one entry per lane
__device__ __forceinline__ int lane_bcast(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float lane_bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_bcast(double v, int l) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

}  // namespace slq
