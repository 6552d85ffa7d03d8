// slq_ring.hip — one object per (RING_F, RING_LPR): the ring-fed tile passes of that panel geometry (slq_ring.hpp) and
// their launcher. Built by __graft_entry__.build_libslq() as
//   hipcc --offload-arch=gfx950 -O3 -c -DRING_F=double -DRING_LPR=32 -DRING_TAG=f64_l32 slq_ring.hip -o ring_f64_l32.o
#include "slq_ring_api.h"
#include "slq_ring.hpp"
#ifndef SLQ_NO_FA  // (A/B builds of other ring geometries: the fused form needs four slots)
#include "slq_ring_fa.hpp"
#endif

#ifndef RING_F
#error "compile with -DRING_F=<double|float> -DRING_LPR=<64|32|16> -DRING_TAG=<f64_l64|...>"
#endif

using namespace slq;
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

namespace {
using F = RING_F;
constexpr int L = RING_LPR;

// steps of up to kRingMaxR ring columns: 16 waves; more: 8 waves (256 VGPRs per wave)
template <int PASS, int RC> constexpr int waves_of() { return RC <= kRingMaxR ? 16 : 8; }

template <int PASS, int RC, int GEO = 0> void launch(const RingArgs &a) {
  constexpr int W = waves_of<PASS, RC>();
  k_ring_pass<F, PASS, 1, RC, L, W, GEO><<<a.grid, dim3(W * 64), RingGeo<L, W, GEO>::kLdsBytes, a.st>>>(
      a.n, a.desc, a.rec, a.xr, (F *)a.ring, a.slot_stride, a.S, a.j, a.coefA, a.coefB, a.gamma, a.part, a.bpad, a.xt, a.fail, a.dbg);
}
template <int PASS, int RC, int GEO = 0> hipError_t prepare() {
  constexpr int W = waves_of<PASS, RC>();
  return hipFuncSetAttribute((const void *)k_ring_pass<F, PASS, 1, RC, L, W, GEO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int PASS, int RC> int vgprs() {
  constexpr int W = waves_of<PASS, RC>();
  hipFuncAttributes at;
  if (hipFuncGetAttributes(&at, (const void *)k_ring_pass<F, PASS, 1, RC, L, W, 0>) != hipSuccess) return -1;
  return at.numRegs + ((int)(at.localSizeBytes > 0) << 16);  // bit 16: the kernel spills to scratch
}

#define RC_SWITCH(PASS, FN, ...)          \
  switch (rc) {                           \
    case 1: FN<PASS, 1>(__VA_ARGS__); break; \
    case 2: FN<PASS, 2>(__VA_ARGS__); break; \
    case 3: FN<PASS, 3>(__VA_ARGS__); break; \
    case 4: FN<PASS, 4>(__VA_ARGS__); break; \
    case 5: FN<PASS, 5>(__VA_ARGS__); break; \
    case 6: FN<PASS, 6>(__VA_ARGS__); break; \
    case 7: FN<PASS, 7>(__VA_ARGS__); break; \
    case 8: FN<PASS, 8>(__VA_ARGS__); break; \
    default: return -1;                   \
  }
}  // namespace

int CAT(slq_ring_launch_, RING_TAG)(const RingArgs &a) {
  const int rc = a.rc;
  switch (a.pass) {
    case PASS_ALPHA:
      if (a.staged) launch<PASS_ALPHA, 0, 1>(a);  // loaders through registers instead of LDS-DMA (slq_ring.hpp: GEO 1)
      else launch<PASS_ALPHA, 0>(a);
      return 0;
    case PASS_SPMM: launch<PASS_SPMM, 0>(a); return 0;
    case PASS_ADOTS: RC_SWITCH(PASS_ADOTS, launch, a) return 0;
    case PASS_UPDATE:
      if (rc == 0) { launch<PASS_UPDATE, 0>(a); return 0; }
      RC_SWITCH(PASS_UPDATE, launch, a) return 0;
    case PASS_UPDATEG: RC_SWITCH(PASS_UPDATEG, launch, a) return 0;
    default: return -1;
  }
}

#ifndef SLQ_NO_FA
// the update pass with the next step's alpha dot fused in (slq_ring_fa.hpp): whole-row panels only
namespace {
template <int RC> int launch_fa(const RingArgs &a) {
  if constexpr (L == 64) {
    k_ring_fa<F, RC><<<a.grid, dim3(16 * 64), RingGeo<64, 16, 0>::kLdsBytes, a.st>>>(a.n, a.desc, a.rec, a.desc_a, a.rec_a, a.xr, (F *)a.ring, a.slot_stride, a.S, a.j, a.coefA,
                                                                                     a.coefB, a.gamma, a.part, a.bpad, a.cnt, a.cnt_rounds, a.gen, a.xcc_tab, a.fail, a.xt);
    return 0;
  } else {
    return -1;
  }
}
template <int RC> hipError_t prepare_fa() {
  if constexpr (L == 64) return hipFuncSetAttribute((const void *)k_ring_fa<F, RC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  else return hipSuccess;
}
template <int RC> int vgprs_fa() {
  if constexpr (L == 64) {
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, (const void *)k_ring_fa<F, RC>) != hipSuccess) return -1;
    return at.numRegs + ((int)(at.localSizeBytes > 0) << 16);
  } else {
    return -1;
  }
}
}  // namespace

int CAT(slq_ring_fa_launch_, RING_TAG)(const RingArgs &a) {
  if (a.pass == PASS_UPDATE && a.rc == 0) return launch_fa<0>(a);
  if (a.pass != PASS_UPDATEG) return -1;
  switch (a.rc) {
    case 1: return launch_fa<1>(a);
    case 2: return launch_fa<2>(a);
    case 3: return launch_fa<3>(a);
    default: return -1;
  }
}
int CAT(slq_ring_fa_vgprs_, RING_TAG)(int rc) {
  switch (rc) {
    case 0: return vgprs_fa<0>();
    case 1: return vgprs_fa<1>();
    case 2: return vgprs_fa<2>();
    case 3: return vgprs_fa<3>();
    default: return -1;
  }
}

#else
int CAT(slq_ring_fa_launch_, RING_TAG)(const RingArgs &) { return -1; }
int CAT(slq_ring_fa_vgprs_, RING_TAG)(int) { return -1; }
namespace {
template <int RC> hipError_t prepare_fa() { return hipSuccess; }
}  // namespace
#endif

hipError_t CAT(slq_ring_prepare_, RING_TAG)() {
  hipError_t e = prepare<PASS_ALPHA, 0>();
  if (e == hipSuccess) e = prepare_fa<0>();
  if (e == hipSuccess) e = prepare_fa<1>();
  if (e == hipSuccess) e = prepare_fa<2>();
  if (e == hipSuccess) e = prepare_fa<3>();
  if (e == hipSuccess) e = prepare<PASS_ALPHA, 0, 1>();
  if (e == hipSuccess) e = prepare<PASS_SPMM, 0>();
  if (e == hipSuccess) e = prepare<PASS_UPDATE, 0>();
#define PREP(R) \
  if (e == hipSuccess) e = prepare<PASS_ADOTS, R>(); \
  if (e == hipSuccess) e = prepare<PASS_UPDATE, R>(); \
  if (e == hipSuccess) e = prepare<PASS_UPDATEG, R>();
  PREP(1) PREP(2) PREP(3) PREP(4) PREP(5) PREP(6) PREP(7) PREP(8)
#undef PREP
  return e;
}

// registers of one variant (diagnostics: scripts/ring_resources.py)
int CAT(slq_ring_vgprs_, RING_TAG)(int pass, int rc) {
  switch (pass) {
    case PASS_ALPHA: return vgprs<PASS_ALPHA, 0>();
    case PASS_SPMM: return vgprs<PASS_SPMM, 0>();
    case PASS_ADOTS: {
      int v = -1;
#define VG(R) if (rc == R) v = vgprs<PASS_ADOTS, R>();
      VG(1) VG(2) VG(3) VG(4) VG(5) VG(6) VG(7) VG(8)
#undef VG
      return v;
    }
    case PASS_UPDATE: {
      int v = -1;
#define VG(R) if (rc == R) v = vgprs<PASS_UPDATE, R>();
      VG(0) VG(1) VG(2) VG(3) VG(4) VG(5) VG(6) VG(7) VG(8)
#undef VG
      return v;
    }
    case PASS_UPDATEG: {
      int v = -1;
#define VG(R) if (rc == R) v = vgprs<PASS_UPDATEG, R>();
      VG(1) VG(2) VG(3) VG(4) VG(5) VG(6) VG(7) VG(8)
#undef VG
      return v;
    }
    default: return -1;
  }
}
