// slq_ring_api.h — host-side entry points of the ring-fed tile passes (slq_ring.hpp). slq_ring.hip is compiled once per
// (element type, lanes per panel row): each object defines one launcher and one attribute setter, named by its tag.
#pragma once
#include <hip/hip_runtime.h>

#include "slq_common.hpp"

struct RingArgs {
  int pass, rc;  // PASS_* of slq_kernels.hpp, ring columns of the step (compile-time variants 0..8)
  int staged;    // PASS_ALPHA: the loader waves bring the tile through their registers instead of LDS-DMA (slq_ring.hpp: GEO 1)
  dim3 grid;
  hipStream_t st;
  int n;
  const int32_t *desc;
  const char *rec;
  slq::TileRanges xr;
  void *ring;
  int64_t slot_stride;
  int S, j;
  const double *coefA, *coefB, *gamma;
  double *part;
  int bpad, xt;
  int *fail;
  unsigned long long *dbg;  // -DSLQ_DEBUG_TIMES builds: the stamp buffer (else null)
  // the update pass with the next step's alpha dot fused in (slq_ring_fa.hpp; slq_ring_fa_launch_*): the operator's interior
  // upper-triangle stream, the chunk counters [NP][8][cnt_rounds] (zeroed at the start of a run), this pass's number within the run
  // (1, 2, ...: the counters' target is gen x workgroups per XCD), the XCC table [8] (zeroed with the counters)
  const int32_t *desc_a = nullptr;
  const char *rec_a = nullptr;
  int *cnt = nullptr;
  int cnt_rounds = 0, gen = 0;
  int *xcc_tab = nullptr;
};

// launch: 0, or -1 when the object has no kernel for (pass, rc). prepare: raises the dynamic-LDS limit of all its kernels.
#define SLQ_RING_DECLARE(TAG)                   \
  int slq_ring_launch_##TAG(const RingArgs &a); \
  hipError_t slq_ring_prepare_##TAG();          \
  int slq_ring_vgprs_##TAG(int pass, int rc);   \
  int slq_ring_fa_launch_##TAG(const RingArgs &a); /* pass: PASS_UPDATEG (rc 1..3) or PASS_UPDATE (rc 0); -1: no such kernel */ \
  int slq_ring_fa_vgprs_##TAG(int rc);
SLQ_RING_DECLARE(f64_l64)
SLQ_RING_DECLARE(f32_l64)
SLQ_RING_DECLARE(f64_l32)
SLQ_RING_DECLARE(f32_l32)
SLQ_RING_DECLARE(f64_l16)
SLQ_RING_DECLARE(f32_l16)
