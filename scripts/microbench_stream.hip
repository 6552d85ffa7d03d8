// Microbenchmark: how fast can MI355X stream the three access shapes the SLQ sweeps use?
//   copy  (1R+1W), triad-in-place (2R+1W: w -= c*q), dot-only (2R)
// over a working set that is either HBM-sized (2 GiB panels) or Infinity-Cache-sized.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mb scripts/microbench_stream.hip ; run: /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// mode 0: copy dst=src ; 1: in-place triad w -= c*q ; 2: read-only dot (w,q) ; 3: in-place scale w *= c
// NT: nontemporal loads/stores
template <int UR, int NT>
__global__ __launch_bounds__(512) void k_stream(d2 *w, const d2 *q, int64_t nvec, int mode, double c, double *sink) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  d2 acc = (d2)0.0;
  for (int64_t i = tid; i < nvec; i += UR * stride) {
    d2 a[UR], b[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t k = i + u * stride;
      if (k < nvec) {
        if (mode != 0) a[u] = NT ? __builtin_nontemporal_load(w + k) : w[k];
        if (mode != 3) b[u] = NT ? __builtin_nontemporal_load(q + k) : q[k];
      }
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t k = i + u * stride;
      if (k < nvec) {
        d2 x;
        if (mode == 0) x = b[u];
        else if (mode == 1) x = a[u] - c * b[u];
        else if (mode == 3) x = a[u] * c;
        else x = a[u] * b[u];
        if (mode != 2) { if (NT) __builtin_nontemporal_store(x, w + k); else w[k] = x; }
        acc += x * x;
      }
    }
  }
  if (acc[0] + acc[1] == 1.2345e-300) sink[0] = acc[0];
}

template <int UR, int NT>
static double run(d2 *w, d2 *q, int64_t nvec, int mode, int blocks, int reps, double *sink) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) k_stream<UR, NT><<<blocks, 512>>>(w, q, nvec, mode, 1e-9, sink);
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) k_stream<UR, NT><<<blocks, 512>>>(w, q, nvec, mode, 1e-9, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const char *names[] = {"copy 1R+1W", "triad 2R+1W inplace", "dot 2R", "scale 1R+1W inplace"};
  const int nbytes_factor[] = {2, 3, 2, 2};
  double *sink; CK(hipMalloc(&sink, 8));
  for (double gib : {2.0, 0.0625, 0.03125}) {
    const int64_t nvec = (int64_t)(gib * (1ll << 30)) / 16;
    d2 *w, *q;
    CK(hipMalloc(&w, nvec * 16)); CK(hipMalloc(&q, nvec * 16));
    CK(hipMemset(w, 0, nvec * 16)); CK(hipMemset(q, 0, nvec * 16));
    printf("== panel %.4f GiB each (%lld vec16)\n", gib, (long long)nvec);
    for (int mode = 0; mode < 4; ++mode) {
      for (int blocks : {256, 512, 1024, 2048, 4096}) {
        const int reps = gib > 1 ? 10 : 200;
        double t1 = run<1, 0>(w, q, nvec, mode, blocks, reps, sink);
        double t4 = run<4, 0>(w, q, nvec, mode, blocks, reps, sink);
        double t8 = run<8, 0>(w, q, nvec, mode, blocks, reps, sink);
        double t4n = run<4, 1>(w, q, nvec, mode, blocks, reps, sink);
        const double gb = nbytes_factor[mode] * nvec * 16 / 1e9;
        printf("  %-22s blocks=%5d  UR1 %7.1f  UR4 %7.1f  UR8 %7.1f  UR4+nt %7.1f  GB/s\n", names[mode], blocks,
               gb / t1 * 1e3, gb / t4 * 1e3, gb / t8 * 1e3, gb / t4n * 1e3);
      }
    }
    CK(hipFree(w)); CK(hipFree(q));
  }
  return 0;
}
