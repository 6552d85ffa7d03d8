// Isolated timing of the SLQ sweep kernels (same code as libslq) on C2-sized panels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../primate_amd/csrc/slq_kernels.hpp"
using namespace slq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename Fn> static double timeit(Fn fn, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  fn(); fn();
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError()); return ms / reps;
}

int main(int argc, char **argv) {
  const int n = 1000000, NP = 2, PW = 128, bpad = 256;
  const size_t slot = (size_t)NP * n * PW;
  double *ring; CK(hipMalloc(&ring, 3 * slot * 8)); CK(hipMemset(ring, 0, 3 * slot * 8));
  double *coef, *part; CK(hipMalloc(&coef, 64 * bpad * 8)); CK(hipMemset(coef, 0, 64 * bpad * 8));
  CK(hipMalloc(&part, (size_t)16 * 4096 * bpad * 8));
  for (int nblk : {256, 512, 1024, 2048}) {
    for (int np : {1, 2}) {
      double t = timeit([&] { k_axpy_norm<double, 64, 0><<<dim3(nblk, np), kBlock>>>(n, ring + slot, ring, coef, part, bpad); }, 10);
      printf("axpy_norm  nblk=%4d panels=%d  %.3f ms  %.1f GB/s\n", nblk, np, t, 3.0 * np * n * PW * 8 / t / 1e6);
      t = timeit([&] { k_axpy_norm<double, 64, 1><<<dim3(nblk, np), kBlock>>>(n, ring + slot, ring, coef, part, bpad); }, 10);
      printf("norm only  nblk=%4d panels=%d  %.3f ms  %.1f GB/s\n", nblk, np, t, 1.0 * np * n * PW * 8 / t / 1e6);
    }
  }
  return 0;
}
