// What v_mfma_f64_16x16x4_f64 sustains on this chip: back-to-back MFMAs on register operands, W waves per SIMD,
// every CU busy. hipcc --offload-arch=gfx950 -O3 -o mb_mfma scripts/microbench_mfma_f64.hip && ./mb_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(double *out, int iters, double a0, double b0) {
  d4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4_t)0.0;
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *out;
  hipMalloc(&out, 8 * 512 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg_per_cu : {1, 2}) {
    for (int threads : {256, 512}) {
      const int iters = 20000, NACC = 8;
      const int grid = 256 * wg_per_cu;
      k<NACC><<<grid, threads>>>(out, 100, 1.0, 2.0);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k<NACC><<<grid, threads>>>(out, iters, 1.0, 2.0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfmas = (double)grid * (threads / 64) * iters * NACC;
      printf("grid %d x %d threads: %.3f ms, %.2f TFLOP/s fp64 MFMA (2048 flop each), %.1f ns per MFMA per SIMD-slot\n", grid, threads, ms,
             mfmas * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * NACC * (double)(threads / 64) * wg_per_cu / 4.0));
    }
  }
  return 0;
}
