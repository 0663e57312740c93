// Issue rate and dependent latency of v_mfma_f64_16x16x4_f64 on gfx950 (what bounds csrc/m355_mvn.hip's solve).
//   chains = 1: every MFMA accumulates into the previous one's result (dependent through SrcC)
//   chains = 2, 4, 8: that many independent accumulators, round robin
//   waves per SIMD 1 or 2 (workgroups of 256 threads = one wavefront per SIMD of a CU; two workgroups per CU for 2)
// Prints cycles per MFMA per SIMD (from the wall clock and the 2.4 GHz shader clock) and the implied chip TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(e)); exit(1);} } while(0)
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int CH> __global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double seed) {
  d4_t acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = d4_t{seed, seed, seed, seed};
  const double a = seed * (threadIdx.x & 15), b = seed * (threadIdx.x >> 4);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678) out[0] = s;
}

template <int CH> void run(int wgs_per_cu, double* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_mfma<CH>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, 100, 1e-30);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_mfma<CH>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters, 1e-30);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_simd = (double)iters * CH * wgs_per_cu;
  const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
  const double tflops = 2048.0 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12;
  printf("chains %d  waves/SIMD %d : %.3f ms  %.1f cycles (2.4 GHz) per MFMA per SIMD  %.1f TFLOP/s chip\n", CH, wgs_per_cu, ms, cyc, tflops);
}

int main() {
  double* out;
  CHECK(hipMalloc(&out, 64));
  for (int w = 1; w <= 2; ++w) {
    run<1>(w, out);
    run<2>(w, out);
    run<4>(w, out);
    run<8>(w, out);
  }
  return 0;
}
