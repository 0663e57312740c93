// Hardware probes (not shipped): accuracy of v_rcp_f64 / v_rsq_f64 seeds and relative issue cost of
// fp64 instructions, to size the Newton steps and the per-element op budget of the log kernels.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(e)); exit(1);} } while(0)

__global__ void k_rcp(const double* x, double* r0, double* r1, double* r2, double* rs, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  r0[i] = r;
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(e, r, r);
  r1[i] = r;
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(e, r, r);
  r2[i] = r;
  rs[i] = __builtin_amdgcn_rsq(d);
}

// throughput: chains of independent ops per thread; 256 threads x 4096 blocks
template <int OP> __global__ __launch_bounds__(256) void k_tp(double* out, int iters, double seed) {
  double a[8];
  for (int j = 0; j < 8; ++j) a[j] = seed + j * 0.125 + threadIdx.x * 1e-6;
  const double c = 1.0000001, d = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (OP == 0) a[j] = __builtin_fma(a[j], c, d);
      if (OP == 1) a[j] = a[j] * c;
      if (OP == 2) a[j] = a[j] + d;
      if (OP == 3) a[j] = __builtin_amdgcn_rcp(a[j]);
      if (OP == 4) { float f = (float)a[j]; f = __builtin_amdgcn_rcpf(f); a[j] = (double)f; }
      if (OP == 5) a[j] = __builtin_amdgcn_frexp_mant(a[j]) + 1.0;
      if (OP == 6) { unsigned long long b = __builtin_bit_cast(unsigned long long, a[j]); b += 0x0000000100000000ull; a[j] = __builtin_bit_cast(double, b); }
      if (OP == 7) a[j] = (a[j] < 1.5) ? a[j] + d : a[j] * c;
      if (OP == 8) a[j] = __builtin_sqrt(a[j]) + 1.0;
      if (OP == 9) a[j] = (double)(int)a[j] + 1.25;
    }
  }
  double s = 0;
  for (int j = 0; j < 8; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void tp(const char* name, double* out, int opsPerIter) {
  const int iters = 2000, blocks = 4096;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_tp<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(k_tp<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  double lane_ops = (double)blocks * 256 * iters * 8 * opsPerIter;
  printf("%-28s %.3f ms  %.3e lane-ops/s\n", name, ms, lane_ops / (ms * 1e-3));
}

int main() {
  const int n = 1 << 20;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = 1.0 + ((s >> 11) * (1.0 / 9007199254740992.0)) * 1.5; }
  double *x, *r0, *r1, *r2, *rs;
  CHECK(hipMalloc(&x, n * 8)); CHECK(hipMalloc(&r0, n * 8)); CHECK(hipMalloc(&r1, n * 8)); CHECK(hipMalloc(&r2, n * 8)); CHECK(hipMalloc(&rs, n * 8));
  CHECK(hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_rcp, dim3(n / 256), dim3(256), 0, 0, x, r0, r1, r2, rs, n);
  std::vector<double> a0(n), a1(n), a2(n), as(n);
  CHECK(hipMemcpy(a0.data(), r0, n * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(a1.data(), r1, n * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(a2.data(), r2, n * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(as.data(), rs, n * 8, hipMemcpyDeviceToHost));
  double e0 = 0, e1 = 0, e2 = 0, es = 0;
  for (int i = 0; i < n; ++i) {
    long double t = 1.0L / (long double)h[i];
    e0 = fmax(e0, (double)fabsl((a0[i] - t) / t)); e1 = fmax(e1, (double)fabsl((a1[i] - t) / t)); e2 = fmax(e2, (double)fabsl((a2[i] - t) / t));
    long double ts = 1.0L / sqrtl((long double)h[i]);
    es = fmax(es, (double)fabsl((as[i] - ts) / ts));
  }
  printf("v_rcp_f64 max rel err: seed %.3e (%.1f bits)  +1 Newton %.3e  +2 Newton %.3e ; v_rsq_f64 seed %.3e\n", e0, -log2(e0), e1, e2, es);
  double* out; CHECK(hipMalloc(&out, 4096 * 256 * 8));
  tp<0>("v_fma_f64", out, 1); tp<1>("v_mul_f64", out, 1); tp<2>("v_add_f64", out, 1); tp<3>("v_rcp_f64", out, 1);
  tp<4>("cvt+v_rcp_f32+cvt", out, 1); tp<5>("frexp_mant+add", out, 1); tp<6>("int add on hi word", out, 1);
  tp<7>("cmp+cndmask (select of add/mul)", out, 1); tp<8>("v_sqrt_f64+add", out, 1); tp<9>("cvt i32<->f64 +add", out, 1);
  return 0;
}
