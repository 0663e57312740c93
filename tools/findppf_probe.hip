// Reduced case for the round-1 observation "find_ppf inlined four times into k_scalar's unrolled body returns a
// different result for the fourth copy of exgaussian's loop" (DESIGN.md): the same translation unit is compiled
// twice, with find_ppf out of line (the library's build) and force-inlined, each evaluated (a) four elements per
// thread in the shape of k_scalar's unrolled body and (b) one element per thread; all four results are compared
// bit for bit, per unroll slot.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPROBE_INLINE=0 -c ... ; same with 1; link both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef PROBE_ORDER
#define PROBE_ORDER 0
#endif
#if PROBE_INLINE
#define M355_FIND_PPF_ATTR static __device__ __forceinline__
#endif
#include "../distributions_amd/csrc/m355_dists4.h"
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(e)); exit(1);} } while(0)
using namespace m355;
using DD = Dist<M355_EXGAUSSIAN>;

#if PROBE_INLINE
#define NAME(x) x##_inl
#else
#define NAME(x) x##_out
#endif

__global__ __launch_bounds__(256) void NAME(k_four)(const double* q, double* out, long long n, const DD::K k) {
  const long long ntiles = n / 1024;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * 1024 + threadIdx.x * 2;
    double v[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { v[u][0] = q[base + u * 512]; v[u][1] = q[base + u * 512 + 1]; }
#if PROBE_ORDER == 1  // slot 3 first
    v[1][1] = DD::eval<M355_PPF>(v[1][1], k);
    v[1][0] = DD::eval<M355_PPF>(v[1][0], k);
    v[0][1] = DD::eval<M355_PPF>(v[0][1], k);
    v[0][0] = DD::eval<M355_PPF>(v[0][0], k);
#else
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      v[u][0] = DD::eval<M355_PPF>(v[u][0], k);
      v[u][1] = DD::eval<M355_PPF>(v[u][1], k);
    }
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      out[base + u * 512] = v[u][0];
      out[base + u * 512 + 1] = v[u][1];
    }
  }
}
__global__ __launch_bounds__(256) void NAME(k_one)(const double* q, double* out, long long n, const DD::K k) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = DD::eval<M355_PPF>(q[i], k);
}
extern "C" void NAME(run)(const double* q, double* o4, double* o1, long long n, const double* p) {
  const DD::K k = DD::derive(p);
  hipLaunchKernelGGL(NAME(k_four), dim3(64), dim3(256), 0, 0, q, o4, n, k);
  hipLaunchKernelGGL(NAME(k_one), dim3(64), dim3(256), 0, 0, q, o1, n, k);
}

#if !PROBE_INLINE
extern "C" void run_inl(const double* q, double* o4, double* o1, long long n, const double* p);
int main() {
  const long long n = 20 * 1024;
  std::vector<double> q(n);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : q) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s >> 11) * 0x1.0p-53; }
  const double params[4][4] = {{0.0, 1.0, 1.0, 0}, {2.0, 0.5, 100.0, 0}, {-1.0, 10.0, 0.5, 0}, {1.0, 3.0, 3.0, 0}};
  double *dq, *d[4];
  CHECK(hipMalloc(&dq, n * 8));
  for (auto& p : d) CHECK(hipMalloc(&p, n * 8));
  CHECK(hipMemcpy(dq, q.data(), n * 8, hipMemcpyHostToDevice));
  std::vector<double> h[4];
  for (auto& v : h) v.resize(n);
  for (const auto& p : params) {
    run_out(dq, d[0], d[1], n, p);
    run_inl(dq, d[2], d[3], n, p);
    CHECK(hipDeviceSynchronize());
    for (int j = 0; j < 4; ++j) CHECK(hipMemcpy(h[j].data(), d[j], n * 8, hipMemcpyDeviceToHost));
    const char* names[4] = {"out-of-line x4", "out-of-line x1", "inlined x4", "inlined x1"};
    for (int j = 1; j < 4; ++j) {
      long long bad = 0, slot[4] = {0, 0, 0, 0};
      for (long long i = 0; i < n; ++i) {
        if (__builtin_memcmp(&h[0][i], &h[j][i], 8) != 0 && !(h[0][i] != h[0][i] && h[j][i] != h[j][i])) {
          ++bad;
          ++slot[((i % 1024) / 512) * 2 + (i & 1)];
          if (bad <= 3) printf("   i=%lld q=%.17g  %s=%.17g  %s=%.17g\n", i, q[i], names[0], h[0][i], names[j], h[j][i]);
        }
      }
      printf("exgaussian(%g, %g, %g): %-15s vs %-15s: %lld of %lld differ (by unroll slot: %lld %lld %lld %lld)\n", p[0], p[1], p[2],
             names[0], names[j], bad, n, slot[0], slot[1], slot[2], slot[3]);
    }
  }
  return 0;
}
#endif
