// Tuning microbenchmark for the HBM-bound elementwise path (not shipped in the library):
// sweeps unroll depth, grid size, non-temporal loads/stores and waves-per-SIMD limits for a plain
// copy (the practical ceiling) and for the Normal/Gamma/Beta logpdf bodies at N = 1e8 fp64.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/stream_tune.hip -o tools/stream_tune
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../distributions_amd/csrc/m355_dists.h"

using namespace m355;
typedef double d2_t __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e = (x);                                                          \
    if (e != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));   \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

struct Copy {
  struct K {
    double dummy;
  };
};

// D: -1 copy, 0..2 library bodies; 100+10*NR+CORR: gamma with flog_fast_t<NR,CORR>;
// 200+10*NR+CORR: beta with flog_fast_t<NR,CORR>
template <int D, class K> __device__ __forceinline__ double body(double x, const K& k) {
  if constexpr (D < 0) return x;
  else if constexpr (D >= 200) {
    constexpr int NR = (D - 200) / 10;
    constexpr bool CORR = (D % 10) != 0;
    const double t = xlogy0_l(k.am1, flog_fast_t<NR, CORR>(x)) + xlogy0_l(k.bm1, flog_fast_t<NR, CORR>(1.0 - x));
    const double r = t - k.lbeta;
    return ((x < 0.0) | (x > 1.0)) ? -kInf : r;
  } else if constexpr (D >= 100) {
    constexpr int NR = (D - 100) / 10;
    constexpr bool CORR = (D % 10) != 0;
    const double t = xlogy0_l(k.am1, flog_fast_t<NR, CORR>(x));
    const double r = (t - x * k.beta) + k.c;
    return (x < 0.0) ? -kInf : r;
  } else return Dist<D>::template eval<M355_LOGPDF>(x, k);
}

template <int D, int UNROLL, bool NT, int WPE, class K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void kern(
    const double* __restrict__ x, double* __restrict__ out, long long n, const K k) {
  constexpr int TILE = 256 * 2 * UNROLL;
  const long long ntiles = n / TILE;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * TILE + threadIdx.x * 2;
    d2_t v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const d2_t* p = reinterpret_cast<const d2_t*>(x + base + u * 512);
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      d2_t r;
      r.x = body<D>(v[u].x, k);
      r.y = body<D>(v[u].y, k);
      d2_t* q = reinterpret_cast<d2_t*>(out + base + u * 512);
      if (NT) __builtin_nontemporal_store(r, q);
      else *q = r;
    }
  }
}

template <int D, int UNROLL, bool NT, int WPE, class K>
float run(const char* name, const double* x, double* out, long long n, const K& k, int blocks) {
  constexpr int TILE = 256 * 2 * UNROLL;
  if (blocks <= 0) blocks = (int)(n / TILE);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((kern<D, UNROLL, NT, WPE, K>), dim3(blocks), dim3(256), 0, 0, x, out, n, k);
  CHECK(hipEventRecord(a));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kern<D, UNROLL, NT, WPE, K>), dim3(blocks), dim3(256), 0, 0, x, out, n, k);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  printf("%-8s unroll=%d nt=%d wpe=%d blocks=%-7d  %.4f ms  %.1f GB/s  %.3e elem/s\n", name, UNROLL, (int)NT, WPE,
         blocks, ms, 16.0 * n / ms / 1e6, n / (ms * 1e-3));
  fflush(stdout);
  return ms;
}

// accuracy of the last launch's output against long double on the host (first 200k elements)
template <class F> void accuracy(const char* name, const std::vector<double>& h, const double* out, F truth) {
  const int m = 200000;
  std::vector<double> o(m);
  CHECK(hipMemcpy(o.data(), out, m * sizeof(double), hipMemcpyDeviceToHost));
  double worst = 0;
  for (int i = 0; i < m; ++i) {
    const long double t = truth((long double)h[i]);
    const double e = (double)(fabsl((long double)o[i] - t) / fmaxl(1.0L, fabsl(t)));
    if (e > worst) worst = e;
  }
  printf("%-8s max |err|/max(1,|truth|) = %.3e\n", name, worst);
}

template <int D, class K> void quick(const char* name, const double* x, double* out, long long n, const K& k) {
  run<D, 1, true, 1>(name, x, out, n, k, 0);
  run<D, 2, true, 1>(name, x, out, n, k, 0);
  run<D, 4, true, 1>(name, x, out, n, k, 0);
  run<D, 4, false, 1>(name, x, out, n, k, 0);
  run<D, 8, true, 1>(name, x, out, n, k, 0);
  run<D, 4, true, 1>(name, x, out, n, k, 8192);
  run<D, 4, true, 1>(name, x, out, n, k, 16384);
  run<D, 2, true, 1>(name, x, out, n, k, 32768);
}

template <int D, class K> void sweep(const char* name, const double* x, double* out, long long n, const K& k) {
  const int grids[] = {0, 1024, 2048, 4096, 8192};
  for (int g : grids) {
    run<D, 1, false, 1>(name, x, out, n, k, g);
    run<D, 2, false, 1>(name, x, out, n, k, g);
    run<D, 4, false, 1>(name, x, out, n, k, g);
    run<D, 4, true, 1>(name, x, out, n, k, g);
    run<D, 2, true, 1>(name, x, out, n, k, g);
  }
  run<D, 8, false, 1>(name, x, out, n, k, 2048);
  run<D, 8, true, 1>(name, x, out, n, k, 0);
  // force high occupancy (caps VGPRs)
  run<D, 2, false, 8>(name, x, out, n, k, 2048);
  run<D, 4, false, 8>(name, x, out, n, k, 2048);
  run<D, 2, true, 8>(name, x, out, n, k, 0);
  run<D, 4, false, 6>(name, x, out, n, k, 2048);
  run<D, 2, false, 6>(name, x, out, n, k, 4096);
  run<D, 1, false, 8>(name, x, out, n, k, 0);
  run<D, 1, true, 8>(name, x, out, n, k, 0);
}

int main() {
  const long long n = 100000000;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (long long i = 0; i < n; ++i) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    h[i] = ((s >> 11) + 0.5) * (1.0 / 9007199254740992.0);  // (0,1)
  }
  double *x, *out;
  CHECK(hipMalloc(&x, n * sizeof(double)));
  CHECK(hipMalloc(&out, n * sizeof(double)));
  CHECK(hipMemcpy(x, h.data(), n * sizeof(double), hipMemcpyHostToDevice));
  // hipMemcpyDtoD as another reference point
  {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipMemcpyAsync(out, x, n * 8, hipMemcpyDeviceToDevice, 0));
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) CHECK(hipMemcpyAsync(out, x, n * 8, hipMemcpyDeviceToDevice, 0));
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemcpyDtoD %.4f ms %.1f GB/s\n", ms / 10, 16.0 * n / (ms / 10) / 1e6);
  }
  Copy::K ck{0.0};
  const double pn[3] = {0.3, 1.7, 0}, pg[3] = {2.5, 1.3, 0}, pb[3] = {2.0, 5.0, 0};
  const auto kn = Dist<M355_NORMAL>::derive(pn);
  const auto kg = Dist<M355_GAMMA>::derive(pg);
  const auto kb = Dist<M355_BETA>::derive(pb);
  auto tg = [&](long double v) { return 1.5L * logl(v) - v * 1.3L + (long double)kg.c; };
  auto tb = [&](long double v) { return logl(v) + 4.0L * logl(1.0L - v) - (long double)kb.lbeta; };
  if (getenv("FULL_SWEEP")) {
    sweep<-1>("copy", x, out, n, ck);
    sweep<M355_NORMAL>("normal", x, out, n, kn);
    sweep<M355_GAMMA>("gamma", x, out, n, kg);
    sweep<M355_BETA>("beta", x, out, n, kb);
  }
  quick<-1>("copy", x, out, n, ck);
  quick<M355_NORMAL>("normal", x, out, n, kn);
  quick<M355_GAMMA>("gamma", x, out, n, kg);
  accuracy("gamma", h, out, tg);
  quick<110>("g_nr1", x, out, n, kg);
  accuracy("g_nr1", h, out, tg);
  quick<111>("g_nr1c", x, out, n, kg);
  accuracy("g_nr1c", h, out, tg);
  quick<120>("g_nr2", x, out, n, kg);
  accuracy("g_nr2", h, out, tg);
  quick<M355_BETA>("beta", x, out, n, kb);
  accuracy("beta", h, out, tb);
  quick<210>("b_nr1", x, out, n, kb);
  accuracy("b_nr1", h, out, tb);
  quick<211>("b_nr1c", x, out, n, kb);
  accuracy("b_nr1c", h, out, tb);
  quick<220>("b_nr2", x, out, n, kb);
  accuracy("b_nr2", h, out, tb);
  quick<221>("b_nr2c", x, out, n, kb);
  accuracy("b_nr2c", h, out, tb);
  CHECK(hipDeviceSynchronize());
  return 0;
}
