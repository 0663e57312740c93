// Where does the overhead of k_binned (csrc/m355_elemwise.h) go?  Copies of the kernel with phases switched off,
// timed on a cheap body (normal.logpdf) next to k_scalar.  Not shipped.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/binned_probe.hip -o tools/binned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../distributions_amd/csrc/m355_elemwise.h"
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(e)); exit(1);} } while(0)
using namespace m355;

// V bit 0: no LDS atomics (identity permutation); bit 1: no mean / min-max passes; bit 2: no un-permute through LDS;
// bit 3: 32-bit keys
template <int D, int FN, int V>
__global__ __launch_bounds__(kBlock) void k_binned_v(const void* __restrict__ x, int xdt, void* __restrict__ out,
                                                     int odt, long long n, const typename Dist<D>::K k) {
  using DD = Dist<D>;
  __shared__ double xs[kBinTile];
  __shared__ double res[kBinTile];
  __shared__ unsigned short idx[kBinTile];
  __shared__ unsigned cnt[kBuckets];
  __shared__ unsigned long long red_min[kBlock / 64], red_max[kBlock / 64];
  __shared__ double red_s[kBlock / 64], red_c[kBlock / 64];
  __shared__ unsigned wave_tot[kBlock / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long long ntiles = n / kBinTile;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kBinTile;
    double v[kBinPer];
    unsigned long long key[kBinPer];
    unsigned long long kmin = ~0ull, kmax = 0ull;
#pragma unroll
    for (int j = 0; j < kBinPer; j += 2) load2<0>(x, xdt, base + (long long)(j / 2) * (kBlock * 2) + t * 2, v[j], v[j + 1]);
    double inv_scale = 1.0;
    if constexpr (!(V & 2)) {
      double sabs = 0.0, cfin = 0.0;
#pragma unroll
      for (int j = 0; j < kBinPer; ++j) {
        const double av = __builtin_fabs(v[j]);
        const bool fin = av < kInf;
        sabs += fin ? av : 0.0;
        cfin += fin ? 1.0 : 0.0;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        sabs += __shfl_xor(sabs, o, 64);
        cfin += __shfl_xor(cfin, o, 64);
      }
      if (lane == 0) { red_s[wave] = sabs; red_c[wave] = cfin; }
      __syncthreads();
      sabs = 0.0; cfin = 0.0;
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) { sabs += red_s[w]; cfin += red_c[w]; }
      inv_scale = (sabs > 0.0 && cfin > 0.0) ? cfin / sabs : 1.0;
    }
#pragma unroll
    for (int j = 0; j < kBinPer; ++j) {
      if constexpr (V & 8) {
        float tt = (float)(__builtin_fabs(v[j]) * inv_scale);
        tt = tt < 1048576.0f ? tt : ((tt == tt) ? 1048576.0f : 0.0f);
        const int mag = (int)(__builtin_bit_cast(unsigned, 1.0f + tt) - 0x3f800000u);
        key[j] = (unsigned)((v[j] < 0.0 ? -mag : mag) + (1 << 30));
      } else {
        double tt = __builtin_fabs(v[j]) * inv_scale;
        tt = tt < 1048576.0 ? tt : ((tt == tt) ? 1048576.0 : 0.0);
        const long long mag = (long long)(__builtin_bit_cast(unsigned long long, 1.0 + tt) - 0x3ff0000000000000ull);
        key[j] = (unsigned long long)((v[j] < 0.0 ? -mag : mag) + (1ll << 62));
      }
      kmin = key[j] < kmin ? key[j] : kmin;
      kmax = key[j] > kmax ? key[j] : kmax;
    }
    if constexpr (!(V & 2)) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(kmin, o, 64), b = __shfl_xor(kmax, o, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
      }
      if (lane == 0) { red_min[wave] = kmin; red_max[wave] = kmax; }
    }
    cnt[t] = 0u;
    __syncthreads();
    if constexpr (!(V & 2)) {
#pragma unroll
      for (int w = 0; w < kBlock / 64; ++w) {
        kmin = red_min[w] < kmin ? red_min[w] : kmin;
        kmax = red_max[w] > kmax ? red_max[w] : kmax;
      }
    } else {
      kmin = (V & 8) ? ((1ull << 30) - (3ull << 23)) : ((1ull << 62) - (3ull << 52));
      kmax = (V & 8) ? ((1ull << 30) + (3ull << 23)) : ((1ull << 62) + (3ull << 52));
    }
    const unsigned long long span = kmax - kmin;
    const int shift = span >> 8 ? 64 - 8 - __builtin_clzll(span) + 0 : 0;
    unsigned bucket[kBinPer], rank[kBinPer];
#pragma unroll
    for (int j = 0; j < kBinPer; ++j) {
      unsigned long long d = key[j] - kmin;
      if constexpr (V & 2) d = (key[j] < kmin) ? 0 : ((key[j] > kmax) ? span : d);
      bucket[j] = (unsigned)(d >> shift);
      if constexpr (V & 1) rank[j] = 0; else rank[j] = atomicAdd(&cnt[bucket[j]], 1u);
    }
    __syncthreads();
    unsigned c = cnt[t], incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned off = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) off += (w < wave) ? wave_tot[w] : 0u;
    cnt[t] = off + incl - c;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kBinPer; ++j) {
      const int local = (j / 2) * (kBlock * 2) + t * 2 + (j & 1);
      const unsigned slot = (V & 1) ? (unsigned)local : cnt[bucket[j]] + rank[j];
      xs[slot] = v[j];
      idx[slot] = (unsigned short)local;
    }
    __syncthreads();
    if constexpr (V & 4) {
#pragma unroll
      for (int j = 0; j < kBinPer; j += 2) {
        const int local = (j / 2) * (kBlock * 2) + t * 2;
        store2<0>(out, odt, base + local, DD::template eval<FN>(xs[local], k), DD::template eval<FN>(xs[local + 1], k));
      }
    } else {
#pragma unroll 1
      for (int j = 0; j < kBinPer; ++j) {
        const int s_ = j * kBlock + t;
        res[idx[s_]] = DD::template eval<FN>(xs[s_], k);
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kBinPer; j += 2) {
        const int local = (j / 2) * (kBlock * 2) + t * 2;
        store2<0>(out, odt, base + local, res[local], res[local + 1]);
      }
    }
    __syncthreads();
  }
}

template <class F> float timeit(F f) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) f();
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / 5;
}

int main() {
  const long long n = 100000000;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (long long i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 8.0; }
  double *x, *o;
  CHECK(hipMalloc(&x, n * 8)); CHECK(hipMalloc(&o, n * 8));
  CHECK(hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice));
  constexpr int D = M355_NORMAL, FN = M355_LOGPDF;
  const double p[4] = {0.3, 1.7, 0, 0};
  const auto k = Dist<D>::derive(p);
  const int bb = (int)((n / kBinTile) > kMaxBlocks ? kMaxBlocks : (n / kBinTile));
  printf("k_scalar            %.4f ms\n", timeit([&] { hipLaunchKernelGGL((k_scalar<D, FN, false, true>), dim3(kMaxBlocks), dim3(kBlock), 0, 0, x, M355_F64, o, M355_F64, n, k, nullptr); }));
  printf("k_binned (shipped)  %.4f ms\n", timeit([&] { hipLaunchKernelGGL((k_binned<D, FN>), dim3(bb), dim3(kBlock), 0, 0, x, M355_F64, o, M355_F64, n, k); }));
#define RUN(V, label) printf("variant %2d %-40s %.4f ms\n", V, label, timeit([&] { hipLaunchKernelGGL((k_binned_v<D, FN, V>), dim3(bb), dim3(kBlock), 0, 0, x, M355_F64, o, M355_F64, n, k); }));
  RUN(0, "copy of the shipped kernel");
  RUN(1, "no atomics");
  RUN(2, "no mean / min-max passes");
  RUN(4, "no un-permute");
  RUN(8, "32-bit keys");
  RUN(3, "no atomics, no passes");
  RUN(7, "no atomics, no passes, no un-permute");
  RUN(10, "32-bit keys, no passes");
  for (int g : {2048, 4096, 8192, 16384})
    printf("shipped, grid %5d  %.4f ms\n", g, timeit([&] { hipLaunchKernelGGL((k_binned<D, FN>), dim3(g), dim3(kBlock), 0, 0, x, M355_F64, o, M355_F64, n, k); }));
  return 0;
}
