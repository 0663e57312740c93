// rvs kernels and their extern "C" entry point (declaration: include/m355_dist.h).
#include <hip/hip_runtime.h>

#include "m355_elemwise.h"
#include "m355_rng.h"

namespace m355 {

M355_DEV void store_draw(void* out, int odt, long long i, double v) {
  if (odt == M355_I64) static_cast<long long*>(out)[i] = (long long)v;
  else if (odt == M355_F32) static_cast<float*>(out)[i] = (float)v;
  else static_cast<double*>(out)[i] = v;
}

// One draw per thread per step; parameters indexed like elementwise operands.  Output-only
// stream: 8 B per draw, coalesced 512 B per wavefront store.
template <int D>
__global__ __launch_bounds__(kBlock) void k_rvs(GenericOps ops, void* __restrict__ out, int odt, long long n,
                                                unsigned long long seed, unsigned stream,
                                                unsigned long long offset) {
  const bool small = n <= 0xffffffffll;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    double pp[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < Dist<D>::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)i, small);
    store_draw(out, odt, i, sample_one<D>(offset + (unsigned long long)i, seed, stream, pp));
  }
}

// Normal with scalar parameters: each thread turns one Philox block into both Box-Muller branches
// and stores 16 bytes (elements 2k, 2k+1); same values as k_rvs<M355_NORMAL>.
__global__ __launch_bounds__(kBlock) void k_rvs_normal_pairs(double* __restrict__ out, long long npairs, double mu,
                                                             double sigma, unsigned long long seed, unsigned stream,
                                                             unsigned long long pair_offset) {
  for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < npairs; k += (long long)gridDim.x * kBlock) {
    const unsigned long long c = pair_offset + (unsigned long long)k;
    const Philox p = philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, stream, (unsigned)seed, (unsigned)(seed >> 32));
    double z0, z1;
    normal_pair(u53(p.c[0], p.c[1]), u53(p.c[2], p.c[3]), z0, z1);
    d2_t v;
    v.x = mu + sigma * z0;
    v.y = mu + sigma * z1;
    __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(out + 2 * k));
  }
}

template <int D> int launch_rvs(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed,
                                unsigned stream, unsigned long long offset, hipStream_t s) {
  const long long want = (n + kBlock - 1) / kBlock;
  const int blocks = (int)(want < 1 ? 1 : (want > kMaxBlocks ? kMaxBlocks : want));
  hipLaunchKernelGGL((k_rvs<D>), dim3(blocks), dim3(kBlock), 0, s, ops, out, odt, n, seed, stream, offset);
  return (int)hipGetLastError();
}

}  // namespace m355

extern "C" int m355_rvs_impl(int dist, int out_dtype, long long n, const m355::GenericOps* ops, unsigned long long seed,
                             unsigned stream_id, unsigned long long offset, void* out, hipStream_t s) {
  using namespace m355;
  if (dist == M355_NORMAL && out_dtype == M355_F64 && ops->p[0].ptr == nullptr && ops->p[1].ptr == nullptr &&
      (offset & 1ull) == 0 && aligned(out, 16) && n >= 2) {
    const long long npairs = n / 2;
    const long long want = (npairs + kBlock - 1) / kBlock;
    const int blocks = (int)(want > kMaxBlocks ? kMaxBlocks : want);
    hipLaunchKernelGGL(k_rvs_normal_pairs, dim3(blocks), dim3(kBlock), 0, s, static_cast<double*>(out), npairs,
                       ops->p[0].scalar, ops->p[1].scalar, seed, stream_id, offset >> 1);
    int rc = (int)hipGetLastError();
    if (rc != 0 || (n & 1) == 0) return rc;
    // odd tail element
    return launch_rvs<M355_NORMAL>(*ops, static_cast<double*>(out) + (n - 1), out_dtype, 1, seed, stream_id,
                                   offset + (unsigned long long)(n - 1), s);
  }
  switch (dist) {
    case M355_NORMAL: return launch_rvs<M355_NORMAL>(*ops, out, out_dtype, n, seed, stream_id, offset, s);
    case M355_GAMMA: return launch_rvs<M355_GAMMA>(*ops, out, out_dtype, n, seed, stream_id, offset, s);
    case M355_BETA: return launch_rvs<M355_BETA>(*ops, out, out_dtype, n, seed, stream_id, offset, s);
    case M355_STUDENTT: return launch_rvs<M355_STUDENTT>(*ops, out, out_dtype, n, seed, stream_id, offset, s);
    case M355_POISSON: return launch_rvs<M355_POISSON>(*ops, out, out_dtype, n, seed, stream_id, offset, s);
    case M355_BINOMIAL: return launch_rvs<M355_BINOMIAL>(*ops, out, out_dtype, n, seed, stream_id, offset, s);
  }
  return -1000;
}
