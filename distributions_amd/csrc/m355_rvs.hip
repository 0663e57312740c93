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
  log_table_init();  // the samplers use flog_tab
  const bool small = n <= 0xffffffffll;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
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
  log_table_init();
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

// Rejection samplers with wavefront-level compaction (PTRS Poisson, BTRS binomial; scalar
// parameters).  A wavefront owns a contiguous chunk of the output.  Every round each lane makes ONE
// attempt for the element it holds; lanes whose attempt was accepted store the draw and take the next
// unassigned element of the chunk (rank among the freed lanes from a ballot + mbcnt prefix count),
// so all 64 lanes keep proposing instead of idling until the slowest lane of a fixed group of 64
// elements is accepted.  A draw still depends only on (seed, stream, element index, attempt
// number), never on the lane that computed it: the output is identical to k_rvs.
template <int D>
__global__ __launch_bounds__(kBlock) void k_rvs_compact(void* __restrict__ out, int odt, long long n, double p0, double p1,
                                                        unsigned long long seed, unsigned stream,
                                                        unsigned long long offset) {
  log_table_init();
  const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  const long long chunk = ((n + nwaves - 1) / nwaves + 63) & ~63ll;
  const long long w1 = (wave + 1) * chunk < n ? (wave + 1) * chunk : n;
  long long next = wave * chunk;  // wave-uniform: first unassigned element
  if (next >= w1) return;
  PtrsK pk;
  BtrsK bk;
  if constexpr (D == M355_POISSON) pk = ptrs_setup(p0);
  else bk = btrs_setup(p0, p1);
  long long idx = -1;  // element held by this lane (-1: none)
  unsigned attempt = 0;
  for (;;) {
    // refill the free lanes from the chunk
    const bool freed = idx < 0;
    const unsigned long long fm = __ballot(freed);
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
    if (freed) {
      const long long cand = next + below;
      if (cand < w1) {
        idx = cand;
        attempt = 0;
      }
    }
    next += __popcll(fm);
    if (next > w1) next = w1;
    const bool active = idx >= 0;
    if (__ballot(active) == 0ull) break;  // chunk exhausted and every lane has stored its draw
    if (active) {
      const unsigned long long c = offset + (unsigned long long)idx;
      const Philox ph = philox4x32_10((unsigned)c, (unsigned)(c >> 32), attempt, stream, (unsigned)seed, (unsigned)(seed >> 32));
      attempt += 1;
      const double u1 = u53(ph.c[0], ph.c[1]), v = u53(ph.c[2], ph.c[3]);
      double k;
      bool ok;
      if constexpr (D == M355_POISSON) {
        ok = ptrs_attempt(pk, u1, v, k);
      } else {
        ok = btrs_attempt(bk, u1, v, k);
        if (bk.flip) k = bk.n - k;
      }
      if (ok || attempt >= (unsigned)kMaxAttempts) {
        store_draw(out, odt, idx, k);
        idx = -1;
      }
    }
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
  const bool scalar2 = ops->p[0].ptr == nullptr && ops->p[1].ptr == nullptr;
  const double q0 = ops->p[0].scalar, q1 = ops->p[1].scalar;
  const bool ptrs = dist == M355_POISSON && ops->p[0].ptr == nullptr && q0 >= 10.0 && q0 < kInf;
  const bool btrs = dist == M355_BINOMIAL && scalar2 && q0 >= 0.0 && q0 < kInf && q1 > 0.0 && q1 < 1.0 &&
                    q0 * (q1 > 0.5 ? 1.0 - q1 : q1) >= 10.0;
  if ((ptrs || btrs) && n >= 4096) {
    const long long want = (n + 4 * kBlock - 1) / (4 * kBlock);  // >= 256 elements per wavefront
    const int blocks = (int)(want > 8192 ? 8192 : want);
    if (ptrs)
      hipLaunchKernelGGL((k_rvs_compact<M355_POISSON>), dim3(blocks), dim3(kBlock), 0, s, out, out_dtype, n, q0, 0.0, seed,
                         stream_id, offset);
    else
      hipLaunchKernelGGL((k_rvs_compact<M355_BINOMIAL>), dim3(blocks), dim3(kBlock), 0, s, out, out_dtype, n, q0, q1, seed,
                         stream_id, offset);
    return (int)hipGetLastError();
  }
  switch (dist) {
#define M355_RVS_CASE(ID, name, np) \
  case ID: return launch_rvs<ID>(*ops, out, out_dtype, n, seed, stream_id, offset, s);
    M355_DIST_LIST(M355_RVS_CASE)
#undef M355_RVS_CASE
  }
  return -1000;
}
