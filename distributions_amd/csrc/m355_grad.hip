// Fused value-and-gradient of the summed log-likelihood: one pass over x gives
// sum(logpdf), sum(d logpdf/dx), sum(d logpdf/d param_j) and, optionally, the per-element partials.
//   k_grad_scalar   x full, contiguous and aligned, scalar parameters: the parameter-only part of the graph
//                   (Grad<D>::pre: lgamma / digamma / logs of parameters and their partials) is evaluated once
//                   per thread, then x is streamed as in k_scalar (16-byte non-temporal loads; dx written the
//                   same way when requested; without dx, x is data only and its partials are not even
//                   compiled in).  8 B read (+ 8 B written with dx) per element.
//   k_grad_generic  any broadcast layout: operands by (i / div) % mod, pre() per element, optional
//                   per-element outputs for every partial (the caller reduces those over broadcast axes).
// Both leave (2 + NP) partial sums per workgroup in `partials` ([slot][block]); k_grad_final adds them in a
// fixed order.
#include <hip/hip_runtime.h>

#include "m355_elemwise.h"
#include "m355_grad.h"

namespace m355 {

struct GradOut {
  void* dx;                 // per-element d/dx or NULL
  void* dparam[kMaxParams];  // per-element d/dparam_j or NULL
  int dtype;                // of the per-element outputs
};
struct ParamVals {
  double p[kMaxParams];
};

template <int S> M355_DEV void block_reduce_multi(double (&acc)[S], double* __restrict__ partials, int nblocks) {
  __shared__ double red[S][kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    double v = acc[s];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[s][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < S) {
    double v = red[threadIdx.x][0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) v += red[threadIdx.x][w];
    partials[(size_t)threadIdx.x * nblocks + blockIdx.x] = v;
  }
}

// Grad<D>::Stats: with x as data and scalar parameters the sum of the graph over a thread's elements is a
// function of a few sufficient statistics (m355_grad.h)
template <class T, class = void> struct has_grad_stats : std::false_type {};
template <class T> struct has_grad_stats<T, std::void_t<typename T::Stats>> : std::true_type {};

template <int D, bool WRITE_DX>
__global__ __launch_bounds__(kBlock) void k_grad_scalar(const void* __restrict__ x, int xdt, void* __restrict__ dx,
                                                        int odt, long long n, const ParamVals pv,
                                                        double* __restrict__ partials) {
  using G = Grad<D>;
  constexpr int S = 2 + G::NP;
  log_table_init();  // dlog / dlog1p / dxlogy0 use flog_tab
  const auto k = grad_pre<G>(pv.p);
  double acc[S];
#pragma unroll
  for (int s = 0; s < S; ++s) acc[s] = 0.0;
  const long long ntiles = n / kTile;
  if constexpr (!WRITE_DX && has_grad_stats<G>::value) {
    if (G::stats_ok(pv.p)) {  // uniform: parameters are kernel arguments
      typename G::Stats st;
      int since = 0;
      auto add = [&](double xv) {
        if (!G::stats_add(st, xv, k)) {  // zero, subnormal, out of support, inf, NaN: the graph itself
          double r[S];
          grad_eval<G, 0u>(xv, k, r);
#pragma unroll
          for (int s = 0; s < S; ++s) acc[s] += r[s];
        }
      };
      for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * kTile + threadIdx.x * kVec;
        double v[kUnroll][kVec];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) load2<false>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          add(v[u][0]);
          add(v[u][1]);
        }
        if (++since == kRenormEvery) {
          since = 0;
          G::stats_renorm(st);
        }
      }
      for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n;
           i += (long long)gridDim.x * kBlock)
        add(load1(x, xdt, i));
      double r[S];
      grad_unpack(G::stats_lp(st, k), r, std::make_integer_sequence<int, 1 + G::NP>{});
#pragma unroll
      for (int s = 0; s < S; ++s) acc[s] += r[s];
      acc[1] = kNaN;
      block_reduce_multi<S>(acc, partials, gridDim.x);
      return;
    }
  }
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kTile + threadIdx.x * kVec;
    double v[kUnroll][kVec];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) load2<false>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      double r0[S], r1[S];
      grad_eval<G, WRITE_DX ? 1u : 0u>(v[u][0], k, r0);
      grad_eval<G, WRITE_DX ? 1u : 0u>(v[u][1], k, r1);
#pragma unroll
      for (int s = 0; s < S; ++s) acc[s] += r0[s] + r1[s];
      if constexpr (WRITE_DX) store2<false>(dx, odt, base + u * (kBlock * kVec), r0[1], r1[1]);
    }
  }
  for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (long long)gridDim.x * kBlock) {
    double r[S];
    grad_eval<G, WRITE_DX ? 1u : 0u>(load1(x, xdt, i), k, r);
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] += r[s];
    if constexpr (WRITE_DX) store1(dx, odt, i, r[1]);
  }
  if constexpr (!WRITE_DX) acc[1] = kNaN;  // sum d/dx is not computed on this path (see include/m355_dist.h)
  block_reduce_multi<S>(acc, partials, gridDim.x);
}

template <int D>
__global__ __launch_bounds__(kBlock) void k_grad_generic(GenericOps ops, GradOut out, long long n,
                                                         double* __restrict__ partials) {
  using G = Grad<D>;
  constexpr int S = 2 + G::NP;
  log_table_init();  // dlog / dlog1p / dxlogy0 use flog_tab
  const bool small = n <= 0xffffffffll;
  double acc[S];
#pragma unroll
  for (int s = 0; s < S; ++s) acc[s] = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const double xv = fetch_operand(ops.x, (unsigned long long)i, small);
    double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < G::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)i, small);
    double r[S];
    grad_eval<G>(xv, grad_pre<G>(pp), r);
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] += r[s];
    if (out.dx) store1(out.dx, out.dtype, i, r[1]);
#pragma unroll
    for (int q = 0; q < G::NP; ++q)
      if (out.dparam[q]) store1(out.dparam[q], out.dtype, i, r[2 + q]);
  }
  block_reduce_multi<S>(acc, partials, gridDim.x);
}

// sums[s] = sum over blocks of partials[s][block], fixed order; one workgroup per slot
__global__ __launch_bounds__(kBlock) void k_grad_final(const double* __restrict__ partials, int nblocks,
                                                        double* __restrict__ sums) {
  const double* p = partials + (size_t)blockIdx.x * nblocks;
  constexpr int kPer = kMaxReduceBlocks / kBlock;
  double v[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int i = threadIdx.x + j * kBlock;
    v[j] = i < nblocks ? p[i] : 0.0;
  }
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < kPer; ++j) acc += v[j];
  acc = block_reduce_sum(acc);
  if (threadIdx.x == 0) sums[blockIdx.x] = acc;
}

struct GradArgs {
  long long n;
  DevOperand x;
  DevOperand p[kMaxParams];
  GradOut out;
  double* sums;
  double* partials;
  hipStream_t stream;
};

template <int D> int launch_grad(const GradArgs& a) {
  using G = Grad<D>;
  if constexpr (!G::kHas) {
    return -1000;
  } else {
    constexpr int S = 2 + G::NP;
    const long long n = a.n;
    const bool x_full = a.x.ptr != nullptr && a.x.div == 1 && a.x.mod >= (unsigned long long)n;
    const size_t xal = a.x.dtype == M355_F32 ? 8 : 16, oal = a.out.dtype == M355_F32 ? 8 : 16;
    bool all_scalar = true, any_dparam = false;
    for (int q = 0; q < G::NP; ++q) {
      all_scalar = all_scalar && a.p[q].ptr == nullptr;
      any_dparam = any_dparam || a.out.dparam[q] != nullptr;
    }
    const bool vec_ok = x_full && all_scalar && !any_dparam && aligned(a.x.ptr, xal) &&
                        (a.out.dx == nullptr || aligned(a.out.dx, oal));
    int blocks;
    if (vec_ok) {
      const long long tiles = n / kTile;
      blocks = (int)(tiles < 1 ? 1 : (tiles > kMaxReduceBlocks ? kMaxReduceBlocks : tiles));
      ParamVals pv;
      for (int q = 0; q < kMaxParams; ++q) pv.p[q] = a.p[q].scalar;
      if (a.out.dx)
        hipLaunchKernelGGL((k_grad_scalar<D, true>), dim3(blocks), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype,
                           a.out.dx, a.out.dtype, n, pv, a.partials);
      else
        hipLaunchKernelGGL((k_grad_scalar<D, false>), dim3(blocks), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype,
                           a.out.dx, a.out.dtype, n, pv, a.partials);
    } else {
      const long long want = (n + kBlock - 1) / kBlock;
      blocks = (int)(want < 1 ? 1 : (want > kMaxReduceBlocks ? kMaxReduceBlocks : want));
      GenericOps ops;
      ops.x = a.x;
      for (int q = 0; q < kMaxParams; ++q) ops.p[q] = a.p[q];
      hipLaunchKernelGGL((k_grad_generic<D>), dim3(blocks), dim3(kBlock), 0, a.stream, ops, a.out, n, a.partials);
    }
    int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    hipLaunchKernelGGL(k_grad_final, dim3(S), dim3(kBlock), 0, a.stream, a.partials, blocks, a.sums);
    return (int)hipGetLastError();
  }
}

}  // namespace m355

// Called by m355_logpdf_grad (m355_api.hip) after argument validation.
extern "C" int m355_grad_impl(int dist, long long n, const m355::GenericOps* ops, int nparams, void* dx,
                              void* const* dparams, int out_dtype, double* sums, double* partials,
                              hipStream_t stream) {
  m355::GradArgs a;
  a.n = n;
  a.x = ops->x;
  for (int q = 0; q < m355::kMaxParams; ++q) {
    a.p[q] = ops->p[q];
    // dparams is a host array of exactly nparams entries (include/m355_dist.h): never read past it
    a.out.dparam[q] = (dparams && q < nparams) ? dparams[q] : nullptr;
  }
  a.out.dx = dx;
  a.out.dtype = out_dtype;
  a.sums = sums;
  a.partials = partials;
  a.stream = stream;
  switch (dist) {
#define M355_GRAD_CASE(ID, name, np) \
  case ID: return m355::launch_grad<ID>(a);
    M355_DIST_LIST(M355_GRAD_CASE)
#undef M355_GRAD_CASE
  }
  return -1000;
}

extern "C" int m355_grad_supported(int dist) { return m355::has_grad(dist) ? 1 : 0; }
