// Elementwise evaluation kernels (gfx950): the replacement for the fused Elemwise loop PyTensor
// compiles for <dist>.<fn>(x, *params), plus the fused sum-logp epilogue.
//
// Three kernels per (dist, fn):
//   k_scalar    x is a full contiguous array, every parameter a scalar (their derived constants
//               are computed once on the host and arrive as kernel arguments).  Grid-stride over tiles of
//               1024 elements; each thread issues kUnroll independent non-temporal 16-byte loads (lane i at
//               base + 16*i: one 1-KiB request per wavefront instruction) before it computes,
//               and stores 16 bytes per lane the same way.  This is the HBM-bound metric path:
//               16 B of traffic per element, nothing else touches memory.
//   k_periodic  parameters of period P along the flat index (x of shape (R, P), params (P,)):
//               each workgroup derives the P parameter-only constants once (lgamma, log, ...)
//               into LDS as struct-of-arrays (conflict-free ds_read_b64) and then streams x as
//               k_scalar does, keeping a running (index mod P) in 32-bit integer adds.
//   k_generic   any other broadcast pattern (full-size parameter arrays, outer broadcast, scalar
//               x, unaligned pointers): one element per thread per step, operands indexed by
//               (i / div) % mod, constants derived per element.
// With REDUCE the result is not stored: per-thread sums -> 64-lane shuffle reduction -> LDS ->
// one double per workgroup in `partials`; k_final adds those in a fixed order (deterministic).
#pragma once
#include <hip/hip_runtime.h>

#include "m355_dist_list.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "m355_dists4.h"
#include "m355_moments.h"

namespace m355 {

struct DevOperand {
  const void* ptr;
  double scalar;
  unsigned long long div, mod;
  int dtype;
};

constexpr int kMaxParams = 4;  // betascaled has four

struct LaunchArgs {
  int fn;
  int out_dtype;
  long long n;
  DevOperand x;
  int nparams;
  DevOperand p[kMaxParams];
  void* out;         // elementwise result (unused with partials)
  double* partials;  // non-null: reduce, one double per workgroup
  int* nblocks_out;  // receives the number of partials written
  double* table = nullptr;      // reduce only: device scratch for a table of derived constants (k_periodic_prod), or null
  long long table_doubles = 0;  // its capacity
  double* ews = nullptr;        // elementwise only: the caller's workspace for the launch-time tables (k_tab, k_coltab), or null
  size_t ews_doubles = 0;       // its capacity
  hipStream_t stream;
};

constexpr int kBlock = 256;
constexpr int kUnroll = 2;
constexpr int kVec = 2;
constexpr int kTile = kBlock * kVec * kUnroll;  // 1024 elements = 8 KiB of f64 in, 8 KiB out
constexpr int kMaxBlocks = 32768;               // grid cap (tools/stream_tune.hip: best of the sweep)
constexpr int kMaxReduceBlocks = 4096;         // sum-logp grid cap: keeps k_final's pass short
constexpr int kMaxPeriodBytes = 64 * 1024;      // LDS table cap for k_periodic

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef long long l2_t __attribute__((ext_vector_type(2)));

// IO: 1 = float64 in and out, 2 = float32 in and out (compile-time), 0 = dtype switch per access (wave-uniform)
template <int IO> M355_DEV void load2(const void* p, int dt, long long i, double& a, double& b) {
  if (IO == 1 || (IO == 0 && dt == M355_F64)) {
    // streamed once: non-temporal (+8-10 % over default-policy loads/stores, tools/stream_tune.hip)
    const d2_t v = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(static_cast<const double*>(p) + i));
    a = v.x;
    b = v.y;
  } else if (IO == 2 || dt == M355_F32) {
    const f2_t v = __builtin_nontemporal_load(reinterpret_cast<const f2_t*>(static_cast<const float*>(p) + i));
    a = (double)v.x;
    b = (double)v.y;
  } else {
    const l2_t v = __builtin_nontemporal_load(reinterpret_cast<const l2_t*>(static_cast<const long long*>(p) + i));
    a = (double)v.x;
    b = (double)v.y;
  }
}
template <int IO> M355_DEV void store2(void* p, int dt, long long i, double a, double b) {
  if (IO == 1 || (IO == 0 && dt == M355_F64)) {
    d2_t v;
    v.x = a;
    v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(static_cast<double*>(p) + i));
  } else {
    f2_t v;
    v.x = (float)a;
    v.y = (float)b;
    __builtin_nontemporal_store(v, reinterpret_cast<f2_t*>(static_cast<float*>(p) + i));
  }
}
M355_DEV double load1(const void* p, int dt, unsigned long long i) {
  if (dt == M355_F64) return static_cast<const double*>(p)[i];
  if (dt == M355_F32) return (double)static_cast<const float*>(p)[i];
  return (double)static_cast<const long long*>(p)[i];
}
M355_DEV void store1(void* p, int dt, long long i, double v) {
  if (dt == M355_F64) static_cast<double*>(p)[i] = v;
  else static_cast<float*>(p)[i] = (float)v;
}

// Sum over the workgroup; the result is valid in thread 0.
M355_DEV double block_reduce_sum(double v) {
  __shared__ double red[kBlock / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = red[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) v += red[w];
  }
  return v;
}

// Dist<D>::kLogTable: the bodies use flog_tab (m355_math.h), whose LDS table every kernel fills first.
template <class T, class = void> struct uses_log_table : std::false_type {};
template <class T> struct uses_log_table<T, std::void_t<decltype(T::kLogTable)>> : std::true_type {};

// Dist<D>::SumAcc: the sum of logpdf over a thread's elements has a cheaper form than summing logpdf()
// (sums of logarithms as products, m355_math.h LogProd); sum_ok(K) says whether it applies to these parameters.
template <class T, class = void> struct has_sum_acc : std::false_type {};
template <class T> struct has_sum_acc<T, std::void_t<typename T::SumAcc>> : std::true_type {};
constexpr int kRenormEvery = 64;  // tiles (4 elements per thread each) between LogProd renormalisations

// Dist<D>::logpdf_f32 / f32_ok: a float32-arithmetic logpdf body for float32 data (k_scalar_f32), valid for the
// parameters f32_ok accepts.
template <class T, class = void> struct has_f32_body : std::false_type {};
template <class T> struct has_f32_body<T, std::void_t<decltype(&T::logpdf_f32)>> : std::true_type {};

// Dist<D>::tab_fn(fn) / eval_tab<FN>: a body for the scalar-parameter kernels that reads an LDS table which
// phi_table_init() fills (normal.logcdf / logsf / sf, m355_dists.h).
template <class T, class = void> struct has_tab_body : std::false_type {};
template <class T> struct has_tab_body<T, std::void_t<decltype(&T::tab_fn)>> : std::true_type {};
template <class DD, int FN> constexpr bool use_tab_body() {
  if constexpr (has_tab_body<DD>::value) return DD::tab_fn(FN);
  else return false;
}
template <class DD, int FN> M355_DEV double eval_scalar(double x, const typename DD::K& k) {
  if constexpr (use_tab_body<DD, FN>()) return DD::template eval_tab<FN>(x, k);
  else return DD::template eval<FN>(x, k);
}

template <int D, int FN, bool REDUCE, int IO>
M355_DEV void scalar_body(const void* __restrict__ x, int xdt, void* __restrict__ out, int odt, long long n,
                          const typename Dist<D>::K& k, double* __restrict__ partials) {
  using DD = Dist<D>;
  if constexpr (uses_log_table<DD>::value) log_table_init();
  if constexpr (use_tab_body<DD, FN>()) phi_table_init();
  const long long ntiles = n / kTile;
  double acc = 0.0;
  if constexpr (!REDUCE && IO == 2 && FN == M355_LOGPDF && has_f32_body<DD>::value) {
    if (DD::f32_ok(k)) {  // uniform: parameters are kernel arguments
      const float* xf = static_cast<const float*>(x);
      float* of = static_cast<float*>(out);
      for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * kTile + threadIdx.x * kVec;
        f2_t v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f2_t*>(xf + base + u * (kBlock * kVec)));
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          f2_t r;
          r.x = DD::logpdf_f32(v[u].x, k);
          r.y = DD::logpdf_f32(v[u].y, k);
          __builtin_nontemporal_store(r, reinterpret_cast<f2_t*>(of + base + u * (kBlock * kVec)));
        }
      }
      for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
        of[i] = DD::logpdf_f32(xf[i], k);
      return;
    }
  }
  if constexpr (REDUCE && FN == M355_LOGPDF && has_sum_acc<DD>::value) {
    if (DD::sum_ok(k)) {  // uniform: parameters are kernel arguments
      typename DD::SumAcc s;
      int since = 0;
      for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * kTile + threadIdx.x * kVec;
        double v[kUnroll][kVec];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) load2<IO>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          DD::sum_add(s, v[u][0], k);
          DD::sum_add(s, v[u][1], k);
        }
        if (++since == kRenormEvery) {
          since = 0;
          DD::sum_renorm(s);
        }
      }
      for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n;
           i += (long long)gridDim.x * kBlock)
        DD::sum_add(s, load1(x, xdt, i), k);
      acc = block_reduce_sum(DD::sum_finish(s, k));
      if (threadIdx.x == 0) partials[blockIdx.x] = acc;
      return;
    }
  }
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kTile + threadIdx.x * kVec;
    double v[kUnroll][kVec];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) load2<IO>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      v[u][0] = eval_scalar<DD, FN>(v[u][0], k);
      v[u][1] = eval_scalar<DD, FN>(v[u][1], k);
      if constexpr (REDUCE) {
        acc += v[u][0] + v[u][1];
      } else {
        store2<IO>(out, odt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
      }
    }
  }
  for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (long long)gridDim.x * kBlock) {
    const double r = eval_scalar<DD, FN>(load1(x, xdt, i), k);
    if constexpr (REDUCE) acc += r;
    else store1(out, odt, i, r);
  }
  if constexpr (REDUCE) {
    acc = block_reduce_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
  }
}

template <int D, int FN, bool REDUCE, bool F64IO>
__global__ __launch_bounds__(kBlock) void k_scalar(const void* __restrict__ x, int xdt,
                                                   void* __restrict__ out, int odt, long long n,
                                                   const typename Dist<D>::K k,
                                                   double* __restrict__ partials) {
  scalar_body<D, FN, REDUCE, F64IO ? 1 : 0>(x, xdt, out, odt, n, k, partials);
}
// logpdf with float32 in and out (8 B per element: the HBM-bound case at half the bytes), dtype fixed at compile time
template <int D, bool REDUCE>
__global__ __launch_bounds__(kBlock) void k_scalar_f32(const void* __restrict__ x, void* __restrict__ out, long long n,
                                                       const typename Dist<D>::K k, double* __restrict__ partials) {
  scalar_body<D, M355_LOGPDF, REDUCE, 2>(x, M355_F32, out, M355_F32, n, k, partials);
}

struct PeriodicParams {
  const double* ptr[kMaxParams];  // NULL: scalar
  double scalar[kMaxParams];
};

template <int D, int FN, bool REDUCE, bool F64IO>
__global__ __launch_bounds__(kBlock) void k_periodic(const void* __restrict__ x, int xdt,
                                                     void* __restrict__ out, int odt, long long n,
                                                     PeriodicParams pa, unsigned P,
                                                     double* __restrict__ partials) {
  constexpr int IO = F64IO ? 1 : 0;
  using DD = Dist<D>;
  using K = typename DD::K;
  constexpr int NK = sizeof(K) / sizeof(double);
  extern __shared__ double lds[];  // [NK][P]
  constexpr bool kProduct = REDUCE && FN == M355_LOGPDF && has_sum_acc<DD>::value;
  __shared__ int all_sum_ok;
  if constexpr (kProduct) {
    if (threadIdx.x == 0) all_sum_ok = 1;
    __syncthreads();
  }
  if constexpr (uses_log_table<DD>::value) log_table_init();
  for (unsigned j = threadIdx.x; j < P; j += kBlock) {
    double pp[kMaxParams];
#pragma unroll
    for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][j] : pa.scalar[q];
    const K k = DD::derive(pp);
    const double* kd = reinterpret_cast<const double*>(&k);
#pragma unroll
    for (int f = 0; f < NK; ++f) lds[f * P + j] = kd[f];
    if constexpr (kProduct) {
      if (!DD::sum_ok(k)) all_sum_ok = 0;  // (every writer stores the same value)
    }
  }
  __syncthreads();
  auto fetch = [&](unsigned idx) {
    K k;
    double* kd = reinterpret_cast<double*>(&k);
#pragma unroll
    for (int f = 0; f < NK; ++f) kd[f] = lds[f * P + idx];
    return k;
  };
  const long long ntiles = n / kTile;
  // running (element index mod P) of this thread's first element in the current tile
  unsigned idx0 = (unsigned)(((unsigned long long)blockIdx.x * kTile + threadIdx.x * kVec) % P);
  const unsigned tile_step = (unsigned)(((unsigned long long)gridDim.x * kTile) % P);
  const unsigned u_step = (unsigned)((kBlock * kVec) % P);
  double acc = 0.0;
  if constexpr (kProduct) {
    // Sum-logp as a product with row-periodic parameters: the launcher makes gridDim.x * kTile a multiple of P where it
    // can, so a thread meets the same kUnroll * kVec columns in every tile it takes and keeps one product-form
    // accumulator (Dist<D>::SumAcc) per column; the logarithms are taken once per thread and column at the end.
    if (tile_step == 0 && all_sum_ok) {
      typename DD::SumAcc sa[kUnroll][kVec];
      K kc[kUnroll][kVec];
      {
        unsigned idx = idx0;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          unsigned idx1 = idx + 1;
          idx1 = idx1 >= P ? idx1 - P : idx1;
          kc[u][0] = fetch(idx);
          kc[u][1] = fetch(idx1);
          idx += u_step;
          idx = idx >= P ? idx - P : idx;
        }
      }
      int since = 0;
      for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long base = tile * kTile + threadIdx.x * kVec;
        double v[kUnroll][kVec];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) load2<IO>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          DD::sum_add(sa[u][0], v[u][0], kc[u][0]);
          DD::sum_add(sa[u][1], v[u][1], kc[u][1]);
        }
        if (++since == 4 * kRenormEvery) {  // one element per accumulator and tile: renormalise every 256 elements
          since = 0;
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            DD::sum_renorm(sa[u][0]);
            DD::sum_renorm(sa[u][1]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) acc += DD::sum_finish(sa[u][0], kc[u][0]) + DD::sum_finish(sa[u][1], kc[u][1]);
      for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
        acc += DD::template eval<FN>(load1(x, xdt, i), fetch((unsigned)((unsigned long long)i % P)));
      acc = block_reduce_sum(acc);
      if (threadIdx.x == 0) partials[blockIdx.x] = acc;
      return;
    }
  }
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kTile + threadIdx.x * kVec;
    double v[kUnroll][kVec];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) load2<IO>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
    unsigned idx = idx0;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      unsigned idx1 = idx + 1;
      idx1 = idx1 >= P ? idx1 - P : idx1;
      v[u][0] = DD::template eval<FN>(v[u][0], fetch(idx));
      v[u][1] = DD::template eval<FN>(v[u][1], fetch(idx1));
      if constexpr (REDUCE) {
        acc += v[u][0] + v[u][1];
      } else {
        store2<IO>(out, odt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
      }
      idx += u_step;
      idx = idx >= P ? idx - P : idx;
    }
    idx0 += tile_step;
    idx0 = idx0 >= P ? idx0 - P : idx0;
  }
  for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (long long)gridDim.x * kBlock) {
    const double r = DD::template eval<FN>(load1(x, xdt, i), fetch((unsigned)((unsigned long long)i % P)));
    if constexpr (REDUCE) acc += r;
    else store1(out, odt, i, r);
  }
  if constexpr (REDUCE) {
    acc = block_reduce_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// Sum-logp as a product with row-periodic parameters, table-free variant.  k_periodic keeps every workgroup's P derived
// constant sets in LDS (48 KiB for beta at P = 1024: two or three workgroups per CU, too few wavefronts to keep the HBM
// stream busy, and every workgroup repeats the P lgamma / log derivations).  When the caller's workspace has room,
// k_derive_table writes the table once to device memory ([field][P]) and k_periodic_prod streams x with no LDS at all:
// the launcher makes gridDim.x * kTile a multiple of P, so a thread meets the same kUnroll * kVec columns in every tile,
// reads their constants once (L2) and keeps one Dist<D>::SumAcc per column.  A column whose coefficients exceed the
// product form's limit (sum_ok) takes logpdf() per element, for that thread's accumulator only.
template <int D>
__global__ __launch_bounds__(kBlock) void k_derive_table(PeriodicParams pa, unsigned P, double* __restrict__ table) {
  using DD = Dist<D>;
  using K = typename DD::K;
  constexpr int NK = sizeof(K) / sizeof(double);
  const unsigned j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= P) return;
  double pp[kMaxParams];
#pragma unroll
  for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][j] : pa.scalar[q];
  const K k = DD::derive(pp);
  const double* kd = reinterpret_cast<const double*>(&k);
#pragma unroll
  for (int f = 0; f < NK; ++f) table[(size_t)f * P + j] = kd[f];
}

template <int D, bool F64IO>
__global__ __launch_bounds__(kBlock) void k_periodic_prod(const void* __restrict__ x, int xdt, long long n, unsigned P,
                                                          const double* __restrict__ table, double* __restrict__ partials) {
  constexpr int IO = F64IO ? 1 : 0;
  using DD = Dist<D>;
  using K = typename DD::K;
  constexpr int NK = sizeof(K) / sizeof(double);
  if constexpr (uses_log_table<DD>::value) log_table_init();
  auto fetch = [&](unsigned idx) {
    K k;
    double* kd = reinterpret_cast<double*>(&k);
#pragma unroll
    for (int f = 0; f < NK; ++f) kd[f] = table[(size_t)f * P + idx];
    return k;
  };
  const long long ntiles = n / kTile;
  typename DD::SumAcc sa[kUnroll][kVec];
  K kc[kUnroll][kVec];
  bool ok[kUnroll][kVec];
  {
    unsigned idx = (unsigned)(((unsigned long long)blockIdx.x * kTile + threadIdx.x * kVec) % P);
    const unsigned u_step = (unsigned)((kBlock * kVec) % P);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      unsigned idx1 = idx + 1;
      idx1 = idx1 >= P ? idx1 - P : idx1;
      kc[u][0] = fetch(idx);
      kc[u][1] = fetch(idx1);
      ok[u][0] = DD::sum_ok(kc[u][0]);
      ok[u][1] = DD::sum_ok(kc[u][1]);
      idx += u_step;
      idx = idx >= P ? idx - P : idx;
    }
  }
  double acc = 0.0;
  int since = 0;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kTile + threadIdx.x * kVec;
    double v[kUnroll][kVec];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) load2<IO>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
      for (int w = 0; w < kVec; ++w) {
        if (ok[u][w]) DD::sum_add(sa[u][w], v[u][w], kc[u][w]);
        else acc += DD::template eval<M355_LOGPDF>(v[u][w], kc[u][w]);
      }
    }
    if (++since == 4 * kRenormEvery) {  // one element per accumulator and tile: renormalise every 256 elements
      since = 0;
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        DD::sum_renorm(sa[u][0]);
        DD::sum_renorm(sa[u][1]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
    for (int w = 0; w < kVec; ++w) acc += ok[u][w] ? DD::sum_finish(sa[u][w], kc[u][w]) : 0.0;
  }
  for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    acc += DD::template eval<M355_LOGPDF>(load1(x, xdt, i), fetch((unsigned)((unsigned long long)i % P)));
  acc = block_reduce_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// Stored results with row-periodic parameters, the same way: when the caller brings a workspace (m355_elemwise_ws),
// k_derive_table writes the P derived constant sets to it once and k_periodic_fixed streams x with no LDS table -- the
// launcher makes gridDim.x * kTile a multiple of P, so a thread meets the same kUnroll * kVec columns in every tile and
// reads their constants once (L2).  k_periodic's table (48 KiB for beta at P = 1024) allowed two or three workgroups per
// CU and every workgroup repeated the P lgamma / log derivations: stored logpdf ran at 48-57 % of HBM where scalar
// parameters reach 77 %.
template <int D, int FN, bool F64IO>
__global__ __launch_bounds__(kBlock) void k_periodic_fixed(const void* __restrict__ x, int xdt, void* __restrict__ out, int odt,
                                                           long long n, unsigned P, const double* __restrict__ table) {
  constexpr int IO = F64IO ? 1 : 0;
  using DD = Dist<D>;
  using K = typename DD::K;
  constexpr int NK = sizeof(K) / sizeof(double);
  if constexpr (uses_log_table<DD>::value) log_table_init();
  auto fetch = [&](unsigned idx) {
    K k;
    double* kd = reinterpret_cast<double*>(&k);
#pragma unroll
    for (int f = 0; f < NK; ++f) kd[f] = table[(size_t)f * P + idx];
    return k;
  };
  const long long ntiles = n / kTile;
  K kc[kUnroll][kVec];
  {
    unsigned idx = (unsigned)(((unsigned long long)blockIdx.x * kTile + threadIdx.x * kVec) % P);
    const unsigned u_step = (unsigned)((kBlock * kVec) % P);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      unsigned idx1 = idx + 1;
      idx1 = idx1 >= P ? idx1 - P : idx1;
      kc[u][0] = fetch(idx);
      kc[u][1] = fetch(idx1);
      idx += u_step;
      idx = idx >= P ? idx - P : idx;
    }
  }
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kTile + threadIdx.x * kVec;
    double v[kUnroll][kVec];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) load2<IO>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      v[u][0] = DD::template eval<FN>(v[u][0], kc[u][0]);
      v[u][1] = DD::template eval<FN>(v[u][1], kc[u][1]);
      store2<IO>(out, odt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
    }
  }
  for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    store1(out, odt, i, DD::template eval<FN>(load1(x, xdt, i), fetch((unsigned)((unsigned long long)i % P))));
}

// ------------------------------------------------------------------------------------------------
// k_binned: the special-function kernels (logcdf / logsf / ppf of gamma, beta, studentt, ...) are pinned at the
// VALU issue limit with lane_util 0.25-0.5: lanes of a wavefront run series / continued fractions / Newton loops
// of different lengths.  For fixed parameters the trip count is a smooth function of x, so each workgroup
// sorts its tile by x in LDS (256 buckets over the tile's range of the key sign(x) (1 + |x| / mean|x|) read as an
// ordered integer: linear in x in the bulk on both sides of zero, logarithmic in the tails; counting sort with LDS
// atomics), evaluates in sorted order -- a wavefront then holds 64 neighbours
// in x, i.e. of similar cost -- and un-permutes through LDS, so global loads and stores stay coalesced and the
// result is the same as k_scalar's, element for element.
constexpr int kBinTile = 2048;  // elements per workgroup tile: 16 KiB x + 4 KiB index + 16 KiB results in LDS
constexpr int kBinPer = kBinTile / kBlock;
constexpr int kBuckets = 256;
constexpr int kBinMinTiles = 256;  // below ~5e5 elements the 1024-element tiles of k_scalar fill the 256 CUs better

// Dist<D>::kMinWaves: register budget of the binned kernel in wavefronts per SIMD.  The betaincinv families inline the
// root finder into the kernel (M355_DEV) so that this budget covers it: 168 VGPRs and a few spills instead of 194
// VGPRs, i.e. three resident wavefronts instead of two -- beta.ppf 7.9 -> 6.1 ms at N = 1e8 (A/B on one box,
// scripts/ab_tile.sh).  A budget cannot be given to a __noinline__ device function.
template <class T, class = void> struct min_waves { static constexpr int value = 1; };
template <class T> struct min_waves<T, std::void_t<decltype(T::kMinWaves)>> { static constexpr int value = T::kMinWaves; };

template <int D, int FN>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(min_waves<Dist<D>>::value))) void k_binned(const void* __restrict__ x, int xdt, void* __restrict__ out,
                                                   int odt, long long n, const typename Dist<D>::K k) {
  using DD = Dist<D>;
  __shared__ double xs[kBinTile];
  __shared__ double res[kBinTile];
  __shared__ unsigned short idx[kBinTile];
  __shared__ unsigned cnt[kBuckets];
  __shared__ unsigned long long red_min[kBlock / 64], red_max[kBlock / 64];
  __shared__ double red_s[kBlock / 64], red_c[kBlock / 64];
  // kLogTable marks logpdf bodies (and pdf = exp(logpdf)): the other functions never touch the table, and without its
  // 4 KiB the tile buffers fit four workgroups per CU instead of three (42128 -> 38032 bytes of LDS)
  if constexpr (uses_log_table<DD>::value && (FN == M355_LOGPDF || FN == M355_PDF)) log_table_init();
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long long ntiles = n / kBinTile;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kBinTile;
    double v[kBinPer];
    unsigned long long key[kBinPer];
    unsigned long long kmin = ~0ull, kmax = 0ull;
#pragma unroll
    for (int j = 0; j < kBinPer; j += 2) load2<0>(x, xdt, base + (long long)(j / 2) * (kBlock * 2) + t * 2, v[j], v[j + 1]);
    // scale of the tile: mean |x| over its finite elements
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
    if (lane == 0) {
      red_s[wave] = sabs;
      red_c[wave] = cfin;
    }
    __syncthreads();
    sabs = 0.0;
    cfin = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      sabs += red_s[w];
      cfin += red_c[w];
    }
    const double inv_scale = (sabs > 0.0 && cfin > 0.0) ? cfin / sabs : 1.0;
    // sort key: sign(x) * (1 + |x| / scale) read as an ordered integer -- linear in x inside the bulk (one binade
    // on each side of zero), logarithmic in the tails; +-inf / NaN are clamped so that they do not stretch the range
#pragma unroll
    for (int j = 0; j < kBinPer; ++j) {
      double tt = __builtin_fabs(v[j]) * inv_scale;
      tt = tt < 1048576.0 ? tt : ((tt == tt) ? 1048576.0 : 0.0);
      const long long mag = (long long)(__builtin_bit_cast(unsigned long long, 1.0 + tt) - 0x3ff0000000000000ull);
      key[j] = (unsigned long long)((v[j] < 0.0 ? -mag : mag) + (1ll << 62));  // contiguous across zero
      kmin = key[j] < kmin ? key[j] : kmin;
      kmax = key[j] > kmax ? key[j] : kmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long a = __shfl_xor(kmin, o, 64), b = __shfl_xor(kmax, o, 64);
      kmin = a < kmin ? a : kmin;
      kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) {
      red_min[wave] = kmin;
      red_max[wave] = kmax;
    }
    cnt[t] = 0u;  // kBuckets == kBlock
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      kmin = red_min[w] < kmin ? red_min[w] : kmin;
      kmax = red_max[w] > kmax ? red_max[w] : kmax;
    }
    const unsigned long long span = kmax - kmin;
    const int shift = span >> 8 ? 64 - 8 - __builtin_clzll(span) + 0 : 0;  // (span >> shift) < 256
    unsigned bucket[kBinPer], rank[kBinPer];
#pragma unroll
    for (int j = 0; j < kBinPer; ++j) {
      bucket[j] = (unsigned)((key[j] - kmin) >> shift);
      rank[j] = atomicAdd(&cnt[bucket[j]], 1u);
    }
    __syncthreads();
    // exclusive prefix sum of the 256 counters (one per thread): wave scan + wave totals
    unsigned c = cnt[t], incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    __shared__ unsigned wave_tot[kBlock / 64];
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned off = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) off += (w < wave) ? wave_tot[w] : 0u;
    cnt[t] = off + incl - c;  // exclusive prefix
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kBinPer; ++j) {
      const unsigned slot = cnt[bucket[j]] + rank[j];
      const int local = (j / 2) * (kBlock * 2) + t * 2 + (j & 1);
      xs[slot] = v[j];
      idx[slot] = (unsigned short)local;
    }
    __syncthreads();
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
    __syncthreads();
  }
  for (long long i = ntiles * kBinTile + (long long)blockIdx.x * kBlock + t; i < n; i += (long long)gridDim.x * kBlock)
    store1(out, odt, i, DD::template eval<FN>(load1(x, xdt, i), k));
}

// ------------------------------------------------------------------------------------------------
// k_colbinned: the same idea for row-periodic parameters (x of shape (R, P), parameters of shape (P,): "config 2,
// broadcast params").  In memory order every lane of a wavefront has its own parameter set, so sorting by x alone
// groups nothing; but a COLUMN of x shares one parameter set.  A workgroup therefore takes a tile of 512 rows x 8
// columns (each row's 8 values are 64 contiguous bytes), sorts each column's 512 values by x in LDS (64 buckets,
// counting sort) and evaluates column by column: the whole workgroup runs one parameter set at a time (its derived
// constants uniform, read once from LDS into scalar registers) and each wavefront pass holds one octile of that
// column's x.  Results return through LDS to their rows and leave as 64-byte row segments.  A workgroup keeps its 8 columns for
// all the row blocks it walks, so the 8 derive() calls (lgamma, log ...) happen once per workgroup.
constexpr int kColW = 8;         // columns per tile
constexpr int kColPer = 2;       // rows per thread
constexpr int kColRows = kColPer * kBlock;  // rows per tile: 512 values per column = 8 wavefront passes (octiles of x)
constexpr int kColBuckets = 64;

typedef __attribute__((address_space(1))) double gdouble;  // a double in global memory (for pointers that arrive generic)
M355_DEV long long uniform_ll(long long v) {  // v is wave-uniform: move it to scalar registers
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
#else
  return v;
#endif
}
M355_DEV double uniform_d(double v) {  // v is wave-uniform: move it to scalar registers
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
#else
  return v;
#endif
}

template <int D, int FN>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(min_waves<Dist<D>>::value))) void k_colbinned(
    const void* __restrict__ x, int xdt, void* __restrict__ out, int odt, long long row_blocks, unsigned P, PeriodicParams pa) {
  using DD = Dist<D>;
  using K = typename DD::K;
  constexpr int NK = sizeof(K) / sizeof(double);
  constexpr int NV = kColPer * kColW;  // values per thread: row t + 256 p, column c at v[p * kColW + c]
  __shared__ double xs[kColW * kColRows];
  __shared__ unsigned cnt[kColW * kColBuckets];
  __shared__ double kt[kColW * NK];
  __shared__ unsigned red_min[kBlock / 64][kColW], red_max[kBlock / 64][kColW];
  if constexpr (uses_log_table<DD>::value && (FN == M355_LOGPDF || FN == M355_PDF)) log_table_init();
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned G = P / kColW;
  const unsigned g = blockIdx.x % G;                               // this workgroup's column group, for all its tiles
  const long long rb0 = blockIdx.x / G, rbstep = gridDim.x / G;    // (gridDim.x is a multiple of G)
  if (t < kColW) {
    double pp[kMaxParams];
#pragma unroll
    for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][g * kColW + t] : pa.scalar[q];
    const K k = DD::derive(pp);
    const double* kd = reinterpret_cast<const double*>(&k);
#pragma unroll
    for (int f = 0; f < NK; ++f) kt[t * NK + f] = kd[f];
  }
  __syncthreads();
  for (long long rb = rb0; rb < row_blocks; rb += rbstep) {
    const long long base = (rb * kColRows + t) * (long long)P + (long long)g * kColW;
    double v[NV];
#pragma unroll
    for (int p = 0; p < kColPer; ++p) {
#pragma unroll
      for (int j = 0; j < kColW; j += 2)
        load2<0>(x, xdt, base + (long long)p * kBlock * P + j, v[p * kColW + j], v[p * kColW + j + 1]);
    }
    // 32-bit sort key per value: the high word of sign(x) (1 + |x|) read as an ordered integer (linear in x below 1,
    // logarithmic above; NaN -> 0); per-column minimum / maximum over the tile's rows
    unsigned key[NV], kmin[kColW], kmax[kColW];
#pragma unroll
    for (int c = 0; c < kColW; ++c) {
      kmin[c] = ~0u;
      kmax[c] = 0u;
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      double tt = __builtin_fabs(v[e]);
      tt = tt < 0x1p500 ? tt : ((tt == tt) ? 0x1p500 : 0.0);
      const int mag = (int)((__builtin_bit_cast(unsigned long long, 1.0 + tt) - 0x3ff0000000000000ull) >> 32);
      key[e] = (unsigned)((v[e] < 0.0 ? -mag : mag) + (1 << 30));
      kmin[e % kColW] = key[e] < kmin[e % kColW] ? key[e] : kmin[e % kColW];
      kmax[e % kColW] = key[e] > kmax[e % kColW] ? key[e] : kmax[e % kColW];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int c = 0; c < kColW; ++c) {
        const unsigned a_ = __shfl_xor(kmin[c], o, 64), b_ = __shfl_xor(kmax[c], o, 64);
        kmin[c] = a_ < kmin[c] ? a_ : kmin[c];
        kmax[c] = b_ > kmax[c] ? b_ : kmax[c];
      }
    }
    if (lane < kColW) {
      unsigned mn = kmin[0], mx = kmax[0];
#pragma unroll
      for (int c = 1; c < kColW; ++c) {
        mn = lane == c ? kmin[c] : mn;
        mx = lane == c ? kmax[c] : mx;
      }
      red_min[wave][lane] = mn;
      red_max[wave][lane] = mx;
    }
    cnt[t] = 0u;
    cnt[t + kBlock] = 0u;  // kColW * kColBuckets == 2 * kBlock
    __syncthreads();
    unsigned bucket[NV], rank[NV];
#pragma unroll
    for (int c = 0; c < kColW; ++c) {
      unsigned mn = red_min[0][c], mx = red_max[0][c];
#pragma unroll
      for (int w = 1; w < kBlock / 64; ++w) {
        mn = red_min[w][c] < mn ? red_min[w][c] : mn;
        mx = red_max[w][c] > mx ? red_max[w][c] : mx;
      }
      const unsigned span = mx - mn;
      const int shift = span >> 6 ? 32 - 6 - __builtin_clz(span) : 0;  // (span >> shift) < 64
#pragma unroll
      for (int p = 0; p < kColPer; ++p) {
        const int e = p * kColW + c;
        bucket[e] = (key[e] - mn) >> shift;
        rank[e] = atomicAdd(&cnt[c * kColBuckets + bucket[e]], 1u);
      }
    }
    __syncthreads();
    // exclusive prefix sums of each column's 64 counters: wavefront w scans columns 2w and 2w + 1, lane = bucket
#pragma unroll
    for (int h = 0; h < kColW / (kBlock / 64); ++h) {
      const int c = wave * (kColW / (kBlock / 64)) + h;
      const unsigned cval = cnt[c * kColBuckets + lane];
      unsigned incl = cval;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      cnt[c * kColBuckets + lane] = incl - cval;
    }
    __syncthreads();
    unsigned slot[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const int c = e % kColW;
      slot[e] = cnt[c * kColBuckets + bucket[e]] + rank[e];
      xs[c * kColRows + slot[e]] = v[e];
    }
    __syncthreads();
    // column by column, kColPer passes each; the pass / wavefront that takes a given octile of x rotates with the
    // column, so that a systematic cost gradient along x (cheap tails, dear middle) does not always load the same
    // wavefront (the wavefronts only meet again at the barrier after the last column)
#pragma unroll 1
    for (int c = 0; c < kColW; ++c) {
      K k;
      double* kd = reinterpret_cast<double*>(&k);
#pragma unroll
      for (int f = 0; f < NK; ++f) kd[f] = uniform_d(kt[c * NK + f]);
#pragma unroll 1
      for (int p = 0; p < kColPer; ++p) {
        const int s_ = c * kColRows + ((p * kBlock + t + 64 * c) & (kColRows - 1));
        xs[s_] = DD::template eval<FN>(xs[s_], k);  // in place: each slot has one reader
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = xs[(e % kColW) * kColRows + slot[e]];
#pragma unroll
    for (int p = 0; p < kColPer; ++p) {
#pragma unroll
      for (int j = 0; j < kColW; j += 2)
        store2<0>(out, odt, base + (long long)p * kBlock * P + j, v[p * kColW + j], v[p * kColW + j + 1]);
    }
    __syncthreads();
  }
}

// Dist<D>::kBinned = 1: every function but logpdf / pdf goes through k_binned; = 2: only ppf / isf (families whose
// cdf is closed-form but whose quantile is a Newton iteration).
template <class T, class = void> struct binning_mode { static constexpr int value = 0; };
template <class T> struct binning_mode<T, std::void_t<decltype(T::kBinned)>> { static constexpr int value = T::kBinned; };
template <class T, int FN> constexpr bool bin_this() {
  constexpr int m = binning_mode<T>::value;
  return (m == 1 && FN != M355_LOGPDF && FN != M355_PDF) || (m == 2 && (FN == M355_PPF || FN == M355_ISF));
}

// ---- Tabulation at launch (scalar parameters, large n): k_tab_build + k_tab ------------------------------------------------
// With the parameters fixed, logcdf / logsf / cdf / sf / ppf / isf are smooth functions of ONE variable, and the exact
// bodies cost 400-3000 operations per element behind divergent loops.  The function is tabulated ONCE per call, by a
// launch of its own (k_tab_build: 64 wavefronts, four cells each), into a workspace the caller brings
// (m355_elemwise_ws, include/m355_dist.h); the evaluation kernel copies the finished table (20 KiB) into LDS:
//   * kTabCells = 256 cells across [ppf(1e-9), ppf(1 - 1e-9)] in a variable w = Dist<>::tab_w(x) in which the function is
//     tame (log x for gamma, logit x for beta, asinh z for studentt; logit q for the quantiles);
//   * per cell a Chebyshev interpolant from kTabNodes = 14 exact evaluations at the cell's Chebyshev nodes, of which the
//     first kTabCoef = 10 coefficients are kept (degree 9): the four that are dropped MEASURE what the truncation loses --
//     and any noise in the node values (a badly conditioned x(w), the exact body's own rounding at large parameters, a
//     staircase in the reference's own formula) -- so a cell is used only if |c_10| + ... + |c_13| <= 1e-14 of its scale;
//     a cell that fails is poisoned with NaN.  (Four, not two: with two, one noisy cell in ~2000 passes by chance -- round
//     3's sweep of the acceptance boxes met one at beta(0.5, 499.5).logsf, 1.5e-12 off the reference's staircase);
//   * the kept coefficients are converted to the monomial basis in the cell's t in [-1, 1) (safe: a validated cell's
//     coefficients fall by >= 25x per degree), ten doubles = 80 bytes per cell: a lookup is one table logarithm, the cell
//     arithmetic, FIVE 16-byte LDS reads and a degree-9 Horner chain.  (Round 2 built a 64 x degree-15 table in every
//     workgroup: 1024 workgroups x 1024 exact evaluations = 60 us per launch, and eight 16-byte reads per element made the
//     kernel LDS-bound.  The 80-byte cell stride spreads eight neighbouring cells over all the banks.)
// An element whose lookup is NaN -- poisoned cell, w outside the table, NaN / out-of-support argument -- takes the exact
// body.  So the table can only replace the exact value where it has shown itself converged; the price of a miss is
// divergence.
constexpr int kTabCells = 256, kTabNodes = 14, kTabCoef = 10, kTabStride = 10, kTabHdr = 8;
constexpr int kTabDoubles = kTabHdr + kTabCells * kTabStride;  // header: [0] w_lo, [1] cells per unit of w (NaN: no table)
constexpr int kTabBuildWaves = kTabCells / 4;                   // k_tab_build: one wavefront (= workgroup) per four cells
constexpr double kTabTailQ = 1e-9;     // the table spans the quantiles [kTabTailQ, 1 - kTabTailQ]
constexpr double kTabCoefTol = 1e-14;  // dropped (k_tab) / last two (k_coltab) coefficients <= kTabCoefTol * scale
template <class T, class = void> struct has_tab_var : std::false_type {};
template <class T> struct has_tab_var<T, std::void_t<decltype(&T::tab_w)>> : std::true_type {};
// optional per-distribution node evaluators (smooth versions of bodies whose reference formula is a staircase somewhere)
template <class T, class = void> struct has_tab_node : std::false_type {};
template <class T> struct has_tab_node<T, std::void_t<decltype(&T::tab_ppf_node)>> : std::true_type {};  // quantile nodes
template <class T, class = void> struct has_tab_ok_fn : std::false_type {};  // per-function restrictions of tab_ok
template <class T> struct has_tab_ok_fn<T, std::void_t<decltype(T::kTabOkFn)>> : std::true_type {};
// Dist<>::tab_ppf_log(k): tabulate log(ppf) instead of ppf for these parameters (shapes below 1: ppf ~ q^(1/alpha) in the
// lower tail, a straight line in logit q once the logarithm is taken); the lookup returns exp(table)
template <class T, class = void> struct has_tab_ppf_log : std::false_type {};
template <class T> struct has_tab_ppf_log<T, std::void_t<decltype(&T::tab_ppf_log)>> : std::true_type {};
template <class T, class = void> struct has_tab_miss : std::false_type {};  // Dist<>::tab_miss<FN>: eval<FN> for parameters inside tab_ok
template <class T> struct has_tab_miss<T, std::void_t<decltype(&T::template tab_miss<M355_LOGCDF>)>> : std::true_type {};
template <class DD, int FN> M355_DEV double tab_miss_eval(double x, const typename DD::K& k) {
  if constexpr (has_tab_miss<DD>::value) return DD::template tab_miss<FN>(x, k);
  else return DD::template eval<FN>(x, k);
}
template <class DD, int FN> M355_DEV bool tab_log_mode(const typename DD::K& k) {
  if constexpr ((FN == M355_PPF || FN == M355_ISF) && has_tab_ppf_log<DD>::value) return DD::tab_ppf_log(k);
  else return false;
}
// Dist<>::tab_q_den(q): the reference's own "1 - q" where its ppf forms it with a rounding of its own (halfstudentt:
// 2 - 2 ((q + 1) / 2)); the quantile lookup then takes logit from (1 - den, den), i.e. follows the reference's argument
template <class T, class = void> struct has_tab_q_den : std::false_type {};
template <class T> struct has_tab_q_den<T, std::void_t<decltype(&T::tab_q_den)>> : std::true_type {};
template <class DD> M355_DEV double tab_logit(double qq) {
  double num = qq, den = 1.0 - qq;
  if constexpr (has_tab_q_den<DD>::value) {
    den = DD::tab_q_den(qq);
    num = 1.0 - den;
  }
  return flog_tab_or_nan(fdiv_t<1, true>(num, den));
}
// Dist<>::tab_value_miss<FN>(r): table values the reference does not reach within the parity bar (its own rounding
// noise, e.g. log of a difference of O(1) numbers): a miss, i.e. the exact body's reproduction of that arithmetic
template <class T, class = void> struct has_tab_value_miss : std::false_type {};
template <class T> struct has_tab_value_miss<T, std::void_t<decltype(&T::template tab_value_miss<M355_LOGCDF>)>> : std::true_type {};
// Dist<>::tab_tol(k): the validation tolerance where the exact body's own rounding noise exceeds 1e-14 inside the box
// tab_ok accepts (beta at a + b in the hundreds): a table cannot show itself closer to its nodes than the nodes are to
// each other
template <class T, class = void> struct has_tab_tol : std::false_type {};
template <class T> struct has_tab_tol<T, std::void_t<decltype(&T::tab_tol)>> : std::true_type {};
template <class DD> M355_DEV double tab_tol_of(const typename DD::K& k) {
  if constexpr (has_tab_tol<DD>::value) return DD::tab_tol(k);
  else return kTabCoefTol;
}
// Dist<>::tab_qfloor(k): the smallest scale a QUANTILE cell is validated against.  Quantile cells are held to
// tol * sum |c_k| (relative to the root); a family whose root is formed as lower + p (upper - lower) carries a rounding
// of ~1e-16 max(|lower|, |upper|) in every node whatever the size of the root, and a narrow cell around a root near
// zero (betascaled on (-1, 3): 256 cells of 0.16 in logit q) was poisoned by it -- 12 % misses, 6.9 ms instead of 0.54.
template <class T, class = void> struct has_tab_polish_lower : std::false_type {};
template <class T> struct has_tab_polish_lower<T, std::void_t<decltype(T::kTabPolishLower)>> : std::true_type {};
template <class T, class = void> struct has_tab_qfloor : std::false_type {};
template <class T> struct has_tab_qfloor<T, std::void_t<decltype(&T::tab_qfloor)>> : std::true_type {};
template <class DD, int FN> M355_DEV double tab_scale_of(double sum, const typename DD::K& k) {
  constexpr bool QDOM = FN == M355_PPF || FN == M355_ISF;
  double scale = ((QDOM && !tab_log_mode<DD, FN>(k)) || sum > 1.0) ? sum : 1.0;
  if constexpr (QDOM && has_tab_qfloor<DD>::value) {
    const double fl = DD::tab_qfloor(k);
    scale = scale < fl ? fl : scale;
  }
  return scale;
}
template <class T, class = void> struct has_tab_cap : std::false_type {};
template <class T> struct has_tab_cap<T, std::void_t<decltype(&T::tab_cap)>> : std::true_type {};
template <class T, class = void> struct has_tab_fnode : std::false_type {};
template <class T> struct has_tab_fnode<T, std::void_t<decltype(T::kTabNode)>> : std::true_type {};  // tab_node<FN>
// cdf / sf through the table of a log function (Dist<>::tab_route, m355_dists.h): the function a (dist, fn) pair tabulates
// and what the lookup does with the table value
template <class T, class = void> struct has_tab_route : std::false_type {};
template <class T> struct has_tab_route<T, std::void_t<decltype(T::template tab_route<M355_CDF>())>> : std::true_type {};
template <class DD, int FN> constexpr TabRoute tab_route_of() {
  if constexpr (has_tab_route<DD>::value && (FN == M355_CDF || FN == M355_SF)) return DD::template tab_route<FN>();
  else return TabRoute{FN, kTabEpiNone};
}
template <int EPI> M355_DEV double tab_epilogue(double r) {  // (NaN, the mark of a miss, passes through)
  if constexpr (EPI == kTabEpiExp) return exp_(r);
  else if constexpr (EPI == kTabEpiOneMinusExp) return 1.0 - exp_(r);
  else return r;
}
template <class DD, int FN> constexpr bool tab_this() {
  return has_tab_var<DD>::value && (FN == M355_LOGCDF || FN == M355_LOGSF || FN == M355_CDF || FN == M355_SF ||
                                    FN == M355_PPF || FN == M355_ISF);
}
// Quantiles are tabulated in w = logit(q) over q in [kTabTailQ, 1 - kTabTailQ]: ppf(sigmoid(w)) is analytic in the strip
// |Im w| < pi.  A node with w > 0 cannot be computed as ppf(q): q = 1 - p rounds where p = sigmoid(-w) does not, so it
// takes ppf(fl(1 - p)) as a start and two Newton steps on logsf(x) = log p, which holds its relative accuracy in the
// tail.  isf(q) is the reference's ppf(1 - q) with the rounded 1 - q (e.g. gamma.py:57-58), so it looks up
// qq = fl(1 - q) -- the same staircase as the exact kernel.
constexpr double kTabLogitMax = 20.723265835946928;  // logit(1 - 1e-9)
// cos(pi a / 28), a = 0 .. 55: the 14 Chebyshev nodes of a k_tab cell are the odd entries below 28, and
// cos(pi m (2 q + 1) / 28) = entry m (2 q + 1) mod 56
constexpr double kCos28[56] = {
    0x1.0000000000000p+0, 0x1.fcc7d8c64135fp-1, 0x1.f329c0558e969p-1, 0x1.e344ad05d3f86p-1, 0x1.cd4bca9cb5c71p-1,
    0x1.b185d590d5a44p-1, 0x1.904c37505de4bp-1, 0x1.6a09e667f3bcdp-1, 0x1.3f3a0e28bedd1p-1, 0x1.106682221cd8ap-1,
    0x1.bc4c04d71abc1p-2, 0x1.5234aca69a9fep-2, 0x1.c7b90e3024582p-3, 0x1.ca9b4332d6f61p-4, 0.0,
    -0x1.ca9b4332d6f61p-4, -0x1.c7b90e3024582p-3, -0x1.5234aca69a9fep-2, -0x1.bc4c04d71abc1p-2, -0x1.106682221cd8ap-1,
    -0x1.3f3a0e28bedd1p-1, -0x1.6a09e667f3bcdp-1, -0x1.904c37505de4bp-1, -0x1.b185d590d5a44p-1, -0x1.cd4bca9cb5c71p-1,
    -0x1.e344ad05d3f86p-1, -0x1.f329c0558e969p-1, -0x1.fcc7d8c64135fp-1, -0x1.0000000000000p+0, -0x1.fcc7d8c64135fp-1,
    -0x1.f329c0558e969p-1, -0x1.e344ad05d3f86p-1, -0x1.cd4bca9cb5c71p-1, -0x1.b185d590d5a44p-1, -0x1.904c37505de4bp-1,
    -0x1.6a09e667f3bcdp-1, -0x1.3f3a0e28bedd1p-1, -0x1.106682221cd8ap-1, -0x1.bc4c04d71abc1p-2, -0x1.5234aca69a9fep-2,
    -0x1.c7b90e3024582p-3, -0x1.ca9b4332d6f61p-4, 0.0, 0x1.ca9b4332d6f61p-4, 0x1.c7b90e3024582p-3,
    0x1.5234aca69a9fep-2, 0x1.bc4c04d71abc1p-2, 0x1.106682221cd8ap-1, 0x1.3f3a0e28bedd1p-1, 0x1.6a09e667f3bcdp-1,
    0x1.904c37505de4bp-1, 0x1.b185d590d5a44p-1, 0x1.cd4bca9cb5c71p-1, 0x1.e344ad05d3f86p-1, 0x1.f329c0558e969p-1,
    0x1.fcc7d8c64135fp-1};
// cos(pi a / 32), a = 0 .. 63: the same for the 16-node cells of k_coltab
constexpr double kCos32[64] = {
    0x1.0000000000000p+0, 0x1.fd88da3d12526p-1, 0x1.f6297cff75cb0p-1, 0x1.e9f4156c62ddap-1,
    0x1.d906bcf328d46p-1, 0x1.c38b2f180bdb1p-1, 0x1.a9b66290ea1a3p-1, 0x1.8bc806b151741p-1,
    0x1.6a09e667f3bcdp-1, 0x1.44cf325091dd6p-1, 0x1.1c73b39ae68c8p-1, 0x1.e2b5d3806f63bp-2,
    0x1.87de2a6aea963p-2, 0x1.294062ed59f06p-2, 0x1.8f8b83c69a60bp-3, 0x1.917a6bc29b42cp-4,
    0.0, -0x1.917a6bc29b42cp-4, -0x1.8f8b83c69a60bp-3, -0x1.294062ed59f06p-2,
    -0x1.87de2a6aea963p-2, -0x1.e2b5d3806f63bp-2, -0x1.1c73b39ae68c8p-1, -0x1.44cf325091dd6p-1,
    -0x1.6a09e667f3bcdp-1, -0x1.8bc806b151741p-1, -0x1.a9b66290ea1a3p-1, -0x1.c38b2f180bdb1p-1,
    -0x1.d906bcf328d46p-1, -0x1.e9f4156c62ddap-1, -0x1.f6297cff75cb0p-1, -0x1.fd88da3d12526p-1,
    -0x1.0000000000000p+0, -0x1.fd88da3d12526p-1, -0x1.f6297cff75cb0p-1, -0x1.e9f4156c62ddap-1,
    -0x1.d906bcf328d46p-1, -0x1.c38b2f180bdb1p-1, -0x1.a9b66290ea1a3p-1, -0x1.8bc806b151741p-1,
    -0x1.6a09e667f3bcdp-1, -0x1.44cf325091dd6p-1, -0x1.1c73b39ae68c8p-1, -0x1.e2b5d3806f63bp-2,
    -0x1.87de2a6aea963p-2, -0x1.294062ed59f06p-2, -0x1.8f8b83c69a60bp-3, -0x1.917a6bc29b42cp-4,
    0.0, 0x1.917a6bc29b42cp-4, 0x1.8f8b83c69a60bp-3, 0x1.294062ed59f06p-2,
    0x1.87de2a6aea963p-2, 0x1.e2b5d3806f63bp-2, 0x1.1c73b39ae68c8p-1, 0x1.44cf325091dd6p-1,
    0x1.6a09e667f3bcdp-1, 0x1.8bc806b151741p-1, 0x1.a9b66290ea1a3p-1, 0x1.c38b2f180bdb1p-1,
    0x1.d906bcf328d46p-1, 0x1.e9f4156c62ddap-1, 0x1.f6297cff75cb0p-1, 0x1.fd88da3d12526p-1};

// T_k(t) = sum_j t[k][j] t^j, k, j < 16 (integers up to 2^14: exact)
struct ChebMono16 {
  double t[16][16];
  constexpr ChebMono16() : t{} {
    t[0][0] = 1.0;
    t[1][1] = 1.0;
    for (int k = 2; k < 16; ++k)
      for (int j = 0; j < 16; ++j) t[k][j] = (j > 0 ? 2.0 * t[k - 1][j - 1] : 0.0) - t[k - 2][j];
  }
};
constexpr ChebMono16 kChebMono16{};

// the function k_tab_build / k_coltab_build tabulate, at table variable w
template <class DD, int FN> M355_DEV double tab_node_value(double w, const typename DD::K& k) {
  constexpr bool QDOM = FN == M355_PPF || FN == M355_ISF;
  double f;
  if constexpr (QDOM && has_tab_node<DD>::value) {
    f = DD::tab_ppf_node(w, k);
  } else if constexpr (!QDOM && has_tab_fnode<DD>::value) {
    f = DD::template tab_node<FN>(w, k);
  } else if constexpr (QDOM) {
    const double e = exp_(-__builtin_fabs(w));
    const double p = fdiv(e, 1.0 + e);  // the smaller of q, 1 - q
    if (w <= 0.0) {
      f = tab_miss_eval<DD, M355_PPF>(p, k);
      if constexpr (has_tab_polish_lower<DD>::value) {
        // families whose LOWER quantiles are the hard direction of their inverse (inversegamma: beta / Qinv(alpha, p))
        // polish those nodes on logcdf as the upper ones are polished on logsf
        const double lp = flog(p);
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
          const double lc = tab_miss_eval<DD, M355_LOGCDF>(f, k);
          const double step = (lc - lp) * exp_(lc - tab_miss_eval<DD, M355_LOGPDF>(f, k));
          if (step - step == 0.0) f -= step;
          if (__builtin_fabs(step) <= 1e-9 * __builtin_fabs(f)) break;
        }
      }
    } else {
      f = tab_miss_eval<DD, M355_PPF>(1.0 - p, k);
      const double lp = flog(p);
#pragma unroll 1
      for (int it = 0; it < 2; ++it) {
        const double lsf = tab_miss_eval<DD, M355_LOGSF>(f, k);
        const double step = (lsf - lp) * exp_(lsf - tab_miss_eval<DD, M355_LOGPDF>(f, k));
        if (step - step == 0.0) f += step;
        // (Newton squares the error: a first step below 1e-9 of the root leaves < 1e-17 for the second)
        if (__builtin_fabs(step) <= 1e-9 * __builtin_fabs(f)) break;
      }
    }
    if (tab_log_mode<DD, FN>(k)) f = flog(f);
  } else {
    f = tab_miss_eval<DD, FN>(DD::tab_x(w, k), k);  // (nodes are evaluated inside tab_ok only)
  }
  return f;
}
// one end (upper: the right one) of the cdf family's table range
template <class DD> M355_DEV double tab_range_end(bool upper, const typename DD::K& k) {
  // (an end whose quantile is not a usable argument -- beta with b < 1: ppf(1 - 1e-9) rounds to 1 -- moves inwards)
  double tq = kTabTailQ, we = kNaN;
#pragma unroll 1
  for (int it = 0; it < 4 && !(we - we == 0.0); ++it, tq *= 100.0)
    we = DD::tab_w(tab_miss_eval<DD, M355_PPF>(upper ? 1.0 - tq : tq, k), k);
  if constexpr (has_tab_cap<DD>::value) {  // (still none: the outermost usable argument)
    if (!(we - we == 0.0)) we = DD::tab_w(DD::tab_cap(upper, k), k);
  }
  return we;
}

template <int D, int FN> __device__ __noinline__ double tab_exact(double x, const typename Dist<D>::K& k) {
  return tab_miss_eval<Dist<D>, FN>(x, k);
}

// One wavefront per four cells: lane = 16 * (cell in the group) + node; lanes 14, 15 of a cell repeat node 13.
template <int D, int FN>
__global__ __launch_bounds__(64) void k_tab_build(const typename Dist<D>::K k, double* __restrict__ table) {
  using DD = Dist<D>;
  constexpr bool QDOM = FN == M355_PPF || FN == M355_ISF;
  constexpr TabRoute RT = tab_route_of<DD, FN>();  // cdf / sf: the table holds RT.fn, the lookup applies RT.epi
  __shared__ double fv[64];
  __shared__ double rng[2];
  log_table_init();  // tab_w may use flog_tab
  const int lane = threadIdx.x, cl = lane >> 4, m = lane & 15;
  if (lane < 2) {
    if constexpr (QDOM) rng[lane] = lane == 0 ? -kTabLogitMax : kTabLogitMax;
    else rng[lane] = tab_range_end<DD>(lane == 1, k);
  }
  __syncthreads();
  const double wlo = rng[0], whi = rng[1];
  const double h = (whi - wlo) * (1.0 / kTabCells);
  const int cell = blockIdx.x * 4 + cl;
  const int jn = m < kTabNodes ? m : kTabNodes - 1;
  const double w = wlo + h * ((double)cell + 0.5 + 0.5 * kCos28[2 * jn + 1]);
  fv[lane] = (h > 0.0) ? tab_node_value<DD, RT.fn>(w, k) : kNaN;  // (an empty or NaN range: no table, see the header)
  __syncthreads();
  // Chebyshev coefficients: c_m = (2 / 14) sum_j f_j cos(pi m (j + 1/2) / 14), c_0 halved
  double c = 0.0;
  if (m < kTabNodes) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < kTabNodes; ++q) acc = __builtin_fma(fv[cl * 16 + q], kCos28[(m * (2 * q + 1)) % 56], acc);
    c = acc * (m == 0 ? 1.0 / kTabNodes : 2.0 / kTabNodes);
  }
  // the parity bar is relative to max(1, |value|); a quantile must hold it relative to the root itself (sum |c_k| >= the
  // cell's sup norm; |c_0| alone would vanish where a quantile changes sign); log mode: an absolute error e of the table
  // is a relative error e of the quantile: scale max(1, |log x|)
  double sum = __builtin_fabs(c), tail = (m >= kTabCoef) ? __builtin_fabs(c) : 0.0;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 16);
    tail += __shfl_xor(tail, o, 16);
  }
  const double scale = tab_scale_of<DD, FN>(sum, k);
  const bool bad = !(tail <= tab_tol_of<DD>(k) * scale);  // NaN / inf coefficients are bad too
  __syncthreads();
  fv[lane] = c;
  __syncthreads();
  if (m < kTabCoef) {  // monomial coefficients a_m = sum_{k < 10} c_k T_k[m]
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < kTabCoef; ++q) acc = __builtin_fma(fv[cl * 16 + q], kChebMono16.t[q][m], acc);
    table[kTabHdr + cell * kTabStride + m] = (m == 0 && bad) ? kNaN : acc;
  }
  if (blockIdx.x == 0 && lane == 0) {
    table[0] = wlo;
    table[1] = (h > 0.0) ? 1.0 / h : kNaN;  // (NaN poisons every lookup)
  }
}

// REDUCE: the fused sum (sum_logcdf / sum_logsf: censored likelihoods) -- nothing stored, one partial per workgroup.
template <int D, int FN, bool F64IO, bool REDUCE = false>
__global__ __launch_bounds__(kBlock) void k_tab(const void* __restrict__ x, int xdt, void* __restrict__ out, int odt,
                                                long long n, const typename Dist<D>::K k, const double* __restrict__ table,
                                                double* __restrict__ partials = nullptr) {
  using DD = Dist<D>;
  constexpr int IO = F64IO ? 1 : 0;
  constexpr bool QDOM = FN == M355_PPF || FN == M355_ISF;
  constexpr TabRoute RT = tab_route_of<DD, FN>();
  __shared__ __attribute__((aligned(16))) double tab[kTabDoubles];
  for (int i = threadIdx.x; i < kTabDoubles; i += kBlock) tab[i] = table[i];
  log_table_init();  // (its barrier publishes `tab` too)
  const double wlo = tab[0], inv_h = tab[1];
  const double* coef = tab + kTabHdr;
  // table value, NaN where the table has nothing to say (w outside it, a poisoned cell, a NaN argument)
  auto lookup = [&](double xv) -> double {
    double wv;
    if constexpr (QDOM) {
      const double qq = (FN == M355_ISF) ? 1.0 - xv : xv;
      wv = tab_logit<DD>(qq);
      if constexpr (has_tab_node<DD>::value) wv = DD::tab_q_miss(qq) ? kNaN : wv;
    } else {
      wv = DD::tab_w(xv, k);
    }
    const double s = (wv - wlo) * inv_h;
    const double fi = __builtin_floor(s);
    const bool in = (s >= 0.0) & (s < (double)kTabCells);
    const double t = __builtin_fma(2.0, s - fi, -1.0);
    const d2_t* c = reinterpret_cast<const d2_t*>(coef + (in ? (int)fi : 0) * kTabStride);
    d2_t cc = c[kTabCoef / 2 - 1];
    double r = __builtin_fma(cc.y, t, cc.x);
#pragma unroll
    for (int q = kTabCoef / 2 - 2; q >= 0; --q) {
      cc = c[q];  // a_{2q}, a_{2q+1}
      r = __builtin_fma(__builtin_fma(r, t, cc.y), t, cc.x);
    }
    if (tab_log_mode<DD, FN>(k)) r = exp_(r);  // (uniform)
    if constexpr (has_tab_value_miss<DD>::value) r = DD::template tab_value_miss<RT.fn>(r) ? kNaN : r;
    r = tab_epilogue<RT.epi>(r);
    return in ? r : kNaN;
  };
  const long long ntiles = n / kTile;
  double acc = 0.0;
  (void)acc;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kTile + threadIdx.x * kVec;
    double v[kUnroll][kVec], r[kUnroll][kVec];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) load2<IO>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
    bool miss = false;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
      for (int q = 0; q < kVec; ++q) {
        r[u][q] = lookup(v[u][q]);
        miss |= r[u][q] != r[u][q];
      }
    }
    if (miss) {
      // rare: the exact body, one out-of-line call site for the four slots (four sites cost 66 more VGPRs around the
      // calls and a wavefront per SIMD); the slot is picked by selects on the uniform e -- indexing v and r with it
      // would move both arrays to scratch memory for every tile
#pragma unroll 1
      for (int e = 0; e < kUnroll * kVec; ++e) {
        double xv = v[0][0], rv = r[0][0];
#pragma unroll
        for (int j = 1; j < kUnroll * kVec; ++j) {
          xv = (e == j) ? v[j / kVec][j % kVec] : xv;
          rv = (e == j) ? r[j / kVec][j % kVec] : rv;
        }
        if (rv != rv) {
          const double ex = tab_exact<D, FN>(xv, k);
#pragma unroll
          for (int j = 0; j < kUnroll * kVec; ++j) r[j / kVec][j % kVec] = (e == j) ? ex : r[j / kVec][j % kVec];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if constexpr (REDUCE) acc += r[u][0] + r[u][1];
      else store2<IO>(out, odt, base + u * (kBlock * kVec), r[u][0], r[u][1]);
    }
  }
  for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const double xv = load1(x, xdt, i);
    double rv = lookup(xv);
    if (rv != rv) rv = tab_exact<D, FN>(xv, k);
    if constexpr (REDUCE) acc += rv;
    else store1(out, odt, i, rv);
  }
  if constexpr (REDUCE) {
    acc = block_reduce_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
  }
}

// ---- The same for row-periodic parameters (x of shape (R, P), parameters of shape (P,)): k_coltab_* ------------------------
// A column of x shares one parameter set, so it can share one table.  All P tables are built once, by the whole grid, into
// the caller's workspace, 64 cells x degree 11 per column (sixteen tables of that size fit a CU's LDS; k_tab's wider
// format would not):
//   k_coltab_ranges   one thread per (column, end): derive, Dist<>::tab_ok, the two quantile evaluations of the range;
//   k_coltab_build    one lane per node (a wavefront = four cells of ONE column, so the exact bodies run with
//                     wavefront-uniform parameters), 16-point DCT through LDS, validation, monomial basis;
//   k_coltab          the lookups.  A workgroup of 1024 threads takes kCtCols = 16 adjacent columns (a row's 16 values are
//                     one 128-byte line), their tables in LDS (115 KiB: one workgroup per CU, four wavefronts per SIMD),
//                     thread t on column t & 15.  Rows are handed out in chunks of 4096 from a counter per column group; a
//                     workgroup serves its home group until that is exhausted and then helps the others (reloading
//                     tables), so groups with slower columns -- warped tables -- do not set the pace.  The kernel calls
//                     nothing: a miss is not stored, it becomes one bit in the workspace (one 64-bit word per thread and
//                     chunk).  With the exact bodies callable from it the kernel needed 248 VGPRs, two wavefronts per
//                     SIMD, and ran at 0.7-1.0 ms per 1e8 elements with VALU and LDS both under 40 % busy;
//   k_coltab_fix      scans the miss words and redoes those elements by the exact body -- from x, which the lookup kernel
//                     left alone there, so out may be x itself;
//   k_coltab_declined the columns whose parameters Dist<>::tab_ok declines (flag in the column's table header).
// Round 2 gave each of 256 workgroups 16 columns = 16384 exact evaluations before its first row.
// per-column table: 64 cells x 16 nodes of which the first 12 coefficients are kept (degree 11; the four that are dropped
// measure the truncation, as in k_tab_build) -- six 16-byte reads per lookup instead of round 2's eight; 112-byte cells
// (a stride of 28 banks spreads eight neighbouring cells over all of them)
constexpr int kCtCells = 64, kCtN = 16, kCtCoef = 12, kCtStride = 14;
constexpr int kCtMaxNK = 12;
// header: [0] w_lo, [1] cells per unit of w, [2] declined, [3] log mode, [4] cell width, [8 ..) the column's derived constants
constexpr int kCtHdr = 8 + kCtMaxNK + 4;
constexpr int kCtTable = kCtHdr + kCtCells * kCtStride;              // 920 doubles per column
constexpr int kCtCols = 16, kCtThreads = 1024, kCtRows = kCtThreads / kCtCols;  // workgroup: 16 columns x 64 rows per pass
// rows per thread and pass (independent lookups in flight per lane): four for the quantiles (their lookup needs no
// per-lane parameters: 71 - 128 VGPRs), two for the cdf family (with four, the per-lane constants of tab_w on top of the
// sixteen coefficient registers of every lookup in flight spilled 76 - 188 bytes per lane under the 128-VGPR budget of a
// 1024-thread workgroup)
#ifndef M355_CT_UNROLL_CDF
#define M355_CT_UNROLL_CDF 2
#endif
template <int FN> constexpr int ct_unroll() { return (FN == M355_PPF || FN == M355_ISF) ? 4 : M355_CT_UNROLL_CDF; }
// rows per thread and pass whose loads are issued together (k_coltab is a stream: with two 8-byte loads per lane in
// flight, 16 wavefronts per CU held 16 KiB of reads in the air and the kernel ran at the 3 TB/s that buys; eight loads
// per lane quadruple it); the lookups then go in groups of ct_unroll
constexpr int kCtLoad = 8;
template <int FN> constexpr int ct_iters() { return 64 / kCtLoad; }                    // passes per chunk: one miss bit per element
template <int FN> constexpr int ct_chunk_rows() { return ct_iters<FN>() * kCtLoad * kCtRows; }  // 4096
constexpr size_t kCtLds = (size_t)kCtCols * kCtTable * sizeof(double);  // dynamic LDS: 117760 B
// workspace: P tables | G chunk counters (+ padding to 8 bytes) | one miss word per (group, chunk, thread) = n / 64 words
template <int FN> inline size_t coltab_chunks(long long R) { return (size_t)((R + ct_chunk_rows<FN>() - 1) / ct_chunk_rows<FN>()); }
template <int FN> inline size_t coltab_ws_doubles(unsigned long long P, long long R) {
  const size_t G = (size_t)(P / kCtCols);
  return (size_t)P * kCtTable + (G + 2) / 2 + G * coltab_chunks<FN>(R) * kCtThreads;
}

template <class DD, int FN> M355_DEV bool coltab_params_ok(const typename DD::K& k) {
  bool ok = DD::tab_ok(k);
  if constexpr (has_tab_ok_fn<DD>::value) ok = ok && DD::template tab_ok_fn<FN>(k);
  return ok;
}

// thread 2 c + e: end e of column c's range; thread 2 c also writes the flags; the first G threads zero the chunk counters
template <int D, int FN>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(min_waves<Dist<D>>::value))) void k_coltab_ranges(
    PeriodicParams pa, unsigned P, double* __restrict__ ws, unsigned* __restrict__ counters, unsigned ncounters) {
  using DD = Dist<D>;
  using K = typename DD::K;
  constexpr bool QDOM = FN == M355_PPF || FN == M355_ISF;
  log_table_init();
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i < ncounters) counters[i] = 0u;
  if (i >= 2 * P) return;
  const unsigned col = i >> 1, e = i & 1;
  double pp[kMaxParams];
#pragma unroll
  for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][col] : pa.scalar[q];
  const K k = DD::derive(pp);
  const bool declined = !coltab_params_ok<DD, FN>(k);
  double we = kNaN;
  if (!declined) {
    if constexpr (QDOM) we = e == 0 ? -kTabLogitMax : kTabLogitMax;
    else we = tab_range_end<DD>(e == 1, k);
  }
  // the pair's two ends meet in lanes 2 c, 2 c + 1 of one wavefront
  const double other = __shfl_xor(we, 1, 64);
  if (e == 0) {
    double* table = ws + (size_t)col * kCtTable;
    const double h = (other - we) * (1.0 / kCtCells);
    table[0] = we;
    table[1] = (h > 0.0) ? 1.0 / h : kNaN;  // (an accepted column whose range came out empty: every lookup a miss)
    table[2] = declined ? 1.0 : 0.0;
    table[3] = tab_log_mode<DD, FN>(k) ? 1.0 : 0.0;
    table[4] = h;
    constexpr int NK = sizeof(K) / sizeof(double);
    static_assert(NK <= kCtMaxNK, "k_coltab: K too large");
    const double* kd = reinterpret_cast<const double*>(&k);
#pragma unroll
    for (int f = 0; f < NK; ++f) table[8 + f] = kd[f];  // k_coltab reads them from LDS: it derives (and calls) nothing
  }
}

// grid = 4 P workgroups of 256 threads: workgroup b builds cells 16 (b & 3) .. + 15 of column b >> 2 (wavefront = four cells)
template <int D, int FN>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(min_waves<Dist<D>>::value))) void k_coltab_build(
    PeriodicParams pa, unsigned P, double* __restrict__ ws) {
  using DD = Dist<D>;
  using K = typename DD::K;
  constexpr bool QDOM = FN == M355_PPF || FN == M355_ISF;
  constexpr TabRoute RT = tab_route_of<DD, FN>();
  __shared__ double fv[kBlock];
  const int t = threadIdx.x, m = t & 15;
  const unsigned col = blockIdx.x >> 2;
  const int cell = (int)(blockIdx.x & 3) * 16 + (t >> 4);
  double* table = ws + (size_t)col * kCtTable;
  if (table[2] != 0.0) return;  // (uniform over the workgroup) declined: k_coltab_declined's column
  log_table_init();
  double pp[kMaxParams];
#pragma unroll
  for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][col] : pa.scalar[q];
  const K k = DD::derive(pp);  // (every thread: the same values)
  const double wlo = table[0], h = table[4];
  {
    const double w = wlo + h * ((double)cell + 0.5 + 0.5 * kCos32[2 * m + 1]);
    fv[t] = (h > 0.0) ? tab_node_value<DD, RT.fn>(w, k) : kNaN;
  }
  __syncthreads();
  double c;
  {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < kCtN; ++q) acc = __builtin_fma(fv[(t & ~15) + q], kCos32[(m * (2 * q + 1)) & 63], acc);
    c = acc * (m == 0 ? 1.0 / kCtN : 2.0 / kCtN);
  }
  double sum = __builtin_fabs(c), tail = (m >= kCtCoef) ? __builtin_fabs(c) : 0.0;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 16);
    tail += __shfl_xor(tail, o, 16);
  }
  const double scale = tab_scale_of<DD, FN>(sum, k);
  const bool bad = !(tail <= tab_tol_of<DD>(k) * scale);
  __syncthreads();
  fv[t] = c;
  __syncthreads();
  if (m < kCtCoef) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < kCtCoef; ++q) acc = __builtin_fma(fv[(t & ~15) + q], kChebMono16.t[q][m], acc);
    table[kCtHdr + cell * kCtStride + m] = (m == 0 && bad) ? kNaN : acc;
  }
}

extern __shared__ __attribute__((aligned(16))) double ct_tab[];  // k_coltab: [kCtCols][kCtTable]

template <int D, int FN, bool F64IO>
__global__ __launch_bounds__(kCtThreads) void k_coltab(const void* __restrict__ x, int xdt, void* __restrict__ out, int odt,
                                                       long long R, unsigned P, const double* __restrict__ ws,
                                                       unsigned* __restrict__ counters, unsigned long long* __restrict__ miss_words) {
  using DD = Dist<D>;
  using K = typename DD::K;
  constexpr int NK = sizeof(K) / sizeof(double);
  static_assert(NK <= kCtMaxNK, "k_coltab: K too large");
  constexpr bool QDOM = FN == M355_PPF || FN == M355_ISF;
  constexpr TabRoute RT = tab_route_of<DD, FN>();
  constexpr int U = ct_unroll<FN>();
  __shared__ unsigned ct_next;                  // the chunk the workgroup has just claimed
  const int t = threadIdx.x;
  const unsigned G = P / kCtCols;
  const unsigned chunks = (unsigned)((R + ct_chunk_rows<FN>() - 1) / ct_chunk_rows<FN>());
  log_table_init();
  const int col = t & (kCtCols - 1), rloc = t >> 4;
  const double* mytab = ct_tab + col * kCtTable;
  const double* mycoef = mytab + kCtHdr;
  const unsigned lane_off = (unsigned)rloc * P + (unsigned)col;  // elements; rloc < 64
  const unsigned home = blockIdx.x % G;
#pragma unroll 1
  for (unsigned s = 0; s < G; ++s) {  // home group first, then the others in order: every workgroup ends after G probes
    const unsigned g = home + s < G ? home + s : home + s - G;
    bool loaded = false;
    K k;
    double wlo = 0.0, inv_h = 0.0;
    bool declined = false, logmode = false;
#pragma unroll 1
    for (;;) {
      __syncthreads();  // (the previous chunk's readers of ct_next / the tables are done)
      if (t == 0) ct_next = atomicAdd(&counters[g], 1u);
      __syncthreads();
      const unsigned ch = ct_next;
      if (ch >= chunks) break;
      if (!loaded) {
        // this group's sixteen tables and parameter sets into LDS
        const double* src = ws + (size_t)g * kCtCols * kCtTable;
        for (int i = t; i < kCtCols * kCtTable; i += kCtThreads) ct_tab[i] = src[i];
        __syncthreads();
        double* kd = reinterpret_cast<double*>(&k);
#pragma unroll
        for (int f = 0; f < NK; ++f) kd[f] = mytab[8 + f];
        wlo = mytab[0];
        inv_h = mytab[1];
        declined = mytab[2] != 0.0;  // no table for this column: k_coltab_declined does it, this loop leaves it alone
        logmode = mytab[3] != 0.0;
        loaded = true;
      }
      auto lookup = [&](double xv) -> double {
        double wv;
        if constexpr (QDOM) {
          const double qq = (FN == M355_ISF) ? 1.0 - xv : xv;
          wv = tab_logit<DD>(qq);
          if constexpr (has_tab_node<DD>::value) wv = DD::tab_q_miss(qq) ? kNaN : wv;
        } else {
          wv = DD::tab_w(xv, k);
        }
        const double s_ = (wv - wlo) * inv_h;
        const double fi = __builtin_floor(s_);
        const bool in = (s_ >= 0.0) & (s_ < (double)kCtCells);
        const double tt = __builtin_fma(2.0, s_ - fi, -1.0);
        const d2_t* c = reinterpret_cast<const d2_t*>(mycoef + (in ? (int)fi : 0) * kCtStride);
        d2_t cc = c[kCtCoef / 2 - 1];
        double r = __builtin_fma(cc.y, tt, cc.x);
#pragma unroll
        for (int q = kCtCoef / 2 - 2; q >= 0; --q) {
          cc = c[q];
          r = __builtin_fma(__builtin_fma(r, tt, cc.y), tt, cc.x);
        }
        if constexpr (QDOM && has_tab_ppf_log<DD>::value) {
          if (__builtin_amdgcn_ballot_w64(logmode) != 0) r = logmode ? exp_(r) : r;  // (skipped when no lane of the wavefront wants it)
        }
        if constexpr (has_tab_value_miss<DD>::value) r = DD::template tab_value_miss<RT.fn>(r) ? kNaN : r;
        r = tab_epilogue<RT.epi>(r);
        return in ? r : kNaN;
      };
      const long long row0 = (long long)ch * ct_chunk_rows<FN>();
      unsigned long long missed = 0;
#pragma unroll 1
      for (int it = 0; it < ct_iters<FN>(); ++it) {
        const long long rowb = row0 + (long long)it * (kCtLoad * kCtRows);     // uniform
        if (rowb >= R) break;
        const long long ubase = rowb * (long long)P + (long long)g * kCtCols;  // uniform
        const long long rows_left = R - rowb;                                  // uniform
        double v[kCtLoad];
#pragma unroll
        for (int u = 0; u < kCtLoad; ++u) {
          const long long ub = ubase + (long long)(u * kCtRows) * (long long)P;
          const bool live = (rloc < rows_left - u * kCtRows) & !declined;
          if constexpr (F64IO) v[u] = live ? __builtin_nontemporal_load((const double*)x + ub + lane_off) : 0.0;
          else v[u] = live ? load1(x, xdt, (unsigned long long)(ub + lane_off)) : 0.0;
        }
#pragma unroll
        for (int b = 0; b < kCtLoad; b += U) {
          double r[U];
#pragma unroll
#ifdef M355_CT_COPY  // (probe: the access pattern alone)
          for (int u = 0; u < U; ++u) r[u] = v[b + u] + wlo;
#else
          for (int u = 0; u < U; ++u) r[u] = lookup(v[b + u]);
#endif
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const long long ub = ubase + (long long)((b + u) * kCtRows) * (long long)P;
            const bool live = (rloc < rows_left - (b + u) * kCtRows) & !declined;
            const bool hit = r[u] == r[u];
            if constexpr (F64IO) {
              if (live & hit) __builtin_nontemporal_store(r[u], (double*)out + ub + lane_off);
            } else {
              if (live & hit) store1(out, odt, ub + lane_off, r[u]);
            }
            missed |= (unsigned long long)(live & !hit) << (it * kCtLoad + b + u);
          }
        }
      }
      miss_words[((size_t)g * chunks + ch) * kCtThreads + t] = missed;
    }
  }
}

// The misses, redone by the exact body: word w = (group, chunk, thread) of k_coltab, bit = pass * unroll + slot.  x is read
// again (k_coltab stored nothing where it missed).
template <int D, int FN>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(min_waves<Dist<D>>::value))) void k_coltab_fix(
    const void* __restrict__ x, int xdt, void* __restrict__ out, int odt, long long R, unsigned P, PeriodicParams pa,
    const unsigned long long* __restrict__ miss_words, unsigned long long nwords) {
  using DD = Dist<D>;
  using K = typename DD::K;
  log_table_init();
  const unsigned chunks = (unsigned)((R + ct_chunk_rows<FN>() - 1) / ct_chunk_rows<FN>());
  for (unsigned long long w = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; w < nwords; w += (unsigned long long)gridDim.x * kBlock) {
    unsigned long long mask = miss_words[w];
    if (mask == 0) continue;
    const unsigned t = (unsigned)(w % kCtThreads);
    const unsigned long long gc = w / kCtThreads;
    const unsigned ch = (unsigned)(gc % chunks), g = (unsigned)(gc / chunks);
    const unsigned col = g * kCtCols + (t & (kCtCols - 1)), rloc = t >> 4;
    double pp[kMaxParams];
#pragma unroll
    for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][col] : pa.scalar[q];
    const K k = DD::derive(pp);
    const long long row0 = (long long)ch * ct_chunk_rows<FN>();
    while (mask) {
      const int b = __builtin_ctzll(mask);
      mask &= mask - 1;
      const long long row = row0 + (long long)b * kCtRows + rloc;  // bit b = pass * kCtLoad + slot: rows row0 + 64 b + rloc
      const long long idx = row * (long long)P + col;
      store1(out, odt, idx, tab_miss_eval<DD, FN>(load1(x, xdt, (unsigned long long)idx), k));  // (a column with a table: tab_ok)
    }
  }
}

// Columns without a table (Dist<>::tab_ok / tab_ok_fn declined the parameters: flag in the column's table header) are
// left untouched by k_coltab and done by this launch: P x kCtDeclChunks workgroups, each returning at once unless its
// column was declined; a declined column's rows then go through the exact body, parameters wavefront-uniform, spread
// over kCtDeclChunks workgroups.
constexpr int kCtDeclChunks = 16;
template <int D, int FN>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(min_waves<Dist<D>>::value))) void k_coltab_declined(
    const void* __restrict__ x, int xdt, void* __restrict__ out, int odt, long long R, unsigned P, PeriodicParams pa,
    const double* __restrict__ ws) {
  using DD = Dist<D>;
  using K = typename DD::K;
  constexpr int NK = sizeof(K) / sizeof(double);
  const unsigned c = blockIdx.x % P, chunk = blockIdx.x / P;
  if (ws[(size_t)c * kCtTable + 2] == 0.0) return;  // k_coltab has this column
  double pp[kMaxParams];
#pragma unroll
  for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][c] : pa.scalar[q];
  const K k0 = DD::derive(pp);  // (every thread: the same values, no barrier)
  log_table_init();
  K k;
  {
    const double* kd = reinterpret_cast<const double*>(&k0);
    double* dst = reinterpret_cast<double*>(&k);
#pragma unroll
    for (int f = 0; f < NK; ++f) dst[f] = uniform_d(kd[f]);
  }
  const long long per = (R + kCtDeclChunks - 1) / kCtDeclChunks;
  const long long r1 = (chunk + 1) * per < R ? (chunk + 1) * per : R;
  for (long long row = chunk * per + threadIdx.x; row < r1; row += kBlock) {
    const long long idx = row * (long long)P + c;
    store1(out, odt, idx, DD::template eval<FN>(load1(x, xdt, (unsigned long long)idx), k));
  }
}

// k_coltab needs 115 KiB of dynamic LDS: above the 64 KiB a kernel gets without asking.  The attribute is per device and
// kernel: asked once for each (remembered per instantiation and device ordinal), not per launch -- hipFuncSetAttribute is
// not a stream operation, so a first call under stream capture is the one that must happen before the capture.
constexpr int kMaxDevicesTracked = 64;
template <int D, int FN, bool F64IO> bool coltab_ready() {
  static std::atomic<unsigned char> state[kMaxDevicesTracked];  // 0 = not asked, 1 = granted, 2 = refused
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevicesTracked) return false;
  unsigned char st = state[dev].load(std::memory_order_acquire);
  if (st == 0) {
    st = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coltab<D, FN, F64IO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)kCtLds) == hipSuccess ? 1 : 2;
    if (st == 2) (void)hipGetLastError();
    state[dev].store(st, std::memory_order_release);
  }
  return st == 1;
}

// ---- Discrete families, scalar parameters, large n: the function of an integer, looked up ------------------------------
// cdf / sf / logcdf / logsf of a count are functions of an integer that lives, for all practical data, in a window of a
// few hundred to two thousand values around the mean; the exact bodies cost 500-5000 operations each (gammainc,
// betainc, sums of pmf terms).  k_dtab has every workgroup evaluate the exact body once per integer of the window
// [lo, lo + len) that Dist<>::dtab_range proposes (host side, from mean and standard deviation: generous, clipped to the
// support, at most kDTabMax values) and then answers integer-valued arguments inside the window from LDS; everything
// else -- outside the window, non-integers, NaN -- takes the exact body.  The looked-up value IS the exact body's value
// for that argument: results are bit-identical to the other kernels'.
template <class T, class = void> struct has_dtab : std::false_type {};
template <class T> struct has_dtab<T, std::void_t<decltype(&T::dtab_range)>> : std::true_type {};
template <class DD, int FN> constexpr bool dtab_this() {
  return has_dtab<DD>::value && (FN == M355_LOGCDF || FN == M355_LOGSF || FN == M355_CDF || FN == M355_SF);
}
template <int D, int FN>
__global__ __launch_bounds__(kBlock) void k_dtab(const void* __restrict__ x, int xdt, void* __restrict__ out, int odt,
                                                 long long n, const typename Dist<D>::K k, double lo, int len) {
  using DD = Dist<D>;
  __shared__ double tab[kDTabMax];
  if constexpr (uses_log_table<DD>::value) log_table_init();
  for (int j = threadIdx.x; j < len; j += kBlock) tab[j] = DD::template eval<FN>(lo + (double)j, k);
  __syncthreads();
  const double flen = (double)len;
  auto lookup = [&](double xv, bool& miss) -> double {
    const double j = xv - lo;
    const bool hit = (j >= 0.0) & (j < flen) & (__builtin_floor(xv) == xv);
    miss |= !hit;
    return tab[hit ? (int)j : 0];
  };
  const long long ntiles = n / kTile;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kTile + threadIdx.x * kVec;
    double v[kUnroll][kVec], r[kUnroll][kVec];
    bool ms[kUnroll][kVec];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) load2<0>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
    bool miss = false;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
      for (int q = 0; q < kVec; ++q) {
        ms[u][q] = false;
        r[u][q] = lookup(v[u][q], ms[u][q]);
        miss |= ms[u][q];
      }
    }
    if (miss) {
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
        for (int q = 0; q < kVec; ++q) {
          if (ms[u][q]) r[u][q] = tab_exact<D, FN>(v[u][q], k);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) store2<0>(out, odt, base + u * (kBlock * kVec), r[u][0], r[u][1]);
  }
  for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const double xv = load1(x, xdt, i);
    bool m = false;
    double rv = lookup(xv, m);
    if (m) rv = tab_exact<D, FN>(xv, k);
    store1(out, odt, i, rv);
  }
}

// Quantiles of the same families: find_ppf_discrete (optimization.py:102-129) itself, step for step, with the cdf and
// pmf values it asks for taken from two tables over the window (exact bodies outside it).  The values are the exact
// bodies' values, the iteration is the same code (Dist<>::ppf_with), so every result -- the cycles and the iterates the
// loop stops at after 100 rounds included -- is bit-identical to the other kernels'; a step costs three LDS reads
// instead of two incomplete gamma / beta functions and a pmf.
template <int D> __device__ __noinline__ double dtab_cdf_exact(double v, const typename Dist<D>::K& k) { return Dist<D>::dtab_cdf(v, k); }
template <int D> __device__ __noinline__ double dtab_pmf_exact(double v, const typename Dist<D>::K& k) { return Dist<D>::dtab_pmf(v, k); }

// (512 threads, two workgroups per CU: every workgroup evaluates the whole window -- up to 2 x 2048 incomplete beta /
// gamma functions -- so fewer, larger workgroups; the 1280 x 256 of the first version spent 0.3 of binomial(1000, 0.3)'s
// 1.9 ms per 1e8 there once betainc's prefactor got its cancellation-free form)
constexpr int kDtabPpfBlock = 512;
template <int D, int FN>
__global__ __launch_bounds__(kDtabPpfBlock) void k_dtab_ppf(const void* __restrict__ x, int xdt, void* __restrict__ out, int odt,
                                                            long long n, const typename Dist<D>::K k, double lo, int len) {
  using DD = Dist<D>;
  constexpr int kBlock = kDtabPpfBlock;  // (shadows the file's 256 inside this kernel)
  __shared__ double tc[kDTabMax], tp[kDTabMax];
  if constexpr (uses_log_table<DD>::value) log_table_init();
  for (int j = threadIdx.x; j < len; j += kBlock) {
    tc[j] = DD::dtab_cdf(lo + (double)j, k);
    tp[j] = DD::dtab_pmf(lo + (double)j, k);
  }
  __syncthreads();
  const double flen = (double)len;
  auto cdf = [&](double v) -> double {
    const double j = v - lo;
    if ((j >= 0.0) & (j < flen) & (__builtin_floor(v) == v)) return tc[(int)j];
    return dtab_cdf_exact<D>(v, k);
  };
  auto pmf = [&](double v) -> double {
    const double j = v - lo;
    if ((j >= 0.0) & (j < flen) & (__builtin_floor(v) == v)) return tp[(int)j];
    return dtab_pmf_exact<D>(v, k);
  };
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const double xv = load1(x, xdt, i);
    const double q = (FN == M355_ISF) ? 1.0 - xv : xv;
    store1(out, odt, i, DD::ppf_with(q, k, cdf, pmf));
  }
}

// The same lookup with row-periodic parameters (x of shape (R, P), parameters (P,)): one window per column, built ONCE per
// call into the caller's workspace (m355_elemwise_ws) and read back through L2 by k_dcol.  Workspace (doubles):
// [0] the longest window of the call (an int, atomicMax: the tables' common stride, so that they stay as compact as the
// data allows -- 1024 windows of ~300 values are 2.4 MB, L2-resident, where a fixed stride of 2048 made them 16 MB);
// [8 + 2p] lo, [8 + 2p + 1] len of column p (-1: no table: the window does not fit, or the parameters are out of domain);
// then tab[p * stride + j] = the exact body at lo + j.  A thread of k_dcol meets the same kUnroll * kVec columns in every
// tile (the launcher makes gridDim.x * kTile a multiple of P, as for k_periodic_fixed), keeps their windows in registers,
// and a lookup is one 8-byte gather.  The looked-up value IS the exact body's value: bit-identical to k_periodic.
constexpr int kDColHdr = 8;
template <int D>
__global__ __launch_bounds__(kBlock) void k_dcol_ranges(PeriodicParams pa, unsigned P, unsigned max_stride, double* __restrict__ ws) {
  using DD = Dist<D>;
  const unsigned p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= P) return;
  double pp[kMaxParams];
#pragma unroll
  for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][p] : pa.scalar[q];
  const typename DD::K k = DD::derive(pp);
  double lo = 0.0;
  int len = 0;
  const bool ok = DD::dtab_range(k, lo, len) && len <= (int)max_stride;
  ws[kDColHdr + 2 * (size_t)p] = ok ? lo : 0.0;
  ws[kDColHdr + 2 * (size_t)p + 1] = ok ? (double)len : -1.0;
  if (ok) atomicMax(reinterpret_cast<int*>(ws), len);
}
template <int D, int FN>
__global__ __launch_bounds__(kBlock) void k_dcol_build(PeriodicParams pa, unsigned P, unsigned max_stride, double* __restrict__ ws) {
  using DD = Dist<D>;
  const unsigned chunks = (max_stride + kBlock - 1) / kBlock;
  const unsigned p = blockIdx.x / chunks, ch = blockIdx.x % chunks;
  const int len = (int)ws[kDColHdr + 2 * (size_t)p + 1];
  if ((int)(ch * kBlock) >= len) return;  // (workgroup-uniform)
  if constexpr (uses_log_table<DD>::value) log_table_init();
  const unsigned stride = (unsigned)*reinterpret_cast<const int*>(ws);
  const double lo = ws[kDColHdr + 2 * (size_t)p];
  double pp[kMaxParams];
#pragma unroll
  for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][p] : pa.scalar[q];
  const typename DD::K k = DD::derive(pp);
  const int j = (int)(ch * kBlock + threadIdx.x);
  if (j < len) ws[kDColHdr + 2 * (size_t)P + (size_t)p * stride + j] = DD::template eval<FN>(lo + (double)j, k);
}
template <int D, int FN>
__device__ __noinline__ double dcol_exact(double x, const PeriodicParams& pa, unsigned col) {
  using DD = Dist<D>;
  double pp[kMaxParams];
#pragma unroll
  for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][col] : pa.scalar[q];
  return DD::template eval<FN>(x, DD::derive(pp));
}
template <int D, int FN>
__global__ __launch_bounds__(kBlock) void k_dcol(const void* __restrict__ x, int xdt, void* __restrict__ out, int odt, long long n,
                                                 unsigned P, PeriodicParams pa, const double* __restrict__ ws) {
  using DD = Dist<D>;
  if constexpr (uses_log_table<DD>::value) log_table_init();
  const unsigned stride = (unsigned)*reinterpret_cast<const int*>(ws);
  const double* hdr = ws + kDColHdr;
  const double* tab = hdr + 2 * (size_t)P;
  const long long ntiles = n / kTile;
  double clo[kUnroll][kVec], clen[kUnroll][kVec];
  unsigned ccol[kUnroll][kVec];
  {
    unsigned idx = (unsigned)(((unsigned long long)blockIdx.x * kTile + threadIdx.x * kVec) % P);
    const unsigned u_step = (unsigned)((kBlock * kVec) % P);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      unsigned idx1 = idx + 1;
      idx1 = idx1 >= P ? idx1 - P : idx1;
      ccol[u][0] = idx;
      ccol[u][1] = idx1;
#pragma unroll
      for (int q = 0; q < kVec; ++q) {
        clo[u][q] = hdr[2 * (size_t)ccol[u][q]];
        clen[u][q] = hdr[2 * (size_t)ccol[u][q] + 1];
      }
      idx += u_step;
      idx = idx >= P ? idx - P : idx;
    }
  }
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long base = tile * kTile + threadIdx.x * kVec;
    double v[kUnroll][kVec], r[kUnroll][kVec];
    bool ms[kUnroll][kVec];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) load2<0>(x, xdt, base + u * (kBlock * kVec), v[u][0], v[u][1]);
    bool miss = false;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
      for (int q = 0; q < kVec; ++q) {
        const double j = v[u][q] - clo[u][q];
        const bool hit = (j >= 0.0) & (j < clen[u][q]) & (__builtin_floor(v[u][q]) == v[u][q]);
        ms[u][q] = !hit;
        miss |= !hit;
        r[u][q] = tab[(size_t)ccol[u][q] * stride + (hit ? (unsigned)j : 0u)];
      }
    }
    if (miss) {
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
        for (int q = 0; q < kVec; ++q) {
          if (ms[u][q]) r[u][q] = dcol_exact<D, FN>(v[u][q], pa, ccol[u][q]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) store2<0>(out, odt, base + u * (kBlock * kVec), r[u][0], r[u][1]);
  }
  for (long long i = ntiles * kTile + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    store1(out, odt, i, dcol_exact<D, FN>(load1(x, xdt, i), pa, (unsigned)((unsigned long long)i % P)));
}
// Column groups with the windows in LDS (P a multiple of 16): k_dcol's gathers miss the L1 -- a workgroup there meets all P
// columns -- so every element pulled a 128-byte line from L2.  k_dcolg gives a workgroup of 1024 threads 16 adjacent
// columns (a row's 16 values are one 128-byte line) and a slice of the rows, copies the 16 windows to LDS when the
// call's longest window fits kDcgS entries (otherwise it reads them from the workspace as k_dcol does) and looks up
// there.  Quantiles (PPF / ISF): find_ppf_discrete itself on two windows per column (cdf and pmf), as k_dtab_ppf does
// with scalar parameters -- the same iterates, bit for bit.
constexpr int kDcgCols = 16, kDcgThreads = 1024, kDcgS = 512, kDcgRowsPerTrip = 4;
template <int FN> constexpr size_t dcolg_lds_bytes() {
  return (size_t)kDcgCols * kDcgS * sizeof(double) * ((FN == M355_PPF || FN == M355_ISF) ? 2 : 1);
}
template <int D>
__global__ __launch_bounds__(kBlock) void k_dcol_build_q(PeriodicParams pa, unsigned P, unsigned max_stride, double* __restrict__ ws) {
  using DD = Dist<D>;
  const unsigned chunks = (max_stride + kBlock - 1) / kBlock;
  const unsigned p = blockIdx.x / chunks, ch = blockIdx.x % chunks;
  const int len = (int)ws[kDColHdr + 2 * (size_t)p + 1];
  if ((int)(ch * kBlock) >= len) return;  // (workgroup-uniform)
  if constexpr (uses_log_table<DD>::value) log_table_init();
  const unsigned stride = (unsigned)*reinterpret_cast<const int*>(ws);
  const double lo = ws[kDColHdr + 2 * (size_t)p];
  double pp[kMaxParams];
#pragma unroll
  for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][p] : pa.scalar[q];
  const typename DD::K k = DD::derive(pp);
  const int j = (int)(ch * kBlock + threadIdx.x);
  if (j < len) {
    double* tab = ws + kDColHdr + 2 * (size_t)P;
    tab[(size_t)p * stride + j] = DD::dtab_cdf(lo + (double)j, k);
    tab[(size_t)P * stride + (size_t)p * stride + j] = DD::dtab_pmf(lo + (double)j, k);
  }
}
template <int D, int FN>
__global__ __launch_bounds__(kDcgThreads) void k_dcolg(const void* __restrict__ x, int xdt, void* __restrict__ out, int odt, long long R,
                                                       unsigned P, unsigned rows_per_wg, PeriodicParams pa,
                                                       const double* __restrict__ ws) {
  using DD = Dist<D>;
  constexpr bool Q = FN == M355_PPF || FN == M355_ISF;
  extern __shared__ __attribute__((aligned(16))) double dcg_lds[];
  if constexpr (uses_log_table<DD>::value) log_table_init();
  const unsigned stride = (unsigned)*reinterpret_cast<const int*>(ws);
  const double* hdr = ws + kDColHdr;
  const double* tab = hdr + 2 * (size_t)P;
  const double* tabp = tab + (size_t)P * stride;  // (quantiles: the pmf windows)
  const bool in_lds = stride <= (unsigned)kDcgS;  // workgroup-uniform
  const unsigned G = P / kDcgCols, g = blockIdx.x % G, chunk = blockIdx.x / G;
  const unsigned col0 = g * kDcgCols, c = threadIdx.x & (kDcgCols - 1), rl = threadIdx.x >> 4;
  if (in_lds) {
    for (unsigned e = threadIdx.x; e < kDcgCols * stride; e += kDcgThreads) {
      const unsigned cc = e / stride, j = e - cc * stride;
      dcg_lds[cc * kDcgS + j] = tab[(size_t)(col0 + cc) * stride + j];
      if constexpr (Q) dcg_lds[(kDcgCols + cc) * kDcgS + j] = tabp[(size_t)(col0 + cc) * stride + j];
    }
  }
  __syncthreads();
  const unsigned col = col0 + c;
  const double lo = hdr[2 * (size_t)col], flen = hdr[2 * (size_t)col + 1];
  const long long r0 = (long long)chunk * rows_per_wg;
  const long long r1 = r0 + rows_per_wg < R ? r0 + rows_per_wg : R;
  constexpr int kRowLanes = kDcgThreads / kDcgCols;  // 64
  auto rows = [&](auto cdf_at, auto pmf_at) {  // cdf_at(j) / pmf_at(j): the windows of this thread's column
    if constexpr (!Q) {
      for (long long r = r0 + rl; r < r1; r += kRowLanes * kDcgRowsPerTrip) {
        double v[kDcgRowsPerTrip], res[kDcgRowsPerTrip];
        bool ms[kDcgRowsPerTrip];
#pragma unroll
        for (int u = 0; u < kDcgRowsPerTrip; ++u) {
          const long long rr = r + u * kRowLanes;
          v[u] = rr < r1 ? load1(x, xdt, rr * (long long)P + col) : lo;
        }
        bool miss = false;
#pragma unroll
        for (int u = 0; u < kDcgRowsPerTrip; ++u) {
          const double j = v[u] - lo;
          const bool hit = (j >= 0.0) & (j < flen) & (__builtin_floor(v[u]) == v[u]);
          ms[u] = !hit & (r + u * kRowLanes < r1);
          miss |= ms[u];
          res[u] = cdf_at(hit ? (unsigned)j : 0u);
        }
        if (miss) {
#pragma unroll
          for (int u = 0; u < kDcgRowsPerTrip; ++u) {
            if (ms[u]) res[u] = dcol_exact<D, FN>(v[u], pa, col);
          }
        }
#pragma unroll
        for (int u = 0; u < kDcgRowsPerTrip; ++u) {
          const long long rr = r + u * kRowLanes;
          if (rr < r1) store1(out, odt, rr * (long long)P + col, res[u]);
        }
      }
    } else {
      double pp[kMaxParams];
#pragma unroll
      for (int q = 0; q < kMaxParams; ++q) pp[q] = (q < DD::NP && pa.ptr[q]) ? pa.ptr[q][col] : pa.scalar[q];
      const typename DD::K k = DD::derive(pp);
      auto cdf = [&](double v) -> double {
        const double j = v - lo;
        if ((j >= 0.0) & (j < flen) & (__builtin_floor(v) == v)) return cdf_at((unsigned)j);
        return dtab_cdf_exact<D>(v, k);
      };
      auto pmf = [&](double v) -> double {
        const double j = v - lo;
        if ((j >= 0.0) & (j < flen) & (__builtin_floor(v) == v)) return pmf_at((unsigned)j);
        return dtab_pmf_exact<D>(v, k);
      };
      for (long long r = r0 + rl; r < r1; r += kRowLanes) {
        const double xv = load1(x, xdt, r * (long long)P + col);
        const double q = (FN == M355_ISF) ? 1.0 - xv : xv;
        store1(out, odt, r * (long long)P + col, DD::ppf_with(q, k, cdf, pmf));
      }
    }
  };
  if (in_lds) {
    const double* tc = dcg_lds + c * kDcgS;
    const double* tp = dcg_lds + (kDcgCols + c) * kDcgS;
    rows([&](unsigned j) { return tc[j]; }, [&](unsigned j) { return tp[j]; });
  } else {
    const double* tc = tab + (size_t)col * stride;
    const double* tp = tabp + (size_t)col * stride;
    rows([&](unsigned j) { return tc[j]; }, [&](unsigned j) { return tp[j]; });
  }
}
template <int D, int FN> bool dcolg_ready() {  // (as coltab_ready: the LDS attribute once per device)
  static std::atomic<unsigned char> state[kMaxDevicesTracked];
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevicesTracked) return false;
  unsigned char st = state[dev].load(std::memory_order_acquire);
  if (st == 0) {
    st = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dcolg<D, FN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)dcolg_lds_bytes<FN>()) == hipSuccess ? 1 : 2;
    if (st == 2) (void)hipGetLastError();
    state[dev].store(st, std::memory_order_release);
  }
  return st == 1;
}
constexpr unsigned kDColMinStride = 64;  // a workspace that cannot hold this many values per column is not used
inline unsigned dcol_stride(size_t ws_doubles, unsigned long long P, int tables = 1) {  // the longest window the workspace can hold for every column
  if (ws_doubles <= 2 * P + kDColHdr) return 0;
  unsigned long long s = (ws_doubles - 2 * P - kDColHdr) / (P * (unsigned long long)tables);
  s = s > (unsigned long long)kDTabMax ? (unsigned long long)kDTabMax : s;
  return (unsigned)(s & ~7ull);
}

struct GenericOps {
  DevOperand x;
  DevOperand p[kMaxParams];
};

M355_DEV double fetch_operand(const DevOperand& o, unsigned long long i, bool small) {
  if (o.ptr == nullptr) return o.scalar;
  unsigned long long idx = i;
  if (small) {
    unsigned j = (unsigned)i;
    if (o.div != 1) j /= (unsigned)o.div;
    if (o.mod <= 0xffffffffull && j >= (unsigned)o.mod) j %= (unsigned)o.mod;
    idx = j;
  } else {
    if (o.div != 1) idx /= o.div;
    if (idx >= o.mod) idx %= o.mod;
  }
  return load1(o.ptr, o.dtype, idx);
}

template <int D, int FN, bool REDUCE>
__global__ __launch_bounds__(kBlock) void k_generic(GenericOps ops, void* __restrict__ out, int odt,
                                                    long long n, double* __restrict__ partials) {
  using DD = Dist<D>;
  if constexpr (uses_log_table<DD>::value) log_table_init();
  const bool small = n <= 0xffffffffll;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const double xv = fetch_operand(ops.x, (unsigned long long)i, small);
    double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < DD::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)i, small);
    const double r = DD::template eval<FN>(xv, DD::derive(pp));
    if constexpr (REDUCE) acc += r;
    else store1(out, odt, i, r);
  }
  if constexpr (REDUCE) {
    acc = block_reduce_sum(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
  }
}

// Dist<D>::kDeviceDerive: derive() needs device-only functions, so scalar parameters go through k_periodic
// with a one-entry table (derived once per workgroup) instead of being derived on the host.
template <class T, class = void> struct device_derive : std::false_type {};
template <class T> struct device_derive<T, std::void_t<decltype(T::kDeviceDerive)>> : std::true_type {};

// k_tab's launch: five workgroups per CU (24 KiB of LDS each), from kTabMinN elements (below, the table's 3584 exact
// evaluations and its second launch cost more than evaluating the elements exactly); m355_set_tab_min_n() moves the
// threshold (tests force the table onto small inputs; a huge value switches it off).  Without a workspace
// (m355_elemwise instead of m355_elemwise_ws) there is nowhere to build a table: the exact kernels serve.
constexpr int kTabBlocks = 1280;
constexpr long long kTabMinN = 1ll << 21;
// (an atomic: m355_set_tab_min_n may run beside launches on other threads.  The environment knobs of this file --
// M355_TAB_MIN_N, M355_TAB_BLOCKS, M355_NO_BINNING -- are for A/B measurements; include/m355_dist.h lists them.)
inline std::atomic<long long>& tab_min_n_ref() {
  static std::atomic<long long> v{[] {
    const char* e = getenv("M355_TAB_MIN_N");
    const long long t = e ? atoll(e) : kTabMinN;
    return t < 0 ? kTabMinN : t;
  }()};
  return v;
}
inline long long tab_min_n() { return tab_min_n_ref().load(std::memory_order_relaxed); }

// M355_NO_BINNING=1 in the environment sends the divergent functions through k_scalar (A/B measurements)
inline bool binning_disabled() {
  static const bool off = [] { const char* e = getenv("M355_NO_BINNING"); return e && e[0] == '1'; }();
  return off;
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// Host-side selection of the kernel for one (dist, fn); returns a hipError_t as int.
template <int D, int FN, bool REDUCE> int launch_fn(const LaunchArgs& a) {
  using DD = Dist<D>;
  const long long n = a.n;
  const bool x_full = a.x.ptr != nullptr && a.x.div == 1 && a.x.mod >= (unsigned long long)n;
  const size_t xal = a.x.dtype == M355_F32 ? 8 : 16, oal = a.out_dtype == M355_F32 ? 8 : 16;
  const bool vec_ok = x_full && aligned(a.x.ptr, xal) && (REDUCE || aligned(a.out, oal));
  // compile-time f64 load/store path for the headline functions; everything else (and f32 /
  // int64 data) goes through the same kernels with a wave-uniform dtype switch per access
  constexpr bool kFastIO = FN <= M355_PPF;
  const bool f64io = kFastIO && a.x.dtype == M355_F64 && (REDUCE || a.out_dtype == M355_F64);
  bool all_scalar = true, periodic_ok = true;
  unsigned long long P = 0;
  for (int q = 0; q < DD::NP; ++q) {
    const DevOperand& o = a.p[q];
    if (o.ptr == nullptr) continue;
    all_scalar = false;
    if (o.div != 1 || o.dtype != M355_F64 || (P != 0 && o.mod != P)) periodic_ok = false;
    P = o.mod;
  }
  periodic_ok = periodic_ok && !all_scalar && P >= 1 && P < (unsigned long long)n &&
                P * sizeof(typename DD::K) <= (size_t)kMaxPeriodBytes;
  const long long tiles = n / kTile;
  const long long cap = REDUCE ? kMaxReduceBlocks : kMaxBlocks;
  int blocks;
  if (vec_ok && (all_scalar || periodic_ok)) {
    long long c = cap;
    if (!all_scalar) {
      // k_periodic: every workgroup derives the whole table (P x lgamma / log ...) before its first tile, so the grid
      // is two rounds of what is resident (the table's LDS bounds that), not one workgroup per three tiles:
      // at P = 1024 the 32768-block grid spent more on deriving tables than on evaluating elements
      const size_t lds_b = (size_t)P * sizeof(typename DD::K) + 8192;
      long long per_cu = (long long)(163840 / lds_b);
      per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
      c = 2 * 256 * per_cu < c ? 2 * 256 * per_cu : c;
    }
    blocks = (int)(tiles < 1 ? 1 : (tiles > c ? c : tiles));
    if (REDUCE && !all_scalar) {
      // k_periodic's product-form sum needs gridDim.x * kTile to be a multiple of P (a thread then keeps its columns):
      // round the grid down to a multiple of P / gcd(P, kTile) where that leaves a grid at all
      unsigned long long a_ = P, b_ = (unsigned long long)kTile;
      while (b_) { const unsigned long long t_ = a_ % b_; a_ = b_; b_ = t_; }
      const long long unit = (long long)(P / a_);
      if (unit <= blocks) blocks = (int)((blocks / unit) * unit);
    }
  } else {
    const long long want = (n + kBlock - 1) / kBlock;
    blocks = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  }
  if (REDUCE && a.nblocks_out) *a.nblocks_out = blocks;
  constexpr bool kDevDerive = device_derive<DD>::value;
  if (kDevDerive && all_scalar) P = 1;
  if (vec_ok && all_scalar && !kDevDerive) {
    if constexpr (!kDevDerive) {
      // parameter-only constants once, on the host; they reach the kernel as arguments (SGPRs)
      const double pp[kMaxParams] = {a.p[0].scalar, a.p[1].scalar, a.p[2].scalar, a.p[3].scalar};
      const typename DD::K k = DD::derive(pp);
      if constexpr (REDUCE && tab_this<DD, FN>()) {
        // fused sum of logcdf / logsf: the table in what the caller's workspace holds beyond the partials
        bool tab_ok = DD::tab_ok(k);
        if constexpr (has_tab_ok_fn<DD>::value) tab_ok = tab_ok && DD::template tab_ok_fn<FN>(k);
        if (n >= tab_min_n() && tab_ok && !binning_disabled() && a.table != nullptr && a.table_doubles >= (long long)kTabDoubles) {
          hipLaunchKernelGGL((k_tab_build<D, FN>), dim3(kTabBuildWaves), dim3(64), 0, a.stream, k, a.table);
          const int bb = (int)(tiles < 1 ? 1 : (tiles > kTabBlocks ? kTabBlocks : tiles));
          if (a.nblocks_out) *a.nblocks_out = bb;
          if (a.x.dtype == M355_F64)
            hipLaunchKernelGGL((k_tab<D, FN, true, true>), dim3(bb), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype, nullptr, M355_F64, n, k,
                               a.table, a.partials);
          else
            hipLaunchKernelGGL((k_tab<D, FN, false, true>), dim3(bb), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype, nullptr, M355_F64, n, k,
                               a.table, a.partials);
          return (int)hipGetLastError();
        }
      }
      if constexpr (!REDUCE && tab_this<DD, FN>()) {
        bool tab_ok = DD::tab_ok(k);
        if constexpr (has_tab_ok_fn<DD>::value) tab_ok = tab_ok && DD::template tab_ok_fn<FN>(k);
        if (n >= tab_min_n() && tab_ok && !binning_disabled() && a.ews != nullptr && a.ews_doubles >= (size_t)kTabDoubles) {
          hipLaunchKernelGGL((k_tab_build<D, FN>), dim3(kTabBuildWaves), dim3(64), 0, a.stream, k, a.ews);
          const long long want = tiles < 1 ? 1 : tiles;
          static const int tb = [] {
            const char* e = getenv("M355_TAB_BLOCKS");
            const int t = e ? atoi(e) : kTabBlocks;
            return t < 1 ? 1 : (t > kMaxBlocks ? kMaxBlocks : t);
          }();
          const int bb = (int)(want > tb ? tb : want);
          if (a.x.dtype == M355_F64 && a.out_dtype == M355_F64) {  // (cdf / sf too: kFastIO is k_scalar's economy)
            hipLaunchKernelGGL((k_tab<D, FN, true, false>), dim3(bb), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype, a.out,
                               a.out_dtype, n, k, a.ews, nullptr);
          } else {
            hipLaunchKernelGGL((k_tab<D, FN, false, false>), dim3(bb), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype, a.out,
                               a.out_dtype, n, k, a.ews, nullptr);
          }
          return (int)hipGetLastError();
        }
      }
      if constexpr (!REDUCE && has_dtab<DD>::value && (FN == M355_PPF || FN == M355_ISF)) {
        double lo = 0.0;
        int len = 0;
        if (n >= tab_min_n() && !binning_disabled() && DD::dtab_range(k, lo, len)) {
          const long long want = (n + kDtabPpfBlock - 1) / kDtabPpfBlock;
          const int bb = (int)(want > 2 * 256 ? 2 * 256 : want);  // 32 KiB of tables, 16 wavefronts per CU by registers
          hipLaunchKernelGGL((k_dtab_ppf<D, FN>), dim3(bb), dim3(kDtabPpfBlock), 0, a.stream, a.x.ptr, a.x.dtype, a.out, a.out_dtype,
                             n, k, lo, len);
          return (int)hipGetLastError();
        }
      }
      if constexpr (!REDUCE && dtab_this<DD, FN>()) {
        double lo = 0.0;
        int len = 0;
        if (n >= tab_min_n() && !binning_disabled() && DD::dtab_range(k, lo, len)) {
          const long long want = tiles < 1 ? 1 : tiles;
          const int bb = (int)(want > 2 * kTabBlocks ? 2 * kTabBlocks : want);
          hipLaunchKernelGGL((k_dtab<D, FN>), dim3(bb), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype, a.out, a.out_dtype, n, k,
                             lo, len);
          return (int)hipGetLastError();
        }
      }
      if constexpr (!REDUCE && bin_this<DD, FN>()) {
        if (n >= (long long)kBinMinTiles * kBinTile && !binning_disabled()) {
          const long long bt = n / kBinTile;
          const int bb = (int)(bt > kMaxBlocks ? kMaxBlocks : bt);
          hipLaunchKernelGGL((k_binned<D, FN>), dim3(bb), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype, a.out,
                             a.out_dtype, n, k);
          return (int)hipGetLastError();
        }
      }
      if constexpr (FN == M355_LOGPDF) {
        if (a.x.dtype == M355_F32 && (REDUCE || a.out_dtype == M355_F32)) {
          hipLaunchKernelGGL((k_scalar_f32<D, REDUCE>), dim3(blocks), dim3(kBlock), 0, a.stream, a.x.ptr, a.out, n, k,
                             a.partials);
          return (int)hipGetLastError();
        }
      }
      if (f64io) {
        if constexpr (kFastIO)
          hipLaunchKernelGGL((k_scalar<D, FN, REDUCE, true>), dim3(blocks), dim3(kBlock), 0, a.stream,
                             a.x.ptr, a.x.dtype, a.out, a.out_dtype, n, k, a.partials);
      } else {
        hipLaunchKernelGGL((k_scalar<D, FN, REDUCE, false>), dim3(blocks), dim3(kBlock), 0, a.stream, a.x.ptr,
                           a.x.dtype, a.out, a.out_dtype, n, k, a.partials);
      }
    }
  } else if (vec_ok && (periodic_ok || all_scalar)) {
    PeriodicParams pa;
    for (int q = 0; q < kMaxParams; ++q) {
      pa.ptr[q] = (q < DD::NP) ? static_cast<const double*>(a.p[q].ptr) : nullptr;
      pa.scalar[q] = a.p[q].scalar;
    }
    if constexpr (REDUCE && FN == M355_LOGPDF && has_sum_acc<DD>::value) {
      // table-free product-form kernel when the workspace can hold the derived constants and the grid can be a
      // multiple of P / gcd(P, kTile)
      constexpr long long NK = sizeof(typename DD::K) / sizeof(double);
      unsigned long long a_ = P, b_ = (unsigned long long)kTile;
      while (b_) { const unsigned long long t_ = a_ % b_; a_ = b_; b_ = t_; }
      const long long unit = (long long)(P / a_);
      long long pb = tiles > kMaxReduceBlocks ? kMaxReduceBlocks : tiles;
      pb = (pb / unit) * unit;
      if (!all_scalar && a.table != nullptr && (long long)P * NK <= a.table_doubles && pb >= 1) {
        hipLaunchKernelGGL((k_derive_table<D>), dim3((unsigned)((P + kBlock - 1) / kBlock)), dim3(kBlock), 0, a.stream, pa,
                           (unsigned)P, a.table);
        if (a.nblocks_out) *a.nblocks_out = (int)pb;
        if (f64io)
          hipLaunchKernelGGL((k_periodic_prod<D, true>), dim3((unsigned)pb), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype, n,
                             (unsigned)P, a.table, a.partials);
        else
          hipLaunchKernelGGL((k_periodic_prod<D, false>), dim3((unsigned)pb), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype, n,
                             (unsigned)P, a.table, a.partials);
        return (int)hipGetLastError();
      }
    }
    const void* xp = a.x.ptr;
    void* op = a.out;
    long long nn = n;
    if constexpr (!REDUCE && has_dtab<DD>::value && FN >= M355_LOGCDF && FN != M355_PDF) {
      // discrete families, row-periodic parameters, large n: one window of exact values per column in the workspace
      // (quantiles: two, cdf and pmf); column groups with the windows in LDS when P is a multiple of 16 (k_dcolg)
      constexpr bool Q = FN == M355_PPF || FN == M355_ISF;
      const unsigned stride = !all_scalar ? dcol_stride(a.ews_doubles, P, Q ? 2 : 1) : 0;
      unsigned long long a_ = P, b_ = (unsigned long long)kTile;
      while (b_) { const unsigned long long t_ = a_ % b_; a_ = b_; b_ = t_; }
      const long long unit = (long long)(P / a_);
      long long pb = tiles > 8 * 256 ? 8 * 256 : tiles;
      pb = (pb / unit) * unit;
      const long long R = !all_scalar ? n / (long long)P : 0;
      const bool groups = P % kDcgCols == 0 && R * (long long)P == n;
      if (a.ews != nullptr && stride >= kDColMinStride && n / 8 >= tab_min_n() && !binning_disabled() &&
          (double)R * (double)kTabMinN >= 4096.0 * (double)tab_min_n() && P <= 0x7fffffffull / (2 * kDTabMax) &&
          (groups ? dcolg_ready<D, FN>() : (!Q && pb >= 1))) {
        const unsigned chunks = (stride + kBlock - 1) / kBlock;
        if (hipMemsetAsync(a.ews, 0, kDColHdr * sizeof(double), a.stream) != hipSuccess) return (int)hipGetLastError();
        hipLaunchKernelGGL((k_dcol_ranges<D>), dim3((unsigned)((P + kBlock - 1) / kBlock)), dim3(kBlock), 0, a.stream, pa, (unsigned)P,
                           stride, a.ews);
        if constexpr (Q)
          hipLaunchKernelGGL((k_dcol_build_q<D>), dim3((unsigned)(P * chunks)), dim3(kBlock), 0, a.stream, pa, (unsigned)P, stride, a.ews);
        else
          hipLaunchKernelGGL((k_dcol_build<D, FN>), dim3((unsigned)(P * chunks)), dim3(kBlock), 0, a.stream, pa, (unsigned)P, stride, a.ews);
        if (groups) {
          const unsigned G = (unsigned)(P / kDcgCols);
          unsigned C = 1024 / G;  // ~4 workgroups per CU in the queue
          C = C < 1 ? 1 : C;
          long long rows_per_wg = (R + C - 1) / C;
          rows_per_wg = ((rows_per_wg + 255) / 256) * 256;
          C = (unsigned)((R + rows_per_wg - 1) / rows_per_wg);
          hipLaunchKernelGGL((k_dcolg<D, FN>), dim3(G * C), dim3(kDcgThreads), dcolg_lds_bytes<FN>(), a.stream, a.x.ptr, a.x.dtype, a.out,
                             a.out_dtype, R, (unsigned)P, (unsigned)rows_per_wg, pa, a.ews);
        } else if constexpr (!Q) {
          hipLaunchKernelGGL((k_dcol<D, FN>), dim3((unsigned)pb), dim3(kBlock), 0, a.stream, a.x.ptr, a.x.dtype, a.out, a.out_dtype, n,
                             (unsigned)P, pa, a.ews);
        }
        return (int)hipGetLastError();
      }
    }
    if constexpr (!REDUCE && tab_this<DD, FN>()) {
      // row-periodic parameters, large n: per-column tables (k_coltab_build + k_coltab) in the caller's workspace; the
      // partial last row, if any, goes on below
      const long long R = !all_scalar ? n / (long long)P : 0;
      const unsigned long long G = P / kCtCols;
      const bool f64 = a.x.dtype == M355_F64 && a.out_dtype == M355_F64;
      // a column's table costs 1024 exact evaluations plus the launches: worth it from ~8192 rows (scaled with
      // m355_set_tab_min_n like the element threshold, so that tests can force the kernels onto small inputs; a million
      // columns of a hundred rows stay on k_colbinned)
      const bool rows_ok = (double)R * (double)kTabMinN >= 8192.0 * (double)tab_min_n();
      if (P % kCtCols == 0 && G >= 1 && G <= 65536ull && R >= 1 && rows_ok && n / 8 >= tab_min_n() && !binning_disabled() &&
          a.ews != nullptr && a.ews_doubles >= coltab_ws_doubles<FN>(P, R) &&
          (f64 ? coltab_ready<D, FN, true>() : coltab_ready<D, FN, false>())) {
        unsigned* counters = reinterpret_cast<unsigned*>(a.ews + (size_t)P * kCtTable);
        unsigned long long* miss_words = reinterpret_cast<unsigned long long*>(a.ews + (size_t)P * kCtTable + (G + 2) / 2);
        const unsigned long long chunks = coltab_chunks<FN>(R);
        const unsigned long long nwords = G * chunks * kCtThreads;
        const unsigned nr = (unsigned)((2 * P > G ? 2 * P : G) + kBlock - 1) / kBlock;
        hipLaunchKernelGGL((k_coltab_ranges<D, FN>), dim3(nr), dim3(kBlock), 0, a.stream, pa, (unsigned)P, a.ews, counters, (unsigned)G);
        hipLaunchKernelGGL((k_coltab_build<D, FN>), dim3((unsigned)(4 * P)), dim3(kBlock), 0, a.stream, pa, (unsigned)P, a.ews);
        unsigned long long m = G * chunks;
        m = m > 256 ? 256 : m;  // one workgroup per CU (LDS)
        if (f64)
          hipLaunchKernelGGL((k_coltab<D, FN, true>), dim3((unsigned)m), dim3(kCtThreads), kCtLds, a.stream, xp, a.x.dtype, op,
                             a.out_dtype, R, (unsigned)P, a.ews, counters, miss_words);
        else
          hipLaunchKernelGGL((k_coltab<D, FN, false>), dim3((unsigned)m), dim3(kCtThreads), kCtLds, a.stream, xp, a.x.dtype, op,
                             a.out_dtype, R, (unsigned)P, a.ews, counters, miss_words);
        const unsigned long long fb = (nwords + kBlock - 1) / kBlock;
        hipLaunchKernelGGL((k_coltab_fix<D, FN>), dim3((unsigned)(fb > 2048 ? 2048 : fb)), dim3(kBlock), 0, a.stream, xp, a.x.dtype, op,
                           a.out_dtype, R, (unsigned)P, pa, miss_words, nwords);
        hipLaunchKernelGGL((k_coltab_declined<D, FN>), dim3((unsigned)(P * kCtDeclChunks)), dim3(kBlock), 0, a.stream, xp, a.x.dtype,
                           op, a.out_dtype, R, (unsigned)P, pa, a.ews);
        const int rc = (int)hipGetLastError();
        const long long done = R * (long long)P;
        if (rc != 0 || done == n) return rc;
        xp = static_cast<const char*>(xp) + done * (a.x.dtype == M355_F32 ? 4 : 8);
        op = static_cast<char*>(op) + done * (a.out_dtype == M355_F32 ? 4 : 8);
        nn = n - done;
        blocks = 1;
      }
    }
    if constexpr (!REDUCE && bin_this<DD, FN>()) {
      // divergent functions with row-periodic parameters: whole blocks of kColRows rows go through k_colbinned (sorted
      // by x within each column); what is left (fewer than kColRows rows and a partial row) through k_periodic below
      const long long row_blocks = !all_scalar ? (nn / (long long)P) / kColRows : 0;
      const unsigned long long G = P / kColW;
      if (row_blocks >= 4 && P % kColW == 0 && G <= (unsigned long long)kMaxBlocks && !binning_disabled()) {
        long long m = 2048 / (long long)G;
        m = m < 1 ? 1 : (m > row_blocks ? row_blocks : m);
        hipLaunchKernelGGL((k_colbinned<D, FN>), dim3((unsigned)(G * m)), dim3(kBlock), 0, a.stream, xp, a.x.dtype, op,
                           a.out_dtype, row_blocks, (unsigned)P, pa);
        const int rc = (int)hipGetLastError();
        const long long done = row_blocks * kColRows * (long long)P;
        if (rc != 0 || done == nn) return rc;
        xp = static_cast<const char*>(xp) + done * (a.x.dtype == M355_F32 ? 4 : 8);
        op = static_cast<char*>(op) + done * (a.out_dtype == M355_F32 ? 4 : 8);
        nn -= done;
        const long long t2 = nn / kTile;
        blocks = (int)(t2 < 1 ? 1 : (t2 > blocks ? blocks : t2));
      }
    }
    if constexpr (!REDUCE && (FN == M355_LOGPDF || FN == M355_PDF)) {
      // stored densities: constants once into the workspace, no LDS table (k_periodic_fixed), when the grid can be a
      // multiple of P / gcd(P, kTile)
      constexpr long long NK = sizeof(typename DD::K) / sizeof(double);
      unsigned long long a_ = P, b_ = (unsigned long long)kTile;
      while (b_) { const unsigned long long t_ = a_ % b_; a_ = b_; b_ = t_; }
      const long long unit = (long long)(P / a_);
      const long long tiles2 = nn / kTile;
      long long pb = tiles2 > 8 * 256 ? 8 * 256 : tiles2;  // eight workgroups per CU: no LDS beyond the log table
      pb = (pb / unit) * unit;
      if (!all_scalar && a.ews != nullptr && (size_t)(P * NK) <= a.ews_doubles && pb >= 1 && nn == n) {
        hipLaunchKernelGGL((k_derive_table<D>), dim3((unsigned)((P + kBlock - 1) / kBlock)), dim3(kBlock), 0, a.stream, pa,
                           (unsigned)P, a.ews);
        if (a.x.dtype == M355_F64 && a.out_dtype == M355_F64)
          hipLaunchKernelGGL((k_periodic_fixed<D, FN, true>), dim3((unsigned)pb), dim3(kBlock), 0, a.stream, xp, a.x.dtype, op,
                             a.out_dtype, nn, (unsigned)P, a.ews);
        else
          hipLaunchKernelGGL((k_periodic_fixed<D, FN, false>), dim3((unsigned)pb), dim3(kBlock), 0, a.stream, xp, a.x.dtype, op,
                             a.out_dtype, nn, (unsigned)P, a.ews);
        return (int)hipGetLastError();
      }
    }
    const size_t lds = (size_t)P * sizeof(typename DD::K);
    if (f64io) {
      if constexpr (kFastIO)
        hipLaunchKernelGGL((k_periodic<D, FN, REDUCE, true>), dim3(blocks), dim3(kBlock), lds, a.stream,
                           xp, a.x.dtype, op, a.out_dtype, nn, pa, (unsigned)P, a.partials);
    } else {
      hipLaunchKernelGGL((k_periodic<D, FN, REDUCE, false>), dim3(blocks), dim3(kBlock), lds, a.stream,
                         xp, a.x.dtype, op, a.out_dtype, nn, pa, (unsigned)P, a.partials);
    }
  } else {
    GenericOps ops;
    ops.x = a.x;
    for (int q = 0; q < kMaxParams; ++q) ops.p[q] = a.p[q];
    hipLaunchKernelGGL((k_generic<D, FN, REDUCE>), dim3(blocks), dim3(kBlock), 0, a.stream, ops, a.out,
                       a.out_dtype, n, a.partials);
  }
  return (int)hipGetLastError();
}

template <int D, bool REDUCE> int launch_dist(const LaunchArgs& a) {
  switch (a.fn) {
    case M355_LOGPDF: return launch_fn<D, M355_LOGPDF, REDUCE>(a);
    case M355_LOGCDF: return launch_fn<D, M355_LOGCDF, REDUCE>(a);
    case M355_LOGSF: return launch_fn<D, M355_LOGSF, REDUCE>(a);
    default: break;
  }
  if constexpr (!REDUCE) {
    switch (a.fn) {
      case M355_PDF: return launch_fn<D, M355_PDF, false>(a);
      case M355_CDF: return launch_fn<D, M355_CDF, false>(a);
      case M355_SF: return launch_fn<D, M355_SF, false>(a);
      default: break;
    }
    if constexpr (supported(D, M355_PPF)) {
      if (a.fn == M355_PPF) return launch_fn<D, M355_PPF, false>(a);
      if (a.fn == M355_ISF) return launch_fn<D, M355_ISF, false>(a);
    }
  }
  return -1000;  // unsupported (dist, fn)
}

// ------------------------------------------------------------------------------------- moments
// out[i] = Mom<D>::eval(stat, params_i) (m355_moments.h).  k_moment_thread: one thread per parameter set (closed
// forms, medians through ppf).  k_moment_wave: one wavefront per parameter set for the stats in Mom<D>::kWave --
// the lanes share the parameter set, split the quadrature / support sums and combine them with xor-shuffles, so
// the 4 x 1000 integrand evaluations of e.g. rice.kurtosis take ~63 evaluations per lane.
struct MomentArgs {
  int stat;
  int out_dtype;
  long long n;
  GenericOps ops;  // ops.x unused
  void* out;
  hipStream_t stream;
};

template <int D>
__global__ __launch_bounds__(kBlock) void k_moment_thread(GenericOps ops, int stat, void* __restrict__ out, int odt, long long n) {
  if constexpr (uses_log_table<Dist<D>>::value) log_table_init();  // integrands / medians may call flog_tab bodies
  const bool small = n <= 0xffffffffll;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < Dist<D>::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)i, small);
    store1(out, odt, i, Mom<D>::template eval<SerialCtx>(stat, pp));
  }
}

template <int D>
__global__ __launch_bounds__(kBlock) void k_moment_wave(GenericOps ops, int stat, void* __restrict__ out, int odt, long long n) {
  if constexpr (uses_log_table<Dist<D>>::value) log_table_init();  // integrands / medians may call flog_tab bodies
  const bool small = n <= 0xffffffffll;
  const long long wave = ((long long)blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * kBlock) >> 6;
  for (long long i = wave; i < n; i += nwaves) {  // uniform per wavefront: every lane reaches the shuffles
    double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < Dist<D>::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)i, small);
    const double r = Mom<D>::template eval<WaveCtx>(stat, pp);
    if ((threadIdx.x & 63) == 0) store1(out, odt, i, r);
  }
}

template <int D> int launch_moment(const MomentArgs& a) {
  if (a.stat < 0 || a.stat >= M355_STAT_COUNT) return -1000;
  if ((Mom<D>::kWave >> a.stat) & 1u) {
    const long long want = (a.n + (kBlock / 64) - 1) / (kBlock / 64);
    const int blocks = (int)(want < 1 ? 1 : (want > kMaxBlocks ? kMaxBlocks : want));
    hipLaunchKernelGGL((k_moment_wave<D>), dim3(blocks), dim3(kBlock), 0, a.stream, a.ops, a.stat, a.out, a.out_dtype, a.n);
  } else {
    const long long want = (a.n + kBlock - 1) / kBlock;
    const int blocks = (int)(want < 1 ? 1 : (want > kMaxBlocks ? kMaxBlocks : want));
    hipLaunchKernelGGL((k_moment_thread<D>), dim3(blocks), dim3(kBlock), 0, a.stream, a.ops, a.stat, a.out, a.out_dtype, a.n);
  }
  return (int)hipGetLastError();
}

}  // namespace m355

// One translation unit per distribution defines its launcher (keeps hipcc compile times parallel).
#define M355_DECLARE_LAUNCH(ID, name, np) int m355_launch_##name(const m355::LaunchArgs& a);
M355_DIST_LIST(M355_DECLARE_LAUNCH)
#undef M355_DECLARE_LAUNCH
#define M355_DECLARE_MOMENT(ID, name, np) int m355_launch_moment_##name(const m355::MomentArgs& a);
M355_DIST_LIST(M355_DECLARE_MOMENT)
#undef M355_DECLARE_MOMENT
#define M355_DEFINE_LAUNCH(ID, name)                                                                  \
  int m355_launch_##name(const m355::LaunchArgs& a) {                                                  \
    return a.partials ? m355::launch_dist<ID, true>(a) : m355::launch_dist<ID, false>(a);              \
  }                                                                                                    \
  int m355_launch_moment_##name(const m355::MomentArgs& a) { return m355::launch_moment<ID>(a); }
