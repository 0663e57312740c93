// Multivariate log-densities: batched Cholesky + triangular solve + fused epilogue
// (mvnormal.py:9-37,80-83; mvstudentt.py:58-68; matrixnormal.py:68-94 of the reference).
//
//   k_cholesky     one workgroup (one or two wavefronts, lane = matrix row) per covariance matrix,
//                  the matrix in LDS with an odd row stride (conflict-free row walks): left-looking
//                  column factorisation -- row i accumulates a[i][j] - sum_{k<j} L[i][k] L[j][k] in
//                  registers (4 independent chains), so a column costs two barriers and no LDS
//                  writes in its inner loop.  Writes L (lower, row-major, upper part zeroed),
//                  1/diag(L) and logdet = 2*sum(log diag) (mvnormal.py:9-14).  A non-positive pivot
//                  gives NaN, as the reference's Cholesky does (posdef is ignored there, quirk Q5).
//                  For a shared covariance it also emits the MFMA operand fragments k_solve_mfma
//                  consumes: -L in 16x16 blocks and the inverses of the diagonal blocks.
//   k_solve_mfma   shared covariance, contiguous observations: blocked forward substitution on
//                  v_mfma_f64_16x16x4_f64.  A wavefront owns tiles of 16 observations (the N
//                  dimension); per 16-row block I of L:  R_I = Delta_I - sum_{J<I} L_IJ Y_J  and
//                  Y_I = inv(L_II) R_I, each 16x16x16 product = 4 MFMAs.  The D layout of the f64
//                  MFMA (row = (lane>>4) + 4*reg) is exactly its B-operand layout for k-step `reg`,
//                  so Y_J feeds the next products straight from registers.  Observation tiles are
//                  contiguous in HBM: coalesced 16-byte loads (next tile prefetched into registers
//                  while the current one is in the MFMA chain), staged through a wave-private LDS
//                  region to reach the MFMA layout.  quad = sum y^2 (mvnormal.py:32) and the logpdf
//                  epilogue are fused.
//   k_solve        one thread per observation: forward substitution L y = x - mu in the order of the
//                  reference's per-row solve (mvnormal.py:30-31), y kept in LDS column-major
//                  ([component][thread]: conflict-free).  Serves batched covariances and the
//                  strided solves of matrixnormal (columns of X-M against L_U, then rows against L_V).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "m355_error.h"
#include "m355_special.h"
#include "../../include/m355_dist.h"

namespace m355 {

constexpr int kMaxDim = 128;
constexpr int kSolveThreads = 64;

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));

// workspace layout per matrix: L (d*d) | invdiag (d) | logdet (1)
__host__ __device__ inline size_t chol_stride(int d) { return (size_t)d * d + d + 1; }
__host__ __device__ inline int nblk16(int d) { return (d + 15) / 16; }
// MFMA operand fragments of one factor: NB(NB+1)/2 blocks x 4 k-steps x 64 lanes
__host__ __device__ inline size_t frag_doubles(int d) { return (size_t)nblk16(d) * (nblk16(d) + 1) / 2 * 256; }

// Phase timestamps for tools/chol_probe.hip only (compiled out of the library).
#ifdef M355_CHOL_PROBE
__device__ long long g_chol_ticks[64];
#define M355_TICK(n) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_chol_ticks[n] = clock64(); } while (0)
#define M355_TICKW(n) do { if (threadIdx.x == 64 && blockIdx.x == 0 && (n) < 64) g_chol_ticks[n] = clock64(); } while (0)
#else
#define M355_TICK(n) do { } while (0)
#define M355_TICKW(n) do { } while (0)
#endif

M355_DEV double readlane_d(double v, int lane) {  // `lane` must be wave-uniform
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// u += (lane N of the caller's row of 16 lanes of r) * m: ONE instruction (v_fmac_f64 with the DPP control row_newbcast,
// gfx90a+), where a v_readlane broadcast costs two instructions plus the FMA.  r must not come out of a VALU
// instruction in the two preceding slots (DPP read hazard): here it is an LDS return (ds_bpermute).
template <int N> M355_DEV void fmac_row_bcast(double& u, double r, double m) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(u) : "v"(r), "v"(m), "n"(N));
}
M355_DEV void fmac_row_bcast_n(double& u, double r, double m, int n) {  // n: a constant after unrolling
  switch (n) {
    case 1: fmac_row_bcast<1>(u, r, m); break;
    case 2: fmac_row_bcast<2>(u, r, m); break;
    case 3: fmac_row_bcast<3>(u, r, m); break;
    case 4: fmac_row_bcast<4>(u, r, m); break;
    case 5: fmac_row_bcast<5>(u, r, m); break;
    case 6: fmac_row_bcast<6>(u, r, m); break;
    case 7: fmac_row_bcast<7>(u, r, m); break;
    case 8: fmac_row_bcast<8>(u, r, m); break;
    case 9: fmac_row_bcast<9>(u, r, m); break;
    case 10: fmac_row_bcast<10>(u, r, m); break;
    case 11: fmac_row_bcast<11>(u, r, m); break;
    case 12: fmac_row_bcast<12>(u, r, m); break;
    case 13: fmac_row_bcast<13>(u, r, m); break;
    case 14: fmac_row_bcast<14>(u, r, m); break;
    default: fmac_row_bcast<15>(u, r, m); break;
  }
}
// v of lane (byte address addr4 / 4), any lane pattern, through the LDS crossbar (no LDS memory, no VALU slot)
M355_DEV double bpermute_d(int addr4, double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(addr4, (int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(addr4, (int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// sqrt(a) and 1/sqrt(a) from the hardware rsq seed (coupled Newton steps, ~1 ulp each); a <= 0 or
// NaN gives NaN for both (a non-positive pivot poisons the factor, as the reference's does).
// (CHECKED = false leaves out the final selects: a < 0 and NaN give NaN through v_rsq_f64 itself, a = 0 and a = inf
// through 0 * inf in the first product -- only -0.0, whose rsq is -inf and whose root comes out as NaN as well, would
// differ in no way either.  The selects are 6 of the ~18 operations.)
template <bool CHECKED = true> M355_DEV void sqrt_rsqrt(double a, double& root, double& inv_root) {
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, a), h, g);
  r = __builtin_fma(-h, g, 0.5);
  h = __builtin_fma(h, r, h);
  if constexpr (CHECKED) {
    const bool ok = a > 0.0;
    root = ok ? g : kNaN;
    inv_root = ok ? 2.0 * h : kNaN;
  } else {
    root = g;
    inv_root = 2.0 * h;
  }
}

// Fragment of block (I, J), J <= I, k-step kk, lane l: element [row 16I + (l&15)][col 16J + (l>>4) + 4kk]
// of -L (J < I) or of inv(L_II) (J == I): the A operand whose k index matches the D/B layout.
//
// Factorisation (256 threads; row i of the matrix belongs to thread i): the matrix is padded to
// dp = 16*NB with an identity block.  Per 16-column panel J:
//   A  U = A[:, panel] - L[:, :16J] L[panel, :16J]^T : the 16 columns are split over the wavefronts
//      (each lane: its row against 4 or 8 columns, register accumulators; one conflict-free row read
//      plus broadcast reads per k), result written back in place;
//   B  the wavefront that owns rows 16J..16J+15 factors the panel in registers: column c takes its
//      pivot and the multipliers L[16J+c][c'] from the owning lanes by v_readlane (no LDS, no
//      barrier): l[c] = (u[c] - sum_{c'<c} l[c'] L[16J+c][c']) / sqrt(pivot);
//   C  rows held by the other row wavefront (d > 64) do the same substitution against the 16x16
//      diagonal block published through LDS.
// Measured phase costs (tools/chol_probe.hip) are in DESIGN.md.
constexpr int kCholThreads = 256;
// Shared covariance, d <= 64: the solve is Y = W (X - mu)^T with W = L^-1 built block by block in k_cholesky -- a plain
// triangular matrix product whose MFMAs depend on each other only through their accumulators (four independent
// accumulations per tile), where blocked substitution is one chain of 40 dependent MFMAs per tile.  For observations
// drawn with the covariance at hand the explicit inverse costs ~2x the rounding error of substitution and stays far
// below 1e-12 (measured to cond(cov) = 1e10 against a long-double solve).  Larger d keeps substitution.
constexpr int kFullInverseMaxNB = 4;

// Arguments of k_mvn_tiles (one observation with its own covariance, d <= 64: factorisation, forward substitution, sum
// of squares and epilogue in one wavefront -- a matrix is read from HBM once and nothing but the result is written).
struct FusedArgs {
  const double* x;   // n x d
  const double* mu;  // d (mu_stride 0) or n x d
  long long mu_stride;
  double* out;
  int mode;          // 1: mvnormal logpdf, 2: mvstudentt logpdf
  double nu, c0;
  long long n;       // number of (covariance, observation) pairs
};

__global__ __launch_bounds__(kCholThreads) void k_cholesky(const double* __restrict__ cov, int d,
                                                            double* __restrict__ ws, double* __restrict__ frags) {
  extern __shared__ double a[];  // dp x S (S odd) | 16x16 diagonal block + 16 reciprocals | NB x 16 x 16 inverses | dp reciprocals
  const int NB = nblk16(d);
  const int dp = 16 * NB, S = dp + 1;
  double* diag = a + (size_t)dp * S;
  double* inv = diag + 272;
  double* rdiag = inv + (NB <= kFullInverseMaxNB ? NB * (NB + 1) / 2 : NB) * 256;
  const double* src = cov + (size_t)blockIdx.x * d * d;
  double* dst = ws + (size_t)blockIdx.x * chol_stride(d);
  const int t = threadIdx.x, wave = t >> 6;
  // 2-D thread layout for the global <-> LDS copies: CW columns x (256 / CW) rows per step
  const int cshift = dp <= 16 ? 4 : (dp <= 32 ? 5 : (dp <= 64 ? 6 : 7));
  const int ccol = t & ((1 << cshift) - 1), crow0 = t >> cshift, crstep = kCholThreads >> cshift;
  M355_TICK(0);
  if (ccol < dp) {
#pragma unroll 8
    for (int r = crow0; r < dp; r += crstep)
      a[r * S + ccol] = (r < d && ccol < d) ? src[r * d + ccol] : (r == ccol ? 1.0 : 0.0);
  }
  __syncthreads();
  M355_TICK(1);
  const int lane = t & 63;
  const int i = t;  // row owned in steps B / C (threads >= dp idle there)
  double* ai = a + (size_t)(i < dp ? i : 0) * S;
  for (int J = 0; J < NB; ++J) {
    M355_TICK(2 + 4 * J);
    const int c0 = 16 * J;
    // A: 16x16 tiles (I, J), I >= J, dealt to the wavefronts; U_IJ = A_IJ - sum_{K<J} L_IK L_JK^T on MFMA
    for (int I = J + wave; J > 0 && I < NB; I += kCholThreads / 64) {
      double* cbase = a + (size_t)(16 * I + (lane >> 4)) * S + c0 + (lane & 15);  // D layout: row (l>>4)+4r, col l&15
      d4_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = cbase[4 * r * S];
      const double* pa = a + (size_t)(16 * I + (lane & 15)) * S + (lane >> 4);  // A: L[16I + (l&15)][k]
      const double* pb = a + (size_t)(c0 + (lane & 15)) * S + (lane >> 4);      // B: L[16J + (l&15)][k]
      for (int k = 0; k < c0; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[k], pb[k], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) cbase[4 * r * S] = acc[r];
    }
    if (J > 0) __syncthreads();
    M355_TICK(3 + 4 * J);
    double u[16], l[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) l[c] = 0.0;
    const bool mine = i >= c0 && i < dp;
#pragma unroll
    for (int c = 0; c < 16; ++c) u[c] = mine ? ai[c0 + c] : 0.0;
    if (wave == (c0 >> 6)) {  // B (wave-uniform branch)
      const int base = c0 & 63;
      const int r = (i & 63) - base;
      double rinv[16];
      // right-looking inside the panel: as soon as column c is known every later column takes its update
      // u[c2] -= l[c] L[16J+c2][c] (15 - c independent FMAs) instead of a chain of c dependent FMAs in front of every
      // pivot.  The updates reach each u[c2] in the same order c = 0, 1, ... as in the left-looking form: bit-identical
      // factors.  (Measured: no faster -- the panel is bound by the issue of its readlanes and FMAs, not by the chain.)
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        double piv;
        sqrt_rsqrt(readlane_d(u[c], base + c), piv, rinv[c]);
        l[c] = (r == c) ? piv : (r > c ? u[c] * rinv[c] : 0.0);
#pragma unroll
        for (int c2 = c + 1; c2 < 16; ++c2) u[c2] = __builtin_fma(-l[c], readlane_d(l[c], base + c2), u[c2]);
      }
      if (r >= 0 && r < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          diag[r * 16 + c] = l[c];
          if (r == c) {
            diag[256 + c] = rinv[c];
            rdiag[c0 + c] = rinv[c];
          }
        }
      }
    }
    if (dp > 64) {
      __syncthreads();
      M355_TICK(4 + 4 * J);
      if (wave != (c0 >> 6) && wave < 2 && i >= c0 + 16 && i < dp) {  // C
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          double sacc = u[c];
#pragma unroll
          for (int cc = 0; cc < c; ++cc) sacc = __builtin_fma(-l[cc], diag[c * 16 + cc], sacc);
          l[c] = sacc * diag[256 + c];
        }
      }
    }
    if (mine) {
#pragma unroll
      for (int c = 0; c < 16; ++c) ai[c0 + c] = l[c];
    }
    __syncthreads();
    M355_TICK(5 + 4 * J);
  }
  M355_TICK(40);
  if (ccol < d) {
#pragma unroll 8
    for (int r = crow0; r < d; r += crstep) dst[r * d + ccol] = (ccol <= r) ? a[r * S + ccol] : 0.0;
  }
  if (i < d) dst[(size_t)d * d + i] = 1.0 / ai[i];
  {
    // logdet = 2 * sum log(diag): per-row logs, then a fixed-order tree over the workgroup
    double lg = (i < d) ? flog(ai[i]) : 0.0;  // log of a NaN diag -> NaN
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lg += __shfl_down(lg, o, 64);
    if ((t & 63) == 0) diag[wave] = lg;
    __syncthreads();
    if (t == 0) dst[(size_t)d * d + d] = 2.0 * (diag[0] + diag[1]);  // rows live in wavefronts 0 and 1 only
  }
  M355_TICK(41);
  if (frags == nullptr) return;
  // Up to NB = 4 the solve multiplies by W = L^-1 outright (k_solve_mfma, kFullInverseMaxNB): every block W_IJ is
  // built here; beyond, only the diagonal blocks are inverted and the solve substitutes across blocks with -L.
  const bool full = NB <= kFullInverseMaxNB;
  auto blk = [&](int I, int J) { return full ? I * (I + 1) / 2 + J : I; };  // slot of block (I, J) in inv[]
  // inverse of each 16x16 lower-triangular diagonal block, one column per thread, in registers:
  //   X[c][c] = 1/L[c][c],  X[r][c] = -(sum_{k=c}^{r-1} L[r][k] X[k][c]) / L[r][r]
  if (t < dp) {
    const int I = t >> 4, c = t & 15;
    const double* Lb = a + (size_t)(16 * I) * S + 16 * I;
    const double* rd = rdiag + 16 * I;
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) acc = __builtin_fma(Lb[r * S + k], (k >= c) ? x[k] : 0.0, acc);
      x[r] = (r == c) ? rd[r] : (r > c ? -acc * rd[r] : 0.0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) inv[blk(I, I) * 256 + r * 16 + c] = x[r];
  }
  __syncthreads();
  M355_TICK(42);
  if (full) {
    // off-diagonal blocks by block forward substitution on the identity, one sub-diagonal s at a time (its blocks
    // are independent: dealt to the wavefronts):  W_IJ = -W_II sum_{K=J}^{I-1} L_IK W_KJ, each 16x16x16 product = 4 MFMAs
    for (int s_ = 1; s_ < NB; ++s_) {
      for (int J = wave; J + s_ < NB; J += kCholThreads / 64) {
        const int I = J + s_;
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        for (int K = J; K < I; ++K) {
          const double* pa = a + (size_t)(16 * I + (lane & 15)) * S + 16 * K + (lane >> 4);  // A: L_IK[row l&15][k]
          const double* pb = inv + blk(K, J) * 256 + (lane >> 4) * 16 + (lane & 15);         // B: W_KJ[k][col l&15]
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk], pb[64 * kk], acc, 0, 0, 0);
        }
        // the D layout of the sum (row (l>>4)+4r, col l&15) is the B layout of k-step r
        const double* pw = inv + blk(I, I) * 256 + (lane & 15) * 16 + (lane >> 4);           // A: W_II[row l&15][k]
        d4_t w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w = __builtin_amdgcn_mfma_f64_16x16x4f64(-pw[4 * kk], acc[kk], w, 0, 0, 0);
        double* po = inv + blk(I, J) * 256 + (lane >> 4) * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) po[64 * r] = w[r];
      }
      __syncthreads();
    }
  }
  {
    // thread = (k-step kk, lane l); blocks b = (I, J) walked in order: the A operand of the solve's MFMAs
    const int l = t & 63, kk = t >> 6;
    const int r = l & 15, c = (l >> 4) + 4 * kk;
    int e = t;
    for (int I = 0; I < NB; ++I) {
      for (int J = 0; J <= I; ++J, e += 256)
        frags[e] = (J == I || full) ? inv[blk(I, J) * 256 + r * 16 + c] : -a[(16 * I + r) * S + 16 * J + c];
    }
  }
  M355_TICK(43);
}

// One observation with its own covariance, d <= 64 (SURVEY 8(d) config 4 variant B): one wavefront per pair, the matrix
// in registers.  The lower triangle is held as 16x16 tiles in the A-operand layout of v_mfma_f64_16x16x4 (lane l,
// element kk: T[l & 15][(l >> 4) + 4 kk]).  A tile in that layout serves as the A operand (= T) and as the B operand
// (= T^T), and the D layout of X Y^T is the A layout of Y X^T, so the trailing updates of a right-looking blocked
// Cholesky,  T_IK -= L_IJ L_KJ^T  as  D = (-L_KJ) L_IJ^T + T_IK,  take every operand straight from registers.
// Panel J (16 columns) is factored in row layout, lane i = matrix row i, reached through a 64 x 17 LDS buffer: column c
// takes its pivot from lane 16J + c and its multipliers L[c2][c] from the lanes below by v_readlane, and since
// l_c = u_c / L_cc is formed in every lane the same sweep is the triangular solve of all the tiles under the diagonal one
// (no block inverse, no MFMA for it).  The observation rides along as one value per lane, b_i: y_c = b_{16J+c} / L_cc
// (a broadcast), b_i -= y_c L_ic for every row below, quad += y_c^2 (mvnormal.py:30-32) -- the forward substitution
// costs five operations per column.  Finished rows (lanes < 16J + c) keep computing on values nobody reads: no selects.
// LDS 8.7 KiB per wavefront (k_mvn_batched of round 2 kept the matrix in LDS: 33 KiB, four wavefronts per CU, 52 us per
// matrix in a chain of LDS round trips).
#ifndef M355_TILES_DPP
#define M355_TILES_DPP 1
#endif
#ifndef M355_TILES_WAVES
#define M355_TILES_WAVES 3  // wavefronts per SIMD the register budget is set for (4 spills 16 VGPRs at d = 64 and is no faster)
#endif
template <int NB>
__global__ __launch_bounds__(64, M355_TILES_WAVES) void k_mvn_tiles(const double* __restrict__ cov, int d, const FusedArgs fa) {
  __shared__ double P[64 * 17];
  const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
  const int prow = lane < 16 * NB ? lane : 16 * NB - 1;  // (NB < 4: the lanes past the matrix shadow its last row)
  // (one workgroup per pair: a wavefront that walks several pairs needs the tile registers across the loop's back edge
  // and spills -- 284 bytes of scratch per lane, 1.45 ms instead of 1.04)
  const long long b = blockIdx.x;
  const double* src = cov + (size_t)b * d * d;
  d4_t T[NB][NB];
  if (d == 16 * NB) {
    // full tiles: one 32-bit lane offset, block-row bases in scalar registers, the rest immediate offsets
    const int loff = r16 * d + g;
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#pragma unroll
      for (int I = J; I < NB; ++I) {
        const double* base = src + (size_t)(16 * I) * d + 16 * J;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) T[I][J][kk] = base[loff + 4 * kk];
      }
    }
  } else {
    // every load is issued unconditionally at a clamped address (no exec-mask branches around the loads); entries
    // outside the d x d matrix are then replaced by the identity
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#pragma unroll
      for (int I = J; I < NB; ++I) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int row = 16 * I + r16, col = 16 * J + g + 4 * kk;
          const double v = src[(size_t)(row < d ? row : d - 1) * d + (col < d ? col : d - 1)];
          T[I][J][kk] = (row < d && col < d) ? v : (row == col ? 1.0 : 0.0);
        }
      }
    }
  }
  double bcur;  // (x - mu)[lane], then the running right-hand side of the forward substitution
  {
    const int cc = lane < d ? lane : d - 1;
    const double v = fa.x[b * d + cc] - fa.mu[(fa.mu_stride ? b * fa.mu_stride : 0) + cc];
    bcur = lane < d ? v : 0.0;
  }
  double mydiag = 1.0, quad = 0.0;
#pragma unroll
  for (int J = 0; J < NB; ++J) {
    // column J of tiles: A layout -> row layout
#pragma unroll
    for (int I = J; I < NB; ++I) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) P[(16 * I + r16) * 17 + g + 4 * kk] = T[I][J][kk];
    }
    __syncthreads();
    double u[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) u[c] = P[prow * 17 + c];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      double piv, rinv;
      sqrt_rsqrt<false>(readlane_d(u[c], 16 * J + c), piv, rinv);
      const double lc = u[c] * rinv;
      P[lane * 17 + c] = lc;  // (row layout, for the trailing tiles and the diagonal; the panel's values are all in u by now)
      const double yc = readlane_d(bcur, 16 * J + c) * rinv;
      quad = __builtin_fma(yc, yc, quad);
      bcur = __builtin_fma(-yc, lc, bcur);
#if M355_TILES_DPP
      // the multipliers L[16J + c2][c] sit in the lanes of row group J: one ds_bpermute pair copies that group's l_c into
      // every group of 16 lanes, and each update is then a single v_fmac_f64 whose first operand is lane c2 of the
      // caller's own row (DPP row_newbcast) -- 1 instruction per update where a v_readlane broadcast took 3
      if (c < 15) {
        const double rep = bpermute_d(4 * (16 * J + r16), lc);
        const double nlc = -lc;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; ++c2) fmac_row_bcast_n(u[c2], rep, nlc, c2);
      }
#else
      // multipliers four at a time: a v_readlane result needs two wait states before a VALU reads it, which the
      // next broadcasts fill (one pair at a time the compiler reuses one SGPR pair and pads every FMA with s_nop)
#pragma unroll
      for (int c0 = c + 1; c0 < 16; c0 += 4) {
        double m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = (c0 + q < 16) ? readlane_d(lc, 16 * J + (c0 + q < 16 ? c0 + q : 15)) : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (c0 + q < 16) u[c0 + q] = __builtin_fma(-lc, m[q], u[c0 + q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    {
      // the sweep left the panel in LDS in row layout (no registers for it: 116 VGPRs, four wavefronts per SIMD):
      // lane i's own diagonal entry L_ii is read back from there (a select per column cost three instructions each), and
      // the tiles under the diagonal one return to the A layout
      __syncthreads();
      if (g == J) mydiag = P[lane * 17 + r16];
      if (J + 1 < NB) {
        d4_t L[NB];
#pragma unroll
        for (int I = J + 1; I < NB; ++I) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) L[I][kk] = P[(16 * I + r16) * 17 + g + 4 * kk];
        }
#pragma unroll
        for (int K = J + 1; K < NB; ++K) {
#pragma unroll
          for (int I = K; I < NB; ++I) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) T[I][K] = __builtin_amdgcn_mfma_f64_16x16x4f64(-L[K][kk], L[I][kk], T[I][K], 0, 0, 0);
          }
        }
        __syncthreads();
      }
    }
  }
  // quad is wave-uniform; log diag(L): lane i holds L_ii (identity padding: log 1 = 0; a NaN pivot -> NaN)
  double lg = flog(mydiag);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);
  if (lane == 0) {
    const double logdet = 2.0 * lg;
    double v = -0.5 * (fa.c0 + logdet + quad);  // mvnormal.py:82-83
    if (fa.mode == 2) v = fa.c0 - 0.5 * (fa.nu + (double)d) * flog1p(quad / fa.nu) - 0.5 * logdet;  // mvstudentt.py:62-64
    fa.out[b] = v;
  }
}

struct SolveArgs {
  const double* x;       // observation r, component i: x[(r / inner) * s1 + (r % inner) * s2 + i * cs]
  const double* mu;      // same with mu_s1 (0 when mu is shared over the batch)
  long long inner, s1, s2, cs, mu_s1, mu_s2, mu_cs;
  const double* chol;    // chol_stride(d) doubles per matrix
  long long l_div, l_mod;  // matrix of observation r: (r / l_div) % l_mod
  double* y_out;         // optional, laid out like x
  double* out;           // optional, one value per observation
  long long n;
  int d;
  int mode;              // 0: quad, 1: mvnormal logpdf, 2: mvstudentt logpdf
  double nu, c0;         // mvstudentt: nu and the nu-only normaliser; mvnormal: c0 = d*log(2 pi)
};

template <bool SHARED> __global__ __launch_bounds__(kSolveThreads) void k_solve(const SolveArgs a) {
  extern __shared__ double y[];  // [d][kSolveThreads]
  const int t = threadIdx.x;
  const long long r = (long long)blockIdx.x * kSolveThreads + t;
  if (r >= a.n) return;  // no barrier below: threads are independent
  const int d = a.d;
  const long long b = r / a.inner, c = r % a.inner;
  const double* xr = a.x + b * a.s1 + c * a.s2;
  const double* mr = a.mu + b * a.mu_s1 + c * a.mu_s2;
  const long long li = SHARED ? 0 : (r / a.l_div) % a.l_mod;  // SHARED: wave-uniform address -> scalar loads
  const double* L = a.chol + (size_t)li * chol_stride(d);
  const double* invdiag = L + (size_t)d * d;
  double quad = 0.0;
  for (int i = 0; i < d; ++i) {
    // s = delta_i - sum_{k<i} L[i][k] y_k, two interleaved chains to hide the FMA latency
    double s0 = xr[i * a.cs] - mr[i * a.mu_cs], s1 = 0.0;
    const double* Li = L + (size_t)i * d;
    int k = 0;
    for (; k + 1 < i; k += 2) {
      s0 = __builtin_fma(-Li[k], y[k * kSolveThreads + t], s0);
      s1 = __builtin_fma(-Li[k + 1], y[(k + 1) * kSolveThreads + t], s1);
    }
    if (k < i) s0 = __builtin_fma(-Li[k], y[k * kSolveThreads + t], s0);
    const double yi = (s0 + s1) * invdiag[i];
    y[i * kSolveThreads + t] = yi;
    quad = __builtin_fma(yi, yi, quad);
    if (a.y_out) a.y_out[b * a.s1 + c * a.s2 + i * a.cs] = yi;
  }
  if (a.out) {
    const double logdet = L[(size_t)d * d + d];
    double v = quad;
    if (a.mode == 1) {
      v = -0.5 * (a.c0 + logdet + quad);  // mvnormal.py:82-83
    } else if (a.mode == 2) {
      // mvstudentt.py:62-64: norm + inner - 0.5*logdet, norm = c0 (nu- and k-only terms)
      v = a.c0 - 0.5 * (a.nu + (double)d) * flog1p(quad / a.nu) - 0.5 * logdet;
    }
    a.out[r] = v;
  }
}

struct MfmaArgs {
  const double* x;      // n x d row-major, contiguous, 16-byte aligned
  const double* mu;     // d doubles (mu_rows 0) or n x d laid out like x
  const double* frags;  // frag_doubles(d) doubles from k_cholesky
  const double* chol;   // chol_stride(d) doubles (logdet at the end)
  double* out;
  long long n;
  int d;
  int mode;  // 0: quad, 1: mvnormal logpdf, 2: mvstudentt logpdf
  double nu, c0;
};

constexpr int kMfmaWaves = 4;

// REGF (NB <= kFullInverseMaxNB): the fragments are the blocks of W = L^-1 and live in registers for the whole kernel
// (NB(NB+1)/2 x 4 doubles per lane: 80 VGPRs at NB = 4); a tile is NB independent accumulations.  Otherwise blocked
// forward substitution with the fragments in LDS (-L blocks, inverse diagonal blocks), T tiles interleaved.
template <int NB, bool MU_ROWS, int T, bool REGF>
__global__ __launch_bounds__(64 * kMfmaWaves) void k_solve_mfma(const MfmaArgs a) {
  constexpr int NBLK = NB * (NB + 1) / 2;
  constexpr int S = 16 * NB + 4;  // staging row stride: the 64 lanes of a D-layout read hit each bank pair twice
  constexpr int NJ = 2 * NB;      // 16-byte loads per lane per tile (16 obs x 16 NB doubles / 128)
  extern __shared__ double lds[];
  double* fr = lds;                                   // NBLK * 256 (unused with REGF)
  double* mu_s = fr + (REGF ? 0 : NBLK * 256);        // 16 * NB (zero padded)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* st = mu_s + 16 * NB + wave * (T * 16 * S);  // this wavefront's T staging tiles of 16 x S
  const int d = a.d;
  const int tile_elems = 16 * d;
  const long long total = a.n * (long long)d;
  const long long ntiles = (a.n + 15) / 16;
  const long long ngroups = (ntiles + T - 1) / T;     // a wavefront takes T consecutive tiles per step
  const long long gstride = (long long)gridDim.x * kMfmaWaves;
  d2_t raw[T][NJ];
  auto fetch = [&](long long group) {
    if ((group * T + T) * tile_elems <= total) {
      // whole tiles (all groups but the last): straight-line loads; lanes past a tile (d < 16 NB) read its element 0 and
      // never use the value.  (With a branch per load the compiler put a wait for ALL outstanding loads in front of
      // each one -- the tail path's narrower load aliased its destination -- one memory latency per load.)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const long long base = (group * T + t) * tile_elems;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int e = 128 * j + 2 * lane;
          const long long g = base + (e < tile_elems ? e : 0);
          d2_t v = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(a.x + g));
          if (MU_ROWS) v -= __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(a.mu + g));
          raw[t][j] = v;
        }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const long long base = (group * T + t) * tile_elems;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int e = 128 * j + 2 * lane;
        const long long g = base + e;
        d2_t v = {0.0, 0.0};
        if (e < tile_elems) {
          if (g + 1 < total) {
            v = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(a.x + g));
            if (MU_ROWS) v -= __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(a.mu + g));
          } else if (g < total) {
            v.x = a.x[g];
            if (MU_ROWS) v.x -= a.mu[g];
          }
        }
        raw[t][j] = v;
      }
    }
  };
  M355_TICK(50);
  long long group = (long long)blockIdx.x * kMfmaWaves + wave;
  if (group < ngroups) fetch(group);  // first tiles in flight while the operand fragments arrive
  double F[REGF ? NBLK * 4 : 1];
  if constexpr (REGF) {
#pragma unroll
    for (int b = 0; b < NBLK * 4; ++b) F[b] = a.frags[b * 64 + lane];
  } else {
    double tmp[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) tmp[b] = a.frags[b * 256 + threadIdx.x];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) fr[b * 256 + threadIdx.x] = tmp[b];
  }
  for (int e = threadIdx.x; e < 16 * NB; e += blockDim.x) mu_s[e] = (!MU_ROWS && e < d) ? a.mu[e] : 0.0;
  for (int e = lane; e < T * 16 * S; e += 64) st[e] = 0.0;  // padding columns stay zero for the whole kernel
  __syncthreads();
  auto frag = [&](int blk, int kk) -> double {
    if constexpr (REGF) return F[blk * 4 + kk];
    else return fr[(blk * 4 + kk) * 64 + lane];
  };

  // flat element e = 128 j + 2 lane (+1) of a tile -> staging offset, fixed for the kernel.
  // e / d through a float reciprocal: exact here (e < 2048, d <= 128: (e + 0.5) / d is never within
  // float rounding of an integer).
  int off0[NJ], off1[NJ];
  const float invd = 1.0f / (float)d;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int e = 128 * j + 2 * lane;
    const int o0 = (int)(((float)e + 0.5f) * invd), c0 = e - o0 * d;
    const int o1 = (int)(((float)e + 1.5f) * invd), c1 = (e + 1) - o1 * d;
    off0[j] = e < tile_elems ? o0 * S + c0 : -1;
    off1[j] = (e + 1) < tile_elems ? o1 * S + c1 : -1;
  }
  const double logdet = a.chol[(size_t)d * d + d];
  const int rbase = (lane & 15) * S + (lane >> 4);
  M355_TICK(51);
  int tk = 52;
  (void)tk;
  for (; group < ngroups; group += gstride) {
    M355_TICK(tk); ++tk;
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (off0[j] >= 0) st[t * 16 * S + off0[j]] = raw[t][j].x;
        if (off1[j] >= 0) st[t * 16 * S + off1[j]] = raw[t][j].y;
      }
    }
    M355_TICK(tk); ++tk;
    if (group + gstride < ngroups) fetch(group + gstride);  // in flight during the MFMA chains below
    __builtin_amdgcn_wave_barrier();
    double quad[T];
#pragma unroll
    for (int t = 0; t < T; ++t) quad[t] = 0.0;
    if constexpr (REGF) {
      // Y_I = sum_{J <= I} W_IJ Delta_J: the B operand of k-step kk of block J is Delta[obs l&15][16J + (l>>4) + 4kk],
      // i.e. the staged tile read in the D layout; consecutive MFMAs go to different accumulators
#pragma unroll
      for (int t = 0; t < T; ++t) {
        d4_t B[NB], acc[NB];
#pragma unroll
        for (int J = 0; J < NB; ++J) {
#pragma unroll
          for (int r = 0; r < 4; ++r) B[J][r] = st[t * 16 * S + rbase + 16 * J + 4 * r] - mu_s[16 * J + (lane >> 4) + 4 * r];
          acc[J] = d4_t{0.0, 0.0, 0.0, 0.0};
        }
#pragma unroll
        for (int J = 0; J < NB; ++J) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int I = J; I < NB; ++I)
              acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag(I * (I + 1) / 2 + J, kk), B[J][kk], acc[I], 0, 0, 0);
          }
        }
#pragma unroll
        for (int I = 0; I < NB; ++I) {
#pragma unroll
          for (int r = 0; r < 4; ++r) quad[t] = __builtin_fma(acc[I][r], acc[I][r], quad[t]);
        }
      }
    } else {
      d4_t Y[T][NB];
#pragma unroll
      for (int I = 0; I < NB; ++I) {
        d4_t acc[T];
        double mu_r[4];  // the mean of the four components this lane holds in the D layout (0 with per-row means)
#pragma unroll
        for (int r = 0; r < 4; ++r) mu_r[r] = mu_s[16 * I + (lane >> 4) + 4 * r];
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] = st[t * 16 * S + rbase + 16 * I + 4 * r] - mu_r[r];
        }
#pragma unroll
        for (int J = 0; J < I; ++J) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const double f = frag(I * (I + 1) / 2 + J, kk);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f, Y[t][J][kk], acc[t], 0, 0, 0);
          }
        }
        d4_t y[T];
#pragma unroll
        for (int t = 0; t < T; ++t) y[t] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double f = frag(I * (I + 1) / 2 + I, kk);
#pragma unroll
          for (int t = 0; t < T; ++t) y[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f, acc[t][kk], y[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
          Y[t][I] = y[t];
#pragma unroll
          for (int r = 0; r < 4; ++r) quad[t] = __builtin_fma(y[t][r], y[t][r], quad[t]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    M355_TICK(tk); ++tk;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      // lanes l, l+16, l+32, l+48 hold the four row groups of observation l & 15
      double q = quad[t];
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const long long r = (group * T + t) * 16 + lane;
      if (lane < 16 && r < a.n) {
        double v = q;
        if (a.mode == 1) {
          v = -0.5 * (a.c0 + logdet + q);  // mvnormal.py:82-83
        } else if (a.mode == 2) {
          v = a.c0 - 0.5 * (a.nu + (double)d) * flog1p(q / a.nu) - 0.5 * logdet;  // mvstudentt.py:62-64
        }
        a.out[r] = v;
      }
    }
  }
}

// ---- Shared covariance, d <= 64: factorisation and solve in ONE launch (round 3) -------------------------------------------
// k_cholesky + k_solve_mfma cost 28 us + 21 us at config 4's stated size (N = 1e5, d = 64): the factorisation is one
// workgroup's latency in front of a solve that could not start.  k_mvn_shared removes the first launch: every workgroup
// (512 threads = two wavefronts per SIMD, one workgroup per CU) factors the covariance itself -- wavefront 0, alone on its
// SIMD, with k_mvn_tiles' register-tile scheme (16x16 tiles in the MFMA A layout, panels swept in row layout, 40 MFMAs
// for the trailing updates; ~16k cycles), while the other seven wait at a barrier with their first observation tiles
// already in flight.  What the solve needs comes straight out of that scheme: a finished tile L_IJ in the A layout IS the
// A operand of  R_I = Delta_I - sum_{J<I} L_IJ Y_J  (negated once), and the inverses of the four 16x16 diagonal blocks
// (Y_I = inv(L_II) R_I) are 16 columns x 4 blocks = 64 independent forward substitutions, one per lane.  The operands
// (NB(NB+1)/2 x 4 doubles per lane: 40 at d = 64) are published through LDS and live in registers for the tile loop,
// which is k_solve_mfma's blocked substitution: a chain of dependent MFMAs per tile, two wavefronts per SIMD filling the
// matrix pipe.  Every workgroup computes the same factor from the same code: results do not depend on the grid.
constexpr int kSharedWaves = 8;
template <int NB> __host__ __device__ constexpr size_t shared_lds_doubles() {
  return 64 * 17 * 2 + 64 + (size_t)NB * (NB + 1) / 2 * 256 + 8 + 16 * NB + (size_t)kSharedWaves * (16 * (16 * NB + 4) + 128);
}
M355_DEV void wave_sync() {  // LDS traffic of ONE wavefront: the hardware keeps its DS operations in order; this keeps the compiler from moving them
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Wavefront 0 of k_mvn_shared: Cholesky of the d x d matrix at `src` (padded to 16 NB with an identity block), operands
// of the solve to fr[(slot(I, J) * 4 + kk) * 64 + lane] (slot = I (I + 1) / 2 + J: -L_IJ for J < I, inv(L_II) for J = I,
// A layout: element [16 I + (lane & 15)][16 J + (lane >> 4) + 4 kk]), logdet = 2 sum log diag to misc[0].
#ifndef M355_SHARED_VARIANT
#define M355_SHARED_VARIANT 0
#endif
// `after_load` runs once the loads of the covariance are in flight (the caller's first observation tile: one memory
// latency for both).
template <int NB, class AfterLoad>
M355_DEV void shared_factor(const double* __restrict__ src, int d, double* __restrict__ Pm, double* __restrict__ Dg,
                            double* __restrict__ rd, double* __restrict__ fr, double* __restrict__ misc, AfterLoad after_load) {
  const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
  M355_TICK(0);
  const int prow = lane < 16 * NB ? lane : 16 * NB - 1;
  d4_t T[NB][NB];
  if (d == 16 * NB) {
    const int loff = r16 * d + g;
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#pragma unroll
      for (int I = J; I < NB; ++I) {
        const double* base = src + (size_t)(16 * I) * d + 16 * J;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) T[I][J][kk] = base[loff + 4 * kk];
      }
    }
  } else {
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#pragma unroll
      for (int I = J; I < NB; ++I) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int row = 16 * I + r16, col = 16 * J + g + 4 * kk;
          const double v = src[(size_t)(row < d ? row : d - 1) * d + (col < d ? col : d - 1)];
          T[I][J][kk] = (row < d && col < d) ? v : (row == col ? 1.0 : 0.0);
        }
      }
    }
  }
  after_load();
  double mydiag = 1.0;
  M355_TICK(1);
#pragma unroll
  for (int J = 0; J < NB; ++J) {
#pragma unroll
    for (int I = J; I < NB; ++I) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) Pm[(16 * I + r16) * 17 + g + 4 * kk] = T[I][J][kk];
    }
    wave_sync();
    double u[16], l[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) u[c] = Pm[prow * 17 + c];
    M355_TICK(2 + 3 * J);
    // One wavefront, alone on its SIMD: what counts is the LATENCY of the chain pivot -> rsqrt -> l_c -> u_{c+1} ->
    // next pivot, so the update of the next column comes first and the other 14 - c follow, free to fill the
    // stalls of the next column's rsqrt chain (k_mvn_tiles, three wavefronts per SIMD, orders for issue instead).
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      double piv, rinv;
      sqrt_rsqrt<false>(readlane_d(u[c], 16 * J + c), piv, rinv);
      const double lc = u[c] * rinv;
      l[c] = lc;
      rd[16 * J + c] = rinv;  // (wave-uniform value and address: one LDS write, nothing kept in registers)
#if M355_SHARED_VARIANT == 0
#pragma unroll
      for (int c0 = c + 1; c0 < 16; c0 += 4) {
        double m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = (c0 + q < 16) ? readlane_d(lc, 16 * J + (c0 + q < 16 ? c0 + q : 15)) : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (c0 + q < 16) u[c0 + q] = __builtin_fma(-lc, m[q], u[c0 + q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#elif M355_SHARED_VARIANT == 1
#pragma unroll
      for (int c0 = c + 1; c0 < 16; c0 += 4) {
        double m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = (c0 + q < 16) ? readlane_d(lc, 16 * J + (c0 + q < 16 ? c0 + q : 15)) : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (c0 + q < 16) u[c0 + q] = __builtin_fma(-lc, m[q], u[c0 + q]);
        }
      }
#else
      if (c + 1 < 16) u[c + 1] = __builtin_fma(-lc, readlane_d(lc, 16 * J + c + 1), u[c + 1]);
#pragma unroll
      for (int c0 = c + 2; c0 < 16; c0 += 4) {
        double m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = (c0 + q < 16) ? readlane_d(lc, 16 * J + (c0 + q < 16 ? c0 + q : 15)) : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (c0 + q < 16) u[c0 + q] = __builtin_fma(-lc, m[q], u[c0 + q]);
        }
      }
#endif
    }
    M355_TICK(3 + 3 * J);
    wave_sync();
    // row layout back to LDS: lane i holds L[i][16 J + c] (rows of the diagonal block: valid for c <= i - 16 J)
#pragma unroll
    for (int c = 0; c < 16; ++c) Pm[lane * 17 + c] = l[c];
    if (g == J) {  // rows of the diagonal block, for its inverse: Dg[block][row][16] (16-byte aligned rows)
#pragma unroll
      for (int c = 0; c < 16; ++c) Dg[lane * 16 + c] = (c <= r16) ? l[c] : 0.0;
    }
    wave_sync();
    if (g == J) mydiag = Pm[lane * 17 + r16];  // L_ii = l[i - 16 J] of lane i (u_c = pivot^2 there): read back, no select chain

    if (J + 1 < NB) {
      d4_t L[NB], Lneg[NB];
#pragma unroll
      for (int I = J + 1; I < NB; ++I) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) L[I][kk] = Pm[(16 * I + r16) * 17 + g + 4 * kk];
        Lneg[I] = -L[I];
        // a finished tile in the A layout IS the solve's operand: published now, not carried in registers to the end
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fr[((I * (I + 1) / 2 + J) * 4 + kk) * 64 + lane] = Lneg[I][kk];
      }
#pragma unroll
      for (int K = J + 1; K < NB; ++K) {
#pragma unroll
        for (int I = K; I < NB; ++I) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) T[I][K] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lneg[K][kk], L[I][kk], T[I][K], 0, 0, 0);
        }
      }
      wave_sync();
    }
    M355_TICK(4 + 3 * J);
  }
  M355_TICK(20);
  {
    double lg = flog(mydiag);  // identity padding: log 1 = 0; a NaN pivot -> NaN
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);
    if (lane == 0) misc[0] = 2.0 * lg;
  }
  wave_sync();
  // inverse of the diagonal blocks: lane (block g, column r16) runs one forward substitution on e_c
  //   X[c][c] = 1 / L[c][c],  X[r][c] = -(sum_{k=c}^{r-1} L[r][k] X[k][c]) / L[r][r]
  {
    // (x[k] = 0 for k < c by construction, so the sums need no select; the rows of L arrive by 16-byte broadcast reads
    // that do not depend on x: the chain is one multiply-add and one multiply per row)
    const int blk = g < NB ? g : NB - 1, c = r16;
    const d2_t* Lb = reinterpret_cast<const d2_t*>(Dg + (16 * blk) * 16);
    const double* rdb = rd + 16 * blk;
    double x[16];
    int dep = 0;  // always 0: see below
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double acc = 0.0;
#pragma unroll
      for (int k2 = 0; 2 * k2 < r; ++k2) {
        const d2_t lv = Lb[r * 8 + k2 + dep];
        acc = __builtin_fma(lv.x, x[2 * k2], acc);
        if (2 * k2 + 1 < r) acc = __builtin_fma(lv.y, x[2 * k2 + 1], acc);
      }
      const double rdr = rdb[r];
      x[r] = (r == c) ? rdr : (r > c ? -acc * rdr : 0.0);
      // The reads do not depend on x: left alone the compiler issues all 64 of them (256 VGPRs: spills) in front of the
      // chain, and neither fences nor scheduling barriers held it back.  Every fourth row the (zero) offset of the
      // next rows' reads passes through an empty asm together with x[r], which makes them wait for it: at most ~30 reads in flight.
      if ((r & 3) == 3) asm volatile("" : "+v"(dep), "+v"(x[r]));
    }
    wave_sync();
    if (g < NB) {  // X back to LDS, row-major with the odd stride of Pm (read below in the A layout)
#pragma unroll
      for (int r = 0; r < 16; ++r) Pm[(16 * blk + r) * 17 + c] = x[r];
    }
    wave_sync();
  }
  M355_TICK(21);
#pragma unroll
  for (int I = 0; I < NB; ++I) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fr[((I * (I + 1) / 2 + I) * 4 + kk) * 64 + lane] = Pm[(16 * I + r16) * 17 + g + 4 * kk];
  }
}

template <int NB, bool MU_ROWS>
__global__ __launch_bounds__(64 * kSharedWaves) void k_mvn_shared(const double* __restrict__ cov, const MfmaArgs a) {
  constexpr int NBLK = NB * (NB + 1) / 2;
  constexpr int S = 16 * NB + 4;  // staging row stride (as k_solve_mfma)
  constexpr int NJ = 2 * NB;      // 16-byte loads per lane per tile
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Pm = lds;                     // 64 x 17: wavefront 0's panel buffer (row layout), then the inverses of the diagonal blocks
  double* Dg = Pm + 64 * 17;            // 64 x 16 (+ padding): rows of the diagonal blocks
  double* rd = Dg + 64 * 17;            // 64: 1 / L_ii
  double* fr = rd + 64;                 // NBLK x 4 x 64 operands of the solve
  double* misc = fr + NBLK * 256;       // [0] logdet
  double* mu_s = misc + 8;              // 16 NB (zero padded)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* st = mu_s + 16 * NB + wave * (16 * S + 128);  // 16 staged rows + a dump slot (16 bytes) per lane
  const int d = a.d;
  const int tile_elems = 16 * d;
  const long long total = a.n * (long long)d;
  const long long ntiles = (a.n + 15) / 16;
  // This workgroup's tiles: blockIdx.x + k gridDim.x, k < t_cnt (neighbouring workgroups read neighbouring tiles: the
  // grid sweeps one window of memory).  Wavefronts 1-7 take three tiles each up front (one staged in LDS, two in
  // registers: all in flight while wavefront 0 factors), wavefront 0 one; what is left goes tile by tile to whoever
  // asks first (an LDS counter), so the late wavefront 0 does not set the pace.
  const long long t_lo = blockIdx.x, t_step = gridDim.x;
  const int t_cnt = (int)((ntiles - t_lo + t_step - 1) / t_step);
  int* next_tile = reinterpret_cast<int*>(misc + 1);
  constexpr int kUpFront = 3 * (kSharedWaves - 1) + 1;
  d2_t raw0[NJ], raw1[NJ];
  auto fetch = [&](d2_t (&raw)[NJ], int k) {  // k: index within the range (k < t_cnt)
    const long long base = (t_lo + k * t_step) * tile_elems;
    if (base + tile_elems <= total) {
      // a whole tile (all but the last): straight-line loads.  Lanes past the tile (d < 16 NB) read element 0 of it and
      // never use the value: with a branch per load the compiler serialised the eight loads (a wait for all outstanding
      // loads in front of each: the tail path's narrower load aliased its destination), one memory latency EACH.
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int e = 128 * j + 2 * lane;
        const bool in = e < tile_elems;
        const long long gi = base + (in ? e : 0);
        d2_t v = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(a.x + gi));
        if (MU_ROWS) v -= __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(a.mu + gi));
        raw[j] = v;  // (stage() never stores the lanes past the tile: no select here, which would wait for the load)
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int e = 128 * j + 2 * lane;
      const long long gi = base + e;
      d2_t v = {0.0, 0.0};
      if (e < tile_elems) {
        if (gi + 1 < total) {
          v = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(a.x + gi));
          if (MU_ROWS) v -= __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(a.mu + gi));
        } else if (gi < total) {
          v.x = a.x[gi];
          if (MU_ROWS) v.x -= a.mu[gi];
        }
      }
      raw[j] = v;
    }
  };
  auto grab = [&]() -> int {  // the next tile of the range, the same value in every lane (>= t_cnt: none left)
    int k = 0;
    if (lane == 0) k = atomicAdd(next_tile, 1);
    return __builtin_amdgcn_readfirstlane(k);
  };
  // (the mean is fetched by the LAST wavefront: wavefront 0's first memory operations are the loads of the covariance)
  if (wave == kSharedWaves - 1 && lane < 16 * NB) mu_s[lane] = (!MU_ROWS && lane < d) ? a.mu[lane] : 0.0;
  for (int e = lane; e < 16 * S; e += 64) st[e] = 0.0;  // padding columns stay zero for the whole kernel
  int off0[NJ], off1[NJ];
  const float invd = 1.0f / (float)d;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {  // flat element of a tile -> staging offset (e / d through a float reciprocal: exact here)
    const int e = 128 * j + 2 * lane;
    const int o0 = (int)(((float)e + 0.5f) * invd), c0 = e - o0 * d;
    const int o1 = (int)(((float)e + 1.5f) * invd), c1 = (e + 1) - o1 * d;
    off0[j] = e < tile_elems ? o0 * S + c0 : 16 * S + 2 * lane;  // (past the tile: the lane's dump slot -- no branch per store)
    off1[j] = (e + 1) < tile_elems ? o1 * S + c1 : 16 * S + 2 * lane + 1;
  }
  const bool pairs = (d & 1) == 0;  // even d: the two elements of a load are neighbours in one row (16-byte aligned: S is even)
  auto stage = [&](const d2_t (&raw)[NJ]) {
    if (pairs) {  // one 16-byte store per load: half the LDS instructions, and no bank is hit more often than it must be
#pragma unroll
      for (int j = 0; j < NJ; ++j) *reinterpret_cast<d2_t*>(st + off0[j]) = raw[j];
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        st[off0[j]] = raw[j].x;
        st[off1[j]] = raw[j].y;
      }
    }
  };
  // kA: the tile in the staging buffer, kB: the one in raw0, kC: the one in raw1 (indices within the range; >= t_cnt:
  // none).  The staged tile goes to LDS BEFORE the factorisation (wavefront 0: behind the loads of the covariance): its
  // load overlaps the factorisation, and wavefront 0 holds no observation registers while it needs them all.
  int kA, kB, kC;
  __builtin_amdgcn_wave_barrier();
  if (wave != 0) {
    kA = wave - 1;
    kB = kA + (kSharedWaves - 1);
    kC = kB + (kSharedWaves - 1);
    // (three tiles' loads issued back to back: B travels in raw1 until A has been staged out of raw0)
    if (kA < t_cnt) fetch(raw0, kA);
    if (kB < t_cnt) fetch(raw1, kB);
    if (kA < t_cnt) stage(raw0);
    if (kC < t_cnt) fetch(raw0, kC);
  } else {
    kA = kUpFront - 1;
    if (lane == 0) *next_tile = kUpFront;  // (read by the others after the barrier below)
    shared_factor<NB>(cov, d, Pm, Dg, rd, fr, misc, [&]() {
      if (kA < t_cnt) {
        fetch(raw0, kA);
        stage(raw0);
      }
    });
    kB = grab();
    kC = grab();
    if (kB < t_cnt) fetch(raw0, kB);
    if (kC < t_cnt) fetch(raw1, kC);
  }
  M355_TICK(22);
  __syncthreads();
  M355_TICK(23);
  if (wave != 0) {  // B to raw0, C to raw1 (long arrived: register moves)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const d2_t t = raw0[j];
      raw0[j] = raw1[j];
      raw1[j] = t;
    }
  }
  double F[NBLK * 4];
#pragma unroll
  for (int b = 0; b < NBLK * 4; ++b) F[b] = fr[b * 64 + lane];
  const double logdet = misc[0];
  const int rbase = (lane & 15) * S + (lane >> 4);
  [[maybe_unused]] int tick_i = 0;
  auto solve_tile = [&](int k) {  // the tile in the staging buffer: blocked forward substitution on the matrix pipe
    const long long tile = t_lo + k * t_step;
    __builtin_amdgcn_wave_barrier();
    M355_TICKW(24 + 4 * tick_i);
    double quad = 0.0;
    d4_t Y[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      d4_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = st[rbase + 16 * I + 4 * r] - mu_s[16 * I + (lane >> 4) + 4 * r];
#pragma unroll
      for (int J = 0; J < I; ++J) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(F[(I * (I + 1) / 2 + J) * 4 + kk], Y[J][kk], acc, 0, 0, 0);
      }
      d4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) y = __builtin_amdgcn_mfma_f64_16x16x4f64(F[(I * (I + 1) / 2 + I) * 4 + kk], acc[kk], y, 0, 0, 0);
      Y[I] = y;
#pragma unroll
      for (int r = 0; r < 4; ++r) quad = __builtin_fma(y[r], y[r], quad);
    }
    __builtin_amdgcn_wave_barrier();
    M355_TICKW(25 + 4 * tick_i);
    quad += __shfl_xor(quad, 16, 64);  // lanes l, l+16, l+32, l+48 hold the four row groups of observation l & 15
    quad += __shfl_xor(quad, 32, 64);
    const long long r = tile * 16 + lane;
    if (lane < 16 && r < a.n) {
      double v = quad;
      if (a.mode == 1) {
        v = -0.5 * (a.c0 + logdet + quad);  // mvnormal.py:82-83
      } else if (a.mode == 2) {
        v = a.c0 - 0.5 * (a.nu + (double)d) * flog1p(quad / a.nu) - 0.5 * logdet;  // mvstudentt.py:62-64
      }
      a.out[r] = v;
    }
    M355_TICKW(26 + 4 * tick_i);
    ++tick_i;
  };
  // Two tiles per trip, so that the roles of raw0 / raw1 alternate without register copies (a copy would wait for the
  // load it copies): behind every MFMA chain the loads of the next two tiles stay in flight.
  while (kA < t_cnt) {
    solve_tile(kA);                       // st = A, raw0 = B, raw1 = C
    if (kB >= t_cnt) break;
    stage(raw0);                          // (after this tile's reads of the buffer: one wavefront, program order)
    M355_TICKW(23 + 4 * tick_i);
    const int kD = kC < t_cnt ? grab() : t_cnt;
    if (kD < t_cnt) fetch(raw0, kD);
    solve_tile(kB);                       // st = B, raw1 = C, raw0 = D
    if (kC >= t_cnt) break;
    stage(raw1);
    M355_TICKW(23 + 4 * tick_i);
    const int kE = kD < t_cnt ? grab() : t_cnt;
    if (kE < t_cnt) fetch(raw1, kE);
    kA = kC;                              // st = C, raw0 = D, raw1 = E
    kB = kD;
    kC = kE;
  }
}

// out[b] = scale * sum_{j<m} v[b*m + j] + offset[b % noff]   (matrixnormal's Frobenius norm and normaliser)
__global__ __launch_bounds__(256) void k_rowsum(const double* __restrict__ v, long long nb, int m, double scale,
                                                const double* __restrict__ offset, long long noff,
                                                double* __restrict__ out) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  double s = 0.0;
  for (int j = 0; j < m; ++j) s += v[b * m + j];
  out[b] = __builtin_fma(scale, s, offset[b % noff]);
}

// offset[b] = -(0.5*m*n*log(2 pi) + 0.5*n*logdet_U[b % nU] + 0.5*m*logdet_V[b % nV])  (matrixnormal.py:92)
__global__ void k_matn_norm(const double* __restrict__ cholU, long long nU, int m, const double* __restrict__ cholV,
                            long long nV, int n, long long nb, double* __restrict__ offset) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const double ldU = cholU[(size_t)(b % nU) * chol_stride(m) + (size_t)m * m + m];
  const double ldV = cholV[(size_t)(b % nV) * chol_stride(n) + (size_t)n * n + n];
  offset[b] = -(0.5 * m * n * 1.8378770664093454836 + 0.5 * n * ldU + 0.5 * m * ldV);
}

}  // namespace m355

namespace {
using m355::fail;
int launched() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? M355_OK : fail(M355_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
}
int run_cholesky(const double* cov, long long ncov, int d, double* ws, double* frags, hipStream_t s) {
  const int threads = m355::kCholThreads;
  const int dp = 16 * m355::nblk16(d);
  const int nb = m355::nblk16(d);
  const size_t ninv = nb <= m355::kFullInverseMaxNB ? (size_t)nb * (nb + 1) / 2 : (size_t)nb;  // inverse blocks kept in LDS
  const size_t lds = ((size_t)dp * (dp + 1) + 272 + ninv * 256 + dp) * sizeof(double);
  if (lds > 64 * 1024) {  // up to ~148 KiB of the CU's 160 KiB LDS for d = 128
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(m355::k_cholesky),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(M355_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(m355::k_cholesky, dim3((unsigned)ncov), dim3(threads), lds, s, cov, d, ws, frags);
  return launched();
}
template <int NB> int launch_tiles(const double* cov, long long n, int d, const m355::FusedArgs& fa, hipStream_t s) {
  hipLaunchKernelGGL(m355::k_mvn_tiles<NB>, dim3((unsigned)n), dim3(64), 0, s, cov, d, fa);
  return launched();
}
int run_batched(const double* cov, long long n, int d, const m355::FusedArgs& fa, hipStream_t s) {
  switch (m355::nblk16(d)) {
    case 1: return launch_tiles<1>(cov, n, d, fa, s);
    case 2: return launch_tiles<2>(cov, n, d, fa, s);
    case 3: return launch_tiles<3>(cov, n, d, fa, s);
    default: return launch_tiles<4>(cov, n, d, fa, s);
  }
}
template <int NB, bool MU_ROWS> int launch_mfma(const m355::MfmaArgs& a, hipStream_t s) {
  constexpr bool REGF = NB <= m355::kFullInverseMaxNB;  // W = L^-1 blocks in registers (80 VGPRs at NB = 4)
  constexpr int T = 1;  // tiles per wavefront step
  const size_t lds = ((REGF ? 0 : (size_t)NB * (NB + 1) / 2 * 256) + 16 * NB + (size_t)m355::kMfmaWaves * T * 16 * (16 * NB + 4)) * sizeof(double);
  auto kern = m355::k_solve_mfma<NB, MU_ROWS, T, REGF>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(M355_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const long long ngroups = ((a.n + 15) / 16 + T - 1) / T;
  long long blocks = (ngroups + m355::kMfmaWaves - 1) / m355::kMfmaWaves;
  // one round of resident workgroups (every wavefront pays the fragment / staging prologue once and then walks its
  // groups): workgroups per CU from the occupancy calculator (registers and LDS decide: 1 at NB = 4, up to 4 at NB = 1)
  // (remembered per device ordinal: occupancy and CU count belong to the device that is current)
  static std::atomic<long long> resident_of[64];
  int dev_now = 0;
  if (hipGetDevice(&dev_now) != hipSuccess || dev_now < 0 || dev_now >= 64) dev_now = 0;
  long long resident = resident_of[dev_now].load(std::memory_order_relaxed);
  if (resident == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * m355::kMfmaWaves, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_now) != hipSuccess || cus < 1) cus = 256;
    (void)hipGetLastError();
    resident = (long long)per_cu * cus;
    resident_of[dev_now].store(resident, std::memory_order_relaxed);
  }
  long long cap = resident;
  if (const char* e = getenv("M355_MVN_BLOCKS")) {  // A/B knob for grid sweeps; ignored unless a positive count
    const long long v = atoll(e);
    if (v >= 1) cap = v;
  }
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * m355::kMfmaWaves), lds, s, a);
  return launched();
}
// k_mvn_shared: one workgroup per CU at most (its wavefront 0 wants a SIMD to itself while it factors), 8 tiles per round
template <int NB, bool MU_ROWS> int launch_shared(const double* cov, const m355::MfmaArgs& a, hipStream_t s) {
  auto kern = m355::k_mvn_shared<NB, MU_ROWS>;
  const size_t lds = m355::shared_lds_doubles<NB>() * sizeof(double);
  static std::atomic<unsigned char> state[64];  // per device: 0 = not asked, 1 = ready, 2 = refused
  static std::atomic<int> cus_of[64];
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(M355_ERR_LAUNCH, "hipGetDevice failed");
  if (state[dev].load(std::memory_order_acquire) == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    cus_of[dev].store(cus);
    const bool ok = lds <= 64 * 1024 ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    (void)hipGetLastError();
    state[dev].store(ok ? 1 : 2, std::memory_order_release);
  }
  if (state[dev].load(std::memory_order_acquire) != 1) return fail(M355_ERR_LAUNCH, "k_mvn_shared: %zu bytes of LDS refused", lds);
  const long long ntiles = (a.n + 15) / 16;
  long long blocks = (ntiles + m355::kSharedWaves - 1) / m355::kSharedWaves;
  const long long cap = cus_of[dev].load();
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * m355::kSharedWaves), lds, s, cov, a);
  return launched();
}
template <bool MU_ROWS> int run_shared(const double* cov, const m355::MfmaArgs& a, hipStream_t s) {
  switch (m355::nblk16(a.d)) {
    case 1: return launch_shared<1, MU_ROWS>(cov, a, s);
    case 2: return launch_shared<2, MU_ROWS>(cov, a, s);
    case 3: return launch_shared<3, MU_ROWS>(cov, a, s);
    default: return launch_shared<4, MU_ROWS>(cov, a, s);
  }
}
template <bool MU_ROWS> int run_mfma(const m355::MfmaArgs& a, hipStream_t s) {
  switch (m355::nblk16(a.d)) {
    case 1: return launch_mfma<1, MU_ROWS>(a, s);
    case 2: return launch_mfma<2, MU_ROWS>(a, s);
    case 3: return launch_mfma<3, MU_ROWS>(a, s);
    case 4: return launch_mfma<4, MU_ROWS>(a, s);
    case 5: return launch_mfma<5, MU_ROWS>(a, s);
    case 6: return launch_mfma<6, MU_ROWS>(a, s);
    case 7: return launch_mfma<7, MU_ROWS>(a, s);
    default: return launch_mfma<8, MU_ROWS>(a, s);
  }
}
// M355_MVN_TWO_LAUNCH=1: the round-2 path (k_cholesky + k_solve_mfma) for d <= 64 too -- A/B measurements and tests
bool two_launch_path() {
  static const bool on = [] { const char* e = getenv("M355_MVN_TWO_LAUNCH"); return e && e[0] == '1'; }();
  return on;
}
int run_solve(const m355::SolveArgs& a, bool shared, hipStream_t s) {
  const long long blocks = (a.n + m355::kSolveThreads - 1) / m355::kSolveThreads;
  const size_t lds = (size_t)a.d * m355::kSolveThreads * sizeof(double);
  if (shared) hipLaunchKernelGGL(m355::k_solve<true>, dim3((unsigned)blocks), dim3(m355::kSolveThreads), lds, s, a);
  else hipLaunchKernelGGL(m355::k_solve<false>, dim3((unsigned)blocks), dim3(m355::kSolveThreads), lds, s, a);
  return launched();
}
}  // namespace

extern "C" size_t m355_mvn_workspace_bytes(int d, int64_t ncov) {
  // the factors, plus the MFMA operand fragments of a shared covariance
  return ((size_t)ncov * m355::chol_stride(d) + (ncov == 1 ? m355::frag_doubles(d) : 0)) * sizeof(double);
}

extern "C" int m355_mvn_logpdf(int kind, int64_t n, int d, const double* x, const double* mu, int64_t mu_stride,
                               const double* cov, int64_t ncov, int64_t cov_div, int64_t cov_mod, double nu,
                               double* out, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace m355;
  if (kind != 0 && kind != 1) return fail(M355_ERR_INVALID, "kind must be 0 (mvnormal) or 1 (mvstudentt)");
  if (d < 1 || d > kMaxDim) return fail(M355_ERR_UNSUPPORTED, "dimension %d outside 1..%d", d, kMaxDim);
  if (n < 0 || ncov < 1 || cov_div < 1 || cov_mod < 1 || cov_mod > ncov) return fail(M355_ERR_INVALID, "bad n / ncov / cov index");
  if (n == 0) return M355_OK;
  if (!x || !mu || !cov || !out || !workspace) return fail(M355_ERR_INVALID, "NULL pointer argument");
  if (workspace_bytes < m355_mvn_workspace_bytes(d, ncov))
    return fail(M355_ERR_WORKSPACE, "workspace of %zu bytes is smaller than the required %zu", workspace_bytes,
                 m355_mvn_workspace_bytes(d, ncov));
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool mfma = ncov == 1 && al16(x) && (mu_stride == 0 || (mu_stride == d && al16(mu)));
  double* frags = mfma ? ws + chol_stride(d) : nullptr;
  const double lognorm_n0 = (double)d * 1.8378770664093454836;
  if (d <= 64 && ncov == n && cov_div == 1 && cov_mod == n && n > 1 && n <= 0x7fffffffLL) {
    // one covariance per observation: fused factorisation + solve, one wavefront per pair, no factor written
    const double lt = kind == 1 ? lgamma((nu + d) / 2.0) - lgamma(nu / 2.0) - 0.5 * d * log(nu * 3.14159265358979323846) : 0.0;
    FusedArgs fa{x, mu, (long long)mu_stride, out, kind == 0 ? 1 : 2, nu, kind == 0 ? lognorm_n0 : lt, (long long)n};
    return run_batched(cov, n, d, fa, s);
  }
  if (mfma && d <= 64 && !two_launch_path()) {
    // shared covariance, d <= 64: factorisation and solve in one launch (k_mvn_shared); the workspace is not touched
    const double lt = kind == 1 ? lgamma((nu + d) / 2.0) - lgamma(nu / 2.0) - 0.5 * d * log(nu * 3.14159265358979323846) : 0.0;
    MfmaArgs m{x, mu, nullptr, nullptr, out, (long long)n, d, kind == 0 ? 1 : 2, nu, kind == 0 ? lognorm_n0 : lt};
    return mu_stride == 0 ? run_shared<false>(cov, m, s) : run_shared<true>(cov, m, s);
  }
  int rc = run_cholesky(cov, ncov, d, ws, frags, s);
  if (rc != M355_OK) return rc;
  const double lognorm_n = (double)d * 1.8378770664093454836;  // k*log(2 pi)
  // gammaln((nu+k)/2) - gammaln(nu/2) - 0.5*k*log(nu*pi)   (mvstudentt.py:62)
  const double lognorm_t = kind == 1 ? lgamma((nu + d) / 2.0) - lgamma(nu / 2.0) - 0.5 * d * log(nu * 3.14159265358979323846) : 0.0;
  if (mfma) {
    MfmaArgs m{x, mu, frags, ws, out, (long long)n, d, kind == 0 ? 1 : 2, nu, kind == 0 ? lognorm_n : lognorm_t};
    return mu_stride == 0 ? run_mfma<false>(m, s) : run_mfma<true>(m, s);
  }
  SolveArgs a{};
  a.x = x;
  a.mu = mu;
  a.inner = 1;
  a.s1 = d;
  a.s2 = 0;
  a.cs = 1;
  a.mu_s1 = mu_stride;
  a.mu_s2 = 0;
  a.mu_cs = 1;
  a.chol = ws;
  a.l_div = cov_div;
  a.l_mod = cov_mod;
  a.y_out = nullptr;
  a.out = out;
  a.n = n;
  a.d = d;
  a.mode = kind == 0 ? 1 : 2;
  a.nu = nu;
  a.c0 = kind == 0 ? lognorm_n : lognorm_t;
  return run_solve(a, ncov == 1, s);
}

extern "C" size_t m355_matrixnormal_workspace_bytes(int64_t nbatch, int m, int n, int64_t nrow, int64_t ncol) {
  return ((size_t)nrow * m355::chol_stride(m) + (size_t)ncol * m355::chol_stride(n) + (size_t)nbatch * m * n +
          (size_t)nbatch * m + (size_t)nbatch) * sizeof(double);
}

// X: nbatch x m x n (row-major, contiguous); mu: m x n shared (mu_batched = 0) or nbatch x m x n;
// rowcov: nrow x m x m, colcov: ncol x n x n with nrow, ncol in {1, nbatch}.
extern "C" int m355_matrixnormal_logpdf(int64_t nbatch, int m, int n, const double* X, const double* mu,
                                        int mu_batched, const double* rowcov, int64_t nrow, const double* colcov,
                                        int64_t ncol, double* out, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  using namespace m355;
  if (m < 1 || n < 1 || m > kMaxDim || n > kMaxDim) return fail(M355_ERR_UNSUPPORTED, "matrix dimensions outside 1..%d", kMaxDim);
  if (nbatch < 0 || (nrow != 1 && nrow != nbatch) || (ncol != 1 && ncol != nbatch)) return fail(M355_ERR_INVALID, "bad batch sizes");
  if (nbatch == 0) return M355_OK;
  if (!X || !mu || !rowcov || !colcov || !out || !workspace) return fail(M355_ERR_INVALID, "NULL pointer argument");
  if (workspace_bytes < m355_matrixnormal_workspace_bytes(nbatch, m, n, nrow, ncol)) return fail(M355_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* wsU = static_cast<double*>(workspace);
  double* wsV = wsU + (size_t)nrow * chol_stride(m);
  double* Y = wsV + (size_t)ncol * chol_stride(n);
  double* rowq = Y + (size_t)nbatch * m * n;
  double* off = rowq + (size_t)nbatch * m;
  int rc = run_cholesky(rowcov, nrow, m, wsU, nullptr, s);
  if (rc != M355_OK) return rc;
  rc = run_cholesky(colcov, ncol, n, wsV, nullptr, s);
  if (rc != M355_OK) return rc;
  // Y = L_U^-1 (X - M): the n columns of each m x n matrix are observations of dimension m
  SolveArgs a{};
  a.x = X;
  a.mu = mu;
  a.inner = n;
  a.s1 = (long long)m * n;
  a.s2 = 1;
  a.cs = n;
  a.mu_s1 = mu_batched ? (long long)m * n : 0;
  a.mu_s2 = 1;
  a.mu_cs = n;
  a.chol = wsU;
  a.l_div = n;
  a.l_mod = nrow;
  a.y_out = Y;
  a.out = nullptr;
  a.n = nbatch * n;
  a.d = m;
  a.mode = 0;
  rc = run_solve(a, nrow == 1, s);
  if (rc != M355_OK) return rc;
  // Z = L_V^-1 Y^T: the m rows of Y are observations of dimension n; per-row sum of squares
  (void)hipMemsetAsync(off, 0, sizeof(double), s);  // a zero "mu"
  SolveArgs b{};
  b.x = Y;
  b.mu = off;
  b.inner = m;
  b.s1 = (long long)m * n;
  b.s2 = n;
  b.cs = 1;
  b.mu_s1 = 0;
  b.mu_s2 = 0;
  b.mu_cs = 0;
  b.chol = wsV;
  b.l_div = m;
  b.l_mod = ncol;
  b.y_out = nullptr;
  b.out = rowq;
  b.n = nbatch * m;
  b.d = n;
  b.mode = 0;
  rc = run_solve(b, ncol == 1, s);
  if (rc != M355_OK) return rc;
  // -0.5 * ||Z||_F^2 - log_norm   (matrixnormal.py:90-94); `off` is rewritten after the solve read it
  const unsigned blocks = (unsigned)((nbatch + 255) / 256);
  hipLaunchKernelGGL(k_matn_norm, dim3(blocks), dim3(256), 0, s, wsU, nrow, m, wsV, ncol, n, nbatch, off);
  hipLaunchKernelGGL(k_rowsum, dim3(blocks), dim3(256), 0, s, rowq, nbatch, m, -0.5, off, nbatch, out);
  return launched();
}
