// rvs kernels and their extern "C" entry point (declaration: include/m355_dist.h).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "m355_elemwise.h"
#include "m355_rng.h"

namespace m355 {

typedef long long ll2_t __attribute__((ext_vector_type(2)));

M355_DEV void store_draw(void* out, int odt, long long i, double v) {
  if (odt == M355_I64) static_cast<long long*>(out)[i] = (long long)v;
  else if (odt == M355_F32) static_cast<float*>(out)[i] = (float)v;
  else static_cast<double*>(out)[i] = v;
}

// One draw per thread per step; parameters indexed like elementwise operands.  Output-only
// stream: 8 B per draw, coalesced 512 B per wavefront store.
// first: the call's element that out[0] holds (a pair kernel leaves an odd last element to this one: its parameters are
// those of element `first`, not of element 0)
// SCALARS: every parameter is a scalar, read once in front of the loop -- a sampler's parameter-only set-up (von Mises' s,
// the Marsaglia-Tsang constants, ...) is then loop-invariant and hoisted; the arithmetic of a draw is unchanged.
template <int D, bool SCALARS = false>
__global__ __launch_bounds__(kBlock) void k_rvs(GenericOps ops, void* __restrict__ out, int odt, long long n,
                                                unsigned long long seed, unsigned stream,
                                                unsigned long long offset, long long first) {
  log_table_init();  // the samplers use flog_tab
  const bool small = first + n <= 0xffffffffll;
  double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
  if constexpr (SCALARS) {
#pragma unroll
    for (int q = 0; q < Dist<D>::NP; ++q) pp[q] = ops.p[q].scalar;
  }
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    if constexpr (!SCALARS) {
#pragma unroll
      for (int q = 0; q < Dist<D>::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)(first + i), small);
    }
    store_draw(out, odt, i, sample_one<D>(offset + (unsigned long long)i, seed, stream, pp));
  }
}

// Families drawn from ONE uniform or ONE normal per element (m355_rng.h one_uniform / one_normal), any parameter layout:
// a thread turns the Philox block of elements 2k, 2k + 1 into both draws and stores 16 bytes.  Same draws as k_rvs.
// SCALARS: every parameter is a scalar -- they are read once in front of the loop, so whatever a family derives from its
// parameters alone (truncatednormal's cdf at the bounds, geometric's log1p(-p), ...) is loop-invariant and the compiler
// hoists it; the arithmetic per draw is unchanged (same draws as the general variant).
template <int D, bool SCALARS = false>
__global__ __launch_bounds__(kBlock) void k_rvs_pairs(GenericOps ops, void* __restrict__ out, int odt, long long npairs,
                                                      unsigned long long seed, unsigned stream, unsigned long long pair_offset) {
  log_table_init();
  const bool small = 2 * npairs <= 0xffffffffll;
  const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  double pa[kMaxParams] = {0.0, 0.0, 0.0, 0.0}, pb[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
  if constexpr (SCALARS) {
#pragma unroll
    for (int q = 0; q < Dist<D>::NP; ++q) pa[q] = pb[q] = ops.p[q].scalar;
  }
  for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < npairs; k += (long long)gridDim.x * kBlock) {
    const unsigned long long c = pair_offset + (unsigned long long)k;
    const Philox b = philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, stream, k0, k1);
    if constexpr (!SCALARS) {
#pragma unroll
      for (int q = 0; q < Dist<D>::NP; ++q) {
        pa[q] = fetch_operand(ops.p[q], 2ull * (unsigned long long)k, small);
        pb[q] = fetch_operand(ops.p[q], 2ull * (unsigned long long)k + 1ull, small);
      }
    }
    double v0, v1;
    if constexpr (one_uniform<D>()) {
      v0 = from_uniform<D>(u53(b.c[0], b.c[1]), pa);
      v1 = from_uniform<D>(u53(b.c[2], b.c[3]), pb);
    } else {
      double z0, z1;
      normal_pair(u53(b.c[0], b.c[1]), u53(b.c[2], b.c[3]), z0, z1);
      v0 = from_normal<D>(z0, pa);
      v1 = from_normal<D>(z1, pb);
    }
    if (odt == M355_I64) {
      ll2_t w;
      w.x = (long long)v0;
      w.y = (long long)v1;
      __builtin_nontemporal_store(w, reinterpret_cast<ll2_t*>(static_cast<long long*>(out) + 2 * k));
    } else {
      d2_t w;
      w.x = v0;
      w.y = v1;
      __builtin_nontemporal_store(w, reinterpret_cast<d2_t*>(static_cast<double*>(out) + 2 * k));
    }
  }
}

// Normal with scalar parameters, fp64 output: the ziggurat of m355_rng.h with the rare path taken out of the loop.  A
// wavefront covers 64 pairs of elements per trip: one Philox block per lane gives the candidates of elements 2k and
// 2k + 1, the threshold test accepts 99.6 % of them at once (16-byte store when both halves pass; the table {threshold,
// x_i} of 16 KiB in LDS), and a candidate that needs the wedge or the tail goes to the wavefront's queue in LDS as
// (element, attempt number).  When 64 have gathered, each lane takes one entry, recomputes its block and runs zig_slow;
// a rejected candidate comes back to the queue with the next attempt number.  Same draws as sample_std_normal (k_rvs).
// Measured (tools/zig_probe.hip, 1e9 draws): the fast path alone 1.49 ms; with 256 layers (98.5 %) 86 % of the trips had
// an open candidate and the queue cost 0.6 ms, hence the larger table.
#ifdef M355_ZIG_BLOCK  // tools/zig_probe.hip
constexpr int kZigBlock = M355_ZIG_BLOCK;
#else
constexpr int kZigBlock = 512;
#endif
constexpr int kZigQueue = 192;  // at most 63 waiting + 128 pushed by one trip; a test batch pushes back at most the 64 it took
// ARR: mu and / or sigma are arrays (any operand layout of m355_elemwise); the fast path fetches them for both halves.
template <bool ARR>
__global__ __launch_bounds__(kZigBlock) void k_rvs_normal_zig(double* __restrict__ out, long long npairs, const DevOperand pmu,
                                                              const DevOperand psigma, unsigned long long seed,
                                                              unsigned stream, unsigned long long pair_offset) {
  const double mu = pmu.scalar, sigma = psigma.scalar;
  const bool small = 2 * npairs <= 0xffffffffll;
  (void)small;
  __shared__ double tab[2 * kZigLayers];
  __shared__ unsigned long long q[kZigBlock / 64][kZigQueue];  // element index relative to `out` | attempt number << 56
  for (int i = threadIdx.x; i < 2 * kZigLayers; i += kZigBlock) tab[i] = kZigTab[i];
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (kZigBlock / 64) + wv;
  const long long nwaves = (long long)gridDim.x * (kZigBlock / 64);
  const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  int queued = 0;  // wave-uniform
  auto push = [&](bool want, unsigned long long entry) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(want);
    if (want) q[wv][queued + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = entry;
    queued += __popcll(m);
  };
  // entries [first, first + cnt) of the queue, one per lane (the caller has already taken them off `queued`)
  auto test = [&](int first, int cnt) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned long long entry = 0;
    if (lane < cnt) entry = q[wv][first + lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every entry is read before push() writes over the same slots
    bool again = false;
    if (lane < cnt) {
      const unsigned long long e = entry & 0x00ffffffffffffffull;
      const unsigned a = (unsigned)(entry >> 56);
      const ZigCtr z{2ull * pair_offset + e, stream, k0, k1};
      unsigned hi, lo;
      zig_bits(z, a, hi, lo);
      const int layer = zig_layer(hi);
      double x;
      bool ok = zig_fast(hi, lo, tab[2 * layer], tab[2 * layer + 1], x);
      if (!ok) ok = zig_slow(z, a, layer, x);
      if (ok || a + 1u >= (unsigned)kMaxAttempts) {
        if constexpr (ARR) out[e] = __builtin_fma(fetch_operand(psigma, e, small), x, fetch_operand(pmu, e, small));
        else out[e] = __builtin_fma(sigma, x, mu);
      } else {
        again = true;
      }
    }
    push(again, entry + (1ull << 56));
  };
  // one trip: pair k of this lane (tail: only where `act`)
  auto trip = [&](long long k, auto tail, bool act) {
    const unsigned long long c = pair_offset + (unsigned long long)k;
    const Philox p = philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, stream, k0, k1);
    const int l0 = zig_layer(p.c[0]), l1 = zig_layer(p.c[2]);
    double x0, x1;
    bool f0 = zig_fast(p.c[0], p.c[1], tab[2 * l0], tab[2 * l0 + 1], x0);
    bool f1 = zig_fast(p.c[2], p.c[3], tab[2 * l1], tab[2 * l1 + 1], x1);
    d2_t v;
    double* dst = out + 2 * k;
    bool live = true;  // (the last, partial trip: the lanes past the end neither store nor queue, but stay in the wave-wide
                       // steps below -- `queued` must remain one value per wavefront)
    if constexpr (decltype(tail)::value) {
      live = act;
      f0 |= !act;
      f1 |= !act;
    }
    if constexpr (ARR) {
      const unsigned long long e = live ? 2ull * (unsigned long long)k : 0ull;
      v.x = __builtin_fma(fetch_operand(psigma, e, small), x0, fetch_operand(pmu, e, small));
      v.y = __builtin_fma(fetch_operand(psigma, e + 1ull, small), x1, fetch_operand(pmu, e + 1ull, small));
    } else {
      v.x = __builtin_fma(sigma, x0, mu);
      v.y = __builtin_fma(sigma, x1, mu);
    }
    // one queue entry per lane in the common case; a lane with both halves open (2e-4) adds the second one
#if defined(M355_ZIG_VARIANT) && (M355_ZIG_VARIANT == 1 || M355_ZIG_VARIANT == 2)  // tools/zig_probe.hip: no queue
    if (M355_ZIG_VARIANT == 1) __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(dst));
    else *reinterpret_cast<d2_t*>(dst) = v;
    return;
#endif
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(!f0), m1 = __builtin_amdgcn_ballot_w64(!f1);
    const unsigned long long any = m0 | m1;
#if defined(M355_ZIG_VARIANT) && M355_ZIG_VARIANT == 3  // tools/zig_probe.hip: no holes (timing only: the later store races)
    __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(dst));
    if (!(f0 & f1)) {
      q[wv][queued + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(any >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)any, 0u))] =
          2ull * (unsigned long long)k + (f0 ? 1ull : 0ull);
    }
    queued += __popcll(any);
    if ((m0 & m1) != 0ull) push(!f0 & !f1, 2ull * (unsigned long long)k + 1ull);
    return;
#endif
    if (f0 & f1) {
      if (live) __builtin_nontemporal_store(v, reinterpret_cast<d2_t*>(dst));
    } else {
      if (f0) dst[0] = v.x;
      if (f1) dst[1] = v.y;
      q[wv][queued + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(any >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)any, 0u))] =
          2ull * (unsigned long long)k + (f0 ? 1ull : 0ull);
    }
    queued += __popcll(any);
    if ((m0 & m1) != 0ull) push(!f0 & !f1, 2ull * (unsigned long long)k + 1ull);
  };
  const long long nfull = npairs >> 6;
  for (long long t = wave; t < nfull; t += nwaves) {
    trip(t * 64 + lane, std::false_type{}, true);
    while (queued >= 64) {  // (back to at most 63 before the next trip adds up to 128)
      queued -= 64;
      test(queued, 64);
    }
  }
  if ((npairs & 63) != 0 && wave == nfull % nwaves) trip(nfull * 64 + lane, std::true_type{}, nfull * 64 + lane < npairs);
  while (queued > 0) {
    const int cnt = queued < 64 ? queued : 64;
    queued -= cnt;
    test(queued, cnt);
  }
}

// Rejection samplers with wavefront-level compaction (PTRS Poisson, BTRS binomial; scalar
// parameters).  A wavefront owns a contiguous chunk of the output.  Every round each lane makes ONE
// attempt for the element it holds; lanes whose attempt was accepted store the draw and take the next
// unassigned element of the chunk (rank among the freed lanes from a ballot + mbcnt prefix count),
// so all 64 lanes keep proposing instead of idling until the slowest lane of a fixed group of 64
// elements is accepted.  A draw still depends only on (seed, stream, element index, attempt
// number), never on the lane that computed it: the output is identical to k_rvs.
// Round 3: the squeeze tests decide ~86 % of the attempts; the rest need three logarithms, and with one such lane in a
// wavefront all 64 paid for them every round.  An undecided attempt now goes to the wavefront's queue in LDS
// (element, attempt number, candidate, us, v) and frees its lane; when 64 have gathered, the logarithmic test runs
// once for all of them, each lane on one queued attempt.  An element whose queued attempt fails goes to a second list
// (element, next attempt number), from which the refill step serves free lanes before it hands out new elements.
constexpr int kQueueCap = 128;  // at most 63 waiting + 64 pushed in one round
template <int D>
__global__ __launch_bounds__(kBlock) void k_rvs_compact(void* __restrict__ out, int odt, long long n, double p0, double p1,
                                                        unsigned long long seed, unsigned stream,
                                                        unsigned long long offset) {
  __shared__ long long q_idx[kBlock / 64][kQueueCap];
  __shared__ unsigned q_att[kBlock / 64][kQueueCap];
  __shared__ double q_k[kBlock / 64][kQueueCap], q_us[kBlock / 64][kQueueCap], q_v[kBlock / 64][kQueueCap];
  __shared__ long long r_idx[kBlock / 64][kQueueCap];  // elements to retry (at most 64 waiting + 64 from one test batch)
  __shared__ unsigned r_att[kBlock / 64][kQueueCap];
  log_table_init();
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (kBlock / 64) + wv;
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  const long long chunk = ((n + nwaves - 1) / nwaves + 63) & ~63ll;
  const long long w1 = (wave + 1) * chunk < n ? (wave + 1) * chunk : n;
  long long next = wave * chunk;  // wave-uniform: first unassigned element
  if (next >= w1) return;
  PtrsK pk;
  BtrsK bk;
  if constexpr (D == M355_POISSON) pk = ptrs_setup(p0);
  else bk = btrs_setup(p0, p1);
  auto block = [&](long long idx, unsigned attempt, double& u1, double& v) {
    const unsigned long long c = offset + (unsigned long long)idx;
    const Philox ph = philox4x32_10((unsigned)c, (unsigned)(c >> 32), attempt, stream, (unsigned)seed, (unsigned)(seed >> 32));
    u1 = u53(ph.c[0], ph.c[1]);
    v = u53(ph.c[2], ph.c[3]);
  };
  auto finish = [&](long long idx, double k) {
    if constexpr (D == M355_BINOMIAL) {
      if (bk.flip) k = bk.n - k;
    }
    store_draw(out, odt, idx, k);
  };
  int retry = 0;  // wave-uniform: entries of the retry list
  // the logarithmic test for `cnt` queued attempts (entries [first, first + cnt) of the queue), one per lane
  auto drain = [&](int first, int cnt) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool again = false;
    long long qi = -1;
    unsigned qa = 0;
    if (lane < cnt) {
      qi = q_idx[wv][first + lane];
      qa = q_att[wv][first + lane];
      const double k = q_k[wv][first + lane];
      bool ok;
      if constexpr (D == M355_POISSON) ok = ptrs_slow(pk, k, q_us[wv][first + lane], q_v[wv][first + lane]);
      else ok = btrs_slow(bk, k, q_us[wv][first + lane], q_v[wv][first + lane]);
      if (ok || qa >= (unsigned)kMaxAttempts) finish(qi, k);  // (qa: the number of the NEXT attempt)
      else again = true;
    }
    const unsigned long long am = __ballot(again);
    if (again) {
      const int at = retry + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
      r_idx[wv][at] = qi;
      r_att[wv][at] = qa;
    }
    retry += __popcll(am);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  long long idx = -1;  // element held by this lane (-1: none)
  unsigned attempt = 0;
  int queued = 0;  // wave-uniform
  for (;;) {
    // refill the free lanes: first from the retry list, then from the chunk
    const bool freed = idx < 0;
    const unsigned long long fm = __ballot(freed);
    const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
    const int nfree = __popcll(fm);
    const int from_retry = nfree < retry ? nfree : retry;
    if (freed) {
      if (below < from_retry) {
        idx = r_idx[wv][retry - 1 - below];
        attempt = r_att[wv][retry - 1 - below];
      } else {
        const long long cand = next + (below - from_retry);
        if (cand < w1) {
          idx = cand;
          attempt = 0;
        }
      }
    }
    retry -= from_retry;
    next += nfree - from_retry;
    if (next > w1) next = w1;
    const bool active = idx >= 0;
    if (__ballot(active) == 0ull) {
      // nothing held and nothing new: test what is queued; failures come back through the retry list
      // (no lane holds anything, so the retry list is empty: at most 127 queued, 64 at a time)
      if (queued > 64) {
        queued -= 64;
        drain(queued, 64);
        continue;
      }
      if (queued > 0) {
        drain(0, queued);
        queued = 0;
      }
      if (retry == 0) break;
      continue;
    }
    bool slow = false;
    double k = 0.0, us = 0.0, v = 0.0;
    const bool can_queue = queued + 64 <= kQueueCap;  // wave-uniform: room for one more round of undecided attempts
    if (active) {
      double u1;
      block(idx, attempt, u1, v);
      attempt += 1;
      int f;
      if constexpr (D == M355_POISSON) f = ptrs_fast(pk, u1, v, k, us);
      else f = btrs_fast(bk, u1, v, k, us);
      if (f == 2 && !can_queue) {  // (both lists full: practically never; decide here, as the per-lane loop does)
        bool ok;
        if constexpr (D == M355_POISSON) ok = ptrs_slow(pk, k, us, v);
        else ok = btrs_slow(bk, k, us, v);
        f = ok ? 1 : 0;
      }
      if (f == 1 || (f == 0 && attempt >= (unsigned)kMaxAttempts)) {
        finish(idx, k);
        idx = -1;
      }
      slow = f == 2;
    }
    const unsigned long long sm = __ballot(slow);
    if (sm != 0ull) {
      if (slow) {
        const int at = queued + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u));
        q_idx[wv][at] = idx;
        q_att[wv][at] = attempt;
        q_k[wv][at] = k;
        q_us[wv][at] = us;
        q_v[wv][at] = v;
        idx = -1;
      }
      queued += __popcll(sm);
      if (queued >= 64 && retry + 64 <= kQueueCap) {
        queued -= 64;
        drain(queued, 64);
      }
    }
  }
}

// PTRS / BTRS with parameters given as ARRAYS (a Poisson regression's mu per observation): the same lane refill, every lane
// carrying the constants of the element it holds (ptrs_setup / btrs_setup when it takes the element).  Elements the
// rejection samplers do not serve (mu < 10, n min(p, q) < 10, degenerate or invalid parameters) are drawn on the spot by
// sample_one's own branch and the lane takes another element next round.  The per-lane loop of k_rvs had a wavefront wait
// for its slowest lane (~3 rounds per 64 draws at 86 % acceptance); a draw still depends only on (seed, stream, element,
// attempt): bit-identical to k_rvs.
template <int D>
__global__ __launch_bounds__(kBlock) void k_rvs_compact_arr(GenericOps ops, void* __restrict__ out, int odt, long long n,
                                                            unsigned long long seed, unsigned stream, unsigned long long offset) {
  static_assert(D == M355_POISSON || D == M355_BINOMIAL, "PTRS / BTRS");
  log_table_init();
  const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  const long long chunk = ((n + nwaves - 1) / nwaves + 63) & ~63ll;
  const long long w1 = (wave + 1) * chunk < n ? (wave + 1) * chunk : n;
  long long next = wave * chunk;  // wave-uniform: first unassigned element
  if (next >= w1) return;
  const bool small = n <= 0xffffffffll;
  PtrsK pk{};
  BtrsK bk{};
  long long idx = -1;  // element held by this lane (-1: none)
  unsigned attempt = 0;
  for (;;) {
    const bool freed = idx < 0;
    const unsigned long long fm = __ballot(freed);
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
    if (freed) {
      const long long cand = next + below;
      if (cand < w1) {
        double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < Dist<D>::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)cand, small);
        bool rejection;
        if constexpr (D == M355_POISSON) {
          rejection = pp[0] >= 10.0 && pp[0] < kInf;
        } else {
          const double pm = pp[1] > 0.5 ? 1.0 - pp[1] : pp[1];
          rejection = pp[0] >= 0.0 && pp[0] < kInf && pp[1] > 0.0 && pp[1] < 1.0 && pp[0] * pm >= 10.0;
        }
        if (rejection) {
          idx = cand;
          attempt = 0;
          if constexpr (D == M355_POISSON) pk = ptrs_setup(pp[0]);
          else bk = btrs_setup(pp[0], pp[1]);
        } else {
          store_draw(out, odt, cand, sample_one<D>(offset + (unsigned long long)cand, seed, stream, pp));
        }
      }
    }
    next += __popcll(fm);
    if (next > w1) next = w1;
    const bool active = idx >= 0;
    if (__ballot(active) == 0ull) {
      if (next >= w1) break;  // chunk exhausted and every lane has stored its draw
      continue;
    }
    if (active) {
      const unsigned long long c = offset + (unsigned long long)idx;
      const Philox ph = philox4x32_10((unsigned)c, (unsigned)(c >> 32), attempt, stream, (unsigned)seed, (unsigned)(seed >> 32));
      attempt += 1;
      const double u1 = u53(ph.c[0], ph.c[1]), v = u53(ph.c[2], ph.c[3]);
      double k;
      bool ok;
      if constexpr (D == M355_POISSON) ok = ptrs_attempt(pk, u1, v, k);
      else ok = btrs_attempt(bk, u1, v, k);
      if (ok || attempt >= (unsigned)kMaxAttempts) {
        if constexpr (D == M355_BINOMIAL) {
          if (bk.flip) k = bk.n - k;
        }
        store_draw(out, odt, idx, k);
        idx = -1;
      }
    }
  }
}

// zi_poisson / zi_binomial with parameters given as ARRAYS (zero-inflated regressions): sample_one's chain -- inflation
// uniform, then the Poisson / binomial draw -- with k_rvs_compact_arr's refill.  A lane whose inflation uniform already
// decides "structural zero" stores 0 and takes another element at once; a base draw the rejection samplers do not serve
// is made on the spot by sample_one.  Attempt 0 is the inflation pair, PTRS / BTRS start at attempt 1 as in sample_one:
// bit-identical to the per-lane loop of k_rvs.  (The families with a gamma stage in front -- negativebinomial,
// zi_negativebinomial, betabinomial -- were tried as a lane state machine over the stages: every round then pays for every
// stage's code, 22.6 -> 32.8 ms per 1e9 for negativebinomial and 40 -> 81 ms for betabinomial; they keep the per-lane loop.)
M355_HD bool btrs_serves(double n, double p) {
  const double pm = p > 0.5 ? 1.0 - p : p;
  return n >= 0.0 && n < kInf && p > 0.0 && p < 1.0 && n * pm >= 10.0;
}
template <int D>
__global__ __launch_bounds__(kBlock) void k_rvs_zi_arr(GenericOps ops, void* __restrict__ out, int odt, long long n,
                                                       unsigned long long seed, unsigned stream, unsigned long long offset) {
  static_assert(D == M355_ZI_POISSON || D == M355_ZI_BINOMIAL, "inflation uniform + PTRS / BTRS");
  constexpr bool kBinomial = D == M355_ZI_BINOMIAL;
  log_table_init();
  const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  const long long chunk = ((n + nwaves - 1) / nwaves + 63) & ~63ll;
  const long long w1 = (wave + 1) * chunk < n ? (wave + 1) * chunk : n;
  long long next = wave * chunk;  // wave-uniform: first unassigned element
  if (next >= w1) return;
  const bool small = n <= 0xffffffffll;
  long long idx = -1;  // element held by this lane (-1: none)
  unsigned attempt = 0;
  PtrsK pk{};
  BtrsK bk{};
  for (;;) {
    const bool freed = idx < 0;
    const unsigned long long fm = __ballot(freed);
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
    if (freed) {
      const long long cand = next + below;
      if (cand < w1) {
        double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < Dist<D>::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)cand, small);
        const unsigned long long c = offset + (unsigned long long)cand;
        const Philox pz = philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, stream, (unsigned)seed, (unsigned)(seed >> 32));
        bool served;
        if constexpr (kBinomial) served = btrs_serves(pp[1], pp[2]);
        else served = pp[1] >= 10.0 && pp[1] < kInf;
        if (!(u53(pz.c[0], pz.c[1]) < pp[0])) {  // sample_one: (u1 < psi) ? base : 0
          store_draw(out, odt, cand, 0.0);
        } else if (!served) {
          store_draw(out, odt, cand, sample_one<D>(c, seed, stream, pp));
        } else {
          idx = cand;
          attempt = 1u;
          if constexpr (kBinomial) bk = btrs_setup(pp[1], pp[2]);
          else pk = ptrs_setup(pp[1]);
        }
      }
    }
    next += __popcll(fm);
    if (next > w1) next = w1;
    const bool active = idx >= 0;
    if (__ballot(active) == 0ull) {
      if (next >= w1) break;  // chunk exhausted and every lane has stored its draw
      continue;
    }
    if (active) {
      const unsigned long long c = offset + (unsigned long long)idx;
      const Philox ph = philox4x32_10((unsigned)c, (unsigned)(c >> 32), attempt, stream, (unsigned)seed, (unsigned)(seed >> 32));
      attempt += 1;
      const double u1 = u53(ph.c[0], ph.c[1]), v = u53(ph.c[2], ph.c[3]);
      double k;
      bool ok;
      if constexpr (kBinomial) ok = btrs_attempt(bk, u1, v, k);
      else ok = ptrs_attempt(pk, u1, v, k);
      if (ok || attempt > (unsigned)kMaxAttempts) {
        if constexpr (kBinomial) {
          if (bk.flip) k = bk.n - k;
        }
        store_draw(out, odt, idx, k);
        idx = -1;
      }
    }
  }
}
template <int D> bool try_rvs_zi_arr(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed,
                                     unsigned stream, unsigned long long offset, hipStream_t s, int* rc) {
  if (n < 4096) return false;  // (array parameters, or scalar ones whose window the inversion kernel declined)
  const long long want = (n + 4 * kBlock - 1) / (4 * kBlock);  // >= 256 elements per wavefront
  const int blocks = (int)(want > 8192 ? 8192 : want);
  hipLaunchKernelGGL((k_rvs_zi_arr<D>), dim3(blocks), dim3(kBlock), 0, s, ops, out, odt, n, seed, stream, offset);
  *rc = (int)hipGetLastError();
  return true;
}

// Marsaglia-Tsang samplers with the same wavefront-level compaction (scalar parameters): gamma and
// its one-gamma relatives, beta / betascaled (two gammas: stage 0 then stage 1 with the other shape's constants -- the
// same code), Student t (one stage: the gamma of shape nu / 2, whose block also yields the numerator).  One Philox block
// and one attempt per lane per round; a lane that finishes its element takes the next one of the chunk.  Attempt
// numbers run on across the stages exactly as in sample_one's per-lane loops: the output is identical to k_rvs.
template <int D> struct MtPlan;  // stages, shapes and the last step of sample_one<D>
template <> struct MtPlan<M355_GAMMA> {
  static constexpr int kStages = 1;
  static M355_HD double shape(const double* p, int) { return p[0]; }
  static M355_DEV double finish(const double* p, double, double g) { return g / p[1]; }
};
template <> struct MtPlan<M355_INVERSEGAMMA> {
  static constexpr int kStages = 1;
  static M355_HD double shape(const double* p, int) { return p[0]; }
  static M355_DEV double finish(const double* p, double, double g) { return p[1] / g; }
};
template <> struct MtPlan<M355_CHISQUARED> {
  static constexpr int kStages = 1;
  static M355_HD double shape(const double* p, int) { return 0.5 * p[0]; }
  static M355_DEV double finish(const double*, double, double g) { return 2.0 * g; }
};
template <> struct MtPlan<M355_SCALEDINVERSECHISQUARED> {
  static constexpr int kStages = 1;
  static M355_HD double shape(const double* p, int) { return 0.5 * p[0]; }
  static M355_DEV double finish(const double* p, double, double g) { return (p[0] * p[1]) / (2.0 * g); }
};
template <> struct MtPlan<M355_BETA> {
  static constexpr int kStages = 2;
  static M355_HD double shape(const double* p, int st) { return p[st]; }
  static M355_DEV double finish(const double*, double ga, double gb) { return ga / (ga + gb); }
};
template <> struct MtPlan<M355_BETASCALED> {
  static constexpr int kStages = 2;
  static M355_HD double shape(const double* p, int st) { return p[st]; }
  static M355_DEV double finish(const double* p, double ga, double gb) { return ga / (ga + gb) * (p[3] - p[2]) + p[2]; }
};
template <> struct MtPlan<M355_STUDENTT> {
  static constexpr int kStages = 1;  // the numerator is the accepted attempt's second normal (mt_attempt<true>)
  static M355_HD double shape(const double* p, int) { return 0.5 * p[0]; }
  static M355_DEV double finish(const double* p, double z0, double g) { return p[1] + p[2] * z0 * fsqrt(0.5 * p[0] / g); }
};
template <> struct MtPlan<M355_HALFSTUDENTT> {
  static constexpr int kStages = 1;  // (the same construction, folded)
  static M355_HD double shape(const double* p, int) { return 0.5 * p[0]; }
  static M355_DEV double finish(const double* p, double z0, double g) { return __builtin_fabs(z0 * fsqrt(0.5 * p[0] / g) * p[1]); }
};
template <int D> constexpr bool mt_wants_z() { return D == M355_STUDENTT || D == M355_HALFSTUDENTT; }

struct MtParams {
  double p[kMaxParams];
};

template <int D>
__global__ __launch_bounds__(kBlock) void k_rvs_mt(void* __restrict__ out, int odt, long long n, const MtParams pa,
                                                   unsigned long long seed, unsigned stream, unsigned long long offset) {
  using Plan = MtPlan<D>;
  log_table_init();
  const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  const long long chunk = ((n + nwaves - 1) / nwaves + 63) & ~63ll;
  const long long w1 = (wave + 1) * chunk < n ? (wave + 1) * chunk : n;
  long long next = wave * chunk;  // wave-uniform: first unassigned element
  if (next >= w1) return;
  // a shape below 1: Gamma(a + 1) U^(1/a), the uniform from the block after the accepted attempt (sample_std_gamma)
  const double a0 = Plan::shape(pa.p, 0), a1 = Plan::shape(pa.p, Plan::kStages - 1);
  const MtK k0 = mt_setup(a0 < 1.0 ? a0 + 1.0 : a0);
  const MtK k1 = mt_setup(a1 < 1.0 ? a1 + 1.0 : a1);
  long long idx = -1;  // element held by this lane (-1: none)
  unsigned attempt = 0;
  int stage = 0;
  double hold = 0.0;  // the first stage's result
  for (;;) {
    const bool freed = idx < 0;
    const unsigned long long fm = __ballot(freed);
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
    if (freed) {
      const long long cand = next + below;
      if (cand < w1) {
        idx = cand;
        attempt = 0;
        stage = 0;
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
      MtK kk;
      kk.d = stage == 0 ? k0.d : k1.d;
      kk.c = stage == 0 ? k0.c : k1.c;
      double g, x;
      bool ok;
      if constexpr (mt_wants_z<D>()) ok = mt_attempt<true>(kk, ph, g, x, &hold);
      else ok = mt_attempt(kk, ph, g, x);
      if (ok || attempt >= (unsigned)kMaxAttempts) {
        const double as = stage == 0 ? a0 : a1;
        if (as < 1.0) {
          const Philox pb = philox4x32_10((unsigned)c, (unsigned)(c >> 32), attempt, stream, (unsigned)seed, (unsigned)(seed >> 32));
          attempt += 1;
          g *= exp_(flog_tab(u53(pb.c[0], pb.c[1])) / as);
        }
        if (stage + 1 < Plan::kStages) {
          hold = g;
          stage = 1;
        } else {
          store_draw(out, odt, idx, Plan::finish(pa.p, hold, g));
          idx = -1;
        }
      }
    }
  }
}

// The one-stage Marsaglia-Tsang families with parameters given as ARRAYS (a gamma regression's shape and rate per
// observation): k_rvs_mt's lane refill with the attempt constants and the parameters carried per lane, as
// k_rvs_compact_arr does for PTRS / BTRS.  Elements whose shape is not positive and finite are drawn by sample_one's own
// branch.  Bit-identical to the per-lane loop of k_rvs.
template <int D>
__global__ __launch_bounds__(kBlock) void k_rvs_mt_arr(GenericOps ops, void* __restrict__ out, int odt, long long n,
                                                       unsigned long long seed, unsigned stream, unsigned long long offset) {
  using Plan = MtPlan<D>;
  static_assert(Plan::kStages == 1, "one gamma per draw");
  log_table_init();
  const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  const long long chunk = ((n + nwaves - 1) / nwaves + 63) & ~63ll;
  const long long w1 = (wave + 1) * chunk < n ? (wave + 1) * chunk : n;
  long long next = wave * chunk;  // wave-uniform: first unassigned element
  if (next >= w1) return;
  const bool small = n <= 0xffffffffll;
  long long idx = -1;  // element held by this lane (-1: none)
  unsigned attempt = 0;
  MtK kk{};
  double shape = 1.0, hold = 0.0;
  double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
  for (;;) {
    const bool freed = idx < 0;
    const unsigned long long fm = __ballot(freed);
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
    if (freed) {
      const long long cand = next + below;
      if (cand < w1) {
#pragma unroll
        for (int q = 0; q < Dist<D>::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)cand, small);
        const double a = Plan::shape(pp, 0);
        if (a > 0.0 && a < kInf) {
          idx = cand;
          attempt = 0;
          shape = a;
          kk = mt_setup(a < 1.0 ? a + 1.0 : a);
        } else {
          store_draw(out, odt, cand, sample_one<D>(offset + (unsigned long long)cand, seed, stream, pp));
        }
      }
    }
    next += __popcll(fm);
    if (next > w1) next = w1;
    const bool active = idx >= 0;
    if (__ballot(active) == 0ull) {
      if (next >= w1) break;  // chunk exhausted and every lane has stored its draw
      continue;
    }
    if (active) {
      const unsigned long long c = offset + (unsigned long long)idx;
      const Philox ph = philox4x32_10((unsigned)c, (unsigned)(c >> 32), attempt, stream, (unsigned)seed, (unsigned)(seed >> 32));
      attempt += 1;
      double g, x;
      bool ok;
      if constexpr (mt_wants_z<D>()) ok = mt_attempt<true>(kk, ph, g, x, &hold);
      else ok = mt_attempt(kk, ph, g, x);
      if (ok || attempt >= (unsigned)kMaxAttempts) {
        if (shape < 1.0) {
          const Philox pb = philox4x32_10((unsigned)c, (unsigned)(c >> 32), attempt, stream, (unsigned)seed, (unsigned)(seed >> 32));
          g *= exp_(flog_tab(u53(pb.c[0], pb.c[1])) / shape);
        }
        store_draw(out, odt, idx, Plan::finish(pp, hold, g));
        idx = -1;
      }
    }
  }
}
template <int D> bool try_rvs_mt_arr(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed,
                                     unsigned stream, unsigned long long offset, hipStream_t s, int* rc) {
  if (n < 4096) return false;
  bool any = false;
  for (int q = 0; q < Dist<D>::NP; ++q) any |= ops.p[q].ptr != nullptr;
  if (!any) return false;
  const long long want = (n + 4 * kBlock - 1) / (4 * kBlock);  // >= 256 elements per wavefront
  const int blocks = (int)(want > 8192 ? 8192 : want);
  hipLaunchKernelGGL((k_rvs_mt_arr<D>), dim3(blocks), dim3(kBlock), 0, s, ops, out, odt, n, seed, stream, offset);
  *rc = (int)hipGetLastError();
  return true;
}

// beta / betascaled with both shapes above 1 (scalar parameters): Cheng's BB attempts (m355_rng.h) with the same lane
// refill -- one stage, one block per attempt.  Same draws as sample_beta_bb in k_rvs.
template <int D>
__global__ __launch_bounds__(kBlock) void k_rvs_bb(void* __restrict__ out, int odt, long long n, const MtParams pa,
                                                   unsigned long long seed, unsigned stream, unsigned long long offset) {
  log_table_init();
  const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  const long long chunk = ((n + nwaves - 1) / nwaves + 63) & ~63ll;
  const long long w1 = (wave + 1) * chunk < n ? (wave + 1) * chunk : n;
  long long next = wave * chunk;  // wave-uniform: first unassigned element
  if (next >= w1) return;
  const BbK k = bb_setup(pa.p[0], pa.p[1]);
  long long idx = -1;  // element held by this lane (-1: none)
  unsigned attempt = 0;
  for (;;) {
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
      double x;
      if (bb_attempt(k, ph, x) || attempt >= (unsigned)kMaxAttempts) {
        if constexpr (D == M355_BETASCALED) x = x * (pa.p[3] - pa.p[2]) + pa.p[2];
        store_draw(out, odt, idx, x);
        idx = -1;
      }
    }
  }
}
template <int D> bool try_rvs_bb(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed,
                                 unsigned stream, unsigned long long offset, hipStream_t s, int* rc) {
  if (n < 4096) return false;
  MtParams pa{};
  for (int q = 0; q < Dist<D>::NP; ++q) {
    if (ops.p[q].ptr != nullptr) return false;
    pa.p[q] = ops.p[q].scalar;
  }
  if (!bb_applies(pa.p[0], pa.p[1])) return false;
  const long long want = (n + 4 * kBlock - 1) / (4 * kBlock);  // >= 256 elements per wavefront
  const int blocks = (int)(want > 8192 ? 8192 : want);
  hipLaunchKernelGGL((k_rvs_bb<D>), dim3(blocks), dim3(kBlock), 0, s, out, odt, n, pa, seed, stream, offset);
  *rc = (int)hipGetLastError();
  return true;
}

// von Mises with scalar parameters (kappa >= 1e-8): Best & Fisher's attempts (m355_rng.h) with the lane refill -- the
// acceptance rate falls to 66 % as kappa grows, and the per-lane loop had a wavefront wait for its slowest lane.  The
// sign uniform is the block after the accepted attempt, as in sample_one: same draws as k_rvs.
__global__ __launch_bounds__(kBlock) void k_rvs_vm(void* __restrict__ out, int odt, long long n, double mu, double kappa,
                                                   unsigned long long seed, unsigned stream, unsigned long long offset) {
  log_table_init();
  const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  const long long chunk = ((n + nwaves - 1) / nwaves + 63) & ~63ll;
  const long long w1 = (wave + 1) * chunk < n ? (wave + 1) * chunk : n;
  long long next = wave * chunk;  // wave-uniform: first unassigned element
  if (next >= w1) return;
  const double s = vm_setup(kappa);
  long long idx = -1;  // element held by this lane (-1: none)
  unsigned attempt = 0;
  for (;;) {
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
      double w;
      if (vm_attempt(s, kappa, u53(ph.c[0], ph.c[1]), u53(ph.c[2], ph.c[3]), w) || attempt >= (unsigned)kMaxAttempts) {
        const Philox pf = philox4x32_10((unsigned)c, (unsigned)(c >> 32), attempt, stream, (unsigned)seed, (unsigned)(seed >> 32));
        store_draw(out, odt, idx, vm_finish(w, u53(pf.c[0], pf.c[1]), mu));
        idx = -1;
      }
    }
  }
}
inline bool try_rvs_vm(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed, unsigned stream,
                       unsigned long long offset, hipStream_t s, int* rc) {
  if (n < 4096 || ops.p[0].ptr != nullptr || ops.p[1].ptr != nullptr) return false;
  const double kappa = ops.p[1].scalar;
  if (!(kappa >= 1e-8) || !(kappa < kInf)) return false;  // (uniform limit, invalid or infinite kappa: sample_one's branches)
  const long long want = (n + 4 * kBlock - 1) / (4 * kBlock);  // >= 256 elements per wavefront
  const int blocks = (int)(want > 8192 ? 8192 : want);
  hipLaunchKernelGGL(k_rvs_vm, dim3(blocks), dim3(kBlock), 0, s, out, odt, n, ops.p[0].scalar, kappa, seed, stream, offset);
  *rc = (int)hipGetLastError();
  return true;
}

// beta / betascaled with parameters given as ARRAYS: the BB constants are set up when a lane takes its element; an
// element with a shape at or below 1 (or a degenerate one) is drawn on the spot by sample_one's gamma-ratio branch.
// Bit-identical to the per-lane loop of k_rvs.
template <int D>
__global__ __launch_bounds__(kBlock) void k_rvs_bb_arr(GenericOps ops, void* __restrict__ out, int odt, long long n,
                                                       unsigned long long seed, unsigned stream, unsigned long long offset) {
  log_table_init();
  const long long wave = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kBlock / 64);
  const long long chunk = ((n + nwaves - 1) / nwaves + 63) & ~63ll;
  const long long w1 = (wave + 1) * chunk < n ? (wave + 1) * chunk : n;
  long long next = wave * chunk;  // wave-uniform: first unassigned element
  if (next >= w1) return;
  const bool small = n <= 0xffffffffll;
  long long idx = -1;  // element held by this lane (-1: none)
  unsigned attempt = 0;
  BbK k{};
  double lower = 0.0, width = 1.0;
  for (;;) {
    const bool freed = idx < 0;
    const unsigned long long fm = __ballot(freed);
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(fm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fm, 0u));
    if (freed) {
      const long long cand = next + below;
      if (cand < w1) {
        double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < Dist<D>::NP; ++q) pp[q] = fetch_operand(ops.p[q], (unsigned long long)cand, small);
        if (bb_applies(pp[0], pp[1])) {
          idx = cand;
          attempt = 0;
          k = bb_setup(pp[0], pp[1]);
          if constexpr (D == M355_BETASCALED) {
            lower = pp[2];
            width = pp[3] - pp[2];
          }
        } else {
          store_draw(out, odt, cand, sample_one<D>(offset + (unsigned long long)cand, seed, stream, pp));
        }
      }
    }
    next += __popcll(fm);
    if (next > w1) next = w1;
    const bool active = idx >= 0;
    if (__ballot(active) == 0ull) {
      if (next >= w1) break;  // chunk exhausted and every lane has stored its draw
      continue;
    }
    if (active) {
      const unsigned long long c = offset + (unsigned long long)idx;
      const Philox ph = philox4x32_10((unsigned)c, (unsigned)(c >> 32), attempt, stream, (unsigned)seed, (unsigned)(seed >> 32));
      attempt += 1;
      double x;
      if (bb_attempt(k, ph, x) || attempt >= (unsigned)kMaxAttempts) {
        if constexpr (D == M355_BETASCALED) x = x * width + lower;
        store_draw(out, odt, idx, x);
        idx = -1;
      }
    }
  }
}
template <int D> bool try_rvs_bb_arr(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed,
                                     unsigned stream, unsigned long long offset, hipStream_t s, int* rc) {
  if (n < 4096) return false;
  bool any = false;
  for (int q = 0; q < Dist<D>::NP; ++q) any |= ops.p[q].ptr != nullptr;
  if (!any) return false;
  const long long want = (n + 4 * kBlock - 1) / (4 * kBlock);  // >= 256 elements per wavefront
  const int blocks = (int)(want > 8192 ? 8192 : want);
  hipLaunchKernelGGL((k_rvs_bb_arr<D>), dim3(blocks), dim3(kBlock), 0, s, ops, out, odt, n, seed, stream, offset);
  *rc = (int)hipGetLastError();
  return true;
}

template <int D> bool try_rvs_mt(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed,
                                 unsigned stream, unsigned long long offset, hipStream_t s, int* rc) {
  if (n < 4096) return false;
  MtParams pa{};
  for (int q = 0; q < Dist<D>::NP; ++q) {
    if (ops.p[q].ptr != nullptr) return false;
    pa.p[q] = ops.p[q].scalar;
  }
  for (int st = 0; st < MtPlan<D>::kStages; ++st) {
    const double a = MtPlan<D>::shape(pa.p, st);
    if (!(a > 0.0) || !(a < kInf)) return false;
  }
  const long long want = (n + 4 * kBlock - 1) / (4 * kBlock);  // >= 256 elements per wavefront
  const int blocks = (int)(want > 8192 ? 8192 : want);
  hipLaunchKernelGGL((k_rvs_mt<D>), dim3(blocks), dim3(kBlock), 0, s, out, odt, n, pa, seed, stream, offset);
  *rc = (int)hipGetLastError();
  return true;
}

// Discrete family with scalar parameters: inversion on the cdf tabulated over the family's window (m355_rng.h: inv_draw).
// Every workgroup evaluates the window itself (len <= 2048 exact cdf values over 512 threads) and builds the guide table;
// then one Philox block per lane and trip gives the draws of elements 2c and 2c + 1 of the global stream (16-byte store
// when the call starts at an even element and `out` allows it).  No rejection, no divergence: ~30 operations per draw.
constexpr int kInvBlock = 512;
constexpr int kInvMaxBlocks = 1024;  // (each workgroup pays for the whole table)
template <int D>
__global__ __launch_bounds__(kInvBlock) void k_rvs_inv(void* __restrict__ out, int odt, long long n,
                                                       const typename Dist<D>::K k, double lo, int len,
                                                       unsigned long long seed, unsigned stream, unsigned long long offset) {
  using DD = Dist<D>;
  __shared__ double c[kDTabMax];
  __shared__ unsigned short g[kInvGuide + 2];
  if constexpr (uses_log_table<DD>::value) log_table_init();
  for (int j = threadIdx.x; j < len; j += kInvBlock) c[j] = DD::dtab_cdf(lo + (double)j, k);
  __syncthreads();
  for (int j = threadIdx.x; j <= kInvGuide; j += kInvBlock) g[j] = (unsigned short)(j < kInvGuide ? inv_guide(c, len, j) : len - 1);
  __syncthreads();
  const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  const unsigned long long first = offset >> 1;  // block of the call's first element
  const long long npairs = (long long)(((offset + (unsigned long long)n - 1ull) >> 1) - first) + 1;
  const long long shift = (long long)(offset & 1ull);  // element 2 * pair - shift of `out` is the first half of the pair
  const bool wide = shift == 0 && odt != M355_F32 && (reinterpret_cast<unsigned long long>(out) & 15ull) == 0ull;
  for (long long pr = (long long)blockIdx.x * kInvBlock + threadIdx.x; pr < npairs; pr += (long long)gridDim.x * kInvBlock) {
    const unsigned long long cc = first + (unsigned long long)pr;
    const Philox ph = philox4x32_10((unsigned)cc, (unsigned)(cc >> 32), 0u, stream, k0, k1);
    const double v0 = inv_draw<D>(c, g, len, lo, ph.c[0], ph.c[1], k);
    const double v1 = inv_draw<D>(c, g, len, lo, ph.c[2], ph.c[3], k);
    const long long e0 = 2 * pr - shift, e1 = e0 + 1;
    if (wide && e1 < n) {
      if (odt == M355_I64) {
        ll2_t w;
        w.x = (long long)v0;
        w.y = (long long)v1;
        __builtin_nontemporal_store(w, reinterpret_cast<ll2_t*>(static_cast<long long*>(out) + e0));
      } else {
        d2_t w;
        w.x = v0;
        w.y = v1;
        __builtin_nontemporal_store(w, reinterpret_cast<d2_t*>(static_cast<double*>(out) + e0));
      }
    } else {
      if (e0 >= 0) store_draw(out, odt, e0, v0);
      if (e1 < n) store_draw(out, odt, e1, v1);
    }
  }
}
template <int D> bool try_rvs_inv(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed,
                                  unsigned stream, unsigned long long offset, hipStream_t s, int* rc) {
  using DD = Dist<D>;
  double pp[kMaxParams] = {0.0, 0.0, 0.0, 0.0};
  for (int q = 0; q < DD::NP; ++q) {
    if (ops.p[q].ptr != nullptr) return false;
    pp[q] = ops.p[q].scalar;
  }
  if (n < 1 || !InvFamily<D>::ok(pp)) return false;
  const typename DD::K k = DD::derive(pp);
  double lo = 0.0;
  int len = 0;
  if (!DD::dtab_range(k, lo, len) || len < 1) return false;
  const long long npairs = n / 2 + 1;
  const long long want = (npairs + 4 * kInvBlock - 1) / (4 * kInvBlock);  // >= 8 draws per thread
  const int blocks = (int)(want > kInvMaxBlocks ? kInvMaxBlocks : want);
  hipLaunchKernelGGL((k_rvs_inv<D>), dim3(blocks), dim3(kInvBlock), 0, s, out, odt, n, k, lo, len, seed, stream, offset);
  *rc = (int)hipGetLastError();
  return true;
}

template <int D> int launch_rvs(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed,
                                unsigned stream, unsigned long long offset, hipStream_t s, long long first = 0) {
  const long long want = (n + kBlock - 1) / kBlock;
  const int blocks = (int)(want < 1 ? 1 : (want > kMaxBlocks ? kMaxBlocks : want));
  bool scalars = n >= 4096;  // (a short call gains nothing from the second instantiation)
  for (int q = 0; q < Dist<D>::NP; ++q) scalars &= ops.p[q].ptr == nullptr;
  if (scalars)
    hipLaunchKernelGGL((k_rvs<D, true>), dim3(blocks), dim3(kBlock), 0, s, ops, out, odt, n, seed, stream, offset, first);
  else
    hipLaunchKernelGGL((k_rvs<D>), dim3(blocks), dim3(kBlock), 0, s, ops, out, odt, n, seed, stream, offset, first);
  return (int)hipGetLastError();
}

// k_rvs_pairs where the family and the call allow it (even first element, 16-byte aligned 8-byte output), the odd last
// element and everything else through k_rvs
template <int D> int launch_rvs_any(const GenericOps& ops, void* out, int odt, long long n, unsigned long long seed,
                                    unsigned stream, unsigned long long offset, hipStream_t s) {
  if constexpr (one_uniform<D>() || one_normal<D>()) {
    if ((offset & 1ull) == 0 && aligned(out, 16) && odt != M355_F32 && n >= 2) {
      const long long npairs = n / 2;
      const long long want = (npairs + kBlock - 1) / kBlock;
      const int blocks = (int)(want > kMaxBlocks ? kMaxBlocks : want);
      bool scalars = true;
      for (int q = 0; q < Dist<D>::NP; ++q) scalars &= ops.p[q].ptr == nullptr;
      if (scalars)
        hipLaunchKernelGGL((k_rvs_pairs<D, true>), dim3(blocks), dim3(kBlock), 0, s, ops, out, odt, npairs, seed, stream, offset >> 1);
      else
        hipLaunchKernelGGL((k_rvs_pairs<D>), dim3(blocks), dim3(kBlock), 0, s, ops, out, odt, npairs, seed, stream, offset >> 1);
      const int rc = (int)hipGetLastError();
      if (rc != 0 || (n & 1) == 0) return rc;
      return launch_rvs<D>(ops, static_cast<char*>(out) + 8 * (n - 1), odt, 1, seed, stream, offset + (unsigned long long)(n - 1), s,
                           n - 1);
    }
  }
  return launch_rvs<D>(ops, out, odt, n, seed, stream, offset, s);
}

}  // namespace m355

extern "C" int m355_rvs_impl(int dist, int out_dtype, long long n, const m355::GenericOps* ops, unsigned long long seed,
                             unsigned stream_id, unsigned long long offset, void* out, hipStream_t s) {
  using namespace m355;
  if (dist == M355_NORMAL && out_dtype == M355_F64 && (offset & 1ull) == 0 && aligned(out, 16) && n >= 2) {
    const long long npairs = n / 2;
    const long long want = (npairs + kZigBlock - 1) / kZigBlock;
    const int blocks = (int)(want > kMaxBlocks / 2 ? kMaxBlocks / 2 : want);
    if (ops->p[0].ptr == nullptr && ops->p[1].ptr == nullptr)
      hipLaunchKernelGGL(k_rvs_normal_zig<false>, dim3(blocks), dim3(kZigBlock), 0, s, static_cast<double*>(out), npairs,
                         ops->p[0], ops->p[1], seed, stream_id, offset >> 1);
    else
      hipLaunchKernelGGL(k_rvs_normal_zig<true>, dim3(blocks), dim3(kZigBlock), 0, s, static_cast<double*>(out), npairs,
                         ops->p[0], ops->p[1], seed, stream_id, offset >> 1);
    int rc = (int)hipGetLastError();
    if (rc != 0 || (n & 1) == 0) return rc;
    // odd tail element
    return launch_rvs<M355_NORMAL>(*ops, static_cast<double*>(out) + (n - 1), out_dtype, 1, seed, stream_id,
                                   offset + (unsigned long long)(n - 1), s, n - 1);
  }
  {  // discrete families with scalar parameters whose window fits: inversion, for every n (one stream per parameter set)
    int rc = 0;
    bool done = false;
    switch (dist) {
#define M355_INV_CASE(ID) \
  case ID: done = try_rvs_inv<ID>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc); break;
      M355_INV_LIST(M355_INV_CASE)
#undef M355_INV_CASE
      default: break;
    }
    if (done) return rc;
  }
  const bool scalar2 = ops->p[0].ptr == nullptr && ops->p[1].ptr == nullptr;
  const double q0 = ops->p[0].scalar, q1 = ops->p[1].scalar;
  const bool ptrs = dist == M355_POISSON && ops->p[0].ptr == nullptr && q0 >= 10.0 && q0 < kInf;
  const bool btrs = dist == M355_BINOMIAL && scalar2 && q0 >= 0.0 && q0 < kInf && q1 > 0.0 && q1 < 1.0 &&
                    q0 * (q1 > 0.5 ? 1.0 - q1 : q1) >= 10.0;
  if ((dist == M355_POISSON || dist == M355_BINOMIAL) && n >= 4096 &&
      (ops->p[0].ptr != nullptr || (dist == M355_BINOMIAL && ops->p[1].ptr != nullptr))) {
    const long long want = (n + 4 * kBlock - 1) / (4 * kBlock);  // >= 256 elements per wavefront
    const int blocks = (int)(want > 8192 ? 8192 : want);
    if (dist == M355_POISSON)
      hipLaunchKernelGGL((k_rvs_compact_arr<M355_POISSON>), dim3(blocks), dim3(kBlock), 0, s, *ops, out, out_dtype, n, seed,
                         stream_id, offset);
    else
      hipLaunchKernelGGL((k_rvs_compact_arr<M355_BINOMIAL>), dim3(blocks), dim3(kBlock), 0, s, *ops, out, out_dtype, n, seed,
                         stream_id, offset);
    return (int)hipGetLastError();
  }
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
  {
    int rc = 0;
    bool done = false;
    switch (dist) {
#define M355_MT1_CASE(ID)                                                                           \
  case ID:                                                                                          \
    done = try_rvs_mt<ID>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc) ||              \
           try_rvs_mt_arr<ID>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc);            \
    break;
      M355_MT1_CASE(M355_GAMMA)
      M355_MT1_CASE(M355_INVERSEGAMMA)
      M355_MT1_CASE(M355_CHISQUARED)
      M355_MT1_CASE(M355_SCALEDINVERSECHISQUARED)
      M355_MT1_CASE(M355_STUDENTT)
      M355_MT1_CASE(M355_HALFSTUDENTT)
#undef M355_MT1_CASE
      case M355_BETA:
        done = try_rvs_bb<M355_BETA>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc) ||
               try_rvs_mt<M355_BETA>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc) ||
               try_rvs_bb_arr<M355_BETA>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc);
        break;
      case M355_BETASCALED:
        done = try_rvs_bb<M355_BETASCALED>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc) ||
               try_rvs_mt<M355_BETASCALED>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc) ||
               try_rvs_bb_arr<M355_BETASCALED>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc);
        break;
      case M355_VONMISES: done = try_rvs_vm(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc); break;
      case M355_ZI_POISSON: done = try_rvs_zi_arr<M355_ZI_POISSON>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc); break;
      case M355_ZI_BINOMIAL: done = try_rvs_zi_arr<M355_ZI_BINOMIAL>(*ops, out, out_dtype, n, seed, stream_id, offset, s, &rc); break;
      default: break;
    }
    if (done) return rc;
  }
  switch (dist) {
#define M355_RVS_CASE(ID, name, np) \
  case ID: return launch_rvs_any<ID>(*ops, out, out_dtype, n, seed, stream_id, offset, s);
    M355_DIST_LIST(M355_RVS_CASE)
#undef M355_RVS_CASE
  }
  return -1000;
}
