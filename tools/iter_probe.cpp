// Host-side convergence probe: iteration counts of the series / continued fractions behind
// logcdf / ppf on the config-2 input distributions.  g++ -O2 -mfma -std=c++17 -DM355_ITER_PROBE
//   tools/iter_probe.cpp -o /tmp/iter_probe
#define M355_ITER_PROBE 1
#include <cstdio>
#include <random>
#include <vector>
#include <algorithm>
namespace m355 { long long g_m355_iters = 0, g_m355_outer = 0; }
#include "../distributions_amd/csrc/m355_dists.h"
using namespace m355;
template <int D, int FN> void run(const char* name, const std::vector<double>& xs, const double* p) {
  auto k = Dist<D>::derive(p);
  std::vector<long long> it;
  long long tot = 0;
  double worst = 0;
  m355::g_m355_outer = 0;
  for (double x : xs) {
    m355::g_m355_iters = 0;
    volatile double r = Dist<D>::template eval<FN>(x, k);
    (void)r;
    it.push_back(m355::g_m355_iters);
    tot += m355::g_m355_iters;
  }
  std::vector<long long> s = it;
  std::sort(s.begin(), s.end());
  size_t wi = std::max_element(it.begin(), it.end()) - it.begin();
  worst = xs[wi];
  printf("%-18s mean %.1f  p50 %lld  p99 %lld  p99.99 %lld  max %lld (x=%.17g)  outer iterations/element %.2f\n", name, (double)tot / xs.size(),
         s[s.size() / 2], s[s.size() * 99 / 100], s[(size_t)(s.size() * 0.9999)], s.back(), worst, (double)m355::g_m355_outer / xs.size());
}
int main() {
  const int n = 400000;
  std::mt19937_64 g(1);
  std::normal_distribution<double> N(0, 1);
  std::gamma_distribution<double> G(2.5, 1 / 1.3), G2(2.0, 1.0), G5(5.0, 1.0), C(2.5, 2.0);
  std::uniform_real_distribution<double> U(1e-12, 1 - 1e-12);
  std::vector<double> xn(n), xg(n), xb(n), xt(n), q(n);
  for (int i = 0; i < n; ++i) {
    xn[i] = N(g); xg[i] = G(g);
    double a = G2(g), b = G5(g); xb[i] = a / (a + b);
    xt[i] = 2.0 * N(g) / std::sqrt(C(g) / 5.0);
    q[i] = U(g);
  }
  const double pg[3] = {2.5, 1.3, 0}, pb[3] = {2.0, 5.0, 0}, pt[3] = {5.0, 0.0, 2.0};
  run<M355_GAMMA, M355_LOGCDF>("gamma.logcdf", xg, pg);
  run<M355_GAMMA, M355_LOGSF>("gamma.logsf", xg, pg);
  run<M355_GAMMA, M355_PPF>("gamma.ppf", q, pg);
  run<M355_BETA, M355_LOGCDF>("beta.logcdf", xb, pb);
  run<M355_BETA, M355_LOGSF>("beta.logsf", xb, pb);
  run<M355_BETA, M355_PPF>("beta.ppf", q, pb);
  run<M355_STUDENTT, M355_LOGCDF>("studentt.logcdf", xt, pt);
  run<M355_STUDENTT, M355_PPF>("studentt.ppf", q, pt);
  return 0;
}
