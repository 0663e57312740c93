/* CPU oracle, C leg: TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/restatement.py).
 *
 * Single-threaded C loops that do per element what the fused Elemwise C loop PyTensor generates
 * for the reference expressions does (parameter-only subexpressions are 0-d graph nodes there and
 * are hoisted here the same way).  Used only by bench.py's `cpu_baseline` leg (kind "port") and by
 * tests/test_oracle_pinning.py, which checks it against oracle/restatement.py.
 * Reference lines: normal.py:79-80, gamma.py:77-82 (with quirk Q2), beta.py:102-108 of
 * /root/reference/pytensor_distributions.
 */
#include <math.h>

void oc_normal_logpdf(long n, const double* x, double mu, double sigma, double* out) {
  const double c = log(sqrt(2.0 * M_PI)) + log(sigma);
  for (long i = 0; i < n; ++i) {
    const double z = (x[i] - mu) / sigma;
    out[i] = -0.5 * (z * z) - c;
  }
}

void oc_gamma_logpdf(long n, const double* x, double alpha, double beta, double* out) {
  const double c = alpha * log(beta) - lgamma(alpha);
  const double am1 = alpha - 1.0;
  for (long i = 0; i < n; ++i) {
    const double xi = x[i];
    const double t = (am1 == 0.0) ? 0.0 : am1 * log(xi);
    out[i] = (xi < 0.0) ? -INFINITY : (t - xi * beta) + c;
  }
}

void oc_beta_logpdf(long n, const double* x, double alpha, double beta, double* out) {
  const double lb = lgamma(alpha) + lgamma(beta) - lgamma(alpha + beta);
  const double am1 = alpha - 1.0, bm1 = beta - 1.0;
  for (long i = 0; i < n; ++i) {
    const double xi = x[i];
    const double t = ((am1 == 0.0) ? 0.0 : am1 * log(xi)) + ((bm1 == 0.0) ? 0.0 : bm1 * log(1.0 - xi));
    out[i] = (xi < 0.0 || xi > 1.0) ? -INFINITY : t - lb;
  }
}

/* pt.sum(...) of the above: sequential float64 accumulation, as a CAReduce loop does. */
double oc_sum(long n, const double* v) {
  double s = 0.0;
  for (long i = 0; i < n; ++i) s += v[i];
  return s;
}
