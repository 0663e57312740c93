"""Generates csrc/m355_temmecoef.h: Temme's uniform asymptotic expansion of the incomplete gamma functions for large a
(N. M. Temme, "The asymptotic expansion of the incomplete gamma functions", SIAM J. Math. Anal. 10 (1979); DLMF 8.12),
the branch SciPy's gammainc / gammaincc take for a > 20 near x ~ a (what pt.gammainc / pt.gammaincc evaluate in
gamma.py:65-74, poisson.py:71-75).

    lambda = x / a,  mu = lambda - 1,  eta^2 / 2 = mu - log(1 + mu),  sign(eta) = sign(mu)
    Q(a, x) = erfc( eta sqrt(a/2)) / 2 + S,    P(a, x) = erfc(-eta sqrt(a/2)) / 2 - S
    S = exp(-a eta^2 / 2) / sqrt(2 pi a) * sum_k c_k(eta) a^-k

Derivation of the coefficients used here.  dQ/d eta = -sqrt(a / 2 pi) exp(-a eta^2/2) f(eta) / Gamma*(a) with
f = eta / mu and Gamma*(a) = Gamma(a) / (sqrt(2 pi / a) a^a e^-a).  Writing 1 / Gamma*(a) = sum_k g_k a^-k and matching
powers of 1/a in the differential equation for S gives
    c_0 = 1/mu - 1/eta,     c_k = ( c_{k-1}'(eta) + g_k eta / mu ) / eta.
Everything is done in exact rational arithmetic on truncated power series in eta: mu(eta) by Lagrange inversion of
eta(mu) = mu sqrt(2 sum_{j>=2} (-mu)^(j-2) / j), g_k from exp(-sum_j B_2j / (2j (2j-1) a^(2j-1))).

The truncation (number of c_k, degree of each) is chosen below and checked in double-precision Horner arithmetic
against 60-digit mpmath over a in [kTemmeMinA, 1e15], |x - a| <= kTemmeMaxSigma a; the maxima are printed into the header.
usage: python tools/gen_temme_coeffs.py > distributions_amd/csrc/m355_temmecoef.h
"""
import math
import sys
from fractions import Fraction as Fr

import mpmath as mp

N = 48  # working series order in eta
MIN_A = 25.0
MAX_SIGMA = 0.3
# degree of c_k(eta) kept, k = 0 .. K-1
DEGREES = [17, 15, 13, 11, 9, 7, 5, 3, 1]


def s_mul(a, b, n=N):
    out = [Fr(0)] * (n + 1)
    for i, x in enumerate(a[: n + 1]):
        if x == 0:
            continue
        for j, y in enumerate(b[: n + 1 - i]):
            out[i + j] += x * y
    return out


def s_inv(a, n=N):  # 1 / a, a[0] != 0
    out = [Fr(0)] * (n + 1)
    out[0] = 1 / a[0]
    for k in range(1, n + 1):
        s = Fr(0)
        for j in range(1, k + 1):
            if j < len(a):
                s += a[j] * out[k - j]
        out[k] = -s / a[0]
    return out


def s_sqrt(a, n=N):  # sqrt(a), a[0] == 1
    out = [Fr(0)] * (n + 1)
    out[0] = Fr(1)
    for k in range(1, n + 1):
        s = Fr(0)
        for j in range(1, k):
            s += out[j] * out[k - j]
        out[k] = (a[k] - s) / 2
    return out


def s_exp(a, n):  # exp(a), a[0] == 0
    out = [Fr(0)] * (n + 1)
    out[0] = Fr(1)
    for k in range(1, n + 1):
        s = Fr(0)
        for j in range(1, k + 1):
            s += j * a[j] * out[k - j]
        out[k] = s / k
    return out


def bernoulli(m):
    B = [Fr(0)] * (m + 1)
    B[0] = Fr(1)
    for n in range(1, m + 1):
        B[n] = -sum(Fr(math.comb(n + 1, k)) * B[k] for k in range(n)) / (n + 1)
    return B


def coefficients():
    # eta / mu = s(mu) = sqrt(2 sum_{j>=2} (-mu)^(j-2) / j)
    s = s_sqrt([2 * Fr((-1) ** j, j + 2) for j in range(N + 1)])
    # Lagrange inversion: mu = sum_n b_n eta^n, b_n = [w^(n-1)] s(w)^-n / n
    sinv = s_inv(s)
    b = [Fr(0)] * (N + 1)
    pw = [Fr(1)] + [Fr(0)] * N
    for n in range(1, N + 1):
        pw = s_mul(pw, sinv)
        b[n] = pw[n - 1] / n
    m = b[1:] + [Fr(0)]  # mu / eta, m[0] == 1
    assert m[0] == 1 and m[1] == Fr(1, 3) and m[2] == Fr(1, 36) and m[3] == Fr(-1, 270)
    minv = s_inv(m)  # eta / mu
    K = len(DEGREES)
    B = bernoulli(2 * K + 2)
    lg = [Fr(0)] * (K + 1)  # -log Gamma*(a) in powers of 1/a
    for j in range(1, K + 1):
        if 2 * j - 1 <= K:
            lg[2 * j - 1] = -B[2 * j] / (2 * j * (2 * j - 1))
    g = s_exp(lg, K)
    assert g[1] == Fr(-1, 12) and g[2] == Fr(1, 288) and g[3] == Fr(139, 51840)
    # c_0 = (1 - m) / (eta m)
    num = [-x for x in m]
    num[0] += 1
    assert num[0] == 0
    c = [s_mul(num[1:] + [Fr(0)], minv)]
    for k in range(1, K):
        d = [(i + 1) * c[-1][i + 1] for i in range(N)] + [Fr(0)]
        num = [x + g[k] * y for x, y in zip(d, minv)]
        assert num[0] == 0, k
        c.append(num[1:] + [Fr(0)])
    assert c[0][0] == Fr(-1, 3) and c[0][1] == Fr(1, 12) and c[1][0] == Fr(-1, 540)
    return [[float(ck[i]) for i in range(DEGREES[k] + 1)] for k, ck in enumerate(c)], c


def temme_double(C, a, x):
    """The evaluation the device performs (plain double arithmetic; log1p(s) - s correctly rounded, as the device's
    log1pmx series gives it): returns (P, Q)."""
    sigma = (x - a) / a
    phi = float(mp.mpf(sigma) - mp.log(1 + mp.mpf(sigma)))
    y = a * phi
    eta = math.copysign(math.sqrt(2.0 * phi), sigma)
    ia = 1.0 / a
    t = 0.0
    for ck in C[::-1]:
        p = 0.0
        for co in ck[::-1]:
            p = p * eta + co
        t = t * ia + p
    small = 0.5 * math.erfc(math.sqrt(y))
    s = t * math.exp(-y) / math.sqrt(2.0 * math.pi * a)
    if x < a:
        p_ = small - s
        return p_, 1.0 - p_
    q_ = small + s
    return 1.0 - q_, q_


def temme_mp(Cx, a, x):
    """The same expansion in 60-digit arithmetic with every computed series term (exact rational coefficients): the
    reference where mpmath's own gammainc is impractical (a > 1e4).  Its truncation error is ~ c_K a^-K, |eta|^N."""
    a, x = mp.mpf(a), mp.mpf(x)
    mu = (x - a) / a
    phi = mu - mp.log(1 + mu)
    eta = mp.sqrt(2 * phi) * (1 if mu >= 0 else -1)
    t = mp.mpf(0)
    for k, ck in enumerate(Cx):
        t += sum(mp.mpf(co.numerator) / co.denominator * eta ** i for i, co in enumerate(ck[: N - 2 * k - 2])) / a ** k
    s = t * mp.exp(-a * phi) / mp.sqrt(2 * mp.pi * a)
    small = mp.erfc(mp.sqrt(a * phi)) / 2
    return (small - s, None) if x < a else (None, small + s)


def check(C, Cx):
    mp.mp.dps = 60
    worst = 0.0
    where = None
    agree = 0.0
    for a in (MIN_A, 33.0, 40.0, 55.5, 80.0, 100.0, 200.0, 1e3, 1e4, 1e6, 1e8, 1e10, 1e12, 1e15):
        for i in range(-30, 31):
            xs = [a * (1.0 + MAX_SIGMA * i / 30.0)]
            if i == 0:  # the region that matters for large a is |x - a| ~ sqrt(a)
                xs += [a + j * math.sqrt(a) for j in (-6.0, -2.5, -0.3, 0.4, 1.7, 5.0) if abs(j) <= MAX_SIGMA * math.sqrt(a)]
            for xx in xs:
                P, Q = temme_double(C, a, xx)
                Pm, Qm = temme_mp(Cx, a, xx)
                if a <= 1e4:  # the reference at the doubles (a, xx) actually used
                    Qr = mp.gammainc(mp.mpf(a), mp.mpf(xx), mp.inf, regularized=True)
                    Pr = mp.gammainc(mp.mpf(a), 0, mp.mpf(xx), regularized=True)
                else:
                    Pr, Qr = Pm, Qm
                got, ref = (P, Pr) if xx < a else (Q, Qr)
                if ref < mp.mpf(10) ** -300:
                    continue
                if a >= 200.0 and a <= 1e4:  # validates the large-a reference against mpmath's gammainc
                    agree = max(agree, float(abs((Pm if xx < a else Qm) - ref) / ref))
                # conditioning: exp(-a phi) carries a phi relative units per relative unit of phi
                e = float(abs(got - ref) / ref)
                cond = max(1.0, float(a * abs(mp.log(1 + mp.mpf(xx - a) / a) - mp.mpf(xx - a) / a)))
                e_scaled = e / cond
                if e_scaled > worst:
                    worst, where = e_scaled, (a, xx, e)
    return worst, where, agree


def main():
    C, Cx = coefficients()
    worst, where, agree = check(C, Cx)
    out = sys.stdout
    out.write("// Generated by tools/gen_temme_coeffs.py -- do not edit.  Temme's uniform expansion of P(a, x) / Q(a, x)\n")
    out.write("// (DLMF 8.12.7-8): Taylor coefficients in eta of c_k(eta), k = 0 .. %d, exact rationals rounded once.\n" % (len(C) - 1))
    out.write("// Checked in double Horner arithmetic against 60-digit mpmath for a in [%g, 1e15], |x - a| <= %g a:\n" % (MIN_A, MAX_SIGMA))
    out.write("// max relative error of the smaller of P, Q, in units of its conditioning max(1, a phi): %.2e\n" % worst)
    out.write("// (worst point a = %g, x = %.17g: relative error %.2e).  Reference: mpmath's gammainc for a <= 1e4, the full\n" % where)
    out.write("// computed expansion in 60-digit arithmetic above (the two agree to %.1e for 200 <= a <= 1e4).\n" % agree)
    out.write("#pragma once\nnamespace m355 {\n")
    out.write("constexpr double kTemmeMinA = %r;\nconstexpr double kTemmeMaxSigma = %r;\n" % (MIN_A, MAX_SIGMA))
    out.write("constexpr int kTemmeK = %d;\n" % len(C))
    for k, ck in enumerate(C):
        out.write("constexpr int kTemmeDeg%d = %d;\n" % (k, len(ck) - 1))
        out.write("constexpr double kTemmeC%d[%d] = {\n" % (k, len(ck)))
        for co in ck:
            out.write("    %s,\n" % float(co).hex())
        out.write("};\n")
    out.write("}  // namespace m355\n")


if __name__ == "__main__":
    main()
