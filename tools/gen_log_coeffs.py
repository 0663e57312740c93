"""Generates the polynomial used by m355_log (csrc/m355_math.h).

log(m) = 2*atanh(s), s = (m-1)/(m+1), m in [sqrt(1/2), sqrt(2)) => t = s*s <= 0.0294373.
We approximate g(t) = (atanh(sqrt(t))/sqrt(t) - 1)/t = 1/3 + t/5 + t^2/7 + ... by interpolation
at Chebyshev nodes (near-minimax) and print the coefficients as C hex-float literals.
"""
import sys
import mpmath as mp

mp.mp.dps = 60


def g(t):
    if t == 0:
        return mp.mpf(1) / 3
    r = mp.sqrt(t)
    return (mp.atanh(r) / r - 1) / t


def fit(deg, tmax):
    n = deg + 1
    nodes = [tmax * (1 + mp.cos(mp.pi * (2 * k + 1) / (2 * n))) / 2 for k in range(n)]
    A = mp.matrix(n, n)
    b = mp.matrix(n, 1)
    for i, t in enumerate(nodes):
        for j in range(n):
            A[i, j] = t**j
        b[i] = g(t)
    c = mp.lu_solve(A, b)
    coeffs = [float(c[j]) for j in range(n)]
    # error of the double-rounded polynomial, relative to log(m)/(2s) = 1 + t*g(t)
    worst = 0
    for k in range(2001):
        t = tmax * mp.mpf(k) / 2000
        p = sum(mp.mpf(coeffs[j]) * t**j for j in range(n))
        err = abs(t * (p - g(t)) / (1 + t * g(t)))
        worst = max(worst, err)
    return coeffs, worst


if __name__ == "__main__":
    s = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
    tmax = s * s * (1 + mp.mpf(2) ** -40)
    for deg in (6, 7, 8):
        coeffs, worst = fit(deg, tmax)
        print(f"// degree {deg}: max rel err {mp.nstr(worst, 3)}")
        for cf in coeffs:
            print(f"    {cf.hex()},  // {cf!r}")
