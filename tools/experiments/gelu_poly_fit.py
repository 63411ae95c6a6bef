#!/usr/bin/env python3
"""Coefficients of a transcendental-free GELU epilogue -- an experiment that was NOT kept (round 4): at degree 9 in s (X = 4.2, |gelu error|
<= 6.5e-5) the fc1 epilogue needs 2 x 9 packed fmas per element pair, 2 026 full-rate VALU instructions per wave and tile against 1 231 + 256
quarter-rate ones (v_rcp_f32, v_exp_f32) of the erf form in csrc/nrv_common.hpp: 8.1k instead of 9.0k issue cycles, for a less accurate GELU.

    Phi(x)   = 1/2 + xc A(s),      gelu(x)  = x Phi(x)
    gelu'(x) = 1/2 + xc R(s)       with xc = clamp(x, -X, X), s = xc^2

A, R: polynomials of degree D in s, weighted-minimax (Lawson) fits of (Phi(x) - 1/2) / x and (gelu'(x) - 1/2) / x on [0, X], constrained to
X A(X^2) = X R(X^2) = 1/2 so that the clamped tails are exactly 0 / 1 (Phi) and 0 / 1 (gelu').  Prints the C arrays and the errors of an
fp32 Horner evaluation against float64 on [-3 X, 3 X].  numpy / scipy only; run on any CPU."""
import sys
import numpy as np
from scipy.special import erf

X = float(sys.argv[1]) if len(sys.argv) > 1 else 3.9
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def Phi(x): return 0.5 * (1 + erf(x / np.sqrt(2)))
def phi(x): return np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


def fit(target, weight):
    x = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000) * X / 2 + X / 2           # (0, X)
    s = x * x
    f = target(x)
    fX = 0.5 / X
    # p(s) = fX + (s - X^2) b(s):  b of degree D - 1, in the Chebyshev basis of s / X^2 for conditioning
    T = np.polynomial.chebyshev.chebvander(2 * s / X ** 2 - 1, D - 1)
    Vb = T * (s - X ** 2)[:, None]
    w0 = weight(x)
    lw = np.ones_like(x)
    for _ in range(200):
        w = w0 * lw
        c, *_ = np.linalg.lstsq(Vb * w[:, None], (f - fX) * w, rcond=None)
        err = np.abs((Vb @ c + fX - f) * w0)
        lw *= (err / err.max()) + 1e-3
        lw /= lw.max()
    # to monomials in s
    b = np.polynomial.chebyshev.cheb2poly(c)                    # in u = 2 s / X^2 - 1
    pu = np.polynomial.polynomial.Polynomial(b)
    u_of_s = np.polynomial.polynomial.Polynomial([-1.0, 2.0 / X ** 2])
    bs = pu(u_of_s)
    ps = bs * np.polynomial.polynomial.Polynomial([-X ** 2, 1.0]) + fX
    return ps.coef


def horner32(c, s):
    acc = np.full_like(s, np.float32(c[-1]))
    for k in c[-2::-1]:
        acc = (acc * s + np.float32(k)).astype(np.float32)
    return acc


cA = fit(lambda x: (Phi(x) - 0.5) / x, lambda x: x)              # error of Phi itself = x * error of A
cR = fit(lambda x: (Phi(x) + x * phi(x) - 0.5) / x, lambda x: x)
x = np.linspace(-3 * X, 3 * X, 400001)
x32 = x.astype(np.float32)
xc = np.clip(x32, np.float32(-X), np.float32(X))
s = (xc * xc).astype(np.float32)
P = (np.float32(0.5) + xc * horner32(cA, s)).astype(np.float32)
g = (x32 * P).astype(np.float32)
dg = (np.float32(0.5) + xc * horner32(cR, s)).astype(np.float32)
tg, tdg = x * Phi(x), Phi(x) + x * phi(x)
print(f"X = {X}, degree {D} in s: max |Phi err| {np.abs(P - Phi(x)).max():.2e}, max |gelu err| {np.abs(g - tg).max():.2e} "
      f"(on |x| <= X: {np.abs(g - tg)[np.abs(x) <= X].max():.2e}), max |gelu' err| {np.abs(dg - tdg).max():.2e}")
rel = np.abs(g - tg) / np.maximum(np.abs(tg), 1e-30)
print(f"max relative gelu error where |gelu| >= 1e-3: {rel[np.abs(tg) >= 1e-3].max():.2e}")
for name, c in (("GELU_A", cA), ("GELU_R", cR)):
    print(f"constexpr float {name}[{D + 1}] = {{" + ", ".join(f"{v:.9e}f" for v in c) + "};")
