#!/usr/bin/env python3
"""Constants of gm_gelu (hipie_amd/csrc/gemm.hip): the erfc form  erf(x) = 1 - P(t) exp(-x^2),  t = 1 / (1 + p x)  of Abramowitz & Stegun
7.1.26 with a sixth-degree P, fitted here (Lawson-weighted least squares -> minimax on [0, 6]), then re-expressed for
Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and checked in emulated fp32 against fp64.  CPU only (numpy / scipy)."""
import numpy as np
from scipy import optimize, special

xs = np.concatenate([np.linspace(0, 1, 4001), np.linspace(1, 6, 6001)])
ref = special.erf(xs)


def model(c, x):
    t = 1.0 / (1.0 + c[0] * x)
    poly = np.zeros_like(x)
    for a in c[:0:-1]:
        poly = (poly + a) * t
    return 1.0 - poly * np.exp(-x * x)


def fit(c):
    w = np.ones_like(xs)
    for _ in range(60):
        c = optimize.least_squares(lambda c_: w * (model(c_, xs) - ref), c, xtol=1e-15, ftol=1e-15, gtol=1e-15).x
        e = np.abs(model(c, xs) - ref)
        w = w * (e / e.max() + 1e-3)
        w /= w.max()
    return c, np.abs(model(c, xs) - ref).max()


handbook = np.array([0.3275911, 0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429])      # A&S 7.1.26
print("A&S 7.1.26 constants: max |erf error| %.2e" % np.abs(model(handbook, xs) - ref).max())
c5, e5 = fit(handbook)
c6, e6 = fit(np.concatenate([c5, [0.0]]))
print("six terms, refitted:  max |erf error| %.2e" % e6)
f = np.float32
pp, a, K = f(c6[0] / np.sqrt(2.0)), [f(v / 2.0) for v in c6[1:]], f(0.5 * 1.4426950408889634)
print("p' = %r\na'  = %s\nK   = %r" % (float(pp), [float(v) for v in a], float(K)))


def gelu32(x):
    ax = np.abs(x)
    t = (f(1.0) / (pp * ax + f(1.0))).astype(np.float32)
    poly = a[5]
    for k in (4, 3, 2, 1, 0):
        poly = (poly * t + a[k]).astype(np.float32)
    h = ((poly * t).astype(np.float32) * np.exp2((-(ax * ax).astype(np.float32) * K).astype(np.float32)).astype(np.float32)).astype(np.float32)
    return (x * np.where(x >= 0, (f(1.0) - h).astype(np.float32), h)).astype(np.float32)


rng = np.random.default_rng(0)
x = np.concatenate([rng.uniform(-12, 12, 3_000_000), rng.normal(0, 1, 3_000_000), np.linspace(-0.01, 0.01, 200001)]).astype(np.float32)
xd = x.astype(np.float64)
want = np.where(xd < 0, 0.5 * xd * special.erfc(-xd / np.sqrt(2.0)), 0.5 * xd * (1.0 + special.erf(xd / np.sqrt(2.0))))
got = gelu32(x).astype(np.float64)
print("fp32 evaluation: max |gelu error| %.2e, max error / |x| %.2e" % (np.abs(got - want).max(), (np.abs(got - want) / np.maximum(np.abs(xd), 1e-30)).max()))
