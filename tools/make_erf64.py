#!/usr/bin/env python3
"""Coefficients of wt_erf64 (wavelets_amd/csrc/wt_f64.h): erf(y) = -expm1(-y * (r(t) + y)), y in [0, 6],
t = y / 3 - 1, r a degree-26 polynomial fitted (weighted Chebyshev least squares on 400 Chebyshev
nodes, 50-digit mpmath reference) to (-log(erfc(y)) - y^2) / y.  Prints the table and the measured
error of the double-precision evaluation (abs <= 3e-16, rel <= 3e-14 against mpmath).

    python tools/make_erf64.py
"""
import mpmath as mp
import numpy as np
from numpy.polynomial import chebyshev as C

mp.mp.dps = 50
DEG = 26
n = 400
tn = np.cos(np.pi * (np.arange(n) + 0.5) / n)
y = (tn + 1) * 3


def r_exact(v):
    v = mp.mpf(float(v))
    return float((-mp.log(mp.erfc(v)) - v * v) / v)


r = np.array([r_exact(v) for v in y])
w = np.array([max(float(mp.erfc(mp.mpf(float(v)))), 1e-10) for v in y]) ** 0.5 * np.maximum(y, 1e-3)
p = C.cheb2poly(C.chebfit(tn, r, DEG, w=w))
yy = np.concatenate([np.linspace(0, 6, 6001)[1:], 10.0 ** np.linspace(-300, -1, 500)])
ref = np.array([float(mp.erf(mp.mpf(float(v)))) for v in yy])
t = yy / 3 - 1
acc = np.zeros_like(t)
for a in p[::-1]:
    acc = acc * t + a
e = -np.expm1(-(yy * (acc + yy)))
print(f"// degree {DEG}: max abs error {np.abs(e - ref).max():.2e}, max rel error {(np.abs(e - ref) / ref).max():.2e}")
print("{" + ",\n ".join(", ".join(float(a).hex() for a in p[i:i + 4]) for i in range(0, len(p), 4)) + "}")
