#!/usr/bin/env python3
"""Coefficients of wt_exp2_64 (wavelets_amd/csrc/wt_device.h): 2^f on [-1/2, 1/2] as a polynomial of
degree DEG, from the Chebyshev interpolant of 2^f at 50 digits (mpmath), converted to the monomial
basis; prints the table and the measured relative error of the double-precision Horner evaluation.

    python tools/make_exp2_64.py [degree]
"""
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 50
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = DEG + 1
# Chebyshev nodes on [-1/2, 1/2] and the interpolating polynomial through them, in mpmath
nodes = [mp.cos(mp.pi * (mp.mpf(k) + mp.mpf(1) / 2) / n) / 2 for k in range(n)]
vals = [mp.power(2, x) for x in nodes]
A = mp.matrix(n, n)
for i, x in enumerate(nodes):
    for j in range(n):
        A[i, j] = x ** j
c = mp.lu_solve(A, mp.matrix(vals))
coef = [float(c[j]) for j in range(n)]
coef[0] = 1.0          # 2^0 = 1 exactly (the interpolant's constant differs by < 1 ulp)
f = np.linspace(-0.5, 0.5, 20001)
acc = np.zeros_like(f)
for a in coef[::-1]:
    acc = acc * f + a
ref = np.array([float(mp.power(2, mp.mpf(float(v)))) for v in f])
print(f"// degree {DEG}: max relative error of the double Horner form {np.abs(acc / ref - 1).max():.2e}")
print("{" + ",\n ".join(", ".join(float(a).hex() for a in coef[i:i + 4]) for i in range(0, len(coef), 4)) + "}")
