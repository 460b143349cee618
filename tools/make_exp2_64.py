#!/usr/bin/env python3
"""Coefficients of wt_exp2_64 (wavelets_amd/csrc/wt_device.h): 2^f on [-1/2, 1/2] as a polynomial of
degree DEG, from the Chebyshev interpolant of 2^f at 50 digits (mpmath), converted to the monomial
basis; prints the table and the measured relative error of the double-precision Horner evaluation.

    python tools/make_exp2_64.py [degree]
    python tools/make_exp2_64.py table      (also the tables and polynomials of the table form, WT_BIL64_TABLE)
"""
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 50
DEG = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "table" else 10
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


def table_form(bits, deg, print_table=True):
    """the table form of wt_math64.h (WT_BIL64_TABLE = 2^bits): 2^(64 g) on |64 g| <= 2^-(bits+1) as a polynomial in g of
    degree `deg` (Chebyshev interpolant, coefficient k scaled by 64^k - exact) and the table 2^(j / 2^bits - 64)"""
    import math
    nt = 1 << bits
    n = deg + 1
    a = mp.mpf(1) / (2 * nt)
    nodes = [a * mp.cos(mp.pi * (mp.mpf(k) + mp.mpf(1) / 2) / n) for k in range(n)]
    A = mp.matrix(n, n)
    for i, x in enumerate(nodes):
        for j in range(n):
            A[i, j] = x ** j
    c = mp.lu_solve(A, mp.matrix([mp.power(2, x) for x in nodes]))
    scaled = [math.ldexp(float(c[k]), 6 * k) for k in range(n)]
    T = [float(mp.power(2, mp.mpf(j) / nt - 64)) for j in range(nt)]
    rng = np.random.default_rng(0)
    t = -rng.random(400000) * 70
    u = np.clip(1 + t / 64, 0, 1)
    M = math.ldexp(1.5, 52 - 6 - bits)
    m = u + M
    e = (m.view(np.int64) & 0xffffffff).astype(np.int64)
    g = u - (m - M)
    p = np.full_like(g, scaled[deg])
    for k in range(deg - 1, -1, -1):
        p = p * g + scaled[k]
    w = ((p * np.array(T)[e & (nt - 1)]).view(np.int64) + ((e >> bits) << 52)).view(np.float64)
    print(f"// table of {nt}, degree {deg}: max relative error of the emulated weight {np.abs(w / np.exp2(np.maximum(t, -64)) - 1).max():.2e}")
    print("C = {" + ", ".join(float(x).hex() for x in scaled) + "}")
    if print_table:
        print("T = {" + ",\n ".join(", ".join(float(x).hex() for x in T[i:i + 4]) for i in range(0, nt, 4)) + "}")


if len(sys.argv) > 1 and sys.argv[1] == "table":
    table_form(5, 5)
    table_form(6, 4)
    table_form(9, 3, print_table=False)      # (the kernel computes these 512 entries itself)
