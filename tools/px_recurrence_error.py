#!/usr/bin/env python3
"""Accuracy of k_composite_px's table recurrence (composite.hip, builder wave) in binary32, on the CPU: along one axis the
Gaussian at pixel pairs p0, p0 + 1, ... is G(u + 2k) = G(u) R(u), R(u + 2k) = R(u) D, seeded at the first covered pair with five
exponentials.  Compared with float64 over random entries (radius 0.5 .. 40 px, centre anywhere a box can reach the tile from),
next to the direct form it replaced (exp2(-(x k - c k)^2) per pixel).  numpy's exp2 stands in for v_exp_f32 (both ~1 ulp).
    python tools/px_recurrence_error.py [entries=400000] [seed=1]"""
import sys

import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
f = np.float32
r = np.exp(rng.uniform(np.log(0.5), np.log(40), n)).astype(f)
c = rng.uniform(-20, 36, n).astype(f)  # splat centre, tile-local pixels
lo, hi = (c - f(1.5) * r).astype(f), (c + f(1.5) * r).astype(f)
a, b = np.maximum(np.ceil(lo - f(0.5)), 0), np.minimum(np.floor(hi - f(0.5)), 15)
ok = a <= b
r, c, a, b = r[ok], c[ok], a[ok].astype(int), b[ok].astype(int)
k = (f(1.6986436005760381) / r).astype(f)
k64 = 1.6986436005760381 / r.astype(np.float64)
exact = lambda x: np.exp2(-(((x + 0.5) - c.astype(np.float64)) * k64) ** 2)
ex = lambda x: np.exp2(x.astype(f)).astype(f)
p0, p1 = a >> 1, b >> 1
u0 = ((f(2) * p0.astype(f) + f(0.5) - c) * k).astype(f)
u1 = (u0 + k).astype(f)
u2 = (u1 + k).astype(f)
k4 = (f(4) * k).astype(f)
G = [ex(-(u0 * u0)), ex(-(u1 * u1))]
R = [ex(-(k4 * u1)), ex(-(k4 * u2))]
D = ex(-(f(2) * (k4 * k).astype(f)))
rec_abs = rec_rel = 0.0
for i in range(8):
    for j in (0, 1):
        x = 2 * (p0 + i) + j
        cov = ((p0 + i) <= p1) & (x >= a) & (x <= b)
        if cov.any():
            e = np.abs(G[j].astype(np.float64) - exact(x))[cov]
            rec_abs, rec_rel = max(rec_abs, e.max()), max(rec_rel, (e / exact(x)[cov]).max())
        G[j] = (G[j] * R[j]).astype(f)
        R[j] = (R[j] * D).astype(f)
dir_abs = dir_rel = 0.0
ck = (c * k).astype(f)
for x in range(16):
    t = ((f(x) + f(0.5)) * k - ck).astype(f)
    cov = (x >= a) & (x <= b)
    e = np.abs(ex(-(t * t)).astype(np.float64) - exact(x))[cov]
    dir_abs, dir_rel = max(dir_abs, e.max()), max(dir_rel, (e / exact(x)[cov]).max())
print(f"{len(r)} entries: recurrence worst abs {rec_abs:.3g} rel {rec_rel:.3g} | direct form worst abs {dir_abs:.3g} rel {dir_rel:.3g}")
