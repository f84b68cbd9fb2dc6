"""Fit of the transcendental-free erf-GELU in csrc/imh_common.h: erf(x / sqrt 2) = x Q(u), u = 2 x^2 / xmax^2 - 1, |x| <= xmax = 3.3 sqrt 2,
Q of degree 10 (Chebyshev least squares at 600 Chebyshev nodes, converted to monomials in u); prints the GELU_Q* constants and the fp32
Horner error.  CPU only."""
import math
import numpy as np
from numpy.polynomial import chebyshev as Ch
from math import erf

xmax, d, n = 3.3 * math.sqrt(2), 10, 600
k = np.arange(n)
tt = 0.5 * (1 + np.cos(np.pi * (k + 0.5) / n)) * xmax ** 2
x = np.sqrt(tt)
f = np.array([erf(v / math.sqrt(2)) / v if v > 0 else math.sqrt(2 / math.pi) for v in x])
mono = Ch.cheb2poly(Ch.chebfit(2 * tt / xmax ** 2 - 1, f, d)).astype(np.float32)
for i, c in enumerate(mono):
    print(f"#define GELU_Q{i} {float(c)!r}f")
xs = np.linspace(-6, 6, 200001).astype(np.float32)
xc = np.clip(xs, -np.float32(xmax), np.float32(xmax))
u = (xc * xc * np.float32(2 / xmax ** 2) - np.float32(1)).astype(np.float32)
q = np.full_like(u, mono[-1])
for c in mono[-2::-1]:
    q = (q * u + c).astype(np.float32)
e = (xc * q).astype(np.float32)
ref = np.array([erf(float(v) / math.sqrt(2)) for v in xs])
g = (0.5 * xs * (1 + e)).astype(np.float32)
print("max |erf error|", np.abs(e - ref).max(), " max |GELU error|", np.abs(g - 0.5 * xs.astype(np.float64) * (1 + ref)).max())
