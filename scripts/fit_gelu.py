"""CPU: fit and verify the logistic-of-odd-polynomial form of the exact-erf GELU used by csrc/k_gemm.hip gelu_erf.

gelu(x) = x * Phi(x);  Phi(x) ~= 1 / (1 + exp(-x * P(x^2))),  P of degree 4 in x^2, fitted on |x| <= 8 by iteratively
re-weighted least squares (towards minimax) on the absolute error of gelu.  Prints the coefficients (also pre-scaled by
-log2 e, as the kernel uses them) and the maximum absolute error of an fp32 evaluation against the fp64 exact value."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import ndtr

x = np.linspace(-8, 8, 200001)
gelu = x * ndtr(x)


def model(c, x):
    x2 = x * x
    p = c[-1]
    for a in c[-2::-1]:
        p = p * x2 + a
    with np.errstate(over="ignore"):
        return x / (1.0 + np.exp(-x * p))


c = np.array([1.5957691, 0.0713548, 0.0, 0.0, 0.0])
w = np.ones_like(x)
for _ in range(30):
    c = least_squares(lambda c: (model(c, x) - gelu) * np.sqrt(w), c, method="lm", xtol=1e-15, ftol=1e-15).x
    e = np.abs(model(c, x) - gelu)
    w = (e / e.max()) ** 2 + 1e-3
print("P coefficients (x^0, x^2, ...):", repr(c))
cs = (-c * 1.4426950408889634).astype(np.float32)
print("kernel constants (-log2e * c):", [f"{v:.9e}f" for v in cs])
xf = np.linspace(-30, 30, 2000001).astype(np.float32)
xc = np.clip(xf, np.float32(-8), np.float32(8))
x2 = xc * xc
p = cs[4] * x2 + cs[3]
for k in (2, 1, 0):
    p = p * x2 + cs[k]
with np.errstate(over="ignore"):
    g = xf * (np.float32(1) / (np.float32(1) + np.exp2(xc * p).astype(np.float32)))
ref = xf.astype(np.float64) * ndtr(xf.astype(np.float64))
err = np.abs(g - ref)
print(f"fp32 evaluation: max abs error {err.max():.3e} at x = {xf[err.argmax()]:.3f}")
