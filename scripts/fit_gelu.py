"""CPU: verify (and optionally re-fit) the logistic-of-odd-polynomial form of the exact-erf GELU used by
csrc/k_gemm.hip gelu_erf.

gelu(x) = x * Phi(x);  Phi(x) ~= 1 / (1 + exp(-x * P(x^2))),  P of degree 4 in x^2.
KERNEL holds the constants compiled into the kernel (already multiplied by -log2 e).  The script evaluates them the
way the kernel does (fp32, no clamp: P(x^2) < 0 for every x, checked below, so the exponent saturates by itself)
against the fp64 exact value over |x| <= 60 and prints the maximum absolute error; with --refit it also runs the iteratively re-weighted least-squares fit they came from
(the fit is not unique: any run that lands below ~1e-5 is as good, the kernel keeps the best one found)."""
import sys
import numpy as np
from scipy.special import ndtr

KERNEL = np.array([-2.302086592e+00, -1.051034182e-01, 2.890509495e-04, 1.012880530e-04, -3.936969279e-06], np.float32)


def max_err(cs):
    xf = np.linspace(-60, 60, 4000001).astype(np.float32)
    xc = xf
    x2 = xc * xc
    p = cs[4] * x2 + cs[3]
    for k in (2, 1, 0):
        p = p * x2 + cs[k]
    assert p.max() < 0, "P(x^2) must stay negative: the kernel relies on it instead of clamping x"
    with np.errstate(over="ignore"):
        g = xf * (np.float32(1) / (np.float32(1) + np.exp2(xc * p).astype(np.float32)))
    ref = xf.astype(np.float64) * ndtr(xf.astype(np.float64))
    err = np.abs(g - ref)
    return err.max(), xf[err.argmax()]


if "--rows" in sys.argv:   # csrc/k_rows.hip gelu_stage: three coefficients, x^2 clamped at 64
    K3 = np.array([-2.301208258e+00, -1.066924557e-01, 1.000115648e-03], np.float32)
    xf = np.linspace(-60, 60, 4000001).astype(np.float32)
    x2 = np.minimum(xf * xf, np.float32(64))
    p = (K3[2] * x2 + K3[1]).astype(np.float32)
    p = (p * x2 + K3[0]).astype(np.float32)
    assert p.max() < 0
    with np.errstate(over="ignore"):
        g = xf * (np.float32(1) / (np.float32(1) + np.exp2(xf * p).astype(np.float32)))
    err = np.abs(g - xf.astype(np.float64) * ndtr(xf.astype(np.float64)))
    print(f"k_rows constants: max abs error of gelu (fp32 evaluation) {err.max():.3e} at x = {xf[err.argmax()]:.3f}")
    assert err.max() < 4e-5
    sys.exit(0)

e, at = max_err(KERNEL)
print(f"kernel constants: max abs error of gelu (fp32 evaluation) {e:.3e} at x = {at:.3f}")
assert e < 1e-5

if "--refit" in sys.argv:
    from scipy.optimize import least_squares
    x = np.linspace(-8, 8, 200001)
    gelu = x * ndtr(x)

    def model(c):
        x2 = x * x
        p = c[-1]
        for a in c[-2::-1]:
            p = p * x2 + a
        with np.errstate(over="ignore"):
            return x / (1.0 + np.exp(-x * p))

    c = np.array([1.5957691, 0.0713548, 0.0, 0.0, 0.0])
    w = np.ones_like(x)
    best = None
    for _ in range(40):
        c = least_squares(lambda c: (model(c) - gelu) * np.sqrt(w), c, method="lm", xtol=1e-15, ftol=1e-15).x
        e = np.abs(model(c) - gelu)
        if best is None or e.max() < best[0]:
            best = (e.max(), c.copy())
        w = (e / e.max()) ** 2 + 1e-3
    cs = (-best[1] * 1.4426950408889634).astype(np.float32)
    print("refit: fp64 max abs error", best[0], " constants:", [f"{v:.9e}f" for v in cs], " fp32 error: %.3e" % max_err(cs)[0])
