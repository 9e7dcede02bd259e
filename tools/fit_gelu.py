"""Fit of the deferred-epilogue GELU of the 4w GEMM (csrc/dvt_vit_gemm4w.inc: w4_gelu).

    gelu(x) ~ x * sigmoid(2 u),  u = x (c1 + c3 x^2 + c5 x^4)      (c3, c5 >= 0: u is monotonic, the tails are exact)

against nn.GELU() = 0.5 x (1 + erf(x / sqrt 2)) in float64, minimax over [-8, 8].  Prints the constants with the factor
2 log2(e) folded in (the kernel evaluates exp2) and the error of the float32 evaluation, also relative to the bf16 ulp of the
result (the output is rounded to bf16)."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

x = np.linspace(-8, 8, 160001)
ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))


def model(c, x=x):
    w = c[0] + x * x * (c[1] + x * x * c[2])
    return x / (1 + np.exp(-2 * x * w))


def cost(c):
    if c[1] < 0 or c[2] < 0:
        return 1e3
    return np.abs(model(c) - ref).max()


best = None
for c5 in (0.0, 1e-4, 5e-4):
    r = minimize(cost, [0.7978845608, 0.0356774, c5], method="Nelder-Mead", options=dict(xatol=1e-12, fatol=1e-14, maxiter=40000))
    if best is None or r.fun < best.fun:
        best = r
c = best.x
print("c1 c3 c5 =", c, "max abs err (float64)", best.fun)
k = 2 * np.log2(np.e)
C = np.float32(c * k)
print("W4_GELU_C1 %.9ef  W4_GELU_C3 %.9ef  W4_GELU_C5 %.9ef" % tuple(C))
x32 = np.linspace(-8, 8, 400001).astype(np.float32)
x2 = x32 * x32
w = (C[2] * x2 + C[1]) * x2 + C[0]
g = x32 / (np.float32(1) + np.exp2(-x32 * w, dtype=np.float32))
ref32 = 0.5 * x32.astype(np.float64) * (1 + erf(x32.astype(np.float64) / np.sqrt(2)))
err = np.abs(g - ref32)
ulp = np.maximum(np.abs(ref32), 1e-30) * 2.0 ** -8  # bf16: 8 bits of mantissa -> spacing 2^-7 .. 2^-8 relative
i = err.argmax()
print("float32 evaluation: max abs err %.3e at x = %.3f (gelu = %.4f); max err / bf16 spacing %.3f" % (
    err[i], x32[i], ref32[i], (err / ulp)[np.abs(ref32) > 1e-3].max()))
tanh = x32 / (1 + np.exp(-2 * 0.7978845608 * (x32 + 0.044715 * x32 ** 3)))
print("for comparison, the standard tanh form: max abs err %.3e" % np.abs(tanh - ref32).max())
