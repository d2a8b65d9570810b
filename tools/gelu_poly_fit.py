"""Coefficients of the bf16-mode GELU epilogue (csrc/common.h: gelu_bf16_parts4).

    gelu(x)  = max(x, 0) + t * P(t)               t = min(|x|, T)       t P(t) ~ r(t) = t (Phi(t) - 1)
    gelu'(x) = 1/2 + copysign(t * Q(t), x)                              t Q(t) ~ e(t) = Phi(t) + t phi(t) - 1/2

Both remainders are smooth on t >= 0 and flat beyond T (r -> 0, e -> 1/2), so one clamp + one Horner chain each replaces
the erf arithmetic (v_rcp_f32 + v_exp_f32 + A&S 7.1.26).  The fit is minimax in the ABSOLUTE error of gelu / gelu' over
the whole real line: Lawson-reweighted least squares on [0, T] with the value at T balanced against the tail (beyond T the
kernel returns the constant t P(T) resp. T Q(T) while the true remainder runs on to its limit).

    python tools/gelu_poly_fit.py            # prints the coefficient tables and the float32 error of the exact kernel formula
"""
import numpy as np
from scipy.special import erf


def Phi(t):
    return 0.5 * (1.0 + erf(t / np.sqrt(2.0)))


def phi(t):
    return np.exp(-t * t / 2.0) / np.sqrt(2.0 * np.pi)


def r(t):
    return t * (Phi(t) - 1.0)


def e(t):
    return Phi(t) + t * phi(t) - 0.5


def fit(f, limit, T, n, c0, iters=400):
    """coefficients c[0..n-1] (ascending) of P with t P(t) ~ f(t) on [0, T], tail target `limit` for t -> inf.  c[0] is
    pinned to the exact slope f'(0), so that gelu(x) -> x / 2 and gelu'(x) - 1/2 -> 2 phi(0) x hold in the RELATIVE sense
    around zero (outputs there are small, and a bf16 output keeps their relative precision)"""
    N = 6000
    t = (np.cos(np.pi * (np.arange(N) + 0.5) / N) + 1.0) * T / 2.0
    V = np.stack([t ** (k + 1) for k in range(1, n)], 1)
    y = f(t) - c0 * t
    # tail: the constant T P(T) is compared with f on [T, inf): worst at the two ends f(T) and `limit`
    Vt = np.stack([T ** (k + 1) for k in range(1, n)], 0)[None, :]
    V = np.concatenate([V, Vt, Vt], 0)
    y = np.concatenate([y, [f(T) - c0 * T], [limit - c0 * T]])
    sc = np.abs(V).max(0)
    w = np.ones(len(y))
    best = None
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(V / sc * w[:, None], y * w, rcond=None)
        c = c / sc
        err = np.abs(V @ c - y)
        if best is None or err.max() < best[1]:
            best = (np.concatenate([[c0], c]), err.max())
        w = w * (err / err.max() + 1e-3) ** 0.5
        w /= w.max()
    return best


def horner32(c, t):
    acc = np.full_like(t, np.float32(c[-1]), dtype=np.float32)
    for k in range(len(c) - 2, -1, -1):
        acc = np.float32(acc * t + np.float32(c[k]))      # (fma in the kernel: one rounding less)
    return acc


def main():
    T, n = 4.0, 8
    cp, ep = fit(r, 0.0, T, n, -0.5)
    cq, eq = fit(e, 0.5, T, n, 2.0 * phi(0.0))
    x = np.linspace(-9, 9, 2_000_001).astype(np.float32)
    t = np.minimum(np.abs(x), np.float32(T))
    g = np.maximum(x, 0) + t * horner32(cp, t)
    d = np.float32(0.5) + np.copysign(t * horner32(cq, t), x)
    xd = x.astype(np.float64)
    g_ref = xd * Phi(xd)
    d_ref = Phi(xd) + xd * phi(xd)
    print(f"T = {T}, {n} coefficients each")
    print("P:", ", ".join(f"{v:.9e}f" for v in cp), f"   fit max err {ep:.2e}")
    print("Q:", ", ".join(f"{v:.9e}f" for v in cq), f"   fit max err {eq:.2e}")
    print(f"float32 evaluation on [-9, 9]: max |gelu err| {np.abs(g - g_ref).max():.3e}   max |gelu' err| {np.abs(d - d_ref).max():.3e}")
    for lo, hi in ((-9, -4), (-4, -1), (-1, 0), (0, 1), (1, 4), (4, 9)):
        m = (x >= lo) & (x < hi)
        print(f"   x in [{lo},{hi}): gelu {np.abs(g - g_ref)[m].max():.2e}  gelu' {np.abs(d - d_ref)[m].max():.2e}")


if __name__ == "__main__":
    main()
