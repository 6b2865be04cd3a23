"""Minimax fits for the transcendental-free GELU / GELU' of the bf16 kernels (csrc/common.h: gelu_poly*, round 5).

Form (every instruction on the vector pipe's full-rate path: v_fma_f32 / v_mul_f32 / v_add_f32, the clamp is the FMA's output modifier):
    s  = clamp01(x / (2 c) + 1/2)                       one v_fma_f32 ... clamp
    w  = s - 1/2  in [-1/2, 1/2]   (= clamp(x, -c, c) / (2 c))
    Phi(x)   ~ 1/2 + w q(w^2),   q(1/4) = 1            (so Phi is exactly 0 / 1 beyond the clamp)
    gelu'(x) ~ 1/2 + w r(w^2),   r(1/4) = 1
Both Phi - 1/2 and gelu' - 1/2 are odd functions, so q and r are polynomials in w^2.

The fit is a linear programme (minimise the largest weighted error on a dense grid); the reported errors are re-evaluated in float32 with
the exact operation order of the kernels.   python tools/fit_gelu_poly.py [c_phi d_phi c_d d_d]
"""
import sys

import numpy as np
from scipy.optimize import linprog
from scipy.special import erf


def targets(x):
    Phi = 0.5 * (1 + erf(x / np.sqrt(2)))
    pdf = np.exp(-x * x / 2) / np.sqrt(2 * np.pi)
    return Phi, Phi + x * pdf


def fit(c, d, kind, X=12.0, n=6001):
    """kind 0: gelu = x Phi, error weighted by 1 / max(1, |x|);  kind 1: gelu', absolute error."""
    x = np.linspace(-X, X, n)
    Phi, dg = targets(x)
    w = np.clip(x / (2 * c) + 0.5, 0, 1) - 0.5
    w2 = w * w
    # unknowns a_0..a_d with the constraint sum a_k (1/4)^k = 1  ->  eliminate a_0 = 1 - sum_{k>=1} a_k 4^-k
    # model = 1/2 + w (1 + sum_{k>=1} a_k (w2^k - 4^-k))
    B = np.stack([w * (w2 ** k - 0.25 ** k) for k in range(1, d + 1)], axis=1)
    base = 0.5 + w
    if kind == 0:
        scale = x / np.maximum(1.0, np.abs(x))
        A = B * scale[:, None]
        b = (Phi - base) * scale
    else:
        A = B
        b = dg - base
    # minimise eps  s.t.  -eps <= A a - b <= eps
    nv = d + 1
    cost = np.zeros(nv)
    cost[-1] = 1
    G = np.block([[A, -np.ones((len(x), 1))], [-A, -np.ones((len(x), 1))]])
    h = np.concatenate([b, -b])
    res = linprog(cost, A_ub=G, b_ub=h, bounds=[(None, None)] * d + [(0, None)], method="highs")
    a = res.x[:d]
    a0 = 1 - sum(a[k - 1] * 0.25 ** k for k in range(1, d + 1))
    return np.concatenate([[a0], a]), res.x[-1]


def eval_f32(x, c, coef, f16=False):
    """the kernels' operation order in float32 (fma emulated in float64 then rounded once)"""
    f = np.float32
    x = x.astype(f)
    r = lambda v: v.astype(f)
    s = np.clip(r(x.astype(np.float64) * np.float64(f(1 / (2 * c))) + 0.5), 0, 1).astype(f)
    w = r(s - f(0.5))
    w2 = r(w * w)
    p = np.full_like(x, f(coef[-1]))
    for k in range(len(coef) - 2, -1, -1):
        p = r(p.astype(np.float64) * w2 + np.float64(f(coef[k])))
    return r(w.astype(np.float64) * p + 0.5)


def report(c0, d0, c1, d1):
    a, e = fit(c0, d0, 0)
    b, e1 = fit(c1, d1, 1)
    x = np.linspace(-16, 16, 400001)
    Phi, dg = targets(x)
    ph = eval_f32(x, c0, a).astype(np.float64)
    g_err = np.abs(x * ph - x * Phi)
    rel = g_err / np.maximum(1, np.abs(x))
    dh = eval_f32(x, c1, b).astype(np.float64)
    d_err = np.abs(dh - dg)
    in8 = np.abs(x) <= 8
    print(f"Phi : c = {c0} d = {d0}  coefficients (a0..ad of q) = {[float(np.float32(v)) for v in a]}")
    print(f"      LP eps {e:.3e};  float32 order: max |gelu err| on [-8, 8] = {g_err[in8].max():.3e}, max |gelu err| / max(1, |x|) on [-16, 16] = {rel.max():.3e}")
    print(f"gelu': c = {c1} d = {d1}  coefficients (b0..bd of r) = {[float(np.float32(v)) for v in b]}")
    print(f"      LP eps {e1:.3e};  float32 order: max |gelu' err| on [-16, 16] = {d_err.max():.3e}")
    return a, b


if __name__ == "__main__":
    if len(sys.argv) == 5:
        report(float(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]))
    else:
        print("scan: (c, d) -> LP error")
        for kind, name in ((0, "gelu (rel. to max(1,|x|))"), (1, "gelu' (abs)")):
            for d in (3, 4, 5, 6, 7):
                best = min(((fit(c, d, kind)[1], c) for c in np.arange(2.5, 6.01, 0.25)))
                print(f"{name:28s} d = {d}: eps {best[0]:.3e} at c = {best[1]}")
