"""tanh(x) = x + x*u*R(u)/Q(u), u = x^2 (the correction form: for small and middle |x| only the last FMA's rounding shows) — fit of
R (degree NR-1) and Q (degree NQ-1, Q(0) = 1) by Lawson-weighted least squares on the RELATIVE error of tanh, evaluated like the
kernel would (fp32 FMAs). usage: python scratch/fit_tanh4.py NR NQ [XMAX]"""
import sys
import numpy as np
from scipy.optimize import least_squares
NR, NQ = int(sys.argv[1]), int(sys.argv[2])
XMAX = float(sys.argv[3]) if len(sys.argv) > 3 else 7.9
N = 8001
x = (np.cos(np.linspace(np.pi, 0, N)) * 0.5 + 0.5) * XMAX; x = x[x > 1e-4]
s = XMAX * XMAX; v = x * x / s
g = (np.tanh(x) / x - 1.0) / (x * x) * s                    # target of R/Q in the scaled variable: tanh/x = 1 + v*g(v)
t = np.tanh(x) / x
V = lambda n: np.stack([v ** k for k in range(n)], axis=1)
def rq(c): return V(NR) @ c[:NR], V(NQ) @ np.r_[1.0, c[NR:]]
def model(c):
    r, q = rq(c); return 1.0 + v * r / q
def lin(w, qprev):
    # (1 + v r/q)/t - 1 ~ 0  ->  v r - (t - 1) q ~ 0, weighted by w / (t q)
    A = np.concatenate([v[:, None] * V(NR), -(t - 1.0)[:, None] * V(NQ)[:, 1:]], axis=1); W = w / (t * qprev)
    c, *_ = np.linalg.lstsq(A * W[:, None], (t - 1.0) * W, rcond=None); return c
w = np.ones_like(v); q = np.ones_like(v); best = None
for it in range(300):
    for _ in range(5):
        c = lin(w, q); r, q = rq(c)
    err = np.abs(model(c) / t - 1)
    if best is None or err.max() < best[0]: best = (err.max(), c.copy())
    w = w * (0.2 + err / err.max()); w /= w.mean()
c = best[1]
res = lambda c: (model(c) / t - 1) * 1e6
c = least_squares(lambda c: np.sign(res(c)) * np.abs(res(c)) ** 4, c, method='lm', max_nfev=4000).x
print("f64 max rel:", np.abs(model(c) / t - 1).max(), "min q", rq(c)[1].min())
a = c[:NR] / s ** (np.arange(NR) + 1); b = np.r_[1.0, c[NR:]] / s ** np.arange(NQ)
f32 = np.float32
def fma(x_, y_, z_): return (x_.astype(np.float64) * np.asarray(y_, np.float64) + np.asarray(z_, np.float64)).astype(f32)
def ev(xs):
    xs = np.clip(xs, f32(-XMAX), f32(XMAX)).astype(f32)
    u = (xs * xs).astype(f32); xu = (xs * u).astype(f32)
    R = np.full_like(u, f32(a[-1]))
    for k in a[-2::-1]: R = fma(R, u, f32(k))
    Q = np.full_like(u, f32(b[-1]))
    for k in b[-2::-1]: Q = fma(Q, u, f32(k))
    return fma(xu, (R * (f32(1) / Q).astype(f32)).astype(f32), xs)
xs = np.concatenate([np.linspace(-12, 12, 4000001), np.logspace(-12, 1.1, 400001), -np.logspace(-12, 1.1, 400001)]).astype(f32)
xs = xs[xs != 0]
ref = np.tanh(xs.astype(np.float64)); y = ev(xs)
rel = np.abs(y - ref) / np.abs(ref)
for lo, hi in ((0, 0.5), (0.5, 1), (1, 2), (2, 4), (4, 13)):
    m = (np.abs(xs) >= lo) & (np.abs(xs) < hi)
    print("  |x| in [%g, %g): max rel %.2e (%.2f ulp of y)  max abs %.2e" % (lo, hi, rel[m].max(), (rel[m] / 2 ** -24).max(), np.abs(y - ref)[m].max()))
print("fp32: max rel %.3e  max abs %.3e  max |y| %.9f  odd: %s" % (rel.max(), np.abs(y - ref).max(), np.abs(y).max(), np.array_equal(ev(-xs), -y)))
print("R (low to high):", ", ".join(repr(float(k)) for k in a))
print("Q (low to high):", ", ".join(repr(float(k)) for k in b))
