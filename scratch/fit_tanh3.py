"""Round 3: can the odd rational tanh (aidax_device.h: P of degree 6, Q of degree 3 in x^2) lose instructions at the same fp32
accuracy? Same Lawson-weighted fit as fit_tanh2.py for other (deg P, deg Q) pairs on [0, 7.9] (the kernel's clamp), evaluated the
way the kernel does it (fp32 FMAs, v_rcp_f32 modelled as a correctly rounded reciprocal). usage: python scratch/fit_tanh3.py NP NQ"""
import sys
import numpy as np
from scipy.optimize import least_squares
NP, NQ = int(sys.argv[1]), int(sys.argv[2])
XMAX = 7.9
N = 8001
x = (np.cos(np.linspace(np.pi, 0, N)) * 0.5 + 0.5) * XMAX; x = x[x > 1e-5]
s = XMAX * XMAX; v = x * x / s; t = np.tanh(x) / x
V = lambda n: np.stack([v ** k for k in range(n)], axis=1)
def pq(c): return V(NP) @ c[:NP], V(NQ) @ np.r_[1.0, c[NP:]]
def lin(w, qprev):
    A = np.concatenate([V(NP), -t[:, None] * V(NQ)[:, 1:]], axis=1); W = w / (t * qprev)
    c, *_ = np.linalg.lstsq(A * W[:, None], t * W, rcond=None); return c
w = np.ones_like(v); q = np.ones_like(v); best = None
for it in range(300):
    for _ in range(5):
        c = lin(w, q); p, q = pq(c)
    err = np.abs(p / q / t - 1)
    if best is None or err.max() < best[0]: best = (err.max(), c.copy())
    w = w * (0.2 + err / err.max()); w /= w.mean()
c = best[1]
res = lambda c: (pq(c)[0] / pq(c)[1] / t - 1) * 1e6
c = least_squares(lambda c: np.sign(res(c)) * np.abs(res(c)) ** 4, c, method='lm', max_nfev=4000).x
p, q = pq(c); print("f64 max rel:", np.abs(p / q / t - 1).max(), "min q", q.min())
a = c[:NP] / s ** np.arange(NP); b = np.r_[1.0, c[NP:]] / s ** np.arange(NQ)
f32 = np.float32
def fma(x_, y_, z_): return (x_.astype(np.float64) * np.float64(y_) + np.float64(z_)).astype(f32) if np.isscalar(y_) else (x_.astype(np.float64) * y_.astype(np.float64) + np.float64(z_)).astype(f32)
def ev(xs):
    xs = np.clip(xs, f32(-XMAX), f32(XMAX)).astype(f32)
    u = (xs * xs).astype(f32)
    P = np.full_like(u, f32(a[-1]))
    for k in a[-2::-1]: P = fma(P, u, f32(k))
    Q = np.full_like(u, f32(b[-1]))
    for k in b[-2::-1]: Q = fma(Q, u, f32(k))
    return ((P * xs).astype(f32) * (f32(1) / Q).astype(f32)).astype(f32)
xs = np.concatenate([np.linspace(-12, 12, 4000001), np.logspace(-12, 1.1, 400001), -np.logspace(-12, 1.1, 400001)]).astype(f32)
xs = xs[xs != 0]
ref = np.tanh(xs.astype(np.float64)); y = ev(xs)
rel = np.abs(y - ref) / np.abs(ref)
print("fp32: max rel %.3e (%.2f ulp)  max abs %.3e  max |y| %.9f  odd: %s" % (rel.max(), rel.max() / 2 ** -24, np.abs(y - ref).max(), np.abs(y).max(), np.array_equal(ev(-xs), -y)))
print("P (low to high):", ", ".join(repr(float(k)) for k in a))
print("Q (low to high):", ", ".join(repr(float(k)) for k in b))
