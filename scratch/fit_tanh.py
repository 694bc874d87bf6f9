# Fit tanh(x) ~= x*P(x^2)/Q(x^2), P deg 6, Q deg 3 (in u=x^2), on [0, XMAX], minimising max RELATIVE error
import numpy as np
from scipy.optimize import least_squares
XMAX=7.90531110763549805  # tanh(x) rounds to 1.0f beyond ~9; clamp here costs <2e-7 abs... evaluate below
NP,NQ=7,4
x=np.cos(np.linspace(0,np.pi,4001))*0.5+0.5; x=np.sort(x)*XMAX; x=x[x>1e-6]
u=x*x; t=np.tanh(x)/x
def model(c,u):
    p=np.polyval(c[:NP][::-1],u); q=np.polyval(np.r_[c[NP:],1.0][::-1]*1.0,u)  # q normalised: highest coeff... see below
    return p/q
# parametrise q = 1 + b1 u + b2 u^2 + b3 u^3 ; p = a0 + ... a6 u^6
def pq(c,u):
    a=c[:NP]; b=np.r_[1.0,c[NP:]]
    p=sum(a[k]*u**k for k in range(NP)); q=sum(b[k]*u**k for k in range(NQ))
    return p,q
def fit(w):
    # linearised: p - t*q = 0  -> solve LS for coefficients, weights w/(t*q_prev)
    c=np.zeros(NP+NQ-1); qprev=np.ones_like(u)
    for it in range(30):
        A=np.stack([u**k for k in range(NP)]+[-t*u**k for k in range(1,NQ)],axis=1)
        rhs=t
        W=w/(t*qprev)
        c,*_=np.linalg.lstsq(A*W[:,None],rhs*W,rcond=None)
        p,q=pq(c,u); qprev=q
    return c
w=np.ones_like(u)
for it in range(60):
    c=fit(w); p,q=pq(c,u); err=np.abs(p/q/t-1)
    w=w*(err/err.mean())**0.5; w/=w.mean()
print("max rel err f64:",err.max())
a=c[:NP]; b=np.r_[1.0,c[NP:]]
# float32 evaluation check over dense grid incl. small x
f32=np.float32
xs=np.concatenate([np.linspace(-XMAX,XMAX,2000001),np.logspace(-8,0,200001)]).astype(f32)
def ev(xs):
    x2=(xs*xs).astype(f32); P=f32(a[-1])
    for k in a[-2::-1]: P=(P*x2+f32(k)).astype(f32)
    P=(P*xs).astype(f32); Q=f32(b[-1])
    for k in b[-2::-1]: Q=(Q*x2+f32(k)).astype(f32)
    return (P/Q).astype(f32)
y=ev(xs); ref=np.tanh(xs.astype(np.float64))
rel=np.abs(y-ref)/np.maximum(np.abs(ref),1e-300)
print("f32 eval: max rel err %.3e (ulp=%.2f) max abs %.3e"%(rel.max(), rel.max()/6e-8, np.abs(y-ref).max()))
print("a =",[float(v) for v in a]); print("b =",[float(v) for v in b])
print("tanh(XMAX)=",np.tanh(XMAX), "model at XMAX", ev(np.array([XMAX],f32)))
