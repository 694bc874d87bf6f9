import numpy as np
from scipy.optimize import least_squares
XMAX=9.0
NP,NQ=7,4
N=6001
x=(np.cos(np.linspace(np.pi,0,N))*0.5+0.5)*XMAX; x=x[x>1e-5]
s=XMAX*XMAX; v=x*x/s; t=np.tanh(x)/x
V=lambda n:np.stack([v**k for k in range(n)],axis=1)
def pq(c): return V(NP)@c[:NP], V(NQ)@np.r_[1.0,c[NP:]]
def lin(w,qprev):
    A=np.concatenate([V(NP), -t[:,None]*V(NQ)[:,1:]],axis=1); W=w/(t*qprev)
    c,*_=np.linalg.lstsq(A*W[:,None],t*W,rcond=None); return c
w=np.ones_like(v); q=np.ones_like(v)
best=None
for it in range(200):
    for _ in range(5):
        c=lin(w,q); p,q=pq(c)
    err=np.abs(p/q/t-1)
    if best is None or err.max()<best[0]: best=(err.max(),c.copy())
    w=w*(0.2+err/err.max()); w/=w.mean()
print("lawson best f64 rel:",best[0])
c=best[1]
# nonlinear polish on relative error (minimise high power norm)
def res(c):
    p,q=pq(c); return (p/q/t-1)*1e6
for pw in (1,):
    r=least_squares(lambda c: np.sign(res(c))*np.abs(res(c))**4, c, method='lm', max_nfev=2000)
    c=r.x
p,q=pq(c); print("polished f64 rel:",np.abs(p/q/t-1).max(), "min q",q.min())
a=c[:NP]/s**np.arange(NP); b=np.r_[1.0,c[NP:]]/s**np.arange(NQ)
# normalise so numbers are mid-range for f32: scale both by b-normaliser
scale=a[0]; 
f32=np.float32
def ev(xs,a,b):
    xs=np.clip(xs,f32(-XMAX),f32(XMAX)).astype(f32)
    x2=(xs*xs).astype(f32); P=f32(a[-1])
    for k in a[-2::-1]: P=(P*x2+f32(k)).astype(f32)
    P=(P*xs).astype(f32); Q=f32(b[-1])
    for k in b[-2::-1]: Q=(Q*x2+f32(k)).astype(f32)
    return (P*(f32(1)/Q).astype(f32)).astype(f32)
xs=np.concatenate([np.linspace(-12,12,3000001),np.logspace(-9,1,300001)]).astype(f32)
ref=np.tanh(xs.astype(np.float64))
for sc in (1.0, 1/200.0):
    y=ev(xs,a*sc,b*sc); rel=np.abs(y-ref)/np.abs(ref)
    print("scale",sc,"f32 max rel %.3e (%.2f ulp) max abs %.3e |y|max %.9f"%(rel.max(),rel.max()/2**-24,np.abs(y-ref).max(),np.abs(y).max()))
print("a =",[float(k) for k in a]); print("b =",[float(k) for k in b])
