import json, glob, numpy as np
f32=np.float32
LOG2E=f32(1.44269504088896340736)
def sig_fast(v):
    e=np.exp2((v*(-LOG2E)).astype(f32)).astype(f32); return (f32(1)/(f32(1)+e)).astype(f32)
def tanh_fast(v):
    return (sig_fast((v*f32(2)).astype(f32))*f32(2)-f32(1)).astype(f32)
def sig_acc(v): return (1/(1+np.exp(-v.astype(np.float64)))).astype(f32)
def tanh_acc(v): return np.tanh(v.astype(np.float64)).astype(f32)
A=[4.89352455891786e-03,6.37261928875436e-04,1.48572235717979e-05,5.12229709037114e-08,-8.60467152213735e-11,2.00018790482477e-13,-2.76076847742355e-16]
B=[4.89352518554385e-03,2.26843463243900e-03,1.18534705686654e-04,1.19825839466702e-06]
def tanh_rat(v):
    x=np.clip(v,f32(-9),f32(9)).astype(f32); x2=(x*x).astype(f32)
    p=f32(A[6])
    for a in A[5::-1]: p=(p*x2+f32(a)).astype(f32)
    p=(p*x).astype(f32)
    q=f32(B[3])
    for b in B[2::-1]: q=(q*x2+f32(b)).astype(f32)
    return (p/q).astype(f32)
def sig_rat(v): return (tanh_rat((v*f32(0.5)).astype(f32))*f32(0.5)+f32(0.5)).astype(f32)
def run(j, sig, tanh, dt=f32):
    L=j['layers']; W=np.array(L[0]['weights'][0],dt); U=np.array(L[0]['weights'][1],dt); b=np.array(L[0]['weights'][2],dt)
    Wd=np.array(L[1]['weights'][0],dt)[:,0]; bd=dt(L[1]['weights'][1][0]); H=U.shape[0]
    x=np.array(j['input_batch'],dt); h=np.zeros(H,dt); c=np.zeros(H,dt); y=np.zeros(len(x),dt)
    for t in range(len(x)):
        z=(h@U + b + W[0]*x[t]).astype(dt)
        i=sig(z[:H]); f=sig(z[H:2*H]); g=tanh(z[2*H:3*H]); o=sig(z[3*H:])
        c=(f*c+i*g).astype(dt); h=(o*tanh(c)).astype(dt); y[t]=h@Wd+bd
    return y
for fn in sorted(glob.glob('tests/golden/models/*.json')):
    j=json.load(open(fn)); gold=np.array(j['output_batch'],np.float64)
    y64=run(j, lambda v:1/(1+np.exp(-v)), np.tanh, np.float64)
    res={}
    for name,(s,t) in dict(acc=(sig_acc,tanh_acc), fast=(sig_fast,tanh_fast), fast_sig_acc_tanh=(sig_fast,tanh_acc), rat=(sig_rat,tanh_rat), fastsig_rattanh=(sig_fast,tanh_rat)).items():
        y=run(j,s,t); res[name]=(np.abs(y-y64).max(), np.abs(y-gold).max())
    print(fn.split('/')[-1][:22], 'gold-vs-f64 %.2e'%np.abs(gold-y64).max(), ' '.join(f"{k}: {v[0]:.1e}/{v[1]:.1e}" for k,v in res.items()))
print("---- own fit")
A2=[0.9999999933888696, 0.13084010352004496, 0.003103956503888039, 1.1154311501654368e-05, -2.0225239996482085e-08, 5.277955823366522e-11, -8.488730763828322e-14]
B2=[1.0, 0.46417337453820245, 0.02449517952619233, 0.00025461456545097517]
def tanh_own(v, XM=f32(7.9)):
    x=np.clip(v,-XM,XM).astype(f32); x2=(x*x).astype(f32)
    p=f32(A2[6])
    for a in A2[5::-1]: p=(p*x2+f32(a)).astype(f32)
    p=(p*x).astype(f32)
    q=f32(B2[3])
    for b in B2[2::-1]: q=(q*x2+f32(b)).astype(f32)
    return (p*(f32(1)/q).astype(f32)).astype(f32)
xs=np.concatenate([np.linspace(-12,12,2000001),np.logspace(-9,1,300001),-np.logspace(-9,1,300001)]).astype(f32)
ref=np.tanh(xs.astype(np.float64)); y=tanh_own(xs); nz=ref!=0
print("own: max rel %.2e  max abs %.2e  max|y| %.9f"%( (np.abs(y-ref)[nz]/np.abs(ref[nz])).max(), np.abs(y-ref).max(), np.abs(y).max()))
y=tanh_rat(xs); print("eigen: max rel %.2e  max abs %.2e max|y| %.9f"%( (np.abs(y-ref)[nz]/np.abs(ref[nz])).max(), np.abs(y-ref).max(), np.abs(y).max()))
def sig_own(v): return (tanh_own((v*f32(0.5)).astype(f32))*f32(0.5)+f32(0.5)).astype(f32)
for fn in sorted(glob.glob('tests/golden/models/*.json')):
    j=json.load(open(fn)); gold=np.array(j['output_batch'],np.float64)
    y64=run(j, lambda v:1/(1+np.exp(-v)), np.tanh, np.float64)
    out=[]
    for name,(s,t) in dict(fastsig_own=(sig_fast,tanh_own), own_all=(sig_own,tanh_own)).items():
        y=run(j,s,t); out.append(f"{name}: {np.abs(y-y64).max():.1e}/{np.abs(y-gold).max():.1e}")
    print(fn.split('/')[-1][:22], ' '.join(out))
print("---- g accurate (own rational), tanh(c) cheap 2*sig(2c)-1")
for fn in sorted(glob.glob('tests/golden/models/*.json')):
    j=json.load(open(fn)); gold=np.array(j['output_batch'],np.float64)
    y64=run(j, lambda v:1/(1+np.exp(-v)), np.tanh, np.float64)
    # custom run with separate functions for g and tanh(c)
    L=j['layers']; dt=f32
    W=np.array(L[0]['weights'][0],dt); U=np.array(L[0]['weights'][1],dt); b=np.array(L[0]['weights'][2],dt)
    Wd=np.array(L[1]['weights'][0],dt)[:,0]; bd=dt(L[1]['weights'][1][0]); H=U.shape[0]
    x=np.array(j['input_batch'],dt)
    for name,(tg,tc) in dict(g_rat_c_fast=(tanh_own,tanh_fast), both_rat=(tanh_own,tanh_own), g_fast_c_rat=(tanh_fast,tanh_own)).items():
        h=np.zeros(H,dt); c=np.zeros(H,dt); y=np.zeros(len(x),dt)
        for t in range(len(x)):
            z=(h@U + b + W[0]*x[t]).astype(dt)
            i=sig_fast(z[:H]); f=sig_fast(z[H:2*H]); g=tg(z[2*H:3*H]); o=sig_fast(z[3*H:])
            c=(f*c+i*g).astype(dt); h=(o*tc(c)).astype(dt); y[t]=h@Wd+bd
        print(fn.split('/')[-1][:22], name, "%.2e / %.2e"%(np.abs(y-y64).max(), np.abs(y-gold).max()))
