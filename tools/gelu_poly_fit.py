import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf
def Phi(x): return 0.5*(1+erf(x/np.sqrt(2)))
def phi(x): return np.exp(-x*x/2)/np.sqrt(2*np.pi)
def fit(f, c, deg):
    # fit g(s) = f(sqrt(s))/sqrt(s) on s in [0, c^2] by Chebyshev interpolation then refine (Remez-lite via lstsq on dense grid weighted)
    n=4000
    t=np.cos(np.pi*(np.arange(n)+0.5)/n)        # cheb nodes in [-1,1]
    s=(t+1)/2*c*c
    x=np.sqrt(s)
    g=np.where(x>1e-6, f(np.maximum(x,1e-6))/np.maximum(x,1e-6), f(1e-6)/1e-6)
    # error of interest: |x * (g_hat - g)| (abs error of f), so weight by x
    w=np.maximum(x,0.05)
    V=C.chebvander(t,deg)
    coef=np.linalg.lstsq(V*w[:,None], g*w, rcond=None)[0]
    # iterate reweighting to approach minimax
    for it in range(60):
        err=(V@coef-g)*x
        a=np.abs(err); 
        w=w*(1+ 2*a/a.max())
        w/=w.max()
        coef=np.linalg.lstsq(V*w[:,None], g*w, rcond=None)[0]
    err=(V@coef-g)*x
    # convert to monomial in s
    p=C.cheb2poly(coef)  # poly in t
    # t = 2 s/c^2 - 1
    from numpy.polynomial import polynomial as P
    tt=np.array([-1.0, 2.0/(c*c)])
    mono=np.zeros(1)
    pw=np.ones(1)
    for k,a in enumerate(p):
        mono=P.polyadd(mono, a*pw)
        pw=P.polymul(pw,tt)
    return mono, np.abs(err).max()
for c in (3.5,4.0,4.5,5.0):
    for deg in (5,6,7,8,9,10):
        m1,e1=fit(lambda x: Phi(x)-0.5, c, deg)
        m2,e2=fit(lambda x: Phi(x)+x*phi(x)-0.5, c, deg)
        print(f"c={c} deg={deg}: Phi err {e1:.2e} (tail {1-Phi(c):.1e})  dGELU err {e2:.2e} (tail {abs(Phi(c)+c*phi(c)-1):.1e})")

print("---- fp32 Horner evaluation check")
def horner32(coef, s):
    # coef low->high, emulate fma in fp32: r = fl32(r*s + c)
    r = np.full_like(s, np.float32(coef[-1]), dtype=np.float32)
    for c in coef[-2::-1]:
        r = (r.astype(np.float64)*s.astype(np.float64) + np.float64(np.float32(c))).astype(np.float32)
    return r
for c,deg in ((4.0,8),(4.0,9),(4.5,9),(4.5,10)):
    mq,eq=fit(lambda x: Phi(x)-0.5, c, deg)
    mr,er=fit(lambda x: Phi(x)+x*phi(x)-0.5, c, deg)
    x=np.linspace(-c,c,400001).astype(np.float32)
    s=(x.astype(np.float64)**2).astype(np.float32)
    Q=horner32(mq,s); R=horner32(mr,s)
    cdf=(0.5+x.astype(np.float64)*Q).astype(np.float32)
    dg=(0.5+x.astype(np.float64)*R).astype(np.float32)
    xx=x.astype(np.float64)
    e1=np.abs(cdf-Phi(xx)).max(); e2=np.abs(dg-(Phi(xx)+xx*phi(xx))).max()
    y=xx*cdf; yt=xx*Phi(xx)
    rel=np.abs(y-yt)/np.maximum(np.abs(yt),1e-30)
    print(f"c={c} deg={deg}: fp32 Phi err {e1:.2e}, dGELU err {e2:.2e}; gelu max abs err {np.abs(y-yt).max():.2e}; rel err at x>-2: {rel[x>-2].max():.2e}, x>-3: {rel[x>-3].max():.2e}")
    print("  Q:", ", ".join(f"{v:.9e}" for v in mq))
    print("  R:", ", ".join(f"{v:.9e}" for v in mr))

print("---- the shipped routine (common.h gelu_both_fast2: the unconstrained c = 4, degree-8 pair at the clamped argument t; gelu = max(x, t) * Phi(t)), emulated in fp32 over |x| <= 20 (ADVICE r4)")
QC = [3.989227094e-01, -6.641059427e-02, 9.877475989e-03, -1.133921717e-03, 9.890799001e-05, -6.294988571e-06, 2.716148569e-07, -7.003438364e-09, 8.063375031e-11]
RC = [7.976095497e-01, -2.648265329e-01, 5.845610030e-02, -8.716320413e-03, 9.073266031e-04, -6.495757385e-05, 3.028347546e-06, -8.218800834e-08, 9.796052989e-10]
x = np.concatenate([np.linspace(-20, 20, 2000001), np.array([-4.0, 4.0, np.nextafter(4.0, 5.0), -np.nextafter(4.0, 5.0)])]).astype(np.float32)
t = np.clip(x, -4.0, 4.0).astype(np.float32)
s = (t.astype(np.float64) ** 2).astype(np.float32)
cdf = (np.float32(0.5) + (t.astype(np.float64) * horner32(QC, s)).astype(np.float32)).astype(np.float32)
dg = (np.float32(0.5) + (t.astype(np.float64) * horner32(RC, s)).astype(np.float32)).astype(np.float32)
xx = x.astype(np.float64)
y = (np.maximum(x, t).astype(np.float64) * cdf).astype(np.float32)
y_r4 = (xx * cdf).astype(np.float32)
for name, m in (('|x| <= 4', np.abs(xx) <= 4.0), ('-20 <= x < -4', (xx < -4.0)), ('4 < x <= 20', (xx > 4.0))):
    yt = xx[m] * Phi(xx[m])
    rel = np.abs(y[m] - yt).max() / 4.0 if name.startswith('4 <') else float('nan')
    print(f"{name:14s}: |Phi err| {np.abs(cdf[m] - Phi(xx[m])).max():.2e}   |GELU' err| {np.abs(dg[m] - (Phi(xx[m]) + xx[m] * phi(xx[m]))).max():.2e}   |gelu err| {np.abs(y[m] - yt).max():.2e}"
          + (f" (relative {np.abs((y[m] - yt) / yt).max():.2e})" if name.startswith('4 <') else '')
          + f"   round 4's x * Phi(t): |gelu err| {np.abs(y_r4[m] - yt).max():.2e}")

# ---- the constrained fit behind the shipped coefficients:  python tools/gelu_poly_fit.py --constrained
# minimax (iteratively re-weighted least squares on Chebyshev nodes) of g(s) = f(sqrt s) / sqrt s on [0, c^2] with the equality constraint
# c g(c^2) = 1/2 eliminated into the constant coefficient, so that the clamped argument gives exactly the limit of Phi / GELU' beyond c
import sys
if '--constrained' in sys.argv:
    from numpy.polynomial import polynomial as P
    def fit_constrained(f, c, deg, target_end):
        # g(s) = f(sqrt s)/sqrt s on s in [0,c^2]; constraint: c*g(c^2) = target_end (equality, by elimination)
        n=6000
        t=np.cos(np.pi*(np.arange(n)+0.5)/n); s=(t+1)/2*c*c; x=np.sqrt(s)
        g=np.where(x>1e-6, f(np.maximum(x,1e-6))/np.maximum(x,1e-6), f(1e-6)/1e-6)
        V=C.chebvander(t,deg); v1=C.chebvander(np.array([1.0]),deg)[0]   # value at t=1 (s=c^2)
        gend=target_end/c
        # eliminate coefficient 0: coef0 = gend - sum_{k>=1} coef_k * v1[k]/v1[0]  (v1[k]=1 for cheb at t=1)
        A=V[:,1:]-V[:,[0]]*v1[1:][None,:]/v1[0]; b=g-V[:,0]*gend/v1[0]
        w=np.maximum(x,0.05)
        ck=np.linalg.lstsq(A*w[:,None], b*w, rcond=None)[0]
        for it in range(80):
            coef=np.concatenate([[ (gend-(ck*v1[1:]).sum())/v1[0] ], ck])
            err=(V@coef-g)*x; a=np.abs(err); w=w*(1+2*a/a.max()); w/=w.max()
            ck=np.linalg.lstsq(A*w[:,None], b*w, rcond=None)[0]
        coef=np.concatenate([[ (gend-(ck*v1[1:]).sum())/v1[0] ], ck])
        err=(V@coef-g)*x
        p=C.cheb2poly(coef); tt=np.array([-1.0,2.0/(c*c)]); mono=np.zeros(1); pw=np.ones(1)
        for k,a in enumerate(p):
            mono=P.polyadd(mono,a*pw); pw=P.polymul(pw,tt)
        return mono, np.abs(err).max()
    def horner32(coef,s):
        r=np.full_like(s,np.float32(coef[-1]),dtype=np.float32)
        for c in coef[-2::-1]:
            r=(r.astype(np.float64)*s.astype(np.float64)+np.float64(np.float32(c))).astype(np.float32)
        return r
    for c,deg in ((4.0,8),(4.5,8),(4.5,9),(5.0,9)):
        mq,eq=fit_constrained(lambda x: Phi(x)-0.5, c, deg, 0.5)
        mr,er=fit_constrained(lambda x: Phi(x)+x*phi(x)-0.5, c, deg, 0.5)
        x=np.linspace(-20,20,2000001).astype(np.float32)
        t=np.clip(x,-c,c).astype(np.float32); s=(t.astype(np.float64)**2).astype(np.float32)
        cdf=(np.float32(0.5)+(t.astype(np.float64)*horner32(mq,s)).astype(np.float32)).astype(np.float32)
        dg=(np.float32(0.5)+(t.astype(np.float64)*horner32(mr,s)).astype(np.float32)).astype(np.float32)
        xx=x.astype(np.float64); y=(xx*cdf).astype(np.float32)
        print(f"c={c} deg={deg}: fit errs {eq:.2e} {er:.2e} | fp32 over |x|<=20: Phi {np.abs(cdf-Phi(xx)).max():.2e}  GELU' {np.abs(dg-(Phi(xx)+xx*phi(xx))).max():.2e}  gelu {np.abs(y-xx*Phi(xx)).max():.2e}; cdf(-20)={cdf[0]:.3e} dg(-20)={dg[0]:.3e} cdf(20)-1={cdf[-1]-1:.3e} dg(20)-1={dg[-1]-1:.3e}")
        print("  Q:", ", ".join(f"{v:.9e}" for v in mq)); print("  R:", ", ".join(f"{v:.9e}" for v in mr))
