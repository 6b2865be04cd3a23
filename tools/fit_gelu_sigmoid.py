import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares
x = np.linspace(-8, 8, 40001)
Phi = 0.5*(1+erf(x/np.sqrt(2)))
phi = np.exp(-x*x/2)/np.sqrt(2*np.pi)
g = x*Phi; dg = Phi + x*phi
def model(a, x):
    x2 = x*x
    p = a[-1]
    for c in a[-2::-1]: p = p*x2 + c
    u = x*p
    s = 1/(1+np.exp(-u))
    return s, p
for deg in (2,3,4):
    a0 = np.array([1.5957691216, 0.0713548162726] + [0.0]*(deg-1))[:deg+0]
    if deg==2: a0=np.array([1.5957691216,0.0713548162726])
    def res(a):
        s,_ = model(a,x)
        return (x*s - g)
    # minimax via iteratively reweighted LS
    a = least_squares(res, a0).x
    w = np.ones_like(x)
    for it in range(60):
        r = res(a)
        w = w*(1+ 4*np.abs(r)/np.abs(r).max())
        w/=w.mean()
        a = least_squares(lambda a: res(a)*np.sqrt(w), a).x
    s,p = model(a,x)
    err = np.abs(x*s-g).max()
    # derivative of approx: s + x*s*(1-s)*(d/dx (x p)) 
    x2=x*x
    # d(x p)/dx = p + x*dp/dx ; p = sum a_i x^(2i) -> x dp/dx = sum 2i a_i x^(2i)
    dp = sum((2*i+1)*a[i]*x2**i for i in range(len(a)))
    dapprox = s + x*s*(1-s)*dp
    derr = np.abs(dapprox-dg).max()
    print(deg, list(a), "gelu err", err, "gelu' err", derr, "rel-to-bf16: ", err/ (2**-9))
