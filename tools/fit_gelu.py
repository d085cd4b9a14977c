import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares, minimize
x = np.linspace(-9, 9, 40001)
g = 0.5*x*(1+erf(x/np.sqrt(2)))
def model(c, x):
    s = x*x
    p = np.zeros_like(x)
    for a in c[::-1]:
        p = p*s + a
    t = np.clip(x*p, -80, 80)
    return x/(1+np.exp(-t))
for nc in (2,3,4):
    c0 = np.array([1.5957691, 0.0713548] + [0.0]*(nc-2))
    # minimax via iterative reweighting (Lawson-like) using least_squares with high p-norm
    best=None
    c=c0.copy()
    for p in (2,4,8,16,32,64):
        f = lambda c: np.sign(model(c,x)-g)*np.abs(model(c,x)-g)**(p/2)
        r = least_squares(f, c, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=2000)
        c = r.x
    err = np.abs(model(c,x)-g)
    print(nc, "coefs", repr(c.tolist()), "max abs err %.3e"%err.max(), "at x=%.2f"%x[err.argmax()])
    # float32 evaluation check
    xf = x.astype(np.float32); cf = c.astype(np.float32)
    s = xf*xf; p = np.zeros_like(xf)
    for a in cf[::-1]: p = p*s + a
    t = xf*p*np.float32(-1.4426950408889634)
    e = np.exp2(np.clip(t,-126,126)).astype(np.float32)
    gf = xf/(np.float32(1)+e)
    print("   fp32 eval max abs err %.3e"%np.abs(gf-g).max())
