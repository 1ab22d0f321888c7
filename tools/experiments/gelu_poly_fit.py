"""can the GELU of the fused epilogues do without v_exp / v_rcp?  Weighted minimax fits (Lawson iteration) of
erf(a / sqrt 2) = a * Q on [0, c], Q a polynomial of degree n in a^2 ("odd") or in a ("full"); printed: max |erf error| and max error
of 2 gelu = |x| * erf error, and the tail 1 - erf(c / sqrt 2) the clamp leaves behind.  Result (round 5): n >= 8 on c = 3.5 for 2e-6,
but c = 3.5 leaves 4.7e-4 * |x| beyond the clamp; c >= 5 needs n >= 10: n + 6 instructions against 13 + 2 transcendental slots today."""
import numpy as np
from scipy.special import erf
from numpy.polynomial import chebyshev as C, polynomial as P
def fit(c, n, odd=True, iters=30):
    # approximate E(a)=erf(a/sqrt2) on [0,c] by a*Q where Q is poly of degree n in u=a^2 (odd=True) or in a (odd=False)
    # weighted minimax (weight a: error of 2gelu = a*err) via iteratively reweighted least squares (Lawson)
    a = np.linspace(1e-6, c, 20001)
    tgt = erf(a/np.sqrt(2))
    var = a*a if odd else a
    V = np.vander(var, n+1, increasing=True) * a[:,None]   # E = a*sum q_k var^k
    wt = a.copy()                                           # error weight
    lw = np.ones_like(a)
    for _ in range(iters):
        W = np.sqrt(lw)*wt
        q,_,_,_ = np.linalg.lstsq(V*W[:,None], tgt*W, rcond=None)
        err = np.abs((V@q - tgt)*wt)
        lw = lw*(err+1e-300); lw/=lw.sum()
    e = (V@q - tgt)
    return q, np.abs(e).max(), np.abs(e*a).max()
for c in (3.5, 4.0, 4.5, 5.0):
    for n in (4,5,6,7,8):
        q, e1, e2 = fit(c, n, True)
        q2, f1, f2 = fit(c, n, False)
        print(f"c={c} n={n}: odd-poly erf err {e1:.2e} gelu2 err {e2:.2e} | full-poly erf err {f1:.2e} gelu2 err {f2:.2e}  tail 1-erf(c)={1-erf(c/np.sqrt(2)):.1e}")
