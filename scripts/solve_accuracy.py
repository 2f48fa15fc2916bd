"""Backward / forward error of the Cholesky solve path on the normal equations of a late IPM iteration.

For a planted C4 member (1024x2048, seed argv[1]) the numpy oracle is traced to the start of its LAST iteration,
M = A.diag(x/z).A^T is formed there (numpy), and M v = r (r = b + A.(d*c), the first sym_solve right-hand side,
newton_equations.rs:220) is solved by
  * the C oracle's Cholesky + substitution (oracle/oracle_linalg.c),
  * lpipm_k_potrf + lpipm_k_chol_solve (GPU), super-block width from LPIPM_SUPER (default 1024),
and compared with a solution refined in extended precision.  Tells whether the explicit-inverse solve
loses digits a substitution keeps.   usage: LPIPM_SUPER=512 python scripts/solve_accuracy.py 73 [it_from_end]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import lp_amd
from lp_amd import synth
from oracle import capi as oracle, oracle_np

seed = int(sys.argv[1]); back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m, n = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1024, 2048)
A, b, c, xs = synth.planted_lp(seed, m, n)
tr = []
r = oracle_np.solve(A, b, c, trace=tr)
x, y, z, tau, kappa = tr[-back]
d = x / z
print(f"seed {seed}: {r.iterations} iterations; iterate {len(tr)-back+1}: d range {d.min():.2e} .. {d.max():.2e}")
M = A @ (d[:, None] * A.T)
rhs = b + A @ (d * c)
LD = np.longdouble
Ml = M.astype(LD)
cf = sla.cho_factor(M, lower=True)
v = sla.cho_solve(cf, rhs).astype(LD)
for _ in range(8):   # refinement with extended-precision residuals
    res = (rhs.astype(LD) - Ml @ v).astype(np.float64)
    v = v + sla.cho_solve(cf, res).astype(LD)
vt = v
nrmM = np.abs(M).sum(1).max()
def report(name, vv):
    res = np.abs(rhs.astype(LD) - Ml @ vv.astype(LD)).max()
    print(f"  {name:28s} backward err {float(res / (nrmM * np.abs(vv).max() + np.abs(rhs).max())):.2e}   "
          f"forward err {float(np.abs(vv.astype(LD) - vt).max() / np.abs(vt).max()):.2e}")
report("scipy cho_solve", sla.cho_solve(cf, rhs))
Lo = M.copy(); oracle.lib().oracle_cholesky(m, Lo.ctypes.data_as(oracle.C.POINTER(oracle.C.c_double)))
vo = np.empty(m); p = lambda a: a.ctypes.data_as(oracle.C.POINTER(oracle.C.c_double))
oracle.lib().oracle_cholesky_solve(m, p(Lo), p(rhs), p(vo))
report("C oracle (substitution)", vo)
ctx = lp_amd.Context(0)
L, info, _ = ctx.k_potrf(M)
V, _ = ctx.k_chol_solve(m, rhs)
report(f"GPU super={os.environ.get('LPIPM_SUPER', '1024')}", V[0])
Lg = np.tril(L).astype(LD); Lor = np.tril(Lo).astype(LD)
fres = lambda F: float(np.abs(np.tril(Ml - F @ F.T)).max() / np.abs(M).max())
dM = np.sqrt(np.diag(M))
fres_s = lambda F: float(np.abs(np.tril((Ml - F @ F.T) / np.outer(dM, dM))).max())     # scaled: |dM_ij| / sqrt(M_ii M_jj)
print(f"  factorisation residual |M - L L^T|/|M|: gpu {fres(Lg):.2e} oracle {fres(Lor):.2e};  scaled by sqrt(M_ii M_jj): gpu {fres_s(Lg):.2e} oracle {fres_s(Lor):.2e}")
print(f"  |L_gpu - L_oracle| / |L|: {np.abs(np.tril(L) - np.tril(Lo)).max() / np.abs(Lo).max():.2e}, info {info}; cond(M) ~ {np.linalg.cond(M):.2e}")
