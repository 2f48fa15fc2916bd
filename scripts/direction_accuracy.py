"""How accurately the device solve path reproduces a Newton direction at a late iterate (what the ratio test sees).
Traces the numpy oracle to iteration `it` of a planted C4 member, forms the predictor's second sym_solve there
(newton_equations.rs:188, :214-225: r1 = rhat_d - rhat_xs/x, rhs = rhat_p + A.(d*r1), v = M^-1 rhs, u = d*(A^T v - r1))
in extended precision, and compares u from (a) the C oracle's Cholesky + substitution, (b) the device kernels
unrefined, (c) refined with the device residual kernel: max_i |u_i - u_true_i| / x_i.
usage: python scripts/direction_accuracy.py <seed> [iteration]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import lp_amd
from lp_amd import synth
from oracle import capi as oracle, oracle_np
seed = int(sys.argv[1]); m, n = 1024, 2048
A, b, c, xs = synth.planted_lp(seed, m, n)
tr = []; r = oracle_np.solve(A, b, c, trace=tr)
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(tr)
x, y, z, tau, kappa = tr[it - 1]
LD = np.longdouble
d = x / z
rP = b * tau - A @ x; rD = c * tau - A.T @ y - z
mu = (x @ z + tau * kappa) / (n + 1)
xs_hat = -(x * z)                                  # gamma = 0, eta = 1
r1 = rD - xs_hat / x
rhs = rP + A @ (d * r1)
M = A @ (d[:, None] * A.T)
Ml = (A.astype(LD) * d.astype(LD)) @ A.T.astype(LD)
rhsl = rP.astype(LD) + A.astype(LD) @ (d.astype(LD) * r1.astype(LD))
cf = sla.cho_factor(M, lower=True)
v = sla.cho_solve(cf, rhs).astype(LD)
for _ in range(10): v = v + sla.cho_solve(cf, (rhsl - Ml @ v).astype(np.float64)).astype(LD)
u_true = (d.astype(LD) * (A.T.astype(LD) @ v - r1.astype(LD))).astype(np.float64)
def score(vv, name):
    u = d * (A.T @ vv - r1)
    e = np.abs(u - u_true) / x
    ratio = np.where(u < 0, x / -np.where(u < 0, u, -1.0), np.inf).min()
    print(f"  {name:26s} max |du_i|/x_i {e.max():.2e} (at x_i = {x[e.argmax()]:.1e}); |dv|/|v| {np.abs(vv - v.astype(np.float64)).max() / np.abs(v).max():.1e}; min ratio {ratio:.6f}")
print(f"seed {seed} iterate {it} of {r.iterations}: mu {mu:.2e}, d range {d.min():.1e}..{d.max():.1e}; true min ratio "
      f"{np.where(u_true < 0, x / -np.where(u_true < 0, u_true, -1.0), np.inf).min():.6f}")
score(sla.cho_solve(cf, rhs), "LAPACK cho_solve")
p = lambda a: a.ctypes.data_as(oracle.C.POINTER(oracle.C.c_double))
Lo = M.copy(); oracle.lib().oracle_cholesky(m, p(Lo)); vo = np.empty(m); oracle.lib().oracle_cholesky_solve(m, p(Lo), p(rhs), p(vo))
score(vo, "C oracle")
ctx = lp_amd.Context(0); ctx.upload_arrays(A, b, c)
Mg, _ = ctx.k_adat(d)
Mg = np.tril(Mg) + np.tril(Mg, -1).T
L, info, _ = ctx.k_potrf(Mg)
V0, _ = ctx.k_chol_solve(m, rhs)
score(V0[0], "device, unrefined")
rho = ctx.k_symv_residual(Mg, V0[0], rhs)
D, _ = ctx.k_chol_solve(m, rho[0])
score(V0[0] + D[0], "device, refined (dd resid)")
rho2 = rhs - Mg @ V0[0]
D2, _ = ctx.k_chol_solve(m, rho2)
score(V0[0] + D2[0], "device, refined (plain)")
print(f"  |M_dev - M_true|/|M| {np.abs(np.tril(Mg - Ml.astype(np.float64))).max() / np.abs(M).max():.1e}; |M_np - M_true|/|M| {np.abs(np.tril(M - Ml.astype(np.float64))).max() / np.abs(M).max():.1e}")
