import numpy as np, sys
from proto_admm import *
from proto_pdas import *
N = 50; B = 512
rng = np.random.default_rng(1)
qp = build_6dof(N)
x0, ref = gen_6dof(N, B, rng)
q, l, u = inst_vectors(qp, x0, ref)
xo, it, dn = admm(qp, q, l, u, rho=1000., eps=1e-5, max_iter=1000)
hard = np.where(it >= 400)[0]
print('hard', len(hard), 'of', B)
xs = np.zeros((len(hard), qp['n']))
for j, b in enumerate(hard):
    xs[j], ys, nit, ok = pdas(qp, q[b], l[b], u[b], xo[b], np.zeros(qp['m']))
    Nx = qp['Nx']; act = np.abs(ys[Nx:]) > 1e-9
    av = act[:Nx].reshape(N+1, 6)
    print(b, ok, 'act pz', av[:, 2].sum(), 'act v', av[:, 3:].sum(), 'act u', act[Nx:].sum())
qh, lh, uh = q[hard], l[hard], u[hard]
Nx = qp['Nx']
def mk(rp, rv_, ru):
    def f(rv, eqr, loose):
        rv = rv.copy()
        idx = np.arange(len(rv))
        box = (idx >= Nx) & (idx < 2*Nx)
        comp = (idx - Nx) % 6
        rv[box & (comp == 2)] = rp
        rv[box & (comp >= 3)] = rv_
        rv[idx >= 2*Nx] = ru
        return rv
    return f
for rho, rp, rvv, ru in [(1000, 1000, 1000, 1000), (1000, 1e4, 1000, 1000), (1000, 1e5, 1000, 1000), (1000, 1e5, 1e4, 1000), (1000, 3e4, 3e3, 1000), (3000, 1e5, 1e4, 3000), (1000, 1e6, 1e4, 1000), (300, 1e5, 3e3, 300),(1000,1e5,1e4,300)]:
    for eps in (1e-3, 1e-6):
        xo2, it2, dn2 = admm(qp, qh, lh, uh, rho=float(rho), eps=eps, rho_rows=mk(rp, rvv, ru))
        err = np.abs(xo2-xs).max(1)
        xo3, it3, dn3 = admm(qp, q[:128], l[:128], u[:128], rho=float(rho), eps=eps, rho_rows=mk(rp, rvv, ru))
        print(f'rho {rho} pz {rp:g} v {rvv:g} u {ru:g} eps {eps:g}: HARD iters mean {it2.mean():.0f} max {it2.max()} err max {err.max():.1e} | first128 iters mean {it3.mean():.0f} max {it3.max()}', flush=True)
