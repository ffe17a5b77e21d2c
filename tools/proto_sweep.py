import numpy as np, sys, time
from proto_admm import *
N = int(sys.argv[1]); B = int(sys.argv[2])
rng = np.random.default_rng(1)
qp = build_6dof(N)
x0, ref = gen_6dof(N, B, rng)
q, l, u = inst_vectors(qp, x0, ref)
xs, it, dn = admm(qp, q, l, u, rho=1000., eps=1e-11, max_iter=40000, check=100)
print('x* iters', it.max(), dn.all())
for rho in [300., 1000., 3000., 1e4, 3e4]:
    for eqm in [1e3]:
        for eps in (1e-3, 1e-4, 1e-5, 1e-6):
            t = time.time()
            xo, it, dn = admm(qp, q, l, u, rho=rho, eps=eps, rho_eq_mult=eqm)
            err = np.abs(xo-xs).max(1)
            print(f'rho={rho:g} eqm={eqm:g} eps={eps:g}: iters mean {it.mean():.0f} p50 {np.median(it):.0f} p99 {np.percentile(it,99):.0f} max {it.max()} solved {dn.mean():.3f} err max {err.max():.2e} med {np.median(err):.2e}  ({time.time()-t:.1f}s)', flush=True)
