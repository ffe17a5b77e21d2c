import numpy as np, sys, itertools
from proto_admm import *
sys.path.insert(0, '..')
from trajectory_generation_amd import synth
N = int(sys.argv[1]); B = int(sys.argv[2]); epss = [float(e) for e in sys.argv[3].split(',')]
qp = build_6dof(N)
x0, ref = synth.mpc6(0x4D5043 + 2, N, 0.1, B)
q, l, u = inst_vectors(qp, x0, ref)
Nx = qp['Nx']
def mk(rp, rv_, ru):
    def f(rv, eqr, loose):
        rv = rv.copy(); idx = np.arange(len(rv))
        box = (idx >= Nx) & (idx < 2*Nx); comp = (idx - Nx) % 6
        rv[box & (comp == 2)] = rp; rv[box & (comp >= 3)] = rv_; rv[idx >= 2*Nx] = ru
        return rv
    return f
def tile_cost(it):
    n = len(it)//16*16
    return it[:n].reshape(-1, 16).max(1).mean()
for rho, mp, mv, mu in itertools.product([500., 1000., 2000.], [1, 10, 30, 100], [1, 3], [1, 3, 10]):
    for eps in epss:
        xo, it, dn = admm(qp, q, l, u, rho=rho, eps=eps, rho_rows=mk(rho*mp, rho*mv, rho*mu), max_iter=2000)
        print(f'N {N} rho {rho:g} mp {mp} mv {mv} mu {mu} eps {eps:g}: mean {it.mean():.1f} p99 {np.percentile(it,99):.0f} max {it.max()} tilemax {tile_cost(it):.1f} solved {dn.mean():.4f}', flush=True)
