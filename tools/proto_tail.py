import numpy as np, sys
from proto_admm import *
N = int(sys.argv[1]); B = int(sys.argv[2])
sys.path.insert(0, '..')
from trajectory_generation_amd import synth
qp = build_6dof(N)
x0, ref = synth.mpc6(0x4D5043 + 2, N, 0.1, B)
q, l, u = inst_vectors(qp, x0, ref)
Nx = qp['Nx']
def mk(rp, rv_, ru, rxy=None):
    def f(rv, eqr, loose):
        rv = rv.copy(); idx = np.arange(len(rv))
        box = (idx >= Nx) & (idx < 2*Nx); comp = (idx - Nx) % 6
        rv[box & (comp == 2)] = rp
        if rxy is not None: rv[box & (comp < 2)] = rxy
        rv[box & (comp >= 3)] = rv_
        rv[idx >= 2*Nx] = ru
        return rv
    return f
def tile_cost(it):
    # cost of 16-instance tiles that run until their slowest member finishes
    n = len(it)//16*16
    return it[:n].reshape(-1, 16).max(1).mean()
for rho, rp, rvv, ru in [(1000,1000,1000,1000),(1000,3000,1000,1000),(1000,1e4,1000,1000),(1000,3e4,1000,1000),(1000,1e4,3000,1000),(1000,1e4,1000,3000),(2000,2e4,2000,2000),(500,1e4,500,500)]:
    for eps in (1e-3,):
        xo, it, dn = admm(qp, q, l, u, rho=float(rho), eps=eps, rho_rows=mk(rp, rvv, ru))
        print(f'rho {rho} pz {rp:g} v {rvv:g} u {ru:g} eps {eps:g}: iters mean {it.mean():.1f} p99 {np.percentile(it,99):.0f} max {it.max()} tile-mean-of-max {tile_cost(it):.1f} solved {dn.mean():.4f}', flush=True)
