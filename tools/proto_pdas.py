import numpy as np
from proto_admm import *

def kkt_residuals(qp, q, l, u, x, y):
    P, A = qp['P'], qp['A']
    Ax = A@x
    rp = max(np.maximum(l-Ax, 0).max(), np.maximum(Ax-u, 0).max())
    rd = np.abs(P@x+q+A.T@y).max()
    # complementarity: y+ (Ax-u) , y- (Ax-l)
    yp = np.maximum(y, 0); ym = np.minimum(y, 0)
    fin_u = u < 1e29; fin_l = l > -1e29
    cs = max(np.abs(yp[fin_u]*(Ax-u)[fin_u]).max(initial=0), np.abs(ym[fin_l]*(Ax-l)[fin_l]).max(initial=0),
             np.abs(yp[~fin_u]).max(initial=0), np.abs(ym[~fin_l]).max(initial=0))
    return rp, rd, cs

def pdas(qp, q, l, u, x, y, max_it=100, tol=1e-9):
    """primal-dual active set refinement starting from ADMM (x,y)."""
    P, A = qp['P'], qp['A']; n, m = qp['n'], qp['m']
    Ax = A@x
    eq = (u-l) < 1e-9
    lo = (~eq) & ((y < -1e-7) | (Ax < l+1e-9))
    up = (~eq) & ((y > 1e-7) | (Ax > u-1e-9))
    for it in range(max_it):
        act = eq | lo | up
        b = np.where(up, u, l)[act]
        Aa = A[act]
        na = Aa.shape[0]
        K = np.block([[P+1e-12*np.eye(n), Aa.T], [Aa, -1e-12*np.eye(na)]])
        sol = np.linalg.solve(K, np.concatenate([-q, b]))
        for _ in range(2):
            r = np.concatenate([-q, b]) - np.block([[P, Aa.T], [Aa, np.zeros((na, na))]])@sol
            sol += np.linalg.solve(K, r)
        xn = sol[:n]; yn = np.zeros(m); yn[act] = sol[n:]
        Ax = A@xn
        # violations
        bad_lo_dual = lo & (yn > tol)      # lower-active must have y<=0
        bad_up_dual = up & (yn < -tol)
        viol_lo = (~act) & (Ax < l-tol)
        viol_up = (~act) & (Ax > u+tol)
        nchg = bad_lo_dual.sum()+bad_up_dual.sum()+viol_lo.sum()+viol_up.sum()
        if nchg == 0:
            return xn, yn, it+1, True
        lo = (lo & ~bad_lo_dual) | viol_lo
        up = (up & ~bad_up_dual) | viol_up
    return xn, yn, max_it, False

if __name__ == '__main__':
    import sys
    N = int(sys.argv[1]); B = int(sys.argv[2])
    rng = np.random.default_rng(1)
    qp = build_6dof(N)
    x0, ref = gen_6dof(N, B, rng)
    q, l, u = inst_vectors(qp, x0, ref)
    P, A = qp['P'], qp['A']
    # moderately accurate ADMM, returning y too -> rerun small variant
    xo, it, dn = admm(qp, q, l, u, rho=1000., eps=1e-5, max_iter=2000)
    nact = []
    for b in range(B):
        # crude y estimate: least squares from stationarity on near-active rows is messy; start with y=0 and active from primal only
        xs, ys, nit, ok = pdas(qp, q[b], l[b], u[b], xo[b], np.zeros(qp['m']))
        rp, rd, cs = kkt_residuals(qp, q[b], l[b], u[b], xs, ys)
        Nx = qp['Nx']
        act = np.abs(ys[Nx:]) > 1e-9
        av = act[:Nx].reshape(N+1, 6)
        au = act[Nx:].reshape(N, 3)
        print(b, ok, nit, f'{rp:.1e} {rd:.1e} {cs:.1e}', 'admm it', it[b], 'err', f'{np.abs(xo[b]-xs).max():.1e}', 'act pz', av[:, 2].sum(), 'act v', av[:, 3:].sum(), 'act u', au.sum())
