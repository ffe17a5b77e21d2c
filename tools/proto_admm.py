"""Throw-away convergence prototype (numpy) used to choose the ADMM variant and rho.

Not part of the product path or of the oracle; kept as the record of how the
fixed-rho settings were selected.
"""
import numpy as np, sys, time

INF = 1e30

def build_6dof(N, dt=0.1, w=7000., r=500., vmax=10., amax=5., minalt=-1., smooth=0.):
    nx, nu = 6, 3
    A = np.eye(6); A[0:3, 3:6] = dt*np.eye(3)
    B = np.zeros((6, 3)); B[3:6, :] = dt*np.eye(3)
    Q = np.diag([w, w, w, 0, 0, 0.]); R = r*np.eye(3)
    Nx, Nu = nx*(N+1), nu*N
    n = Nx+Nu; m = 2*Nx+Nu
    P = np.zeros((n, n))
    for i in range(N+1): P[nx*i:nx*i+nx, nx*i:nx*i+nx] = Q
    for i in range(N): P[Nx+nu*i:Nx+nu*i+nu, Nx+nu*i:Nx+nu*i+nu] = R
    if smooth > 0:
        D = np.zeros((nu*(N-1), Nu))
        for i in range(N-1):
            D[nu*i:nu*i+nu, nu*i:nu*i+nu] = -np.eye(nu)
            D[nu*i:nu*i+nu, nu*(i+1):nu*(i+1)+nu] = np.eye(nu)
        P[Nx:, Nx:] += smooth*D.T@D
    Ac = np.zeros((m, n))
    Ac[:Nx, :Nx] = -np.eye(Nx)
    for i in range(1, N+1):
        Ac[nx*i:nx*i+nx, nx*(i-1):nx*i] = A
        Ac[nx*i:nx*i+nx, Nx+nu*(i-1):Nx+nu*i] = B
    Ac[Nx:2*Nx, :Nx] = np.eye(Nx)
    Ac[2*Nx:, Nx:] = np.eye(Nu)
    xmin = np.array([-INF, -INF, minalt, -vmax, -vmax, -vmax]); xmax = np.array([INF, INF, INF, vmax, vmax, vmax])
    l = np.concatenate([np.zeros(Nx), np.tile(xmin, N+1), np.tile([-amax]*3, N)])
    u = np.concatenate([np.zeros(Nx), np.tile(xmax, N+1), np.tile([amax]*3, N)])
    return dict(P=P, A=Ac, l=l, u=u, Q=Q, nx=nx, nu=nu, N=N, n=n, m=m, Nx=Nx)

def gen_6dof(N, B, rng, dt=0.1):
    p0 = np.concatenate([rng.uniform(-50, 50, (B, 2)), rng.uniform(5, 30, (B, 1))], 1)
    v0 = rng.uniform(-3, 3, (B, 3))
    pt = p0 + rng.uniform(-8, 8, (B, 3)); pt[:, 2] = np.maximum(pt[:, 2], 2.0)
    vt = rng.uniform(-3, 3, (B, 3))
    x0 = np.concatenate([p0, v0], 1)
    ref = np.zeros((B, N+1, 6))
    for i in range(N+1):
        ref[:, i, 0:3] = pt + vt*dt*i
        ref[:, i, 3:6] = vt
    return x0, ref.reshape(B, -1)

def inst_vectors(qp, x0, ref):
    B = x0.shape[0]; N = qp['N']; nx = qp['nx']
    q = np.zeros((B, qp['n']))
    for i in range(N+1):
        q[:, nx*i:nx*i+nx] = -(ref[:, nx*i:nx*i+nx] @ qp['Q'].T)
    l = np.tile(qp['l'], (B, 1)); u = np.tile(qp['u'], (B, 1))
    l[:, :nx] = -x0; u[:, :nx] = -x0
    return q, l, u

def ruiz(P, A, iters=10):
    n = P.shape[0]; m = A.shape[0]
    D = np.ones(n); E = np.ones(m); c = 1.0
    Ps = P.copy(); As = A.copy()
    for _ in range(iters):
        # column inf-norms of KKT = [P A'; A 0]
        dn = np.maximum(np.abs(Ps).max(0), np.abs(As).max(0))
        en = np.abs(As).max(1)
        dn[dn < 1e-4] = 1.0; en[en < 1e-4] = 1.0
        dn = np.clip(dn, 1e-4, 1e4); en = np.clip(en, 1e-4, 1e4)
        dl = 1/np.sqrt(dn); el = 1/np.sqrt(en)
        Ps = dl[:, None]*Ps*dl[None, :]; As = el[:, None]*As*dl[None, :]
        D *= dl; E *= el
        # cost scaling (q=0 at setup)
        cn = np.abs(Ps).max(0).mean()
        cn = 1.0 if cn < 1e-4 else np.clip(cn, 1e-4, 1e4)
        ct = 1/cn
        Ps *= ct; c *= ct
    return D, E, c, Ps, As

def admm(qp, q, l, u, rho=0.1, sigma=1e-6, alpha=1.6, eps=1e-3, max_iter=4000, check=25,
         scaled=False, rho_eq_mult=1e3, rho_rows=None, verbose=False):
    P, A = qp['P'], qp['A']; n, m = qp['n'], qp['m']
    B = q.shape[0]
    if scaled:
        D, E, c, Ps, As = ruiz(P, A)
    else:
        D = np.ones(n); E = np.ones(m); c = 1.0; Ps, As = P, A
    qs = c*q*D; ls = l*E; us = u*E
    ls = np.maximum(ls, -INF*E); us = np.minimum(us, INF*E)
    # rho vec from template bounds
    lt, ut = qp['l'], qp['u']
    rv = np.full(m, float(rho))
    loose = (lt < -INF*1e-6*1e6+1) & (ut > INF-1)
    loose = (lt <= -1e29) & (ut >= 1e29)
    eqr = (ut - lt) < 1e-4
    eqr[:qp['nx']] = True
    rv[eqr] = rho*rho_eq_mult
    rv[loose] = 1e-6
    if rho_rows is not None:
        rv = rho_rows(rv, eqr, loose)
    K = Ps + sigma*np.eye(n) + As.T@(rv[:, None]*As)
    M = np.linalg.inv(K)
    x = np.zeros((B, n)); z = np.zeros((B, m)); y = np.zeros((B, m))
    done = np.zeros(B, bool); iters = np.full(B, max_iter)
    xo = np.zeros((B, n))
    for k in range(1, max_iter+1):
        rhs = sigma*x - qs + (rv*z - y)@As
        xt = rhs@M.T
        zt = xt@As.T
        x = alpha*xt + (1-alpha)*x
        s = alpha*zt + (1-alpha)*z + y/rv
        zn = np.clip(s, ls, us)
        y = y + rv*(alpha*zt + (1-alpha)*z - zn)
        z = zn
        if k % check == 0:
            # unscaled residuals
            xu = x*D; Ax = (x@As.T)/E; zu = z/E; yu = y*E/c
            rp = np.abs(Ax - zu).max(1)
            Px = (x@Ps.T)/D/c; Aty = (y@As)/D/c
            rd = np.abs(Px + q + Aty).max(1)
            ep = eps + eps*np.maximum(np.abs(Ax).max(1), np.abs(zu).max(1))
            ed = eps + eps*np.maximum.reduce([np.abs(Px).max(1), np.abs(Aty).max(1), np.abs(q).max(1)])
            ok = (rp <= ep) & (rd <= ed) & ~done
            iters[ok] = k; xo[ok] = xu[ok]; done |= ok
            if verbose: print(k, done.sum(), rp.max(), rd.max())
            if done.all(): break
    xo[~done] = (x*D)[~done]
    return xo, iters, done

if __name__ == '__main__':
    N = int(sys.argv[1]); B = int(sys.argv[2])
    rng = np.random.default_rng(0)
    qp = build_6dof(N)
    x0, ref = gen_6dof(N, B, rng)
    q, l, u = inst_vectors(qp, x0, ref)
    xs, it, dn = admm(qp, q, l, u, rho=float(sys.argv[3]) if len(sys.argv) > 3 else 100., eps=1e-10, max_iter=20000, check=100)
    print('x* iters', it.max(), dn.all())
    for scaled in (False, True):
        for rho in ([10, 100, 1000] if not scaled else [0.01, 0.1, 1, 10]):
            for eps in (1e-3, 1e-5):
                t = time.time()
                xo, it, dn = admm(qp, q, l, u, rho=rho, eps=eps, scaled=scaled)
                err = np.abs(xo-xs).max(1)
                print(f'scaled={scaled} rho={rho} eps={eps}: iters mean {it.mean():.0f} p50 {np.median(it):.0f} p99 {np.percentile(it,99):.0f} max {it.max()} solved {dn.mean():.3f} err max {err.max():.2e} med {np.median(err):.2e}  ({time.time()-t:.1f}s)')
