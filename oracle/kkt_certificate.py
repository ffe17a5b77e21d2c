"""Solver-independent optimum of the reference's QPs (TEST INFRASTRUCTURE ONLY).

The QPs built by the reference (src/mpc_6dof.cpp:329-519, src/mpc_12state.cpp:314-1014) are
convex with a unique minimiser (R > 0 on the inputs and the dynamics rows determine the
states).  Any solver run to convergence -- the reference's OSQP included -- ends at that point,
so a point whose KKT residuals vanish to rounding is the limit of the reference's solutions as
eps -> 0.  `refine` finds it with a primal-dual active-set iteration started from an
approximate solution; `kkt_residuals` is the certificate.  Neither uses ADMM.
"""
import numpy as np

INF = 1e29


def kkt_residuals(P, A, q, l, u, x, y):
    """(primal infeasibility, stationarity, complementarity), inf-norms, for
    min 1/2 x'Px + q'x  s.t. l <= Ax <= u  with multipliers y (y>0: upper active)."""
    Ax = A @ x
    rp = max(np.maximum(l - Ax, 0).max(initial=0.0), np.maximum(Ax - u, 0).max(initial=0.0))
    rd = np.abs(P @ x + q + A.T @ y).max(initial=0.0)
    yp, ym = np.maximum(y, 0), np.minimum(y, 0)
    fu, fl = u < INF, l > -INF
    cs = max(np.abs(yp[fu] * (Ax - u)[fu]).max(initial=0.0), np.abs(ym[fl] * (Ax - l)[fl]).max(initial=0.0),
             np.abs(yp[~fu]).max(initial=0.0), np.abs(ym[~fl]).max(initial=0.0))
    return rp, rd, cs


def refine(P, A, q, l, u, x_approx, y_approx=None, max_iter=200, tol=1e-9):
    """Primal-dual active-set refinement.  Returns (x, y, iterations, converged).
    The initial active set comes from the approximate multipliers when given (rows whose
    multiplier is clearly non-zero), otherwise from the rows sitting on a bound."""
    n, m = P.shape[0], A.shape[0]
    Ax = A @ x_approx
    eq = (u - l) < 1e-9
    if y_approx is not None:
        ys = 1e-7 * (1.0 + np.abs(y_approx).max())
        lo = (~eq) & (l > -INF) & (y_approx < -ys)
        up = (~eq) & (u < INF) & (y_approx > ys)
    else:
        lo = (~eq) & (l > -INF) & (Ax < l + 1e-7)
        up = (~eq) & (u < INF) & (Ax > u - 1e-7)
    x = x_approx.copy()
    y = np.zeros(m) if y_approx is None else y_approx.copy()
    for it in range(max_iter):
        act = eq | lo | up
        b = np.where(up, u, l)[act]
        Aa = A[act]
        na = Aa.shape[0]
        K = np.block([[P, Aa.T], [Aa, np.zeros((na, na))]])
        rhs = np.concatenate([-q, b])
        # least-squares correction of the current point: at a degenerate vertex the active
        # rows are linearly dependent and the multipliers are not unique (x still is); the
        # minimum-norm correction keeps them next to the approximate multipliers
        sol = np.concatenate([x, y[act]])
        for _ in range(2):
            sol = sol + np.linalg.lstsq(K, rhs - K @ sol, rcond=1e-13)[0]
        x = sol[:n]; y = np.zeros(m); y[act] = sol[n:]
        Ax = A @ x
        bad_lo = lo & (y > tol)
        bad_up = up & (y < -tol)
        v_lo = (~act) & (Ax < l - tol)
        v_up = (~act) & (Ax > u + tol)
        if not (bad_lo.any() or bad_up.any() or v_lo.any() or v_up.any()):
            return x, y, it + 1, True
        lo = (lo & ~bad_lo) | v_lo
        up = (up & ~bad_up) | v_up
    return x, y, max_iter, False
