"""Second, independent (numpy) restatement of the reference's QP construction, used only to
cross-check the C oracle in the CPU tests.  Written from src/mpc_6dof.cpp:243-519 and
src/mpc_12state.cpp:74-1014 with Kronecker products instead of the reference's block loops."""
import numpy as np

INF = 1e30


def _chain(Ad, Bd, Q, R, N, smooth):
    nx, nu = Bd.shape
    Nx, Nu = nx * (N + 1), nu * N
    P = np.zeros((Nx + Nu, Nx + Nu))
    P[:Nx, :Nx] = np.kron(np.eye(N + 1), Q)
    P[Nx:, Nx:] = np.kron(np.eye(N), R)
    if smooth is not None:
        Dm = np.zeros((N - 1, N))
        Dm[np.arange(N - 1), np.arange(N - 1)] = -1.0
        Dm[np.arange(N - 1), np.arange(1, N)] = 1.0
        Ud = np.kron(Dm, np.eye(nu))
        P[Nx:, Nx:] += Ud.T @ (smooth * np.eye(nu * (N - 1))) @ Ud
    Aeq = np.hstack([-np.eye(Nx) + np.kron(np.eye(N + 1, k=-1), Ad),
                     np.vstack([np.zeros((nx, Nu)), np.kron(np.eye(N), Bd)])])
    A = np.vstack([Aeq, np.eye(Nx + Nu)])
    return P, A


def qp6(N, dt, w, r, smooth, min_alt, vmax, amax):
    Ad = np.eye(6); Ad[:3, 3:] = dt * np.eye(3)
    Bd = np.vstack([np.zeros((3, 3)), dt * np.eye(3)])
    Q = np.diag([w, w, w, 0, 0, 0.0]); R = r * np.eye(3)
    P, A = _chain(Ad, Bd, Q, R, N, smooth)
    xmin = np.array([-INF, -INF, min_alt, -vmax[0], -vmax[1], -vmax[2]])
    xmax = np.array([INF, INF, INF, vmax[0], vmax[1], vmax[2]])
    l = np.concatenate([np.zeros(6 * (N + 1)), np.tile(xmin, N + 1), np.tile(-np.asarray(amax), N)])
    u = np.concatenate([np.zeros(6 * (N + 1)), np.tile(xmax, N + 1), np.tile(np.asarray(amax), N)])
    return P, A, l, u, Q


def _a3(dt):
    return np.array([[1, dt, dt * dt / 2], [0, 1, dt], [0, 0, 1.0]])


def qp_axis(N, dt, w, r, smooth, pmax, vmax, amax, jmax):
    Q = np.diag([w, 0, 0.0]); R = np.array([[r]])
    P, A = _chain(_a3(dt), np.array([[0], [0], [dt]]), Q, R, N, smooth)
    b = np.array([pmax, vmax, amax])
    l = np.concatenate([np.zeros(3 * (N + 1)), np.tile(-b, N + 1), np.full(N, -jmax)])
    u = np.concatenate([np.zeros(3 * (N + 1)), np.tile(b, N + 1), np.full(N, jmax)])
    return P, A, l, u, Q


def qp_xy(N, dt, w, r, smooth, vmax, amax, jmax):
    Ad = np.kron(np.eye(2), _a3(dt))
    Bd = np.zeros((6, 2)); Bd[2, 0] = dt; Bd[5, 1] = dt
    Q = np.diag([w, 0, 0, w, 0, 0.0]); R = r * np.eye(2)
    P, A = _chain(Ad, Bd, Q, R, N, smooth)
    n = A.shape[1]
    s3, s2 = np.sqrt(3) / 2, np.sqrt(2) / 2
    Mv = np.array([[s3, 0.5], [s3, -0.5], [0.5, s3], [-0.5, s3]])
    Ma = np.array([[s2, s2], [s2, -s2]])
    mixed = np.zeros((6 * N, n))
    for i in range(N):
        c = 6 * (i + 1)
        mixed[6 * i:6 * i + 4, [c + 1, c + 4]] = Mv
        mixed[6 * i + 4:6 * i + 6, [c + 2, c + 5]] = Ma
    A = np.vstack([A, mixed])
    b = np.array([INF, vmax, amax, INF, vmax, amax])
    mb = np.array([vmax] * 4 + [amax] * 2)
    l = np.concatenate([np.zeros(6 * (N + 1)), np.tile(-b, N + 1), np.full(2 * N, -jmax), np.tile(-mb, N)])
    u = np.concatenate([np.zeros(6 * (N + 1)), np.tile(b, N + 1), np.full(2 * N, jmax), np.tile(mb, N)])
    return P, A, l, u, Q


def xy_bounds(z_sol, N, xy_vmax, xy_amax, z_amax):
    xm = np.zeros(6 * N); mm = np.zeros(6 * N)
    for i in range(1, N + 1):
        vz, az = z_sol[3 * i + 1], z_sol[3 * i + 2]
        if vz < 0:
            vb = xy_vmax
        else:
            d = az / z_amax
            vb = xy_vmax * np.sqrt(1 - d * d) if 1 - d * d >= 0 else 0.9 * xy_vmax
        xm[6 * (i - 1):6 * i] = [INF, vb, xy_amax, INF, vb, xy_amax]
        mm[6 * (i - 1):6 * i] = [vb] * 4 + [xy_amax] * 2
    return xm, mm
