"""Seeded synthetic MPC instances (counter-based: instance k, draw c depend only on (seed, k, c)),
so that every rank of a sharded run, the C++ harness and the CPU oracle generate the same batch.

6-DOF:    p_xy ~ U(-50,50), p_z ~ U(5,30), v ~ U(-3,3)^3; target = p + U(-8,8)^3 with z >= 2,
          target velocity ~ U(-3,3)^3; ref_i = [p_t + v_t dt i, v_t].
12-state: x,y ~ U(-50,50), v_xy ~ U(-3,3), a_xy ~ U(-1,1); z ~ U(5,30), v_z ~ U(-2,2),
          a_z ~ U(-1,1); yaw ~ U(-pi,pi), v_yaw ~ U(-.5,.5), a_yaw = 0; target = pos + U(-8,8) in
          x,y and U(-3,3) in z, constant velocity U(-3,3), zero acceleration; yaw reference unused
          (it is overwritten by computeYawRefTrajectory, mpc_12state.cpp:541-549).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def uniform(seed, stream, ctr):
    """U[0,1) for arrays of instance indices `stream` and draw counters `ctr`."""
    with np.errstate(over="ignore"):
        stream = np.asarray(stream, dtype=np.uint64)
        ctr = np.asarray(ctr, dtype=np.uint64)
        h = _splitmix64(np.uint64(seed) ^ _splitmix64(stream * np.uint64(0xD1342543DE82EF95) + ctr))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _u(seed, k, c, lo, hi):
    return lo + (hi - lo) * uniform(seed, k, np.full_like(k, c))


def mpc6(seed, N, dt, B, first=0):
    """x0 [B, 6], ref [B, 6(N+1)] (instance-major)."""
    k = np.arange(first, first + B, dtype=np.uint64)
    p0 = np.stack([_u(seed, k, 0, -50, 50), _u(seed, k, 1, -50, 50), _u(seed, k, 2, 5, 30)], 1)
    v0 = np.stack([_u(seed, k, 3 + a, -3, 3) for a in range(3)], 1)
    pt = p0 + np.stack([_u(seed, k, 6 + a, -8, 8) for a in range(3)], 1)
    pt[:, 2] = np.maximum(pt[:, 2], 2.0)
    vt = np.stack([_u(seed, k, 9 + a, -3, 3) for a in range(3)], 1)
    x0 = np.concatenate([p0, v0], 1)
    ref = np.zeros((B, N + 1, 6))
    for i in range(N + 1):
        ref[:, i, 0:3] = pt + vt * dt * i
        ref[:, i, 3:6] = vt
    return x0, ref.reshape(B, -1)


def mpc12(seed, N, dt, B, first=0):
    """x0 [B, 12], ref [B, 12(N+1)] (instance-major)."""
    k = np.arange(first, first + B, dtype=np.uint64)
    x = np.zeros((B, 12))
    spec = [(-50, 50), (-3, 3), (-1, 1), (-50, 50), (-3, 3), (-1, 1), (5, 30), (-2, 2), (-1, 1),
            (-np.pi, np.pi), (-0.5, 0.5)]
    for c, (lo, hi) in enumerate(spec):
        x[:, c] = _u(seed, k, c, lo, hi)
    pt = np.stack([x[:, 0] + _u(seed, k, 11, -8, 8), x[:, 3] + _u(seed, k, 12, -8, 8),
                   x[:, 6] + _u(seed, k, 13, -3, 3)], 1)
    vt = np.stack([_u(seed, k, 14 + a, -3, 3) for a in range(3)], 1)
    ref = np.zeros((B, N + 1, 12))
    for i in range(N + 1):
        for a in range(3):
            ref[:, i, 3 * a] = pt[:, a] + vt[:, a] * dt * i
            ref[:, i, 3 * a + 1] = vt[:, a]
    return x, ref.reshape(B, -1)


def stress(model, x0, ref, N):
    """The geometry of the reference's 12-state test main on every instance of a batch
    (test/test_mpc_12state.cpp:146-167): the vehicle starts at rest at its drawn position and
    the target hovers 4 m ahead in x, 0.5 m in y and 40 m above -- further than the horizon can
    cover, so velocity, acceleration and input limits are active along most of it."""
    nx = 6 if model == "6dof" else 12
    pos = (0, 1, 2) if model == "6dof" else (0, 3, 6)
    x0 = x0.copy()
    keep = np.zeros(nx, dtype=bool)
    keep[list(pos)] = True
    if model != "6dof":
        keep[9] = True                       # yaw
    x0[:, ~keep] = 0.0
    r = np.zeros((x0.shape[0], N + 1, nx))
    for c, off in zip(pos, (4.0, 0.5, 40.0)):
        r[:, :, c] = (x0[:, c] + off)[:, None]
    return x0, r.reshape(x0.shape[0], -1)


def shard(B, rank, world):
    """Contiguous slab [first, first + count) of a batch of B instances for `rank` of `world`."""
    base, rem = divmod(B, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count
