"""Generates tests/golden/*.npz (run from the repo root: python tests/golden/make_golden.py).

The reference ships no golden vectors and cannot be built offline (Eigen, OSQP 25b6b39 and
osqp-eigen v0.8.0 are absent), so every fixture here is produced by the CPU oracle and then
replaced by the exact optimum of the same QP: ADMM to a tight tolerance, a primal-dual
active-set refinement, and a KKT-residual certificate (oracle/kkt_certificate.py).  PARITY
UNPINNED with respect to the real OsqpEigen binary; pinned with respect to the optimum both
converge to.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # noqa: E402
from oracle import kkt_certificate as K   # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 0x4D5043


def exact(qp, q, l, u, rho):
    s = O.fixed_rho_settings(rho, eps_abs=1e-8, eps_rel=1e-8, max_iter=40000, warm_start=0, linsys=1)
    w = O.Osqp(qp["P"], qp["A"], qp["q"], qp["l"], qp["u"], s)
    w.update(q=q, l=l, u=u)
    r = w.solve()
    x, y, it, ok = K.refine(qp["P"], qp["A"], q, l, u, r["x"], r["y"])
    rp, rd, cs = K.kkt_residuals(qp["P"], qp["A"], q, l, u, x, y)
    scale = 1.0 + np.abs(q).max()
    assert ok and rp < 1e-9 and rd < 1e-9 * scale and cs < 1e-7 * scale, (ok, rp, rd, cs)
    return x, y, (rp, rd, cs)


def golden6(name, p, x0, ref, rho=1000.0):
    qp = O.build_qp(p)
    N = p.N
    X, U, cert = [], [], []
    for b in range(x0.shape[0]):
        q, l, u = O.qp6_vectors(p, x0[b], ref[b])
        x, y, c = exact(qp, q, l, u, rho)
        X.append(x[:6 * (N + 1)]); U.append(x[6 * (N + 1):]); cert.append(c)
    np.savez_compressed(os.path.join(OUT, name), N=N, dt=p.dt, state_weight=p.state_weight,
                        input_weight=p.input_weight, smooth_input_weight=p.smooth_input_weight,
                        enable_control_smoothing=p.enable_control_smoothing, min_altitude=p.min_altitude,
                        max_vel=np.array(p.max_vel[:]), max_accel=np.array(p.max_accel[:]),
                        x0=x0, ref=ref, X=np.array(X), U=np.array(U), kkt=np.array(cert))
    print(name, "ok", np.array(cert).max(0))


def golden12(name, p, x12, ref12, rho=(1000.0, 1000.0, 200.0)):
    N = p.N
    qps = {f: O.build_qp(p, f) for f in (O.FAM_XY, O.FAM_Z, O.FAM_YAW)}
    out = dict(X=[], U=[], z_sol=[], xy_max=[], mixed_max=[], xy_sol=[], yaw_ref=[], yaw_sol=[])
    for b in range(x12.shape[0]):
        xyref, zref, yawref = O.split_ref12(N, ref12[b])
        q, l, u = O.qp12_vectors(p, O.FAM_Z, x12[b, 6:9], zref)
        zs, _, _ = exact(qps[O.FAM_Z], q, l, u, rho[1])
        xm, mm = O.xy_bounds_from_z(p, zs[:3 * (N + 1)])
        q, l, u = O.qp12_vectors(p, O.FAM_XY, x12[b, 0:6], xyref, xm, mm)
        xys, _, _ = exact(qps[O.FAM_XY], q, l, u, rho[0])
        yr = O.yaw_ref_from_xy(N, xyref, xys[:6 * (N + 1)])
        q, l, u = O.qp12_vectors(p, O.FAM_YAW, x12[b, 9:12], yr)
        ys, _, _ = exact(qps[O.FAM_YAW], q, l, u, rho[2])
        X = np.zeros(12 * (N + 1)); U = np.zeros(4 * N)
        for i in range(N + 1):
            X[12 * i:12 * i + 6] = xys[6 * i:6 * i + 6]
            X[12 * i + 6:12 * i + 9] = zs[3 * i:3 * i + 3]
            X[12 * i + 9:12 * i + 12] = ys[3 * i:3 * i + 3]
        for i in range(N):
            U[4 * i:4 * i + 2] = xys[6 * (N + 1) + 2 * i:6 * (N + 1) + 2 * i + 2]
            U[4 * i + 2] = zs[3 * (N + 1) + i]; U[4 * i + 3] = ys[3 * (N + 1) + i]
        for k, v in (("X", X), ("U", U), ("z_sol", zs), ("xy_max", xm), ("mixed_max", mm), ("xy_sol", xys),
                     ("yaw_ref", yr), ("yaw_sol", ys)):
            out[k].append(v)
    np.savez_compressed(os.path.join(OUT, name), N=N, dt=p.dt,
                        params=np.array([getattr(p, f[0]) for f in p._fields_[3:]]),
                        param_names=np.array([f[0] for f in p._fields_[3:]]),
                        enable_control_smoothing=p.enable_control_smoothing,
                        x0=x12, ref=ref12, **{k: np.array(v) for k, v in out.items()})
    print(name, "ok")


def main(only):
    want = lambda name: not only or name in only
    if want("harness"):
        harness()
    # seeded random instances (config/mpc.yaml weights, smoothing off): 32 per family and horizon
    for N in (5, 10, 20, 50):
        if want("6dof_N%d" % N):
            p = O.mpc6_params(N=N)
            x0, ref = O.synth_mpc6(SEED + 2, N, 0.1, 32)
            golden6("rand_6dof_N%d.npz" % N, p, x0, ref)
    for N in (5, 10, 20, 50):
        if want("12state_N%d" % N):
            p = O.mpc12_params(N=N)
            x12, ref12 = O.synth_mpc12(SEED + 3, N, 0.1, 32)
            golden12("rand_12state_N%d.npz" % N, p, x12, ref12)


def harness():
    # harness case of test/test_mpc_6dof.cpp:16-83 (smoothing on with weight 10: ctor defaults)
    p = O.mpc6_params(N=20, dt=0.1, state_weight=7000.0, input_weight=500.0, smooth_input_weight=10.0,
                      enable_control_smoothing=True, min_altitude=-1.0)
    x0 = np.array([[0.1, 0, 0, 0, 0, 0.0]]); ref = np.zeros((1, 21, 6)); ref[0, :, 0] = 0.1; ref[0, :, 2] = 1.0
    golden6("harness_6dof.npz", p, x0, ref.reshape(1, -1))
    # harness case of test/test_mpc_12state.cpp:14-168
    p = O.mpc12_params(N=10, dt=0.1, state_weight=(1000.0, 1000.0, 1000.0), input_weight=(10.0, 10.0, 1.0),
                       max_vel=(12.0, 5.0, 5.0), max_accel=(5.0, 5.0, 10.0), max_jerk=(10.0, 10.0, 10.0))
    x12 = np.array([[0.03, 0, 0, -0.04, 0, 0, 0.1, 0, 0, 0, 0, 0.0]])
    ref12 = np.zeros((1, 11, 12)); ref12[0, :, 6] = 40.0; ref12[0, :, 0] = 4.0; ref12[0, :, 3] = 0.5
    golden12("harness_12state.npz", p, x12, ref12.reshape(1, -1), rho=(100.0, 100.0, 50.0))


if __name__ == "__main__":
    # python tests/golden/make_golden.py [harness] [6dof_N50] [12state_N20] ...  (no argument: everything)
    main(set(sys.argv[1:]))


# ---------------------------------------------------------------------------------------------
# fuzz205_wild_trial{33,41,56}.npz are not made by this script: they are three draws of the
# differential fuzzer (FUZZ_WILD=1 FUZZ_NO_SCHEDULE=1 FUZZ_DUMP=dir FUZZ_DUMP_TRIALS=33,41,56
# python tools/fuzz_vs_oracle.py 205 60, on a GPU box) in which the yaw columns of engine and oracle
# parted by 2e-4 ... 0.4 rad, cut down to the instances that differed most, a few that agreed and
# a few that failed: inputs (x, ref), model parameters and solver options as JSON text -- no outputs.
