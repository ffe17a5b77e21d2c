"""Step-size policies for the batched fixed-factorisation ADMM, tried out in torch before they are
built into the kernel: a menu of pre-factored operators (base rho x per-class multipliers) and a
rule that moves an instance from one to another at a termination check.  Prints mean / p99 / max
iteration counts per policy on seeded instances of one workload.  (Development tool: plain torch
fp64 matmuls, the QP data from the CPU oracle; not part of the product.)

  python tools/rho_policy_probe.py --model 6dof --N 50 --B 16384 --eps 3e-9
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 0x4D5043
INF = 1e26


def problem(model, N, B, seed_off, first=0, stage=None):
    """Dense P, A and the per-instance q, l, u of B seeded instances (6-DOF, or one stage of the
    12-state chain with the upstream solutions taken from the oracle run tight)."""
    from oracle import oracle as O
    O.build()
    if model == "6dof":
        p = O.mpc6_params(N=N)
        d = O.build_qp(p)
        P, A, q0, l0, u0 = d['P'], d['A'], d['q'], d['l'], d['u']
        x0, ref = O.synth_mpc6(SEED + seed_off, N, 0.1, B, first)
        n, m = P.shape[0], A.shape[0]
        q = np.zeros((B, n)); l = np.tile(l0, (B, 1)); u = np.tile(u0, (B, 1))
        Q = np.diag(P)[:6]
        R = ref.reshape(B, N + 1, 6)
        q[:, :6 * (N + 1)] = -(R * Q[None, None, :]).reshape(B, -1)
        l[:, :6] = -x0; u[:, :6] = -x0
        cls = np.full(m, -1)
        Nx = 6 * (N + 1)
        for i in range(N + 1):
            for a in range(6):
                cls[Nx + 6 * i + a] = 0 if a < 3 else 1
        cls[2 * Nx:] = 3
        return P, A, q, l, u, cls
    # one stage of the 12-state chain; the upstream solutions come from the oracle's own chain run tight
    fam = {"xy": O.FAM_XY, "z": O.FAM_Z, "yaw": O.FAM_YAW}[stage]
    p = O.mpc12_params(N=N)
    d = O.build_qp(p, fam)
    P, A = d["P"], d["A"]
    n, m, nx, nu = d["n"], d["m"], d["nx"], d["nu"]
    x12, ref12 = O.synth_mpc12(SEED + seed_off, N, 0.1, B, first)
    O.set_rho_class_scale(1.0, 10.0, 3.0, 3.0)
    tight = {k: O.fixed_rho_settings(r, warm_start=0, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000)
             for k, r in (("xy", 500.0), ("z", 500.0), ("yaw", 200.0))}
    chain = O.Mpc12(p, tight)
    q = np.zeros((B, n)); l = np.zeros((B, m)); u = np.zeros((B, m))
    keep = []
    for b in range(B):
        r = chain.loop(x12[b], ref12[b])
        xyref, zref, yawref = O.split_ref12(N, ref12[b])
        x = x12[b]
        if stage == "z":
            q[b], l[b], u[b] = O.qp12_vectors(p, fam, x[6:9], zref)
        elif stage == "xy":
            if r["status"][0] != 1:
                continue
            q[b], l[b], u[b] = O.qp12_vectors(p, fam, x[0:6], xyref, r["xy_max"], r["mixed_max"])
        else:
            if not r["ok"]:
                continue
            q[b], l[b], u[b] = O.qp12_vectors(p, fam, x[9:12], r["yaw_ref"])
        keep.append(b)
    O.set_rho_class_scale(1.0, 1.0, 1.0, 1.0)
    keep = np.array(keep)
    q, l, u = q[keep], l[keep], u[keep]
    cls = np.full(m, -1)
    Nx, Nu = nx * (N + 1), nu * N
    for i in range(N + 1):
        for a in range(nx):
            cls[Nx + nx * i + a] = a % 3
    cls[2 * Nx:2 * Nx + Nu] = 3
    for r_ in range(2 * Nx + Nu, m):
        cls[r_] = 1 if (r_ - 2 * Nx - Nu) % 6 < 4 else 2
    return P, A, q, l, u, cls


def rho_vector(l0, u0, cls, rho, scale):
    """(the engine types the rows from the init-time bounds: x0 = 0, widest boxes)"""
    m = len(l0)
    r = np.full(m, rho)
    eq = (u0 - l0) < 1e-4
    loose = (l0 < -INF) & (u0 > INF)
    for k in range(4):
        r[(cls == k)] = rho * scale[k]
    r[eq] = 1e3 * rho
    r[loose] = 1e-6
    return r


def run(dev, P, A, q, l, u, menu, rule, eps, max_iter, check=25, sigma=1e-6, alpha=1.6, start_op=0, verbose=False):
    """menu: list of rho vectors (m,).  rule(it, op, pri_n, dua_n, running) -> new op index tensor."""
    import torch
    dt = torch.float64
    B, n = q.shape
    m = A.shape[0]
    At = torch.tensor(A.T.copy(), dtype=dt, device=dev)      # n x m : z = x @ At
    Am = torch.tensor(A, dtype=dt, device=dev)               # m x n : A'w = w @ Am
    Pm = torch.tensor(P, dtype=dt, device=dev)
    K = []
    for r in menu:
        Kd = P + sigma * np.eye(n) + A.T @ (r[:, None] * A)
        K.append(torch.tensor(np.linalg.inv(Kd), dtype=dt, device=dev))
    rhos = torch.tensor(np.stack(menu), dtype=dt, device=dev)   # ops x m
    q = torch.tensor(q, dtype=dt, device=dev); l = torch.tensor(l, dtype=dt, device=dev); u = torch.tensor(u, dtype=dt, device=dev)
    x = torch.zeros((B, n), dtype=dt, device=dev); z = torch.zeros((B, m), dtype=dt, device=dev); y = torch.zeros_like(z)
    op = torch.full((B,), start_op, dtype=torch.long, device=dev)
    iters = torch.zeros(B, dtype=torch.long, device=dev)
    running = torch.ones(B, dtype=torch.bool, device=dev)
    qn = q.abs().max(1).values
    switches = torch.zeros(B, dtype=torch.long, device=dev)
    for it in range(1, max_iter + 1):
        idx = running.nonzero().squeeze(1)
        if idx.numel() == 0:
            break
        xr, zr_, yr, opr = x[idx], z[idx], y[idx], op[idx]
        rho = rhos[opr]
        rhs = sigma * xr - q[idx] + (rho * zr_ - yr) @ Am
        xt = torch.empty_like(rhs)
        for k in range(len(menu)):
            mk = opr == k
            if mk.any():
                xt[mk] = rhs[mk] @ K[k]
        zt = xt @ At
        xn = alpha * xt + (1 - alpha) * xr
        zrel = alpha * zt + (1 - alpha) * zr_
        zn = torch.minimum(torch.maximum(zrel + yr / rho, l[idx]), u[idx])
        yn = yr + rho * (zrel - zn)
        x[idx], z[idx], y[idx] = xn, zn, yn
        if it % check == 0:
            Ax = xn @ At
            pri = (Ax - zn).abs().max(1).values
            Px = xn @ Pm
            Aty = yn @ Am
            dua = (Px + q[idx] + Aty).abs().max(1).values
            pn = torch.maximum(Ax.abs().max(1).values, zn.abs().max(1).values)
            dn = torch.maximum(torch.maximum(Px.abs().max(1).values, Aty.abs().max(1).values), qn[idx])
            conv = (pri < eps + eps * pn) & (dua < eps + eps * dn)
            iters[idx[conv]] = it
            running[idx[conv]] = False
            keep = ~conv
            if keep.any():
                new = rule(it, opr[keep], pri[keep] / (pn[keep] + 1e-10), dua[keep] / (dn[keep] + 1e-10))
                switches[idx[keep]] += (new != opr[keep]).long()
                op[idx[keep]] = new
            if verbose and it % (check * 20) == 0:
                print("  it %d running %d" % (it, int(keep.sum())), flush=True)
    iters[running] = max_iter
    return iters.cpu().numpy(), switches.cpu().numpy(), x.cpu().numpy()


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="6dof")
    ap.add_argument("--N", type=int, default=50)
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--seed-off", type=int, default=4)
    ap.add_argument("--eps", default="3e-9", help="tolerance, or a comma-separated list of them")
    ap.add_argument("--max-iter", type=int, default=8000)
    ap.add_argument("--rho", type=float, default=1000.0)
    ap.add_argument("--set", default="a")
    ap.add_argument("--stage", default=None, help="12state: z, xy or yaw")
    ap.add_argument("--base", default=None, help="class multipliers of the first operator, e.g. 1,10,3,3")
    ap.add_argument("--grid", default=None,
                    help="set 'g': semicolon-separated schedules, each 'k1:p,v,a,i[/k2:p,v,a,i]' with multipliers relative to --base")
    args = ap.parse_args()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    P, A, q, l, u, cls = problem(args.model, args.N, args.B, args.seed_off, args.first, args.stage)
    from oracle import oracle as O
    if args.model == "6dof":
        d0 = O.build_qp(O.mpc6_params(N=args.N))
    else:
        d0 = O.build_qp(O.mpc12_params(N=args.N), {"xy": O.FAM_XY, "z": O.FAM_Z, "yaw": O.FAM_YAW}[args.stage])
    l0, u0 = d0["l"], d0["u"]
    print("instances %d, n %d, m %d" % (q.shape[0], q.shape[1], l.shape[1]), flush=True)
    rv = lambda mult, scale: rho_vector(l0, u0, cls, args.rho * mult, scale)
    stay = lambda it, op, rp, rd: op
    policies = {}
    tight, loose = (96.0, 3.0, 1.0, 2.0), (12.0, 3.0, 1.0, 4.0)
    if args.set == "a":
        policies["fixed tight (96,3,1,2)"] = ([rv(1, tight)], stay, 0)
        policies["fixed loose (12,3,1,4)"] = ([rv(1, loose)], stay, 0)
        for k1, s1 in ((300, (400.0, 3.0, 1.0, 2.0)), (500, (400.0, 3.0, 1.0, 2.0)), (300, (400.0, 10.0, 1.0, 2.0)), (200, (200.0, 6.0, 1.0, 2.0))):
            def sched(it, op, rp, rd, k1=k1):
                return torch.where(torch.tensor(it >= k1, device=op.device), torch.ones_like(op), op)
            policies["tight -> %s after %d" % (s1, k1)] = ([rv(1, tight), rv(1, s1)], sched, 0)
        # OSQP's estimate on a geometric menu of base rho (class multipliers kept), moved one step per decision
        for f, tol, every in ((4.0, 4.0, 100), (4.0, 2.0, 100), (3.0, 3.0, 50), (5.0, 5.0, 200)):
            mults = [f ** k for k in range(-3, 4)]
            menu = [rv(mu, tight) for mu in mults]
            def osqp_rule(it, op, rp, rd, tol=tol, every=every, top=len(mults) - 1):
                if it % every:
                    return op
                est = torch.sqrt(rp / (rd + 1e-10))
                up = (est > tol) & (op < top)
                dn = (est < 1.0 / tol) & (op > 0)
                return op + up.long() - dn.long()
            policies["osqp-estimate menu x%g tol %g every %d (tight classes)" % (f, tol, every)] = (menu, osqp_rule, 3)
    if args.set == "b":
        def sched_rule(bounds):
            def r(it, op, rp, rd):
                tgt = sum(1 for b in bounds if it >= b)
                return torch.maximum(op, torch.full_like(op, tgt))
            return r
        for pos in (200.0, 400.0, 800.0, 1600.0):
            for k1 in (100, 200, 300):
                policies["tight -> pos x%g after %d" % (pos, k1)] = ([rv(1, tight), rv(1, (pos, 3.0, 1.0, 2.0))], sched_rule([k1]), 0)
        for (k1, p1, k2, p2) in ((150, 400.0, 500, 1600.0), (200, 400.0, 600, 3200.0), (100, 200.0, 300, 800.0), (200, 800.0, 600, 6400.0),
                                 (100, 300.0, 300, 1000.0), (200, 400.0, 500, 1600.0)):
            policies["tight -> pos x%g after %d -> x%g after %d" % (p1, k1, p2, k2)] = (
                [rv(1, tight), rv(1, (p1, 3.0, 1.0, 2.0)), rv(1, (p2, 3.0, 1.0, 2.0))], sched_rule([k1, k2]), 0)
        for k1 in (200, 300):
            policies["tight -> all x4 after %d" % k1] = ([rv(1, tight), rv(1, (384.0, 12.0, 4.0, 8.0))], sched_rule([k1]), 0)
            policies["tight -> pos x400, vel x12 after %d" % k1] = ([rv(1, tight), rv(1, (400.0, 12.0, 1.0, 2.0))], sched_rule([k1]), 0)
            policies["tight -> pos x400, inp x8 after %d" % k1] = ([rv(1, tight), rv(1, (400.0, 3.0, 1.0, 8.0))], sched_rule([k1]), 0)
            policies["tight -> pos x400, inp x0.5 after %d" % k1] = ([rv(1, tight), rv(1, (400.0, 3.0, 1.0, 0.5))], sched_rule([k1]), 0)
        # estimate-gated: at k1 only instances whose primal residual dominates move up
        for k1, thr in ((200, 1.0), (200, 2.0), (100, 1.0)):
            def gated(it, op, rp, rd, k1=k1, thr=thr):
                if it < k1 or it % 100:
                    return op
                est = torch.sqrt(rp / (rd + 1e-10))
                return torch.where((est > thr) & (op < 2), op + 1, op)
            policies["estimate-gated (> %g) every 100 from %d: pos 96 -> 400 -> 1600" % (thr, k1)] = (
                [rv(1, tight), rv(1, (400.0, 3.0, 1.0, 2.0)), rv(1, (1600.0, 3.0, 1.0, 2.0))], gated, 0)
    if args.set == "c":
        def sched_rule(bounds):
            def r(it, op, rp, rd):
                tgt = sum(1 for b in bounds if it >= b)
                return torch.maximum(op, torch.full_like(op, tgt))
            return r
        for mult in (0.25, 0.5, 1.0, 2.0):
            for pos in (24.0, 96.0, 384.0):
                for vel, inp in ((3.0, 2.0), (1.0, 1.0), (3.0, 4.0)):
                    policies["base x%g (pos %g, vel %g, inp %g) -> pos abs 800 after 300" % (mult, pos, vel, inp)] = (
                        [rv(mult, (pos, vel, 1.0, inp)), rv(mult, (800.0 / mult, vel, 1.0, inp))], sched_rule([300]), 0)
    if args.set == "d":
        def sched_rule(bounds):
            def r(it, op, rp, rd):
                tgt = sum(1 for b in bounds if it >= b)
                return torch.maximum(op, torch.full_like(op, tgt))
            return r
        for pos0, inp in ((6.0, 4.0), (12.0, 4.0), (12.0, 2.0), (24.0, 4.0), (24.0, 2.0), (48.0, 2.0)):
            for k1 in (100, 200, 300):
                for pos1 in (400.0, 800.0, 1600.0):
                    policies["(pos %g, vel 3, inp %g) -> pos %g after %d" % (pos0, inp, pos1, k1)] = (
                        [rv(1, (pos0, 3.0, 1.0, inp)), rv(1, (pos1, 3.0, 1.0, inp))], sched_rule([k1]), 0)
        for (pos0, k1, p1, k2, p2) in ((12.0, 100, 100.0, 300, 800.0), (12.0, 100, 200.0, 400, 1600.0), (24.0, 150, 200.0, 400, 1600.0),
                                       (12.0, 75, 96.0, 250, 800.0), (12.0, 100, 100.0, 200, 800.0)):
            policies["(pos %g, vel 3, inp 4) -> pos %g after %d -> pos %g after %d" % (pos0, p1, k1, p2, k2)] = (
                [rv(1, (pos0, 3.0, 1.0, 4.0)), rv(1, (p1, 3.0, 1.0, 4.0)), rv(1, (p2, 3.0, 1.0, 4.0))], sched_rule([k1, k2]), 0)
    if args.set == "g":
        def sched_rule(bounds):
            def r(it, op, rp, rd):
                tgt = sum(1 for b in bounds if it >= b)
                return torch.maximum(op, torch.full_like(op, tgt))
            return r
        base = tuple(float(v) for v in args.base.split(","))
        policies["fixed %s" % (base,)] = ([rv(1, base)], stay, 0)
        for spec in args.grid.split(";"):
            spec = spec.strip()
            if not spec:
                continue
            bounds, menu = [], [rv(1, base)]
            for ph in spec.split("/"):
                k, mult = ph.split(":")
                mult = [float(v) for v in mult.split(",")]
                bounds.append(int(k))
                menu.append(rv(1, tuple(b * m_ for b, m_ in zip(base, mult))))
            policies["%s -> %s" % (base, spec)] = (menu, sched_rule(bounds), 0)
    for eps in [float(v) for v in str(args.eps).split(",")]:
        print("eps %g" % eps, flush=True)
        for name, (menu, rule, start) in policies.items():
            t0 = time.time()
            it, sw, _ = run(dev, P, A, q, l, u, menu, rule, eps, args.max_iter, start_op=start)
            done = it[it < args.max_iter]
            print("%-62s mean %7.1f  p50 %5d  p99 %6d  max %6d (solved: %6d)  unsolved %d  switches mean %.2f  (%.0f s)"
                  % (name, it.mean(), np.median(it), np.percentile(it, 99), it.max(), done.max() if len(done) else 0,
                     int((it >= args.max_iter).sum()), sw.mean(), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
