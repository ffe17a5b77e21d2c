"""Exact iteration needs of the three 12-state stages over a grid of rho and per-class multipliers
(termination check after every iteration; development aid).  usage: sweep12.py N B"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
N, B = int(sys.argv[1]), int(sys.argv[2])
x0, ref = synth.mpc12(0x4D5043 + 3, N, 0.1, B)
scales = [(1, 1, 1, 1), (1, 3, 1, 3), (1, 3, 3, 3), (1, 10, 3, 3), (1, 3, 1, 10), (1, 10, 10, 10), (1, 10, 3, 10), (1, 1, 1, 10)]
rhos = [(r, r, r) for r in (100., 200., 500., 1000., 2000., 5000.)]
if len(sys.argv) > 3 and sys.argv[3] == "fine":
    rhos = [(500., 500., 200.), (500., 1000., 200.), (1000., 500., 200.), (500., 500., 500.)]
    scales = [(1, 10, 3, 3), (1, 10, 3, 5), (1, 15, 3, 3), (1, 10, 5, 3), (1, 7, 3, 3), (1, 10, 2, 3), (1, 10, 3, 2), (1, 20, 5, 3)]
for rho3 in rhos:
    rho = rho3[0]
    for sc in scales:
        opts = tg.make_opts(eps_abs=1e-3, eps_rel=1e-3, check_termination=1, check_fine=0, rho_scale=sc, max_iter=1000)
        eng = tg.BatchedMPC12(tg.mpc12_params(N=N), opts, rho_xyz=rho3)
        buf = eng.alloc(B)
        buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
        buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
        eng.solve(buf); torch.cuda.synchronize()
        it = buf["iters"].cpu().numpy().reshape(-1, buf["ld"])[:, :B]
        st = buf["status"].cpu().numpy().reshape(-1, buf["ld"])[:, :B]
        for k, name in enumerate(("z", "xy", "yaw")):
            ok = st[k] == 1
            v = it[k][ok]
            if len(v) == 0:
                continue
            q = np.percentile(v, [50, 90, 99, 99.9, 100])
            tile = np.where(ok, it[k], 0).reshape(-1, 16).max(1)
            print(f"{name} rho={rho3[k]:g} of {rho3} scale={sc}: ok {ok.mean():.4f} mean {v.mean():.1f} p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {q[4]:.0f} tilemax {tile.mean():.1f}", flush=True)
