"""Exact number of ADMM iterations every instance needs (termination check after every iteration),
for a given per-class rho scale (development aid): iter_need.py [6dof|12state] N B p,v,a,u [eps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
model, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
scale = tuple(float(v) for v in sys.argv[4].split(","))
eps = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-3
opts = tg.make_opts(eps_abs=eps, eps_rel=eps, check_termination=1, check_fine=0, rho_scale=scale)
if model == "6dof":
    eng = tg.BatchedMPC6(tg.mpc6_params(N=N), opts); x0, ref = synth.mpc6(0x4D5043 + 2, N, 0.1, B)
else:
    eng = tg.BatchedMPC12(tg.mpc12_params(N=N), opts); x0, ref = synth.mpc12(0x4D5043 + 3, N, 0.1, B)
buf = eng.alloc(B)
buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
eng.solve(buf); torch.cuda.synchronize()
it = buf["iters"].cpu().numpy().reshape(-1, buf["ld"])[:, :B]
for k in range(it.shape[0]):
    v = it[k]
    q = np.percentile(v, [50, 90, 99, 99.9, 100])
    tile = v.reshape(-1, 16).max(1)
    print(f"{model} N={N} scale={scale} eps={eps} stage{k}: mean {v.mean():.1f} p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {q[4]:.0f}; tile-max mean {tile.mean():.1f}; "
          f"<=25: {(v<=25).mean():.4f} <=30: {(v<=30).mean():.4f} <=35: {(v<=35).mean():.4f} <=50: {(v<=50).mean():.4f}", flush=True)
