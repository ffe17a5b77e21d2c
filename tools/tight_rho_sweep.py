"""Step sizes for tight tolerances (development aid; GPU box): 6-DOF, mean / max iterations at a given
eps over a grid of base rho multipliers, class scales and relaxation, on a held-out seed.
usage: tight_rho_sweep.py [N [B [eps,eps,... [mult,mult,... [short]]]]]   (short: built-in class scales, alpha 1.6 only)"""
import os, sys, itertools, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
epss = [float(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1e-8]
mults = [float(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0.03, 0.1, 0.3, 1.0, 3.0, 10.0]
scales = [None, (1, 1, 1, 1), (3, 1, 1, 1)] if len(sys.argv) <= 5 else [None]
alphas = [1.6, 1.8] if len(sys.argv) <= 5 else [1.6]
x0, ref = synth.mpc6(77, N, 0.1, B, 0)
base = tg.BatchedMPC6(tg.mpc6_params(N=N)).rho
print("default rho", base, flush=True)
for eps, mult, scale, alpha in itertools.product(epss, mults, scales, alphas):
    kw = dict(rho=base * mult, eps_abs=eps, eps_rel=eps, alpha=alpha, max_iter=100000)
    if scale:
        kw["rho_scale"] = scale
    eng = tg.BatchedMPC6(tg.mpc6_params(N=N), tg.make_opts(**kw))
    buf = eng.alloc(B)
    buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
    buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
    eng.solve(buf); torch.cuda.synchronize()
    it = buf["iters"][:B].double(); st = buf["status"][:B]
    print("eps %.0e rho x%-5g scale %-14s alpha %.1f: mean %8.1f  p99 %8.0f  max %7d  solved %.4f" % (
        eps, mult, scale or "built-in", alpha, float(it.mean()), float(it.quantile(0.99)), int(it.max()), float((st == 1).double().mean())), flush=True)
