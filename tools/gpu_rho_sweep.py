"""Sweep of the per-class rho multipliers on the GPU engine (development aid).
usage: gpu_rho_sweep.py model N B eps[,eps...]"""
import os, sys, itertools, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth

def tilemax(it):
    n = len(it) // 16 * 16
    return it[:n].reshape(-1, 16).max(1).mean()

def main(model, N, B, epss, grid):
    if model == "6dof":
        x0, ref = synth.mpc6(0x4D5043 + 2, N, 0.1, B)
    else:
        x0, ref = synth.mpc12(0x4D5043 + 3, N, 0.1, B)
    for rho, sp, sv, sa, su in grid:
        for eps in epss:
            opts = tg.make_opts(rho=rho, eps_abs=eps, eps_rel=eps, rho_scale=(sp, sv, sa, su), max_iter=4000)
            eng = tg.BatchedMPC6(tg.mpc6_params(N=N), opts) if model == "6dof" else tg.BatchedMPC12(tg.mpc12_params(N=N), opts)
            buf = eng.alloc(B)
            buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
            buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
            eng.solve(buf); torch.cuda.synchronize()
            t = time.perf_counter(); eng.solve(buf); torch.cuda.synchronize(); dt = time.perf_counter() - t
            it = buf["iters"].cpu().numpy().reshape(-1, buf["ld"])[:, :B]
            st = buf["status"].cpu().numpy().reshape(-1, buf["ld"])[:, :B]
            for k in range(it.shape[0]):
                i = it[k]
                print(f"{model} N={N} stage{k} rho={rho:g} scale=({sp:g},{sv:g},{sa:g},{su:g}) eps={eps:g}: mean {i.mean():.1f} p99 {np.percentile(i,99):.0f} max {i.max()} tilemax {tilemax(i):.1f} solved {np.mean(st[k]==1):.4f} batch {dt*1e3:.2f} ms", flush=True)

if __name__ == "__main__":
    model, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    epss = [float(e) for e in sys.argv[4].split(",")]
    fine = len(sys.argv) > 5 and sys.argv[5] == "fine"
    if model == "6dof" and fine:
        grid = [(r, p, v, 1, u) for r, p, v, u in itertools.product([1000.], [10, 15, 20, 30, 50], [1, 2, 3], [2, 3, 5])]
    elif model == "6dof":
        grid = [(r, p, v, 1, u) for r, p, v, u in itertools.product([300., 1000.], [1, 3, 10, 30, 100], [1, 3], [1, 3, 10])]
    else:
        grid = [(r, p, v, a, u) for r, p, v, a, u in itertools.product([0.], [1, 10, 30], [1, 3], [1, 3], [1, 3, 10])]
    main(model, N, B, epss, grid)
