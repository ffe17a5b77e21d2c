"""Per-iteration cost of each 12-state stage kernel (eps = 0: every tile runs max_iter iterations).
usage: iter_cost12.py N B [iters] [lowrank]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
N, B = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lowrank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
good = tg.BatchedMPC12(tg.mpc12_params(N=N))
x0, ref = synth.mpc12(1, N, 0.1, B)
buf = good.alloc(B)
buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
good.solve(buf); torch.cuda.synchronize()          # a real upstream solution in X
eng = tg.BatchedMPC12(tg.mpc12_params(N=N), tg.make_opts(eps_abs=0.0, eps_rel=0.0, max_iter=iters,
                                                         check_termination=iters, eps_prim_inf=0.0, lowrank=lowrank))
X = buf["X"].clone()
for stage, n, blocks in (("z", 4 * N + 3, 1), ("xy", 4 * N + 3, 2), ("yaw", 4 * N + 3, 1)):
    buf["X"].copy_(X)
    eng.solve_stage(stage, buf); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        eng.solve_stage(stage, buf)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"12state N={N} B={B} lowrank={lowrank} stage {stage}: {dt/iters*1e6:.2f} us per tile-iteration, {B*iters*blocks*2*n*n/dt/1e12:.2f} TFLOP/s", flush=True)
