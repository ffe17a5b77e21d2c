"""Per-iteration cost of the ADMM kernel: eps = 0 so that every workgroup runs exactly max_iter
iterations (development aid).  usage: iter_cost.py [6dof|12state] N B [max_iter]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth

def run(model, N, B, iters=200, reps=5, check=25, lowrank=0):
    opts = tg.make_opts(eps_abs=0.0, eps_rel=0.0, max_iter=iters, check_termination=check, eps_prim_inf=0.0, check_fine=0, lowrank=lowrank)
    if model == "6dof":
        eng = tg.BatchedMPC6(tg.mpc6_params(N=N), opts); x0, ref = synth.mpc6(1, N, 0.1, B)
        flops = 3 * 2 * (3 * N + 2) ** 2
    else:
        eng = tg.BatchedMPC12(tg.mpc12_params(N=N), opts); x0, ref = synth.mpc12(1, N, 0.1, B)
        flops = 4 * 2 * (4 * N + 3) ** 2
    buf = eng.alloc(B)
    buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
    buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
    eng.solve(buf); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): eng.solve(buf)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    wgs = (B + 15) // 16
    waves = max(1.0, wgs / 256.0)
    print(f"{model} N={N} B={B} check={check} lowrank={lowrank}: {dt*1e3:.3f} ms / {iters} iters = {dt/iters*1e6:.2f} us/iter/launch; "
          f"per-WG-iteration {dt/iters*1e6/waves:.2f} us (WGs/CU {waves:.1f}); {B*iters*flops/dt/1e12:.2f} TFLOP/s", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 200,
            check=int(sys.argv[5]) if len(sys.argv) > 5 else 25, lowrank=int(sys.argv[6]) if len(sys.argv) > 6 else 0)
    else:
        for B in (4096, 8192, 16384, 65536):
            run("6dof", 20, B)
        for B in (4096, 16384):
            run("6dof", 50, B, 100)
        for B in (4096, 16384):
            run("12state", 20, B, 100)
