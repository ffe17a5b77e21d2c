"""One 6-DOF solve with a fixed number of iterations and a chosen check interval (eps 0: nothing
exits early), for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes that separate the traffic of the
iterations from that of the termination checks (development aid).
usage: traffic_probe.py N B iters check"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
N, B, iters, chk = (int(v) for v in sys.argv[1:5])
eng = tg.BatchedMPC6(tg.mpc6_params(N=N), tg.make_opts(eps_abs=0.0, eps_rel=0.0, max_iter=iters, check_termination=chk, eps_prim_inf=0.0))
x0, ref = synth.mpc6(1, N, 0.1, B)
buf = eng.alloc(B)
buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
for _ in range(3):
    eng.solve(buf)
torch.cuda.synchronize()
print("done", int(buf["iters"][:B].max()))
