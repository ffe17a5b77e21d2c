"""Phase times of an ADMM iteration from a -DTG_STAMP build (development aid).
  hipcc ... -DTG_STAMP -DTG_VARIANTS=<mask> -shared -o trajectory_generation_amd/lib/dev/libtg_stamp.so ...
  TG_MPC_LIB=.../libtg_stamp.so python tools/phase_stamps.py 6dof 50 [iters]
Prints, per wave of workgroup 0, shader-clock cycles per iteration spent in: rhs, barrier, MFMA stage 1,
barrier, MFMA stage 2, barrier, projection, end-of-iteration barrier (+ checks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
model, N = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
B = 4096
opts = tg.make_opts(eps_abs=0.0, eps_rel=0.0, max_iter=iters, check_termination=iters, eps_prim_inf=0.0)
eng = tg.BatchedMPC6(tg.mpc6_params(N=N), opts)
x0, ref = synth.mpc6(1, N, 0.1, B)
buf = eng.alloc(B)
buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
eng.solve(buf); torch.cuda.synchronize()
acc = buf["U"][0, :64].cpu().numpy().reshape(8, 8) / iters
names = ["rhs", "bar", "mfma1", "bar", "mfma2", "bar", "proj", "bar+"]
print("cycles per iteration (kernel info %s)" % (eng.kernel_info(),))
print("wave " + " ".join("%8s" % n for n in names) + "    total")
for w in range(8):
    print("%4d " % w + " ".join("%8.0f" % v for v in acc[w]) + " %8.0f" % acc[w].sum())
