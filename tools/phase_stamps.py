"""Phase times of an ADMM iteration from a -DTG_STAMP build (development aid).
  hipcc ... -DTG_STAMP -DTG_VARIANTS=<mask> -shared -o trajectory_generation_amd/lib/dev/libtg_stamp.so ...
  TG_MPC_LIB=.../libtg_stamp.so python tools/phase_stamps.py 6dof 50 [iters [check_interval [real]]]
Prints, per wave of workgroup 0, shader-clock cycles per iteration spent in: rhs, barrier, MFMA stage 1,
barrier, MFMA stage 2, barrier, projection, end-of-iteration barrier (+ checks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
model, N = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
CHK = int(sys.argv[4]) if len(sys.argv) > 4 else iters
REAL = len(sys.argv) > 5 and sys.argv[5] == "real"     # eps 1e-3 on a batch of 32768: exits and slot refills included
B = 32768 if REAL else 4096
opts = tg.make_opts(phase_iters=(-1, -1)) if REAL else tg.make_opts(eps_abs=0.0, eps_rel=0.0, max_iter=iters, check_termination=CHK, eps_prim_inf=0.0, phase_iters=(-1, -1))
if model == "6dof":
    eng = tg.BatchedMPC6(tg.mpc6_params(N=N), opts)
    x0, ref = synth.mpc6(1, N, 0.1, B)
else:                      # 12z / 12xy / 12yaw: one stage of the 12-state chain (upstream solution: zeros)
    eng = tg.BatchedMPC12(tg.mpc12_params(N=N), opts)
    x0, ref = synth.mpc12(1, N, 0.1, B)
buf = eng.alloc(B)
buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
if model == "6dof":
    eng.solve(buf)
else:
    eng.solve_stage(model[2:], buf)
torch.cuda.synchronize()
acc = buf["U"][0, :16].cpu().numpy().reshape(2, 8)
ck = buf["X"][0, :16].cpu().numpy().reshape(2, 8)
acc = acc / (ck[0][7] * 25 if REAL else iters)
names = ["rhs", "bar", "mfma1", "bar", "mfma2", "bar", "proj", "bar+"]
print("cycles per iteration (kernel info %s)" % (eng.kernel_info() if model == "6dof" else eng.kernel_info(model[2:]),))
print("wave " + " ".join("%8s" % n for n in names) + "    total")
for w in range(2):
    print("%4d " % (4 * w) + " ".join("%8.0f" % v for v in acc[w]) + " %8.0f" % acc[w].sum())
print("cycles per termination check: barrier, A' dy, x and y to LDS | residual loops | reductions | exit writes | restore x, y, w | refill + load")
for w in range(2):
    n = max(ck[w][7], 1.0)
    print("%4d " % (4 * w) + " ".join("%8.0f" % (v / n) for v in ck[w][:6]) + "   total %8.0f  (%d checks)" % (ck[w][:6].sum() / n, n))
