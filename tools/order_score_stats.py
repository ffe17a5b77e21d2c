"""How well the ordering score (largest distance of the reference from where the initial state coasts
to) predicts the iteration count of the seeded N = 50 batch: iteration statistics per score quantile
(development aid; GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
N, B = 50, 65536
eng = tg.BatchedMPC6(tg.mpc6_params(N=N), tg.make_opts())
x0, ref = synth.mpc6(0x4D5043 + 4, N, 0.1, B)
buf = eng.alloc(B)
buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
eng.solve(buf)
torch.cuda.synchronize()
it = buf["iters"][:B].cpu().numpy()
dt = 0.1
k = np.arange(N + 1) * dt
r = ref.reshape(B, N + 1, 6)
sc = np.zeros(B)
for ax in range(3):
    coast = x0[:, ax, None] + k[None, :] * x0[:, 3 + ax, None]
    sc = np.maximum(sc, np.abs(r[:, :, ax] - coast).max(1))
order = np.argsort(sc)
print("iters: mean %.2f, >25: %.3f, >50: %.4f, >100: %.5f, max %d" % (it.mean(), (it > 25).mean(), (it > 50).mean(), (it > 100).mean(), it.max()))
for lo, hi in ((0, .1), (.1, .25), (.25, .5), (.5, .75), (.75, .9), (.9, .9375), (.9375, 1.0)):
    s = it[order[int(lo * B):int(hi * B)]]
    print("score quantile %.4f-%.4f: mean %.2f  >25 %.4f  >50 %.5f  >100 %.5f  max %d" % (lo, hi, s.mean(), (s > 25).mean(), (s > 50).mean(), (s > 100).mean(), s.max()))
