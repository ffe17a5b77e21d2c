"""Where in the batch do the slow instances sit, and how long does the batch take without them?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth

N, B = 50, 32768
fine = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng = tg.BatchedMPC6(tg.mpc6_params(N=N), tg.make_opts(check_fine=fine))
x0, ref = synth.mpc6(1234 + 2, N, 0.1, B)
buf = eng.alloc(B)
buf["x0"][:, :B] = torch.as_tensor(x0.T, device="cuda"); buf["ref"][:, :B] = torch.as_tensor(ref.T, device="cuda")

def timeit(n=10):
    eng.solve(buf); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): eng.solve(buf)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

ms = timeit()
it = buf["iters"][:B].cpu().numpy()
slow = np.argsort(-it)[:12]
print("fine", fine, "ms", ms, "sum iters", it.sum(), "slow:", [(int(i), int(it[i])) for i in slow])
# replace the slow instances by copies of instance 0
for i in slow:
    buf["x0"][:, i] = buf["x0"][:, 0]; buf["ref"][:, i] = buf["ref"][:, 0]
ms2 = timeit()
it2 = buf["iters"][:B].cpu().numpy()
print("without them: ms", ms2, "sum iters", it2.sum(), "max", it2.max())
