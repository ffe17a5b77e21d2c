"""Closed-loop 12-state step (warm solve + gated plant + reference shift) of four sub-fleets: launched
call by call against replayed from one HIP graph per sub-fleet (development aid; GPU box)."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trajectory_generation_amd as tg
from trajectory_generation_amd import synth
N, B, depth, steps = 20, 8192, 4, 300
Bh = B // depth
engs = [tg.BatchedMPC12(tg.mpc12_params(N=N), tg.make_opts()) for _ in range(depth)]
bufs = [e.alloc(Bh) for e in engs]
streams = [torch.cuda.Stream() for _ in range(depth)]
def fill():
    for h, hb in enumerate(bufs):
        x0, ref = synth.mpc12(0x4D5043 + 5, N, 0.1, Bh, h * Bh)
        hb["x0"][:, :Bh] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
        hb["ref"][:, :Bh] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
def one(h):
    engs[h].solve(bufs[h], warm=True, stream=streams[h])
    engs[h].plant_step(bufs[h], stream=streams[h])
    engs[h].ref_shift(bufs[h], stream=streams[h])
fill()
for _ in range(3):
    for h in range(depth): one(h)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    for h in range(depth): one(h)
torch.cuda.synchronize()
t_calls = (time.perf_counter() - t0) / steps
xa = [hb["x0"].clone() for hb in bufs]
# the same with one graph per sub-fleet
fill()
for _ in range(3):
    for h in range(depth): one(h)
torch.cuda.synchronize()
graphs = []
for h in range(depth):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[h]):
        engs[h].solve(bufs[h], warm=True)
        engs[h].plant_step(bufs[h])
        engs[h].ref_shift(bufs[h])
    graphs.append(g)
torch.cuda.synchronize()
# capture does not execute: the state is where the three warm-up steps left it
t0 = time.perf_counter()
for _ in range(steps):
    for h in range(depth):
        with torch.cuda.stream(streams[h]):
            graphs[h].replay()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / steps
same = all(torch.equal(a, hb["x0"]) for a, hb in zip(xa, bufs))
print("per step: calls %.3f ms (%.2f M solves/s), graphs %.3f ms (%.2f M solves/s), final states identical: %s" % (
    t_calls * 1e3, B / t_calls / 1e6, t_graph * 1e3, B / t_graph / 1e6, same))
