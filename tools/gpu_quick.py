"""Quick GPU sanity run: engine vs CPU oracle on seeded instances (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import trajectory_generation_amd as tg
from oracle import oracle as O

def run6(N, B, eps, smooth=False):
    p = tg.mpc6_params(N=N, enable_control_smoothing=smooth)
    eng = tg.BatchedMPC6(p, tg.make_opts(eps_abs=eps, eps_rel=eps))
    x0, ref = O.synth_mpc6(0x4D5043 + 2, N, 0.1, B)
    t = time.time(); r = eng.solve_host(x0, ref); tgpu = time.time() - t
    op = O.mpc6_params(N=N, enable_control_smoothing=smooth)
    s = O.fixed_rho_settings(eng.rho, eps_abs=eps, eps_rel=eps, warm_start=0)
    nb = min(B, 256)
    t = time.time(); ro = O.mpc6_batch(op, s, x0[:nb], ref[:nb]); tcpu = time.time() - t
    dX = np.abs(r['X'][:nb] - ro['X']).max(); dU = np.abs(r['U'][:nb] - ro['U']).max()
    print(f"6dof N={N} B={B} eps={eps} smooth={smooth} rho={eng.rho}: status ok {np.mean(r['status']==1):.3f} iters gpu mean {r['iters'].mean():.1f} max {r['iters'].max()} | iters equal {np.mean(r['iters'][:nb]==ro['iters']):.3f} dX {dX:.2e} dU {dU:.2e} | gpu {tgpu*1e3:.1f} ms cpu/solve {tcpu/nb*1e6:.0f} us", flush=True)

def run12(N, B, eps):
    p = tg.mpc12_params(N=N)
    eng = tg.BatchedMPC12(p, tg.make_opts(eps_abs=eps, eps_rel=eps))
    x0, ref = O.synth_mpc12(0x4D5043 + 3, N, 0.1, B)
    t = time.time(); r = eng.solve_host(x0, ref); tgpu = time.time() - t
    op = O.mpc12_params(N=N)
    nb = min(B, 128)
    X = np.zeros((nb, 12*(N+1))); U = np.zeros((nb, 4*N)); its = np.zeros((nb,3),int); sts=np.zeros((nb,3),int)
    # per-stage rho: oracle object per rho is awkward; use one oracle per stage by running the whole loop with stage rho = engine's
    rhos = [eng.rho('z'), eng.rho('xy'), eng.rho('yaw')]
    print('  rhos z/xy/yaw', rhos)
    same = len(set(rhos)) == 1
    if same:
        s = O.fixed_rho_settings(rhos[0], eps_abs=eps, eps_rel=eps, warm_start=0)
        ro = O.mpc12_batch(op, s, x0[:nb], ref[:nb])
        dX = np.abs(r['X'][:nb] - ro['X']).max(); dU = np.abs(r['U'][:nb] - ro['U']).max()
        print(f"12st N={N} B={B} eps={eps}: ok {np.mean((r['status']==1).all(1)):.3f} iters mean {r['iters'].mean(0)} max {r['iters'].max(0)} | iters equal {np.mean(r['iters'][:nb]==ro['iters']):.3f} dX {dX:.2e} dU {dU:.2e} gpu {tgpu*1e3:.1f} ms", flush=True)
    else:
        print(f"12st N={N} B={B} eps={eps}: ok {np.mean((r['status']==1).all(1)):.3f} iters mean {r['iters'].mean(0)} max {r['iters'].max(0)} gpu {tgpu*1e3:.1f} ms", flush=True)

if __name__ == '__main__':
    print(torch.cuda.get_device_name(0), flush=True)
    run6(20, 64, 1e-3)
    run6(20, 100, 1e-5)
    run6(20, 64, 1e-3, smooth=True)
    run6(5, 37, 1e-3)
    run6(50, 64, 1e-3)
    run12(20, 64, 1e-3)
    run12(10, 30, 1e-4)
    # timing
    for N, B in ((20, 4096), (50, 4096), (50, 32768)):
        p = tg.mpc6_params(N=N)
        eng = tg.BatchedMPC6(p, tg.make_opts())
        x0, ref = O.synth_mpc6(1, N, 0.1, B)
        buf = eng.alloc(B)
        buf['x0'][:, :B] = torch.from_numpy(x0.T.copy()).cuda(); buf['ref'][:, :B] = torch.from_numpy(ref.T.copy()).cuda()
        for _ in range(3): eng.solve(buf)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(10): eng.solve(buf)
        torch.cuda.synchronize(); dt = (time.time() - t) / 10
        it = buf['iters'][:B].cpu().numpy()
        print(f"time 6dof N={N} B={B}: {dt*1e3:.3f} ms/batch -> {B/dt:.3e} solves/s, iters mean {it.mean():.1f} max {it.max()} ok {np.mean(buf['status'][:B].cpu().numpy()==1):.4f}", flush=True)
