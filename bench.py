"""Benchmark of the hot path: batched MPC::mpcLoop on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; instances are sharded, no
   collective in the data path, RCCL only reduces the timing / statistics)

A step is one pass of the hot path over one batch that is already resident in HBM: QP vector
assembly from (x0, reference) + ADMM solve + trajectory extraction for every instance.
Default workload: mpc_6dof, 4096 instances, N = 20, eps 1e-3 (the configuration the headline
metric is quoted on).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4D5043
WORKLOADS = {
    # name: (model, B per GPU, N, seed offset)
    "mpc_6dof_B4096_N20": ("6dof", 4096, 20, 2),
    "mpc_12state_B4096_N20": ("12state", 4096, 20, 3),
    "mpc_6dof_B32768_N50": ("6dof", 32768, 50, 4),
    # closed loop (per-GPU share of the 65536-vehicle, 500-step configuration): a step = warm-started
    # solve + plant update + reference shift
    "rollout_12state_B8192_N20": ("12state", 8192, 20, 5),
    "rollout_6dof_B8192_N20": ("6dof", 8192, 20, 5),
}
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix peak (v_mfma_f64_16x16x4: 2048 flop / 64 clk / SIMD at 2.4 GHz)
FP64_MFMA_SUSTAINED_TFLOPS = 75.7   # tools/microbench/mfma_valu_overlap.hip on the box: register-only MFMA, two chains per SIMD


def flops_per_iteration(model, N):
    """Dense reduced-KKT apply actually executed per ADMM iteration and instance: 2 * size^2
    per block (6-DOF: three per-axis blocks of 3N+2, the monolithic 9N+6 form of the reference
    would be 3x this; 12-state: Z and yaw one block of 4N+3 each, XY two blocks of 4N+3 -- the
    sum and the difference of its two interchangeable axes -- where the monolithic 8N+6 form
    would be 2x that)."""
    if model == "6dof":
        return 3 * 2 * (3 * N + 2) ** 2
    return sum(stage_flops_12state(N))


def stage_flops_12state(N):
    """Executed flops per iteration and instance of the stages in status-row order (z, xy, yaw)."""
    blk = 2 * (4 * N + 3) ** 2
    return [blk, 2 * blk, blk]


def cpu_baseline(model, N, seed_off, eps, rho, rho_rows, check=25, check_fine=5, smoothing=False, stress=False):
    """Reference CPU path restated (OSQP algorithm with its default settings, one instance at a
    time on one core, as the reference runs it) on a bounded sample of the same workload."""
    from oracle import oracle as O
    from trajectory_generation_amd.synth import stress as synth_stress
    O.build()
    sample = 384 if model == "6dof" else 256
    if N >= 50:
        sample //= 8
    out = {}
    fixed = lambda r: O.fixed_rho_settings(r, warm_start=0, eps_abs=eps, eps_rel=eps, check_termination=check,
                                           check_fine=check_fine)
    for kind, mk in (("osqp_defaults", lambda: O.default_settings(warm_start=0, eps_abs=eps, eps_rel=eps)),
                     ("engine_algorithm", None)):
        if mk:
            O.set_rho_class_scale(1.0, 1.0, 1.0, 1.0)    # OSQP: one rho for all inequality rows
        else:
            O.match_rho_classes(rho_rows)                # the engine's fixed per-class step sizes
        if model == "6dof":
            p = O.mpc6_params(N=N, enable_control_smoothing=smoothing)
            x0, ref = O.synth_mpc6(SEED + seed_off, N, 0.1, sample)
            if stress:
                x0, ref = synth_stress(model, x0, ref, N)
            s = mk() if mk else fixed(rho)
            t = time.perf_counter(); r = O.mpc6_batch(p, s, x0, ref); dt = time.perf_counter() - t
        else:
            p = O.mpc12_params(N=N, enable_control_smoothing=smoothing)
            x0, ref = O.synth_mpc12(SEED + seed_off, N, 0.1, sample)
            if stress:
                x0, ref = synth_stress(model, x0, ref, N)
            s = mk() if mk else {k: fixed(rho[i]) for i, k in enumerate(("xy", "z", "yaw"))}
            t = time.perf_counter(); r = O.mpc12_batch(p, s, x0, ref); dt = time.perf_counter() - t
        out[kind] = dict(value=sample / dt, iters_mean=float(np.mean(r["iters"])), solved=float(r["n_ok"]) / sample)
    return {"value": out["osqp_defaults"]["value"], "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": "%d instances of the same seeded workload, one at a time on one core; OSQP algorithm "
                      "restated with its default settings (Ruiz scaling 10, adaptive rho every 50 iterations, "
                      "eps as the GPU run), dense Cholesky of the reduced KKT matrix; upstream OSQP 25b6b39 is not "
                      "available offline" % sample,
            "iters_mean": out["osqp_defaults"]["iters_mean"], "solved_frac": out["osqp_defaults"]["solved"],
            "same_algorithm_as_gpu": out["engine_algorithm"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="mpc_6dof_B4096_N20", choices=sorted(WORKLOADS))
    ap.add_argument("--eps", type=float, default=1e-3)
    ap.add_argument("--max-iter", type=int, default=4000, help="ADMM iteration budget per solve (OSQP default 4000)")
    ap.add_argument("--check", type=int, default=25, help="termination check interval (OSQP default 25)")
    ap.add_argument("--grid-cap", type=int, default=0, help="most workgroups per solve (0: what the GPU holds at once, -1: one per tile of 16 instances)")
    ap.add_argument("--lowrank", type=int, default=0, help="K^-1 as diagonal tiles + low-rank rest: 0 where it pays, 1 wherever possible, -1 dense")
    ap.add_argument("--check-fine", type=int, default=None, help="check interval after the first check (0: OSQP's fixed interval)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path with several ranks on one GPU")
    ap.add_argument("--smoothing", action="store_true", help="input smoothing term on (the test mains' setting; config/mpc.yaml has it off)")
    ap.add_argument("--stress", action="store_true", help="every instance in the geometry of the reference's test main: target 40 m above, limits active")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import trajectory_generation_amd as tg
    from trajectory_generation_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"    # where the few statistics are reduced
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    model, B, N, seed_off = WORKLOADS[args.workload]
    if args.check_fine is None:
        # One tile per CU (4096 instances): after the first check test every 5 iterations, so a
        # tile of 16 instances does not wait for the next multiple of 25.  Batches of many tiles per
        # CU refill their instance slots from the batch instead, no slot waits for its tile, and the
        # cheaper schedule is OSQP's own.  Closed loop and the 12-state chain keep OSQP's interval
        # as well: their iterates feed a next problem (next step's initial state, the XY bounds,
        # the yaw reference), and iterates nearer the tolerance make more of those infeasible or
        # slow to certify as such.
        args.check_fine = 5 if args.workload == "mpc_6dof_B4096_N20" else 0
    opts = tg.make_opts(eps_abs=args.eps, eps_rel=args.eps, max_iter=args.max_iter, check_termination=args.check, check_fine=args.check_fine, grid_cap=args.grid_cap, lowrank=args.lowrank)
    first = rank * B                       # weak scaling: every rank owns its own slab of B instances
    if model == "6dof":
        eng = tg.BatchedMPC6(tg.mpc6_params(N=N, enable_control_smoothing=args.smoothing), opts, device=local)
        x0, ref = synth.mpc6(SEED + seed_off, N, 0.1, B, first)
    else:
        eng = tg.BatchedMPC12(tg.mpc12_params(N=N, enable_control_smoothing=args.smoothing), opts, device=local)
        x0, ref = synth.mpc12(SEED + seed_off, N, 0.1, B, first)
    if args.stress:
        x0, ref = synth.stress(model, x0, ref, N)
    buf = eng.alloc(B)
    buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
    buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()

    def barrier():
        if world > 1:
            dist.barrier()

    closed = args.workload.startswith("rollout")

    def step():
        if closed:
            eng.solve(buf, warm=True)
            eng.plant_step(buf)
            eng.ref_shift(buf)
        else:
            eng.solve(buf)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); barrier()
    # HIP events on the launch stream (the engine launches on the null stream, torch's current one)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-batch latency distribution and kernel time (HIP events on the launch stream),
    # outside the timed region
    lat = []
    tg.timing_enable(True); tg.timing_read(reset=True)
    for _ in range(max(5, min(args.steps, 20))):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        step()
        torch.cuda.synchronize(); lat.append((time.perf_counter() - t1) * 1e3)
    kern_ms, kern_launches = tg.timing_read(reset=True)
    tg.timing_enable(False)

    # the same step when the boundary hands over host buffers: H2D of x0 / ref (pinned, SoA),
    # solve, D2H of X / U / status; reported beside the resident-in-HBM figure, never as `value`
    hx0 = buf["x0"].cpu().pin_memory(); href = buf["ref"].cpu().pin_memory()
    hX = torch.empty_like(buf["X"], device="cpu").pin_memory(); hU = torch.empty_like(buf["U"], device="cpu").pin_memory()
    hst = torch.empty_like(buf["status"], device="cpu").pin_memory()
    pcie = []
    for _ in range(5):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        buf["x0"].copy_(hx0, non_blocking=True); buf["ref"].copy_(href, non_blocking=True)
        eng.solve(buf, warm=closed)
        hX.copy_(buf["X"], non_blocking=True); hU.copy_(buf["U"], non_blocking=True); hst.copy_(buf["status"], non_blocking=True)
        torch.cuda.synchronize(); pcie.append(time.perf_counter() - t1)
    pcie_rate = B / float(np.median(pcie))

    iters = buf["iters"].cpu().numpy().reshape(-1, buf["ld"])[:, :B]
    status = buf["status"].cpu().numpy().reshape(-1, buf["ld"])[:, :B]
    stats = torch.tensor([float(iters.sum()), float((status == 1).all(0).sum()), float(B)],
                         dtype=torch.float64, device=red_dev)
    imax = torch.tensor([float(iters.max())], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(imax, op=dist.ReduceOp.MAX)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_inst = float(stats[2].item())
    value = total_inst * args.steps / elapsed
    stages = iters.shape[0]
    # roofline of the dominant kernel (admm_kernel), rank 0's launches
    launches_per_step = stages
    flops_launch = float(iters.sum()) * flops_per_iteration(model, N) / launches_per_step
    if model != "6dof":   # per-stage sizes differ: weight the stage iteration counts
        flops_launch = sum(float(iters[k].sum()) * f for k, f in enumerate(stage_flops_12state(N))) / 3
    # flops the matrix cores actually execute (2048 per v_mfma_f64_16x16x4_f64, padding rows and
    # lock-step iterations of finished tile members included as executed work is per tile): the
    # low-rank operator form executes fewer than the dense product the algorithmic figure counts
    if model == "6dof":
        info = [eng.kernel_info()]
    else:
        info = [eng.kernel_info(k) for k in ("z", "xy", "yaw")]
    executed = sum(float(iters[k].sum()) / 16.0 * info[k][0] * 2048.0 for k in range(stages)) / launches_per_step
    # average admm_kernel launch over the timed region: event time / launches (the other kernels of a
    # step -- xy bounds, yaw reference, plant, shift -- are < 1 % of it); the per-launch event pairs
    # taken after the timed region are reported beside it
    avg_kernel_ms = region_ms / (args.steps * launches_per_step)
    per_launch_event_ms = kern_ms / max(kern_launches, 1)
    achieved = flops_launch / (avg_kernel_ms * 1e-3) / 1e12 if avg_kernel_ms > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(args.workload)
        except Exception:
            traffic = None
    out = {
        "metric": "MPC QP solves/sec (whole node) at fixed tolerance; p50 per-batch solve latency",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "model": None, "instances_per_gpu": B, "horizon_N": N,
                   "eps_abs": args.eps, "eps_rel": args.eps, "max_iter": args.max_iter, "check_termination": args.check, "check_fine": args.check_fine, "grid_cap": args.grid_cap, "lowrank": args.lowrank,
                   "smoothing": bool(args.smoothing), "inputs": "stress" if args.stress else "seeded", "rho": eng.rho if model == "6dof" else
                   [eng.rho(k) for k in ("xy", "z", "yaw")], "parallelism": "instances sharded x%d" % world},
        "p50_batch_ms": float(np.median(lat)),
        "iters_mean": float(stats[0].item()) / total_inst / stages, "iters_max": float(imax.item()),
        "solved_frac": float(stats[1].item()) / total_inst,
        "iters_hist": {str(int(k)): int(v) for k, v in zip(*np.unique(iters, return_counts=True))},
        "pcie_inclusive_solves_per_s_rank0": pcie_rate,
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "kernel": "admm_kernel",
                     "avg_kernel_ms": avg_kernel_ms, "launches_timed": args.steps * launches_per_step,
                     "per_launch_event_ms": per_launch_event_ms,
                     "sustained_mfma_microbench": FP64_MFMA_SUSTAINED_TFLOPS,
                     "flops_per_launch": flops_launch,
                     "operator_form": ["lowrank" if i[1] else "dense" for i in info],
                     "mfma_per_tile_iteration": [i[0] for i in info],
                     "executed_mfma_flops_per_launch_at_least": executed,
                     "executed_frac_at_least": executed / (avg_kernel_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if avg_kernel_ms > 0 else 0.0},
    }
    del out["config"]["model"]
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(model, N, seed_off, args.eps, out["config"]["rho"], eng.rho_rows(), args.check, args.check_fine,
                                           args.smoothing, args.stress)
        if closed:
            out["cpu_baseline"]["sample"] += "; cold-started single solves of the same problems (the closed-loop GPU " \
                                             "steps are warm-started)"
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
