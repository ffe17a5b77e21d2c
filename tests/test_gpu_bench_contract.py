"""The benchmark's one-line JSON contract, single rank and two ranks (gloo rehearsal of the
sharded path with both ranks on GPU 0)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_rank():
    """The default run: the configuration the headline target is quoted on (6-DOF, N = 50, 262144
    instances), with the roofline of the executed matrix-core work, the CPU baseline and the
    parity figures against the restated OSQP-default path at the same eps."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, cwd=ROOT)
    d = _last_json(r.stdout.decode())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["workload"] == "mpc_6dof_B262144_N50" and "model" not in d["config"]
    assert d["config"]["horizon_N"] == 50 and d["config"]["instances_total"] == 262144 and d["config"]["check_fine"] == 0
    assert d["scaling"] == "strong" and d["timed_s"] >= 2.0 and d["timed_steps"] % 5 == 0 and d["ranks_seen"] == 1
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # frac counts the matrix-core flops actually executed: MFMAs per tile-iteration x 2048 / 16 per instance-iteration
    per_it = sum(rf["mfma_per_tile_iteration"]) * 2048 / 16
    # ... shared out over the launches of a solve (one per operator of the step-size schedule)
    assert rf["launches_per_solve"] == [3] and rf["launches_timed"] == 3 * d["timed_steps"]
    assert abs(rf["executed_flops_per_launch"] - per_it * d["iters_mean"] * 262144 / 3) < 1e-6 * rf["executed_flops_per_launch"]
    assert 0 < rf["frac"] < 1 and rf["dense_equivalent_tflops"] > rf["achieved"] and d["solved_frac"] == 1.0
    assert d["value"] > 1e6
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "solves/s"
    assert cb["single_core"]["value"] > 0 and "LDL" in cb["sample"]
    par = d["parity"]
    assert par["solved_by_both"] >= 0.9 * par["instances"] and par["instances"] >= 256
    # both sides stopped at eps 1e-3; the engine's iterates are the ones nearer the optimum
    # (max over 2048 trajectories of 456 entries each, N = 50: the far end of the horizon is loosely determined)
    assert par["gpu_vs_optimum_maxabs"] < 2.5 and par["gpu_vs_optimum_maxabs"] <= par["osqp_defaults_vs_optimum_maxabs"]
    assert par["gpu_vs_osqp_defaults_maxabs"] <= par["gpu_vs_optimum_maxabs"] + par["osqp_defaults_vs_optimum_maxabs"] + 1e-12
    # both sides run to convergence: within the target's 1e-4
    assert par["converged"]["solved_by_both"] >= 0.9 * par["converged"]["instances"] >= 28
    assert par["converged_gpu_vs_converged_osqp_defaults_maxabs"] < par["converged"]["target"] == 1e-4
    # what a receding-horizon user applies, x(1) and u(0), is part of the whole-horizon figure
    fs = par["first_step_maxabs"]
    assert 0 <= fs["gpu_vs_optimum"] <= par["gpu_vs_optimum_maxabs"] and 0 <= fs["gpu_vs_osqp_defaults"] <= par["gpu_vs_osqp_defaults_maxabs"]
    assert fs["converged_gpu_vs_converged_osqp_defaults"] <= par["converged_gpu_vs_converged_osqp_defaults_maxabs"]
    # second leg: the rate at the accuracy the target names -- the loosest tolerance whose every entry of
    # X and U over the whole batch lies within 1e-4 of a converged run, timed over whole batches; an
    # eighth of the node's target rate (1e6 solves/s on eight GPUs) is the least one GPU must deliver
    at = d["at_target_accuracy"]
    assert at["target"] == 1e-4 and at["eps"] is not None and at["eps"] < 1e-3 and at["converged_solved_frac"] == 1.0
    assert at["max_abs_vs_converged"] <= 1e-4 and at["first_step_maxabs_vs_converged"] <= at["max_abs_vs_converged"]
    assert at["value"] >= 1.25e5 and at["steps"] >= 3 and at["timed_s"] >= 1.0 and at["solved_frac"] == 1.0
    assert abs(at["value"] - 262144 / (at["ms_per_step"] * 1e-3)) < 1e-6 * at["value"]
    assert at["ladder"][-1]["eps"] == at["eps"] and all(l["max_abs_vs_converged"] > 1e-4 for l in at["ladder"][:-1])
    assert 0 < at["roofline_frac"] < 1 and at["iters_mean"] > d["iters_mean"]
    pt = par["at_target_accuracy"]
    assert pt["eps"] == at["eps"] and pt["gpu_vs_converged_osqp_defaults_maxabs"] < 1e-4


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` without a launcher starts its two ranks itself (gloo rehearsal,
    both on GPU 0) and prints one line for the two of them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "10",
                        "--warmup", "2", "--workload", "mpc_6dof_B4096_N20", "--no-target-leg"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    d = _last_json(r.stdout.decode())
    assert r.returncode == 0 and d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["config"]["instances_total"] == 2 * 4096 and d["solved_frac"] == 1.0


def test_bench_small_batch_line_carries_the_finer_check_schedule_as_a_second_figure():
    """mpc_6dof_B4096_N20 keeps OSQP's schedule of checks for its value; the figure with convergence
    also tested every 5 iterations is a second key."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2",
                        "--workload", "mpc_6dof_B4096_N20", "--no-cpu-baseline", "--no-target-leg"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    d = _last_json(r.stdout.decode())
    cf = d["check_fine_5"]
    assert d["config"]["check_fine"] == 0 and cf["solved_frac"] == 1.0 and cf["value"] > 1e6
    assert cf["iters_mean"] <= d["iters_mean"] and cf["iters_max"] <= d["iters_max"]


def test_bench_two_ranks_shard_the_instances():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5",
           "--warmup", "1", "--backend", "gloo", "--workload", "mpc_6dof_B32768_N50", "--no-target-leg"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    d = _last_json(r.stdout.decode())
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    # both ranks' slabs are counted: 2 x 32768 instances per step
    assert abs(d["value"] - 2 * 32768 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert sum(d["iters_hist_rank0"].values()) == 32768 and d["solved_frac"] == 1.0


def test_bench_fixed_total_is_shared_out():
    """The default workload keeps 262144 instances in total: two ranks own 131072 each."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29573", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--backend", "gloo"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    d = _last_json(r.stdout.decode())
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["ranks_seen"] == 2
    assert d["config"]["instances_total"] == 262144 and d["config"]["instances_rank0"] == 131072
    # every rank runs the second leg on its share; the figures are those of the whole batch
    at = d["at_target_accuracy"]
    assert at["max_abs_vs_converged"] <= 1e-4 and at["solved_frac"] == 1.0
    assert abs(at["value"] - 262144 / (at["ms_per_step"] * 1e-3)) < 1e-6 * at["value"]


def test_bench_closed_loop_reports_failures_per_step():
    """The closed-loop workloads carry the per-step failure figures: share of vehicles whose solve
    failed, hit the iteration budget or was certified infeasible in each of the timed steps."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "2",
                        "--workload", "rollout_12state_B8192_N20", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    d = _last_json(r.stdout.decode())
    cl = d["closed_loop"]
    # the fleet runs as four sub-fleets of 2048 vehicles, each on its own handle and stream
    assert cl["steps"] == 40 and cl["vehicles_rank0"] == 8192 and d["config"]["batches_in_flight"] == 4
    assert d["config"]["instances_per_handle"] == 2048 and sum(d["iters_hist_rank0"].values()) == 3 * 8192
    for k in ("failed_frac_per_step", "max_iter_hit_frac_per_step", "infeasible_frac_per_step"):
        assert 0.0 <= cl[k]["mean"] <= cl[k]["max"] <= 1.0
    assert cl["failed_frac_per_step"]["max"] < 0.05
    assert cl["failed_frac_per_step"]["mean"] >= cl["max_iter_hit_frac_per_step"]["mean"]
    assert len(cl["failed_frac_series"]) >= 10 and cl["iters_mean_per_step"]["last"] <= cl["iters_mean_per_step"]["first"]


def test_bench_small_batches_are_kept_in_flight():
    """mpc_12state_B4096_N20 by default keeps several batches in flight (one handle, buffer set and
    stream each); --pipeline 1 is the strictly serial figure, and the latency keys are measured
    with one batch in flight either way."""
    out = {}
    for depth in (0, 1):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "2",
                            "--workload", "mpc_12state_B4096_N20", "--no-cpu-baseline", "--no-target-leg", "--pipeline", str(depth)],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
        out[depth] = _last_json(r.stdout.decode())
    a, b = out[0], out[1]
    assert a["config"]["batches_in_flight"] > 1 and b["config"]["batches_in_flight"] == 1
    assert a["iters_hist_rank0"] == b["iters_hist_rank0"] and a["solved_frac"] == b["solved_frac"]
    assert a["value"] > 1.2 * b["value"]
    assert abs(b["one_in_flight_solves_per_s"] / b["value"] - 1.0) < 0.25
