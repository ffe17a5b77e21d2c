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
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    d = _last_json(r.stdout.decode())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["workload"] == "mpc_6dof_B4096_N20" and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0 < rf["frac"] < 1 and d["solved_frac"] == 1.0 and d["value"] > 1e6
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "solves/s"


def test_bench_two_ranks_shard_the_instances():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10",
           "--warmup", "2", "--backend", "gloo"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    d = _last_json(r.stdout.decode())
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    # both ranks' slabs are counted: 2 x 4096 instances per step
    assert abs(d["value"] - 2 * 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert sum(d["iters_hist"].values()) == 4096 and d["solved_frac"] == 1.0
