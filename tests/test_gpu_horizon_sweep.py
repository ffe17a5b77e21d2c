"""Every horizon N = 1 .. 50 and a sample of the longer ones the engine accepts (6-DOF to 81,
12-state to 60), both families, a ragged batch of 37: statuses
and iteration counts equal the CPU oracle's in the engine's mode and the solutions agree to 1e-9.
(The horizons between the benchmark's walk through every kernel variant, operator form, ring
depth and LDS layout the dispatch can choose; N = 43 and 44 once asked for more LDS than a
workgroup may have.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_every_horizon_matches_the_oracle(oracle):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the engine has no CPU path")
    import trajectory_generation_amd as tg
    B = 37
    forms = set()
    for N in range(1, 51):
        eng = tg.BatchedMPC6(tg.mpc6_params(N=N))
        x0, ref = oracle.synth_mpc6(11, N, 0.1, B)
        r = eng.solve_host(x0, ref)
        oracle.match_engine(eng)
        ro = oracle.mpc6_batch(oracle.mpc6_params(N=N), oracle.fixed_rho_settings(eng.rho, warm_start=0), x0, ref)
        np.testing.assert_array_equal(r["status"], ro["status"], err_msg="6-DOF N=%d" % N)
        np.testing.assert_array_equal(r["iters"], ro["iters"], err_msg="6-DOF N=%d" % N)
        ok = ro["status"] == 1
        assert ok.any() and np.abs(r["X"][ok] - ro["X"][ok]).max() < 1e-9 and np.abs(r["U"][ok] - ro["U"][ok]).max() < 1e-9, N
        forms.add(("6dof",) + tuple(eng.kernel_info()))

        e12 = tg.BatchedMPC12(tg.mpc12_params(N=N))
        x12, ref12 = oracle.synth_mpc12(12, N, 0.1, B)
        r12 = e12.solve_host(x12, ref12)
        oracle.match_engine(e12)
        s = {k: oracle.fixed_rho_settings(e12.rho(k), warm_start=0) for k in ("xy", "z", "yaw")}
        ro12 = oracle.mpc12_batch(oracle.mpc12_params(N=N), s, x12, ref12)
        np.testing.assert_array_equal(r12["status"], ro12["status"], err_msg="12-state N=%d" % N)
        np.testing.assert_array_equal(r12["iters"], ro12["iters"], err_msg="12-state N=%d" % N)
        ok = (ro12["status"] == 1).all(1)
        assert ok.any() and np.abs(r12["X"][ok] - ro12["X"][ok]).max() < 1e-8, N
        forms.add(("12state",) + tuple(e12.kernel_info("xy")))
    oracle.set_rho_class_scale(1.0, 1.0, 1.0, 1.0)
    assert len(forms) >= 20      # many different operator forms / MFMA counts were exercised


def test_horizons_past_50_match_the_oracle_and_the_limit_is_an_error(oracle):
    """The 160 KiB of LDS hold the iterates of 16 instances up to N = 81 (6-DOF, dense operators from
    N = 53) and N = 60 (12-state: the XY stage); one step further create() succeeds and the solve
    declines with a message instead of launching."""
    import trajectory_generation_amd as tg
    B = 37
    for N in (51, 52, 53, 63, 64, 68, 69, 81):
        eng = tg.BatchedMPC6(tg.mpc6_params(N=N))
        x0, ref = oracle.synth_mpc6(11, N, 0.1, B)
        r = eng.solve_host(x0, ref)
        oracle.match_engine(eng)
        ro = oracle.mpc6_batch(oracle.mpc6_params(N=N), oracle.fixed_rho_settings(eng.rho, warm_start=0), x0, ref)
        np.testing.assert_array_equal(r["status"], ro["status"], err_msg="6-DOF N=%d" % N)
        np.testing.assert_array_equal(r["iters"], ro["iters"], err_msg="6-DOF N=%d" % N)
        ok = ro["status"] == 1
        assert ok.any() and np.abs(r["X"][ok] - ro["X"][ok]).max() < 1e-9 and np.abs(r["U"][ok] - ro["U"][ok]).max() < 1e-9, N
    for N in (51, 56, 60):
        e12 = tg.BatchedMPC12(tg.mpc12_params(N=N))
        x12, ref12 = oracle.synth_mpc12(12, N, 0.1, B)
        r12 = e12.solve_host(x12, ref12)
        oracle.match_engine(e12)
        s = {k: oracle.fixed_rho_settings(e12.rho(k), warm_start=0) for k in ("xy", "z", "yaw")}
        ro12 = oracle.mpc12_batch(oracle.mpc12_params(N=N), s, x12, ref12)
        np.testing.assert_array_equal(r12["status"], ro12["status"], err_msg="12-state N=%d" % N)
        np.testing.assert_array_equal(r12["iters"], ro12["iters"], err_msg="12-state N=%d" % N)
        ok = (ro12["status"] == 1).all(1)
        assert ok.any() and np.abs(r12["X"][ok] - ro12["X"][ok]).max() < 1e-8, N
    oracle.set_rho_class_scale(1.0, 1.0, 1.0, 1.0)
    with pytest.raises(Exception, match="horizon too long"):
        x0, ref = oracle.synth_mpc6(11, 82, 0.1, B)
        tg.BatchedMPC6(tg.mpc6_params(N=82)).solve_host(x0, ref)
    with pytest.raises(Exception, match="horizon too long"):
        x12, ref12 = oracle.synth_mpc12(12, 61, 0.1, B)
        tg.BatchedMPC12(tg.mpc12_params(N=61)).solve_host(x12, ref12)


def test_random_model_parameters_match_the_oracle():
    """Thirty random parameter sets per family (weights over two decades, dt, limits that differ
    per axis, smoothing on and off, horizons 3 .. 50): every solve launches -- whichever kernel
    variant, constant path and operator form the parameters select -- and agrees with the CPU
    oracle in the engine's mode (tools/param_sweep.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "param_sweep.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600, cwd=root)
    out = r.stdout.decode()
    assert r.returncode == 0 and "problems: 0" in out, out[-3000:]


def test_differential_fuzz_against_the_oracle():
    """Random model parameters x solver options (tolerances, relaxation, iteration budget, check
    interval, finer checks, step-size scales, operator form) x grid sizes x inputs with a share in
    the stress geometry, both families, plus warm-started triples: statuses and iteration counts
    equal the CPU oracle's, trajectories agree to 1e-8 (tools/fuzz_vs_oracle.py, 25 draws of a fixed
    seed; the tool takes other seeds for longer runs)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_vs_oracle.py"), "4", "25"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900, cwd=root)
    out = r.stdout.decode()
    assert r.returncode == 0 and "problems: 0" in out, out[-3000:]
