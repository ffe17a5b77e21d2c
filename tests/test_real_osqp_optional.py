"""Pins the CPU oracle's OSQP restatement against the real OSQP library where one is installed.
This image has none (no osqp.h, no libosqp): oracle/real_osqp/Makefile then builds nothing and
every test here is skipped.  On a machine with OSQP (`make -C oracle/real_osqp
OSQP_PREFIX=/path/to/install`) they compare, on the committed golden instances, status, iteration
count and solution of the restated default path with the library's."""
import glob
import os

import numpy as np
import pytest

from oracle.real_osqp import real_osqp as R

pytestmark = pytest.mark.skipif(not R.available(), reason="real OSQP is not installed (optional harness)")


def _instances(oracle, golden_dir, pattern):
    for f in sorted(glob.glob(os.path.join(golden_dir, pattern))):
        g = np.load(f)
        p = oracle.mpc6_params(N=int(g["N"]), dt=float(g["dt"]), state_weight=float(g["state_weight"]),
                               input_weight=float(g["input_weight"]), smooth_input_weight=float(g["smooth_input_weight"]),
                               enable_control_smoothing=bool(g["enable_control_smoothing"]),
                               min_altitude=float(g["min_altitude"]), max_vel=tuple(g["max_vel"]),
                               max_accel=tuple(g["max_accel"]))
        yield f, p, g


def test_restatement_matches_real_osqp_on_the_6dof_goldens(oracle, golden_dir):
    for f, p, g in _instances(oracle, golden_dir, "*6dof*.npz"):
        qp = oracle.build_qp(p)
        real = R.RealOsqp(qp["P"], qp["A"], qp["q"], qp["l"], qp["u"])
        mine = oracle.Osqp(qp["P"], qp["A"], qp["q"], qp["l"], qp["u"], oracle.default_settings())
        for b in range(min(8, g["x0"].shape[0])):
            q, l, u = oracle.qp6_vectors(p, g["x0"][b], g["ref"][b])
            real.update(q, l, u); mine.update(q=q, l=l, u=u)
            a, c = real.solve(), mine.solve()
            assert a["status"] == c["status"], (f, b)
            # same algorithm, same data: the iteration counts agree up to one check interval
            # (different LDL' orderings round differently), the iterates to well below eps
            assert abs(a["iter"] - c["iter"]) <= 25, (f, b, a["iter"], c["iter"])
            assert np.abs(a["x"] - c["x"]).max() < 1e-3, (f, b)
            # and both sit within the solver tolerance of the certified optimum
            xs = np.concatenate([g["X"][b], g["U"][b]])
            assert np.abs(a["x"] - xs).max() < 0.5
