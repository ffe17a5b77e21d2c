"""MI355X-native batched MPC engine for the hot path of mzahana/trajectory_generation:
horizon-QP assembly and the OSQP-style ADMM solve of the mpc_6dof and mpc_12state models.

The compute path is libtg_mpc.so (hand-written gfx950 HIP kernels behind a C ABI,
include/tg_mpc.h).  There is no CPU fallback: loading fails loudly when the library is absent.
"""
from . import _capi
from .engine import (BatchedMPC6, BatchedMPC12, make_opts, mpc6_params, mpc12_params,
                     phase_launches, timing_enable, timing_read)

__all__ = ["BatchedMPC6", "BatchedMPC12", "make_opts", "mpc6_params", "mpc12_params",
           "phase_launches", "timing_enable", "timing_read", "MPC", "MPC12STATE"]


def __getattr__(name):
    if name == "MPC":
        from .mpc_6dof import MPC
        return MPC
    if name == "MPC12STATE":
        from .mpc_12state import MPC12STATE
        return MPC12STATE
    raise AttributeError(name)
