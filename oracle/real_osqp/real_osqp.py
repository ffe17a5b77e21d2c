"""ctypes view of oracle/_ref/libreal_osqp.so, the optional harness around the real OSQP library
(test infrastructure).  available() is False wherever OSQP is not installed -- this image
included -- and the tests that use it are skipped there."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "_ref", "libreal_osqp.so")
_lib = None


def build():
    """Runs the Makefile, which builds only when osqp.h is found."""
    subprocess.run(["make", "-s", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
    return os.path.exists(LIB)


def available():
    return os.path.exists(LIB)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.real_osqp_setup.restype = C.c_void_p
        L.real_osqp_setup.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, C.c_double, C.c_double, C.c_int]
        L.real_osqp_update.argtypes = [C.c_void_p, dp, dp, dp]
        L.real_osqp_solve.argtypes = [C.c_void_p, dp, dp, ip, dp, dp]
        L.real_osqp_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class RealOsqp:
    def __init__(self, P, A, q, l, u, eps_abs=0.0, eps_rel=0.0, max_iter=0):
        self.n, self.m = P.shape[0], A.shape[0]
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self._h = lib().real_osqp_setup(self.n, self.m, _p(c(P)), _p(c(A)), _p(c(q)), _p(c(l)), _p(c(u)),
                                        eps_abs, eps_rel, max_iter)
        if not self._h:
            raise RuntimeError("osqp_setup failed")

    def update(self, q, l, u):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        if lib().real_osqp_update(self._h, _p(c(q)), _p(c(l)), _p(c(u))):
            raise ValueError("osqp update failed")

    def solve(self):
        x = np.zeros(self.n); y = np.zeros(self.m)
        it = C.c_int(); pri = C.c_double(); dua = C.c_double()
        st = lib().real_osqp_solve(self._h, _p(x), _p(y), C.byref(it), C.byref(pri), C.byref(dua))
        return dict(status=st, x=x, y=y, iter=it.value, pri_res=pri.value, dua_res=dua.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().real_osqp_free(self._h)
            self._h = None
