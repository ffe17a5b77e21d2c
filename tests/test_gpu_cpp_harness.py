"""The reference's two test mains, ported to C++ with assertions (tests/cpp/*.cpp), built with g++
against include/tg_mpc_classes.hpp + the C ABI and run as they are on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["test_mpc_6dof", "test_mpc_12state"])
def test_ported_cpp_main(name):
    exe = os.path.join(BIN, name)
    assert os.path.exists(exe), "build it with `make -C tests/cpp` (done by __graft_entry__.build())"
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and (name + ": OK") in out, out[-3000:]


def test_c_header_is_plain_c_and_cpp_harness_compiles():
    """include/tg_mpc.h parses as C99; the C++ look-alike classes compile and link against the
    library without a GPU."""
    hdr = os.path.join(ROOT, "include", "tg_mpc.h")
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", hdr])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-B"],
                          stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(BIN, "test_mpc_6dof"))
