"""The entry point that allocates nothing (tg_mpc12_solve_batch_static): through the C ABI, two
calls in a row, the second on a smaller batch, each compared bit for bit with tg_mpc12_solve_batch
on the same inputs; oversize batches are refused."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x4D5043


def _fill(buf, x0, ref):
    import torch
    B = x0.shape[0]
    buf["x0"].zero_(); buf["ref"].zero_()
    buf["x0"][:, :B] = torch.from_numpy(np.ascontiguousarray(x0.T)).cuda()
    buf["ref"][:, :B] = torch.from_numpy(np.ascontiguousarray(ref.T)).cuda()
    for k in ("X", "U"):
        buf[k].fill_(-3.0)
    buf["status"].fill_(77); buf["iters"].fill_(-1)


def test_static_entry_point_matches_the_allocating_one():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the engine has no CPU path")
    import trajectory_generation_amd as tg
    from trajectory_generation_amd import _capi as capi
    from trajectory_generation_amd import synth
    N, ld = 20, 1200
    lib = capi.load()
    assert lib.tg_mpc12_static_scratch_bytes(N, ld) == 256 + (12 * ld + 255) // 256 * 256 + 8 * (29 * N + 13) * ld
    # (the static entry point runs every instance under the handle's first operator: the comparison is
    # with the allocating entry point on a handle whose step-size schedule is off)
    one = tg.make_opts(phase_iters=(-1, -1))
    ref_eng = tg.BatchedMPC12(tg.mpc12_params(N=N), one)
    eng = tg.BatchedMPC12(tg.mpc12_params(N=N), one)
    # a handle that has never solved anything: the static entry point must not make it allocate
    fresh = tg.BatchedMPC12(tg.mpc12_params(N=N), one)
    x0, ref = synth.mpc12(SEED + 3, N, 0.1, 200, 0)
    a, b = ref_eng.alloc(200, ld=ld), fresh.alloc(200, ld=ld)
    _fill(a, x0, ref); _fill(b, x0, ref)
    ref_eng.solve(a)                                  # (the runtime loads the kernels' code at their first launch)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    fresh.solve_static(b)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, "tg_mpc12_solve_batch_static changed the free device memory"
    for k in ("status", "iters", "X", "U"):
        assert torch.equal(a[k], b[k]), k
    # first call: 1100 instances including one whose XY stage is infeasible (gated yaw stage);
    # second call, right after it: 300 other instances
    for B, first in ((1100, 1000), (300, 5000)):
        x0, ref = synth.mpc12(SEED + 3, N, 0.1, B, first)
        a, b = ref_eng.alloc(B, ld=ld), eng.alloc(B, ld=ld)
        _fill(a, x0, ref); _fill(b, x0, ref)
        ref_eng.solve(a)
        eng.solve_static(b)
        torch.cuda.synchronize()
        for k in ("status", "iters", "X", "U"):
            assert torch.equal(a[k], b[k]), (B, k)
        st = b["status"][:, :B].cpu().numpy()
        assert (st[:, :] != 77).all() and (b["status"][:, B:] == 77).all()
        if first == 1000:
            assert tuple(st[:, 92]) == (1, -3, -10)           # instance 1092
        assert ((st == 1).all(0)).mean() > 0.99
    # no status array from the caller: the gate rows come from the scratch as well
    x0, ref = synth.mpc12(SEED + 3, N, 0.1, 64, 1040)
    a, b = ref_eng.alloc(64, ld=ld), eng.alloc(64, ld=ld)
    _fill(a, x0, ref); _fill(b, x0, ref)
    ref_eng.solve(a)
    vp = lambda t: C.c_void_p(t.data_ptr())
    capi.check(lib.tg_mpc12_solve_batch_static(eng._h, 64, ld, vp(b["x0"]), vp(b["ref"]), vp(b["X"]), vp(b["U"]),
                                               None, None, None))
    torch.cuda.synchronize()
    assert torch.equal(a["X"], b["X"]) and torch.equal(a["U"], b["U"])
    # a grid smaller than the batch: the slots are refilled from the queue counters in the static array
    # (cleared by every call), bit-identical to the allocating path with the same grid
    for e in (ref_eng, eng):
        o = e.get_opts("xy")
        o.grid_cap = 8
        e.set_opts(o)
    for B, first in ((1100, 1000), (700, 3000)):
        x0, ref = synth.mpc12(SEED + 3, N, 0.1, B, first)
        a, b = ref_eng.alloc(B, ld=ld), eng.alloc(B, ld=ld)
        _fill(a, x0, ref); _fill(b, x0, ref)
        ref_eng.solve(a)
        eng.solve_static(b)
        torch.cuda.synchronize()
        for k in ("status", "iters", "X", "U"):
            assert torch.equal(a[k], b[k]), (B, k, "queue")
        assert ((b["status"][:, :B] == 1).all(0)).double().mean().item() > 0.99
    # beyond the static array: refused, nothing launched, the allocating entry point still serves it
    big_ld = 8192
    assert lib.tg_mpc12_static_scratch_bytes(N, big_ld) > 32 << 20
    c = eng.alloc(big_ld, ld=big_ld)
    with pytest.raises(capi.TgError) as e:
        eng.solve_static(c)
    assert "static scratch" in str(e.value)
    x0, ref = synth.mpc12(SEED + 3, N, 0.1, 128, 0)
    _fill(c, x0, ref)
    eng.solve(c, B=128)
    torch.cuda.synchronize()
    assert (c["status"][:, :128] == 1).all()
