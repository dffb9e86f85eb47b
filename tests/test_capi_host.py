"""Host-side checks of the C ABI that need no GPU: the library loads, exports every
declared symbol, mirrors the reference's literals and rejects bad descriptors."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ilqr_amd import capi, problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    header = open(os.path.join(ROOT, "include", "alilqr.h")).read()
    declared = set(re.findall(r"\b(alilqr_[a-z_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.alilqr_abi_version() == capi.ABI_VERSION


@pytest.mark.parametrize("mode", ["legacy", "fast", "new_al"])
def test_default_params_are_the_reference_literals(mode):
    lib = capi.load_library()
    got = capi.Params()
    assert lib.alilqr_default_params(capi.MODES[mode], C.byref(got)) == 0
    want = capi.default_params(mode)
    for name, _ in capi.Params._fields_:
        assert getattr(got, name) == getattr(want, name), name
    if mode == "legacy":   # cilqr/ilqr.py:27,92,117,155-174
        assert (want.max_outer, want.max_inner, want.inner_tol) == (20, 20, 1e-8)
        assert (want.viol_stop, want.viol_lambda, want.mu_gain, want.quu_reg, want.init_r) == (1e-3, 1e-1, 8.0, 1e-9, 20.0)
    if mode == "fast":     # cilqr/fast_ilqr.py:91
        assert want.quu_reg == 0.0


def test_bad_descriptors_are_rejected_before_any_device_work():
    lib = capi.load_library()
    desc, x0, goals, Q, R, cons = problems.lat_legacy(4)
    null = [None] * 17

    def call(d):
        return lib.alilqr_solve_batch(C.byref(d), *null)

    assert call(desc) == capi.E_ARG                       # null pointers
    assert b"must not be null" in lib.alilqr_last_error()
    bad = capi.make_problem("lat", "legacy", 50, 4, 1.0, 0.1, v=10.0)
    bad.abi_version = 99
    assert call(bad) == capi.E_ARG
    bad = capi.make_problem("lat", "legacy", 50, 4, 1.0, 0.1, v=10.0)
    bad.c = 7
    assert call(bad) == capi.E_ARG
    bad = capi.make_problem("lat", "legacy", 0, 4, 1.0, 0.1, v=10.0)
    assert call(bad) == capi.E_ARG
    bad = capi.make_problem("lat", "legacy", 50, 4, 1.0, 0.1, v=10.0)
    bad.goal_batch = 3
    assert call(bad) == capi.E_ARG
    bad = capi.make_problem("dyn_mid", "legacy", 50, 4, 1.0, 0.1)
    assert call(bad) == capi.E_ARG
    assert lib.alilqr_workspace_bytes(C.byref(desc)) > 0
    # the two extra legacy nodes: LON (3 states, 8 rows), LAT_INTERIOR (state rows only: c = 2n)
    assert lib.alilqr_workspace_bytes(C.byref(capi.make_problem("lon", "legacy", 50, 4, 0.0, 0.1))) > 0
    ok = capi.make_problem("lat_interior", "legacy", 50, 4, 1.0, 0.1, v=10.0, k=0.3)
    assert ok.c == 8 and lib.alilqr_workspace_bytes(C.byref(ok)) > 0
    ok.c = 10
    assert lib.alilqr_workspace_bytes(C.byref(ok)) == capi.E_ARG and b"c = 2n" in lib.alilqr_last_error()
    assert lib.alilqr_workspace_bytes(C.byref(capi.make_problem("lon", "fast", 50, 4, 0.0, 0.1, c=8))) == capi.E_ARG
    empty = capi.make_problem("lat", "legacy", 50, 0, 1.0, 0.1, v=10.0)
    assert call(empty) == 0                               # an empty batch is valid and touches nothing


def test_lanes_for_answers_without_a_device():
    """The lane width the library will use is a function of the descriptor alone: small LEGACY models keep 32 lanes at
    any batch size, the other generations narrow their groups as the batch grows; a bad descriptor is an error."""
    lib = capi.load_library()
    width = lambda *a, **k: lib.alilqr_lanes_for(C.byref(capi.make_problem(*a, **k)))
    assert width("lat", "legacy", 50, 4096, 1.0, 0.1, v=10.0) == 32
    assert width("lat", "legacy", 50, 81920, 1.0, 0.1, v=10.0) == 32
    seen = [width("dyn_mid", "fast", 100, b, 3.0, 0.1, v=10.0, k=0.001, c=18) for b in (1024, 8192, 16384)]
    assert seen == [16, 8, 4]
    for b in (64, 4096, 16384, 65536, 327680):
        assert width("dyn", "legacy", 100, b, 3.0, 0.1, v=10.0, k=0.001) in (1, 4, 8, 16, 32)
        assert width("new_bicycle", "new_al", 50, b, 3.0, 0.1, v=10.0, k=0.001, c=14) in (8, 16)
    bad = capi.make_problem("lat", "legacy", 50, 4, 1.0, 0.1, v=10.0)
    bad.c = 7
    assert lib.alilqr_lanes_for(C.byref(bad)) == capi.E_ARG


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ilqr_amd import batch
    desc, x0, goals, Q, R, cons = problems.lat_legacy(2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        batch.solve_batch(desc, x0, goals, Q, R, cons)


def test_problem_generators_are_seeded():
    a = problems.lat_start_states(8, 0)
    b = problems.lat_start_states(16, 0)
    assert np.array_equal(a, b[:8])
    assert np.all(a[:, 0] == 0) and np.all(np.abs(a[:, 1]) <= 1) and np.all(np.abs(a[:, 2]) <= 0.2)
