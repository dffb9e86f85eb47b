"""Benchmark configurations 3, 4 and 5 at their full sizes on one GPU, through the C ABI.

The oracle cannot solve 10^4..10^5 problems in a test, so the full-size runs are held to properties that do not
depend on the size:
  * every problem ends in exactly one of the two stopping states and none is left unwritten;
  * a random sample of the batch, solved alone (another grid, another lane-group width, other neighbours in the
    wave), gives the same bits: a problem's result does not depend on the batch it travels in;
  * that sample matches the CPU oracle to 1e-6 relative on x, u and the final cost wherever the oracle reproduces its
    own solve from a start state moved by 1e-13 (see test_gpu_configs.py);
  * multi-start: the unperturbed start of each scenario equals the plain solve of the scenario, and the gathered
    best cost is the minimum over the scenario's starts.
"""
import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
from ilqr_amd import batch, capi, problems, sharding

pytestmark = pytest.mark.gpu
TOL = 1e-6
SAMPLE = 48


def to_np(out):
    return {k: v.cpu().numpy() for k, v in out.items()}


def subset(prob, idx, u_offset=None):
    desc = type(prob[0]).from_buffer_copy(prob[0])
    desc.batch = len(idx)
    assert desc.goal_batch == 1
    sub = (desc, np.ascontiguousarray(prob[1][idx])) + tuple(prob[2:6])
    return sub, (None if u_offset is None else np.ascontiguousarray(u_offset[idx]))


def worst_err(o, ref, i, j=None):
    return gu.solve_err(o, ref, i, j)


def check_full_batch(prob, kw, floor, u_offset=None, seed=0):
    desc = prob[0]
    B = desc.batch
    full = to_np(batch.solve_batch(*prob, extras=False, u_offset=u_offset, **kw))
    flags = full["status"][:, 2]
    done = ((flags & capi.FLAG_VIOL_OK) != 0).astype(int) + ((flags & capi.FLAG_OUTER_CAP) != 0).astype(int)
    assert (done == 1).all(), np.nonzero(done != 1)[0][:8]
    assert (full["status"][:, 0] >= 1).all() and (full["status"][:, 1] >= 1).all()
    nonfinite = (flags & capi.FLAG_NONFINITE) != 0
    assert np.array_equal(nonfinite, ~np.isfinite(full["cost"]))
    assert nonfinite.mean() < 0.01

    idx = np.sort(np.random.default_rng(seed).choice(B, SAMPLE, replace=False))
    sub, uo = subset(prob, idx, u_offset)
    alone = to_np(batch.solve_batch(*sub, extras=False, u_offset=uo, **kw))
    for k in ("x", "u", "cost", "status"):
        assert np.array_equal(alone[k], full[k][idx], equal_nan=True), k

    ref = ol.solve_batch(*sub, u_offset=uo, **kw)
    ok = gu.oracle_stable_mask(sub, ref, probes=4, seed=3, **({} if uo is None else dict(u_offset=uo)), **kw)
    assert gu.pinned(f"full_sizes/sample of {full['cost'].shape[0]} ({sub[0].model}/{sub[0].mode})", ok) >= floor
    for i in np.nonzero(ok)[0]:
        assert worst_err(alone, ref, i) < TOL, (i, idx[i], alone["status"][i], ref["status"][i])
    return full


def test_config3_kinematic_rectangles_batch_65536():
    prob, kw = problems.kin_rect(65536)
    full = check_full_batch(prob, kw, 0.8)
    # the rest stops at the outer-iteration cap (rows activated by the rectangles keep the violation above 1e-3)
    assert ((full["status"][:, 2] & capi.FLAG_VIOL_OK) != 0).mean() > 0.3


@pytest.mark.parametrize("gen", ["legacy", "fast_box_linear"])
def test_config4_dynamic_h100_batch_16384(gen):
    if gen == "legacy":
        prob, floor = problems.full_legacy("dyn", 16384, 100, 2), 0.9
    else:
        prob, floor = problems.fast_box_linear(16384), 0.85
    check_full_batch(prob, {}, floor)


def test_config5_multi_start_256_starts_1024_scenarios():
    scen, starts = 1024, 256
    desc, x0, goals, Q, R, cons, uo = problems.multi_start(scen, starts)
    prob = (desc, x0, goals, Q, R, cons)
    full = check_full_batch(prob, {}, 0.9, u_offset=uo)
    # start 0 of every scenario carries no perturbation: same bits as the plain solve of the 1024 scenarios
    base, _ = subset(prob, np.arange(0, scen * starts, starts))
    plain = to_np(batch.solve_batch(*base, extras=False))
    for k in ("x", "u", "cost", "status"):
        assert np.array_equal(plain[k], full[k][::starts], equal_nan=True), k
    bc, bi = sharding.best_start(full["cost"], starts)
    per = np.where(np.isfinite(full["cost"]), full["cost"], np.inf).reshape(scen, starts)
    assert np.array_equal(bc, per.min(axis=1)) and np.array_equal(bi, per.argmin(axis=1))
    assert (bc <= per[:, 0]).all()
