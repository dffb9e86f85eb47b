"""GPU parity on the remaining benchmark configurations (kinematic + static linear table, dynamic model at
H = 100 in both Python generations, multi-start), each against the CPU oracle through the C ABI.

Tolerance 1e-6 relative on x, u and the final cost, asserted on the solves the oracle itself reproduces from a start
state moved by 1e-13 (the rest amplify one rounding error beyond any fixed tolerance; their fraction is bounded
below per configuration).  NEW_AL cases are parity-unpinned against the C++ reference (see test_gpu_new_al.py).
"""
import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
from ilqr_amd import batch, capi, problems, sharding

pytestmark = pytest.mark.gpu
TOL = 1e-6


def to_np(out):
    return {k: v.cpu().numpy() for k, v in out.items()}


def worst_err(o, ref, i):
    return gu.solve_err(o, ref, i)


def stable_mask(prob, ref, u_offset=None, probes=4):
    """Solves the oracle reproduces to 1e-8 from `probes` start states moved by ~1e-13."""
    kw = {} if u_offset is None else dict(u_offset=u_offset)
    return gu.oracle_stable_mask(prob, ref, probes=probes, seed=4, **kw)


CASES = {
    # name: (generator, fraction of solves that must be pinned, compare iteration counts)
    "kin_linear_new_al": (lambda: problems.kin_linear(48), 0.6, True),
    "dyn_legacy_h100": (lambda: problems.full_legacy("dyn", 32, 100, 2), 0.9, True),
    # FastILQR stops its inner loop on |dJ| < 1e-8, i.e. on rounding noise of the cost: results are pinned,
    # iteration counts are not.  Benchmark configuration 4b is the H = 100 case (converging weights, see
    # problems.fast_box_linear; reference vectors in tests/golden/fast_box_linear_h100.npz).
    "fast_box_plus_linear_h50": (lambda: problems.fast_box_linear(32, H=50), 0.8, False),
    "fast_box_plus_linear_h100": (lambda: problems.fast_box_linear(32), 0.75, False),
    # PROPERTY TEST, not parity: the shipped demo weights (Q * 1e3, L = 1) at H = 100, on which nothing - the reference
    # included - reproduces a solve from a start state moved by 1e-13; held to the stopping state and the converged cost
    "fast_box_plus_linear_h100_stiff_weights": (lambda: problems.fast_box_linear(32, stiff=True), 0.0, False),
}


@pytest.mark.parametrize("name", list(CASES))
def test_configuration_matches_oracle(name):
    gen, floor, counts = CASES[name]
    prob = gen()
    ref = ol.solve_batch(*prob)
    o = to_np(batch.solve_batch(*prob))
    ok = stable_mask(prob, ref)
    assert gu.pinned(f"configs/{name}", ok) >= floor, name
    for i in np.nonzero(ok)[0]:
        assert worst_err(o, ref, i) < TOL, (name, i, o["status"][i], ref["status"][i])
    assert gu.rel_err(o["x_init"], ref["x_init"]) < 1e-12
    if counts:
        assert (o["status"][ok] == ref["status"][ok]).all(axis=1).mean() > 0.9
    # every solve, pinned or not: same stopping state, and converged costs that agree to 1e-3
    conv = ((o["status"][:, 2] & capi.FLAG_VIOL_OK) != 0) & ((ref["status"][:, 2] & capi.FLAG_VIOL_OK) != 0)
    assert conv.mean() > 0.8
    rel = np.abs(o["cost"][conv] - ref["cost"][conv]) / np.maximum(1.0, np.abs(ref["cost"][conv]))
    assert np.mean(rel < 1e-3) > 0.8, np.sort(rel)[-5:]


@pytest.mark.parametrize("mode", ["new_al", "legacy"])
def test_multi_start_matches_oracle_and_picks_the_same_winner(mode):
    scen, starts, H = 6, 16, 50
    if mode == "new_al":
        desc, x0, goals, Q, R, cons, uo = problems.multi_start(scen, starts, H=H)
    else:
        desc, _, goals, Q, R, cons = problems.lat_legacy(scen * starts, H=H)
        _, x0, _, _, _, _, uo = problems.multi_start(scen, starts, H=H)
    prob = (desc, x0, goals, Q, R, cons)
    ref = ol.solve_batch(*prob, u_offset=uo)
    o = to_np(batch.solve_batch(*prob, u_offset=uo))
    assert gu.rel_err(o["u_init"], ref["u_init"]) < 1e-12
    # start 0 of every scenario is the unperturbed problem
    base = to_np(batch.solve_batch(*prob))
    assert np.array_equal(o["x"][::starts], base["x"][::starts])
    assert np.abs(o["u_init"][1] - o["u_init"][0]).max() > 1e-3
    ok = stable_mask(prob, ref, u_offset=uo)
    assert gu.pinned(f"configs/multi_start/{mode}", ok) >= 0.9
    for i in np.nonzero(ok)[0]:
        assert worst_err(o, ref, i) < TOL, (mode, i)
    # arg-min over the starts of a scenario: same best cost where every start of the scenario is pinned
    bc, bi = sharding.best_start(o["cost"], starts)
    rc, ri = sharding.best_start(ref["cost"], starts)
    for s in range(scen):
        if ok[s * starts:(s + 1) * starts].all():
            assert abs(bc[s] - rc[s]) <= TOL * max(1.0, abs(rc[s]))
            second = np.partition(ref["cost"][s * starts:(s + 1) * starts], 1)[1]
            if second - rc[s] > 1e-5 * max(1.0, abs(rc[s])):
                assert bi[s] == ri[s]


@pytest.mark.parametrize("name", ["lat_rect_demo", "kin_rect_config3"])
@pytest.mark.parametrize("lanes", ["auto", "1", "8"])
def test_rectangle_obstacle_activation_matches_oracle(name, lanes, tuning):
    """Row activation by rectangle obstacles (new_al_ilqr.h:525-625): the reference's lateral demo and benchmark
    configuration 3 (kinematic model + linear table + rectangles)."""
    if lanes != "auto":
        tuning(lanes=int(lanes))
    prob, kw = (problems.new_al_lat_rect(32) if name == "lat_rect_demo" else problems.kin_rect(32))
    ref = ol.solve_batch(*prob, **kw)
    o = to_np(batch.solve_batch(*prob, **kw))
    desc, x0 = prob[0], prob[1]
    bump = np.zeros_like(x0); bump[:, 1] = 1e-13
    ref2 = ol.solve_batch(desc, x0 + bump, *prob[2:6], **kw)
    ok = np.array([worst_err(ref2, ref, i) < 1e-8 for i in range(desc.batch)])
    assert gu.pinned(f"configs/rect/{name}/lanes={lanes}", ok) >= 0.7
    for i in np.nonzero(ok)[0]:
        assert worst_err(o, ref, i) < TOL, (name, i, o["status"][i], ref["status"][i])
    assert (o["status"][ok] == ref["status"][ok]).all(axis=1).mean() > 0.9
    assert gu.rel_err(o["lam"][ok], ref["lam"][ok]) < 1e-5
    # the obstacles did something: the solve without them differs
    free = to_np(batch.solve_batch(*prob))
    assert np.abs(free["x"] - o["x"]).max() > 1.0
    # rectangles nobody enters: bit-identical to the plain kernel variant
    far = [np.array([[1032.0, 1032, 1028, 1028], [13, 15, 15, 13]])]
    same = to_np(batch.solve_batch(*prob, left_obs=far, first_row=kw["first_row"]))
    for k in ("x", "u", "cost", "status"):
        assert np.array_equal(same[k], free[k]), k


@pytest.mark.parametrize("name", ["fast_box", "fast_box_linear", "new_al_box6", "new_al_box6_active", "new_al_lat4", "kin_linear",
                                  "lat_rect", "kin_rect", "fast_equality"])
def test_single_entry_row_codes_do_not_change_bits(name, tuning):
    """Box tables are evaluated row by row from the bound's column, and rows of a NEW_AL table without A (B) entries
    from one sum of zero products per node, instead of the dense products.  Same bits as the dense rows (the dense_rows
    tuning override), diverging solves included."""
    kw = {}
    if name == "fast_box":
        prob = problems.fast_box(24)
    elif name == "fast_box_linear":
        prob = problems.fast_box_linear(12)
    elif name == "new_al_box6":
        prob = problems.new_al_box(24)
    elif name == "new_al_box6_active":
        prob = problems.new_al_box(24, y_max=20.0)
    elif name == "new_al_lat4":
        prob = problems.new_al_lat(24)
    elif name == "kin_linear":
        prob = problems.kin_linear(24)          # a table with dense rows: those keep the dense code
    elif name == "lat_rect":
        prob, kw = problems.new_al_lat_rect(16)
    elif name == "kin_rect":
        prob, kw = problems.kin_rect(16)
    else:
        g = gu.load("fast_equality")
        prob = gu.problem_from_golden(g)
    x0 = np.array(prob[1], dtype=np.float64)
    x0[0, 1] = np.inf if name in ("fast_box", "new_al_box6", "new_al_lat4", "kin_rect") else x0[0, 1]   # a non-finite point: the dense code / NaN sums
    prob = (prob[0], x0) + tuple(prob[2:])
    tuning(dense_rows=1)
    dense = to_np(batch.solve_batch(*prob, **kw))
    tuning(dense_rows=0)
    coded = to_np(batch.solve_batch(*prob, **kw))
    for k in dense:
        assert np.array_equal(dense[k], coded[k], equal_nan=True), (name, k)
