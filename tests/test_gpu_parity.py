"""Parity of the HIP solve path (through the C ABI) against the CPU oracle and the golden vectors.

Tolerance (stated by the task for this fp64 path): 1e-6 relative on x, u and the
final AL cost.  An AL-iLQR solve is a long chain of accept/reject decisions; on
some start states the reference ITSELF does not reproduce its own result to
1e-6 when the start state moves by 1e-13 (its iterates pass through a chaotic
stretch).  Those solves cannot be pinned by anything; they are identified by
re-running the oracle from start states moved by ~1e-13 in three random
directions and are excluded from the 1e-6 assertion, never silently: the share
of pinned solves is asserted too (about 92 % of the 4096-problem batch).
"""
import os

import numpy as np
import pytest

import golden_util as gu
import oracle_lib as ol
from ilqr_amd import batch, capi, problems

pytestmark = pytest.mark.gpu
TOL = 1e-6
KEYS = ("x", "u", "node_x", "node_u", "x_init", "u_init")


def to_np(out):
    return {k: v.cpu().numpy() for k, v in out.items()}


def worst_err(o, ref, i):
    return gu.solve_err(o, ref, i, keys=KEYS)


def oracle_stable_mask(desc, x0, goals, Q, R, cons, ref, probes=3):
    """Solves the oracle reproduces to 1e-8 from start states moved by ~1e-13 (three random directions)."""
    rng = np.random.default_rng(1)
    scale = np.zeros(x0.shape[1]); scale[1:4] = [1e-13, 1e-13, 1e-14]
    ok = np.ones(desc.batch, dtype=bool)
    for _ in range(probes):
        bump = rng.uniform(-1, 1, size=x0.shape) * scale
        ref2 = ol.solve_batch(desc, x0 + bump, goals, Q, R, cons, extras=True)
        ok &= np.array([worst_err(ref2, ref, i) < 1e-8 for i in range(desc.batch)])
    return ok


@pytest.fixture(params=["1", "8", "32"])
def lanes(request, tuning):
    tuning(lanes=int(request.param))
    return int(request.param)


GOLDEN_SETS = ["lat_legacy", "dyn_legacy", "kin_legacy", "kin_legacy_h50", "fast_box", "fast_linear", "fast_box_linear_h100",
               "fast_equality", "lon_legacy", "interior_legacy"]
# share of a set's solves that must be held to 1e-6 at the END of the solve against the reference's own vectors
END_FLOOR = {"lat_legacy": 0.8, "dyn_legacy": 0.8, "kin_legacy": 0.9, "kin_legacy_h50": 0.7, "fast_box": 0.1, "fast_linear": 0.9,
             "fast_box_linear_h100": 0.8, "fast_equality": 0.9, "lon_legacy": 0.7, "interior_legacy": 0.7}
PROBES = 8
_ORACLE_CACHE = {}


def _oracle_side(name, prob, g):
    """Oracle results of a golden set (computed once per session): the solve, its per-iteration trace, and the mask of
    solves it reproduces to 1e-8 from PROBES start states moved by ~1e-13 in random directions."""
    if name not in _ORACLE_CACHE:
        ref = ol.solve_batch(*prob)
        cost, trials = ol.solve_batch_traced(*prob, capacity=1024)
        probe = gu.oracle_stable_mask(prob, ref, probes=PROBES, keys=("x", "u", "node_x", "node_u"))
        _ORACLE_CACHE[name] = (ref, cost, trials, probe)
    return _ORACLE_CACHE[name]


def _gpu_stopped(sub, stop, **kw):
    return to_np(batch.solve_batch(*sub, stop_after=stop, **kw))


@pytest.mark.parametrize("name", GOLDEN_SETS)
def test_golden_sets(name, lanes):
    """GPU vs the reference's own vectors and vs the oracle, every lane-group width.  EVERY solve of a set gets a 1e-6
    assertion:
      * at the END of the solve, against the reference's vectors and the oracle, when the reference reproduced the solve
        from a moved start state (golden `stable`) and the oracle reproduces it from eight more, or the GPU simply took
        the same decisions as the oracle (equal iteration, outer-iteration and roll-out counts);
      * otherwise - a solve whose end neither the reference nor the oracle reproduces: a late line search between costs
        equal to the last bit, iterates that amplify one rounding error - at the last inner iteration up to which the
        GPU's and the oracle's traces agree, on the iterate itself (fast_box, the reference's own demo weights: all but
        two or three of its 16 solves are of that kind);
      * sets that hold the reference's own mid-solve iterates (fast_box, fast_box_linear_h100): against those as well."""
    g = gu.load(name)
    prob = gu.problem_from_golden(g)
    B = prob[0].batch
    ref, or_cost, or_trials, probe = _oracle_side(name, prob, g)
    o = to_np(batch.solve_batch(*prob, trace=1024))
    same = (o["status"] == ref["status"]).all(axis=1)
    at_end = np.asarray(g["stable"], dtype=bool) & (probe | same)
    frac = gu.pinned(f"golden/{name}/lanes={lanes}: end of solve vs reference vectors", at_end)
    assert frac >= END_FLOOR[name], (name, frac)
    gold = {k: g[k] for k in KEYS + ("cost",)}
    for i in np.nonzero(at_end)[0]:
        assert worst_err(o, gold, i) < TOL, (name, i, "vs reference vectors", worst_err(o, gold, i))
        assert worst_err(o, ref, i) < TOL, (name, i, "vs oracle")
        assert gu.rel_err(o["lam"][i], ref["lam"][i]) < TOL
        assert o["status"][i, 1] == ref["status"][i, 1]
    rest = np.nonzero(~at_end)[0]
    gu.mid_solve_check(f"golden/{name}/lanes={lanes}", prob, rest, o["trials"], o["iter_cost"], o["status"][:, 0],
                       or_trials, or_cost, _gpu_stopped, ol.solve_batch)
    # the end of those solves is still a converged answer of the same problem: costs agree to 1e-3 where both are finite
    for i in rest:
        if np.isfinite(o["cost"][i]) and np.isfinite(ref["cost"][i]) and g["stable"][i]:
            assert abs(o["cost"][i] - ref["cost"][i]) <= 1e-3 * max(1.0, abs(ref["cost"][i])), (name, i, "final cost")
    if "snap_iters" in g.files:
        gu.check_iterates(g, f"{name}/lanes={lanes}", lambda stop: _gpu_stopped(prob, stop), o["trials"], o["iter_cost"], "gpu")


def test_node_evaluation_kernel_matches_reference_vectors():
    """alilqr_linearize_batch vs the per-call known answers recorded from the reference nodes."""
    k = gu.load("nodes_kat")
    for tag, model, mode in (("legacy_lat", "lat", "legacy"), ("legacy_kin", "kin", "legacy"),
                             ("legacy_dyn", "dyn", "legacy"), ("fast_box", "dyn_mid", "fast"),
                             ("fast_linear", "dyn_mid", "fast")):
        pre = tag + "_"
        X, U, LAM, MU = k[pre + "X"], k[pre + "U"], k[pre + "LAM"], k[pre + "MU"]
        N = X.shape[0]
        if mode == "legacy":
            cons = capi.pack_box(k[pre + "state_min"], k[pre + "state_max"], k[pre + "control_min"], k[pre + "control_max"])
            c, v = None, float(k[pre + "v"])
        else:
            cons = capi.pack_linear(k[pre + "A"], k[pre + "B"], k[pre + "C"])
            c, v = k[pre + "A"].shape[0], 0.0
        desc = capi.make_problem(model, mode, 1, N, float(k[pre + "L"]), float(k[pre + "dt"]), v=v, k=float(k[pre + "k"]), c=c)
        n, m = desc.n, desc.m
        x = np.zeros((N, 2, n)); x[:, 0] = X
        u = np.zeros((N, 2, m)); u[:, 0] = U
        lam = np.zeros((N, 2, desc.c)); lam[:, 0] = LAM
        out = to_np(batch.linearize_batch(desc, x, u, lam, MU, np.tile(k[pre + "goal"], (2, 1)), k[pre + "Q"], k[pre + "R"], cons))
        for key in ("f", "fx", "fu", "l", "lx", "lu", "lxx", "luu"):
            assert gu.rel_err(out[key][:, 0], k[pre + key]) < 1e-12, (tag, key)


def test_full_lat_batch_against_oracle():
    """BASELINE config 2: 4096 lat problems, rng(0) start states, every solve compared with the oracle."""
    B = 4096
    desc, x0, goals, Q, R, cons = problems.lat_legacy(B, 50, seed=0)
    ref = ol.solve_batch(desc, x0, goals, Q, R, cons)
    stable = oracle_stable_mask(desc, x0, goals, Q, R, cons, ref)
    o = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    err = np.array([worst_err(o, ref, i) for i in range(B)])
    pinned = gu.pinned("parity/config 2: 4096 lateral problems", stable)
    print(f"pinned (reference-reproducible) solves: {pinned:.3f}; of those within 1e-6: {(err[stable] < TOL).mean():.4f}; "
          f"all solves within 1e-6: {(err < TOL).mean():.4f}")
    assert pinned > 0.80
    assert np.all(err[stable] < TOL), np.nonzero(stable & ~(err < TOL))[0][:10]
    # decision sequences: the same number of backward passes and roll-outs on (nearly) every pinned solve; a
    # last-ulp difference of the device sin/cos/tan can add or drop an iteration that changes nothing above 1e-6
    same = (o["status"][stable, 0] == ref["status"][stable, 0]) & (o["status"][stable, 3] == ref["status"][stable, 3])
    print(f"identical iteration / roll-out counts on {same.mean():.4f} of the pinned solves")
    assert same.mean() > 0.90


def test_size_independent_properties_at_full_size():
    """No oracle at this size: determinism, order independence, dynamic feasibility, cost consistency."""
    B = 4096
    desc, x0, goals, Q, R, cons = problems.lat_legacy(B, 50, seed=11)
    a = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    b = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k                    # run-to-run bit identical
    perm = np.random.default_rng(5).permutation(B)
    c = to_np(batch.solve_batch(desc, x0[perm], goals, Q, R, cons))
    for k in a:
        assert np.array_equal(a[k][perm], c[k], equal_nan=True), k              # problem order / group assignment irrelevant
    # the node trajectory is a roll-out of its own controls (x_{t+1} = f(x_t, u_t)), checked with the node kernel
    H, n, m = desc.horizon, desc.n, desc.m
    u_pad = np.concatenate([a["node_u"], np.zeros((B, 1, m))], axis=1)
    lin = to_np(batch.linearize_batch(desc, a["node_x"], u_pad, a["lam"], a["mu"], goals, Q, R, cons))
    assert np.max(np.abs(lin["f"][:, :H] - a["node_x"][:, 1:])) < 1e-12
    # reported cost = sum of node costs at the reported trajectory and multipliers
    assert np.allclose(lin["l"].sum(axis=1), a["cost"], rtol=1e-12, atol=0)
    assert np.all(a["status"][:, 0] >= 1) and np.all(a["status"][:, 1] <= desc.prm.max_outer)


def test_lane_group_width_does_not_change_results(tuning):
    desc, x0, goals, Q, R, cons = problems.lat_legacy(257, 50, seed=3)
    outs = []
    for g in ("1", "4", "16", "32"):
        tuning(lanes=int(g))
        outs.append(to_np(batch.solve_batch(desc, x0, goals, Q, R, cons)))
    for o in outs[1:]:
        for k in o:
            assert np.array_equal(o[k], outs[0][k], equal_nan=True), k


@pytest.mark.parametrize("B", [0, 1, 3, 65])
def test_empty_and_ragged_batches(B, lanes):
    desc, x0, goals, Q, R, cons = problems.lat_legacy(B, 50, seed=2)
    o = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    assert o["x"].shape == (B, 51, 4)
    if B:
        ref = ol.solve_batch(desc, x0, goals, Q, R, cons)
        stable = oracle_stable_mask(desc, x0, goals, Q, R, cons, ref)
        gu.pinned(f"parity/ragged B={B}/lanes={lanes}", stable)
        for i in np.nonzero(stable)[0]:
            assert worst_err(o, ref, i) < TOL


def test_persistent_groups_drain_a_batch_larger_than_the_grid(tuning):
    """Fewer resident groups than problems: every problem is still solved exactly once."""
    tuning(lanes=32, max_waves=8)                     # 16 groups for 100 problems
    desc, x0, goals, Q, R, cons = problems.lat_legacy(100, 50, seed=9)
    o = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    tuning(lanes=32)
    full = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    for k in o:
        assert np.array_equal(o[k], full[k], equal_nan=True), k


def test_per_problem_goals_and_per_node_tables():
    """goal_batch = B and table_nodes = H+1 take the same path as the shared tables when the rows repeat."""
    B, H = 9, 50
    desc, x0, goals, Q, R, cons = problems.lat_legacy(B, H, seed=4)
    base = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    d2 = capi.make_problem("lat", "legacy", H, B, 1.0, 0.1, v=10.0, goal_batch=B, table_nodes=H + 1)
    o = to_np(batch.solve_batch(d2, x0, np.tile(goals, (B, 1, 1)), np.tile(Q, (H + 1, 1, 1)), np.tile(R, (H + 1, 1, 1)),
                                np.tile(cons, (H + 1, 1))))
    for k in base:
        assert np.array_equal(base[k], o[k], equal_nan=True), k
    # a different goal list per problem really is used: shift problem 3's lateral goal
    g3 = np.tile(goals, (B, 1, 1)); g3[3, :, 1] += 0.5
    o3 = to_np(batch.solve_batch(d2, x0, g3, np.tile(Q, (H + 1, 1, 1)), np.tile(R, (H + 1, 1, 1)), np.tile(cons, (H + 1, 1))))
    ref3 = ol.solve_batch(d2, x0, g3, np.tile(Q, (H + 1, 1, 1)), np.tile(R, (H + 1, 1, 1)), np.tile(cons, (H + 1, 1)))
    assert not np.allclose(o3["x"][3], base["x"][3])
    assert worst_err(o3, ref3, 3) < TOL or not oracle_stable_mask(d2, x0, g3, np.tile(Q, (H + 1, 1, 1)), np.tile(R, (H + 1, 1, 1)), np.tile(cons, (H + 1, 1)), ref3)[3]
    assert np.array_equal(o3["x"][0], base["x"][0])


def test_non_finite_start_state_is_flagged_not_hidden():
    desc, x0, goals, Q, R, cons = problems.lat_legacy(4, 50, seed=6)
    x0[2, 1] = np.nan
    o = to_np(batch.solve_batch(desc, x0, goals, Q, R, cons))
    assert o["status"][2, 2] & capi.FLAG_NONFINITE
    assert not (o["status"][0, 2] & capi.FLAG_NONFINITE) and np.isfinite(o["cost"][[0, 1, 3]]).all()


def test_extension_is_the_code_that_ran():
    """The HIP library is loaded in-tree; nothing else can have produced the results above."""
    lib = capi.load_library()
    assert os.path.samefile(capi.LIB_PATH, os.path.join(os.path.dirname(batch.__file__), "csrc", "libalilqr_hip.so"))
    assert lib.alilqr_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libalilqr_hip.so" in maps


def test_solves_on_different_streams_overlap_and_match():
    """Each stream has its own scratch in the library: batches enqueued on three streams at once (the way bench.py
    keeps the machine full) give exactly the results of the same batches solved one after the other."""
    import torch
    probs = [problems.lat_legacy(256, 50, seed=s) for s in (21, 22, 23)]
    serial = [to_np(batch.solve_batch(*p)) for p in probs]
    solvers = [batch.BatchSolver(*p, extras=True) for p in probs]
    streams = [torch.cuda.Stream() for _ in probs]
    torch.cuda.synchronize()
    outs = []
    for rep in range(2):
        for sv, st in zip(solvers, streams):
            with torch.cuda.stream(st):
                outs.append(sv.run(st.cuda_stream))
    torch.cuda.synchronize()
    for sv, ref in zip(solvers, serial):
        o = to_np(sv.out)
        for k in ("x", "u", "cost", "status"):
            assert np.array_equal(o[k], ref[k]), k


@pytest.mark.parametrize("tag,model,mode", [("legacy_lat", "lat", "legacy"), ("legacy_dyn", "dyn", "legacy"),
                                             ("fast_box", "dyn_mid", "fast"), ("fast_linear", "dyn_mid", "fast"),
                                             ("fast_equality", "dyn_mid", "fast"),
                                             ("legacy_lon", "lon", "legacy"), ("legacy_lat_interior", "lat_interior", "legacy")])
def test_node_aux_kernel_matches_reference_vectors(tag, model, mode):
    """alilqr_node_aux_batch: constraint values, one multiplier update and the violation term against the
    reference's golden points (and the oracle for the legacy violation term, which the goldens do not hold)."""
    k = gu.load("nodes_kat_extra" if model in ("lon", "lat_interior") else "nodes_kat")
    pre = tag + "_"
    X, U, LAM, MU = k[pre + "X"], k[pre + "U"], k[pre + "LAM"], k[pre + "MU"]
    N = X.shape[0]
    if mode == "legacy":
        cons = capi.pack_box(k[pre + "state_min"], k[pre + "state_max"], k[pre + "control_min"], k[pre + "control_max"])
        kk = float(k[pre + "control_max"][0]) if model == "lat_interior" else float(k[pre + "k"])
        desc = capi.make_problem(model, mode, 1, N, float(k[pre + "L"]), float(k[pre + "dt"]), v=float(k[pre + "v"]), k=kk)
    else:
        cons = capi.pack_linear(k[pre + "A"], k[pre + "B"], k[pre + "C"])
        desc = capi.make_problem(model, mode, 1, N, float(k[pre + "L"]), float(k[pre + "dt"]), k=float(k[pre + "k"]),
                                 c=k[pre + "A"].shape[0],
                                 cons_kind=capi.CONS_EQUALITY if tag == "fast_equality" else capi.CONS_LINEAR)
    n, m = desc.n, desc.m
    goals = np.tile(k[pre + "goal"], (2, 1))
    x = np.zeros((N, 2, n)); x[:, 0] = X
    u = np.zeros((N, 2, m)); u[:, 0] = U
    lam = np.zeros((N, 2, desc.c)); lam[:, 0] = LAM
    lin = to_np(batch.linearize_batch(desc, x, u, lam, MU, goals, k[pre + "Q"], k[pre + "R"], cons))
    for key in ("f", "fx", "fu", "l", "lx", "lu", "lxx", "luu"):
        assert gu.rel_err(lin[key][:, 0], k[pre + key]) < 1e-11, (tag, key)
    o = to_np(batch.node_aux_batch(desc, x, u, lam, MU, goals, k[pre + "Q"], k[pre + "R"], cons))
    assert gu.rel_err(o["cval"][:, 0], k[pre + "cons"]) < 1e-12
    assert gu.rel_err(o["lam_new"][:, 0], k[pre + "lam_new"]) < 1e-12
    ref = ol.node_aux_batch(desc, x, u, lam, MU, goals, k[pre + "Q"], k[pre + "R"], cons)
    assert gu.rel_err(o["viol"][:, 0], ref["viol"][:, 0]) < 1e-12
    if mode == "fast":
        assert gu.rel_err(o["viol"][:, 0], k[pre + "viol"]) < 1e-12


def test_lds_resident_linearisation_records_do_not_change_bits(tuning):
    """The 32-lane LEGACY kernels keep their linearisation records in LDS where they fit; the no_lds tuning override forces
    the HBM path.  Same arithmetic, same bits."""
    outs = []
    # lateral: records + accumulated controls in LDS; dynamic H = 60: records; dynamic H = 100: packed records
    for prob in (problems.lat_legacy(96, 50, seed=31), problems.full_legacy("dyn", 48, 60, 32), problems.full_legacy("dyn", 24, 100, 2)):
        tuning(no_lds=0)
        a = to_np(batch.solve_batch(*prob))
        tuning(no_lds=1)
        b = to_np(batch.solve_batch(*prob))
        tuning(no_lds=0)
        for k in ("x", "u", "cost", "status", "lam"):
            assert np.array_equal(a[k], b[k], equal_nan=True), k
