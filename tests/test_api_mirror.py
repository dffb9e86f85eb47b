"""Host logic of the reference-shaped Python classes (no GPU): signatures, validation, data packing."""
import copy

import numpy as np
import pytest

from ilqr_amd import capi, problems
from ilqr_amd.nodes import (BoxConstraint, Constraints, FastBicycleNode, FastILQRNode, FullBicycleDynamicNode,
                            FullBicycleKinematicNode, ILQRNode, LatBicycleKinematicNode, LinearConstraints)
from ilqr_amd.solvers import ILQR, FastILQR


def lat_nodes(H=5):
    goals = problems.s_curve_goals(10.0, 0.1, H, False)
    sb = np.array([[-1000, -1000, -2 * np.pi, -10], [1000, 1000, 2 * np.pi, 10]])
    cb = np.array([[-0.1], [0.1]])
    return [LatBicycleKinematicNode(L=1, dt=0.1, v=10, state_bounds=sb, control_bounds=cb, goal=g,
                                    Q=np.diag([1e-3, 1e-1, 1e1, 1e-9]), R=np.array([[50]])) for g in goals]


def test_setters_validate_like_the_reference():
    nd = lat_nodes()[0]
    assert (nd.state_dim, nd.control_dim, nd.constraint_dim) == (4, 1, 10)
    with pytest.raises(ValueError, match="Invalid state dimension"):
        nd.state = np.zeros(3)
    with pytest.raises(ValueError, match="Invalid control dimension"):
        nd.control = np.zeros(2)
    with pytest.raises(ValueError, match="Invalid goal dimension"):
        nd.goal = np.zeros(5)
    nd.state = [1, 2, 3, 4]
    assert isinstance(nd.state, np.ndarray)
    assert issubclass(LatBicycleKinematicNode, ILQRNode) and issubclass(FastBicycleNode, FastILQRNode)
    assert issubclass(BoxConstraint, LinearConstraints) and issubclass(LinearConstraints, Constraints)


def test_legacy_node_data_helpers():
    nd = lat_nodes()[0]
    nd.state = np.array([1.0, 2.0, 4.0, 0.2])        # theta = 4 wraps to 4 - 2 pi
    nd.control = np.array([0.3])
    c = nd.constraints()
    assert abs(nd.state[2] - (4.0 - 2 * np.pi)) < 1e-12
    assert c.shape == (10,) and abs(c[8] - 0.2) < 1e-15 and abs(c[9] + 0.4) < 1e-15
    nd.set_lambda(np.zeros(10)); nd.set_mu(3.0)
    nd.update_Imu()
    assert np.array_equal(np.diag(nd.Imu) != 0, c > 0)
    nd.update_lambda()
    assert np.allclose(nd.get_lambda(), np.maximum(0, 3.0 * c))
    Jx, Ju = nd.constraint_jacobian()
    assert Jx.shape == (10, 4) and Ju.shape == (10, 1) and Ju[8, 0] == 1 and Ju[9, 0] == -1
    assert not nd.is_state_within_constraints()
    assert np.array_equal(nd.constraint_table(), capi.pack_box(nd.state_min, nd.state_max, nd.control_min, nd.control_max))


def test_constraints_objects_hold_tables_and_refuse_host_evaluation():
    box = BoxConstraint(-np.ones(3), np.ones(3), -np.ones(2) / 2, np.ones(2) / 2)
    assert box.A.shape == (10, 3) and box.B.shape == (10, 2) and np.allclose(box.C, [-1, -1, -1, -1, -1, -1, -.5, -.5, -.5, -.5])
    assert np.array_equal(box.projection(np.array([1.0, -2.0])), [0.0, -2.0])
    # a 3-state table has no GPU node kernel in this generation: the AL calls say so instead of computing on the host
    with pytest.raises(NotImplementedError):
        box.augmented_lagrangian_cost(np.zeros(3), np.zeros(2))

    class Mine(Constraints):
        def constrains(self, x, u): return x
        def constrains_jacobian(self, x, u): return None, None
        def constrains_hessian(self, x, u): return None, None, None
    with pytest.raises(NotImplementedError):
        Mine(2).max_violation(np.zeros(2), np.zeros(1))


def test_solver_describes_the_node_list():
    nodes = lat_nodes(6)
    s = ILQR(nodes)
    assert s.horizon == 6
    desc, goals, Q, R, cons = s._describe(3, 20, 1e-8)
    assert (desc.model, desc.mode, desc.horizon, desc.batch, desc.table_nodes, desc.c) == (0, 0, 6, 3, 1, 10)
    assert goals.shape == (1, 7, 4) and Q.shape == (1, 4, 4) and cons.shape == (1, 10)
    assert desc.prm.max_inner == 20 and desc.prm.quu_reg == 1e-9
    nodes[2].Q = np.diag([1.0, 1.0, 1.0, 1.0])              # one node differs -> per-node tables
    desc, goals, Q, R, cons = s._describe(1, 20, 1e-8)
    assert desc.table_nodes == 7 and Q.shape == (7, 4, 4)
    nodes[3].dt = 0.2
    with pytest.raises(ValueError, match="same on every node"):
        s._describe(1, 20, 1e-8)
    with pytest.raises(TypeError):
        FastILQR(lat_nodes(3))._describe(1, 20, 1e-8)
    with pytest.raises(ValueError, match="Invalid state dimension"):
        ILQR(lat_nodes(3)).solve_batch(np.zeros((2, 5)))


def test_fast_node_packs_its_constraint_object():
    goals = problems.s_curve_goals(10.0, 0.1, 4, True)
    box = BoxConstraint(np.array([-1000, -1000, -7, -10, -100, -10.0]), np.array([30, 10, 7, 10, 100, 10.0]),
                        np.array([-0.2, -1.0]), np.array([0.2, 1.0]))
    nodes = [FastBicycleNode(1.0, 0.1, 0.001, g, np.eye(6), np.eye(2), copy.deepcopy(box)) for g in goals]
    desc, g, Q, R, cons = FastILQR(nodes)._describe(1, 20, 1e-8)
    assert (desc.model, desc.mode, desc.c) == (3, 1, 16) and cons.shape == (1, 16 * 9)
    assert nodes[0].constraints_obj is not nodes[1].constraints_obj
    nodes[0].state = np.array([31.0, 0, 0, 0, 10, 0])
    assert nodes[0].constraints[0] == 1.0                  # x - x_max
    eq = LinearConstraints(np.zeros((1, 6)), np.zeros((1, 2)), np.zeros(1), is_equality=True)
    eq_nodes = [FastBicycleNode(1.0, 0.1, 0.0, g_, np.eye(6), np.eye(2), copy.deepcopy(eq)) for g_ in goals]
    d2 = FastILQR(eq_nodes)._describe(1, 20, 1e-8)[0]
    assert d2.cons_kind == capi.CONS_EQUALITY and desc.cons_kind == capi.CONS_LINEAR
    with pytest.raises(ValueError):
        FastILQR([eq_nodes[0]] + nodes[1:])._describe(1, 20, 1e-8)


def test_longitudinal_and_interior_nodes_host_side():
    """cilqr/lon_node.py and cilqr/lat_bicycle_interior_node.py: dimensions, constraint helpers, descriptor."""
    from ilqr_amd.nodes import LatBicycleInteriorNode, LongitudinalNode
    lon = [LongitudinalNode(dt=0.1, state_bounds=np.array([[-10, 0, -2], [10, 14, 2.0]]), control_bounds=np.array([[-3], [3.0]]),
                            goal=np.array([t, 10.0, 0]), Q=np.eye(3), R=np.eye(1)) for t in range(4)]
    nd = lon[0]
    assert (nd.state_dim, nd.control_dim, nd.constraint_dim) == (3, 1, 8)
    nd.state = np.array([11.0, 5.0, 4.0]); nd.control = np.array([-4.0])      # no angle wrap on a[2] = 4
    c = nd.constraints()
    assert np.allclose(c, [1, -9, 2, -21, -5, -6, -7, 1]) and nd.state[2] == 4.0
    assert not nd.is_state_within_constraints()
    desc = ILQR(lon)._describe(5, 20, 1e-8)[0]
    assert (desc.model, desc.n, desc.m, desc.c, desc.horizon, desc.batch) == (capi.MODELS["lon"], 3, 1, 8, 3, 5)
    with pytest.raises(ValueError, match="Invalid state dimension"):
        nd.state = np.zeros(4)

    sb = np.array([[-1000, -1000, -0.05, -0.05], [1000, 1000, 0.05, 0.05]])
    itr = [LatBicycleInteriorNode(L=1.0, dt=0.1, v=10.0, state_bounds=sb, control_bounds=np.array([[-0.3], [0.3]]),
                                  goal=np.zeros(4), Q=np.eye(4), R=np.eye(1)) for _ in range(3)]
    nd = itr[0]
    assert (nd.state_dim, nd.control_dim, nd.constraint_dim) == (4, 1, 8) and nd.lambda_.shape == (8,)
    nd.state = np.array([0.0, 0.0, 4.0, 0.1])
    c = nd.constraints()
    assert c.shape == (8,) and abs(nd.state[2] - (4.0 - 2 * np.pi)) < 1e-12 and abs(c[3] - 0.05) < 1e-15
    Jx, Ju = nd.constraint_jacobian()
    assert Jx.shape == (8, 4) and not Ju.any()
    nd.set_mu(2.0); nd.update_lambda()
    assert np.allclose(nd.get_lambda(), np.maximum(0, 2.0 * c))
    desc, goals, Q, R, cons = ILQR(itr)._describe(2, 20, 1e-8)
    assert (desc.model, desc.c, desc.k) == (capi.MODELS["lat_interior"], 8, 0.3) and cons.shape == (1, 10)


def test_flat_module_names_of_the_reference_import(monkeypatch):
    import os, sys, importlib
    compat = os.path.join(os.path.dirname(capi.__file__), "compat")
    monkeypatch.syspath_prepend(compat)
    for mod, name in (("lat_bicycle_node", "LatBicycleKinematicNode"), ("ilqr", "ILQR"), ("fast_ilqr", "FastILQR"),
                      ("box_constrains", "BoxConstraint"), ("fast_bicycle_node", "FastBicycleNode"),
                      ("lon_node", "LongitudinalNode"), ("lat_bicycle_interior_node", "LatBicycleInteriorNode")):
        sys.modules.pop(mod, None)
        m = importlib.import_module(mod)
        assert hasattr(m, name)
        sys.modules.pop(mod, None)


def test_pybind_shaped_module_surface(monkeypatch):
    """Names and constructor shapes of al_ilqr_cpp/ilqr_pybind.cc:37-79 for the linear / box families."""
    import os
    import sys
    monkeypatch.syspath_prepend(os.path.join(os.path.dirname(capi.__file__), "compat"))
    sys.modules.pop("ilqr_pybind", None)
    import ilqr_pybind as ip
    box = ip.BoxConstraints6_2([-1] * 6, [1] * 6, [-2, -3], [2, 3])
    assert (box.get_state_dim(), box.get_control_dim(), box.get_constraint_dim()) == (6, 2, 16)
    A, B = box.constraints_jacobian()
    assert np.array_equal(A[:6], np.eye(6)) and np.array_equal(A[6:12], -np.eye(6)) and not A[12:].any()
    assert np.array_equal(B[12:14], np.eye(2)) and np.array_equal(B[14:], -np.eye(2))
    assert np.array_equal(box.Cv, [-1] * 6 + [-1] * 6 + [-2, -3, -2, -3])
    assert getattr(box, "lambda").shape == (16,) and box.mu == 1.0
    setattr(box, "lambda", np.arange(16.0))
    box.update_mu(8.0)
    node = ip.NewBicycleNodeBoxConstraints6_2(3.0, 0.1, 0.001, np.zeros(6), np.eye(6), np.eye(2), box)
    # the constraint object is copied into the node (new_bicycle_node.h:169-170)
    box.update_mu(1.0)
    assert node.constraints_obj.mu == 8.0 and node.constraints_obj.lambda_[3] == 3.0
    assert node.normalize_angle(0.5) == 0.5 and abs(node.normalize_angle(4.0) - (4.0 - 2 * np.pi)) < 1e-15
    assert node.normalize_angle(-4.0) == float(np.fmod(-4.0 + np.pi, 2 * np.pi) - np.pi)   # C fmod: stays below -pi
    with pytest.raises(TypeError):
        ip.BoxConstraints6_2([-1] * 5, [1] * 6, [-2, -3], [2, 3])
    with pytest.raises(TypeError):
        ip.NewBicycleNodeBoxConstraints6_2(3.0, 0.1, 0.001, np.zeros(6), np.eye(6), np.eye(2), ip.BoxConstraints4_1([0] * 4, [1] * 4, [0], [1]))
    lin = ip.LinearConstraints4_1_10(np.zeros((10, 4)), np.zeros((10, 1)), np.zeros(10))
    lat = ip.NewLatBicycleNodeLinearConstraints4_1_10(3.0, 0.1, 0.001, 10.0, np.zeros(4), np.eye(4), np.eye(1), lin)
    s = ip.NewALILQR4_1([lat] * 5, np.zeros(4))
    assert s.get_x_list().shape == (4, 5) and s.get_u_list().shape == (1, 4)
    desc, goals, Q, R, T = s._describe(7, 50, 100, 1e-3)
    assert (desc.model, desc.mode, desc.horizon, desc.batch, desc.c, desc.table_nodes) == (capi.MODELS["lat_k"], 2, 4, 7, 10, 1)
    assert desc.prm.viol_stop == 1e-3 and desc.prm.viol_lambda == 5e-3 and desc.prm.mu_gain == 100.0
    with pytest.raises(TypeError):
        ip.NewALILQR6_2([lat] * 5, np.zeros(6))


def test_pybind_module_exports_every_name_of_the_reference_module(monkeypatch):
    """The class names bound in al_ilqr_cpp/ilqr_pybind.cc:39-79, abstract bases included, and the call surface of the
    constraint classes (constraints_bind.h:14-93)."""
    import os
    import sys
    monkeypatch.syspath_prepend(os.path.join(os.path.dirname(capi.__file__), "compat"))
    sys.modules.pop("ilqr_pybind", None)
    import ilqr_pybind as ip
    for name in ("Constraints6_2_16", "Constraints6_2_5", "Constraints4_1_10", "Constraints4_1_6", "Constraints4_1_3",
                 "LinearConstraints6_2_16", "LinearConstraints4_1_10", "LinearConstraints4_1_6", "BoxConstraints6_2",
                 "BoxConstraints4_1", "QuadraticConstraints6_2_5", "QuadraticConstraints4_1_3", "ILQRNode6_2", "ILQRNode4_1",
                 "NewBicycleNodeBoxConstraints6_2", "NewBicycleNodeQuadraticConstraints6_2_5",
                 "NewLatBicycleNodeQuadraticConstraints4_1_3", "NewLatBicycleNodeLinearConstraints4_1_6",
                 "NewALILQR6_2", "NewALILQR4_1"):
        assert hasattr(ip, name), name
    assert ip.Constraints6_2_16.get_constraint_dim() == 16 and ip.ILQRNode4_1.get_state_dim() == 4
    for cls in (ip.LinearConstraints4_1_6, ip.BoxConstraints6_2, ip.QuadraticConstraints4_1_3):
        for method in ("constraints", "parallel_constraints", "constraints_jacobian", "constraints_hessian", "projection",
                       "projection_jacobian", "update_mu", "set_current_constraints_index", "get_current_constraints_index"):
            assert callable(getattr(cls, method)), (cls.__name__, method)
    lin = ip.LinearConstraints4_1_6(np.zeros((6, 4)), np.zeros((6, 1)), np.zeros(6))
    v = np.array([1.0, -2.0, 0.0, 3.0, -0.5, -0.0])
    assert np.array_equal(lin.projection(v), [0.0, -2.0, 0.0, 0.0, -0.5, -0.0])                 # x.cwiseMin(0)
    assert np.array_equal(lin.projection_jacobian(v), np.diag([0.0, 1.0, 0.0, 0.0, 1.0, 0.0]))   # diag(x < 0)
    hxx, huu, hxu = lin.constraints_hessian(np.zeros(4), np.zeros(1))
    assert len(hxx) == 6 and hxx[0].shape == (4, 4) and huu[0].shape == (1, 1) and hxu[0].shape == (4, 1) and not hxx[0].any()
