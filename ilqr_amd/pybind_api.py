"""The ``ilqr_pybind`` module of the reference, on the GPU (al_ilqr_cpp/ilqr_pybind.cc:37-79).

Same class names (``Class<n>_<m>[_<c>]``), constructors and solver methods as the pybind11 module:

    cons  = BoxConstraints6_2(state_min, state_max, control_min, control_max)
    nodes = [NewBicycleNodeBoxConstraints6_2(L, dt, k, goal, Q, R, cons) for goal in goals]
    s     = NewALILQR6_2(nodes, init_state)
    s.optimize(max_outer_iter, max_inner_iter, max_violation)
    x, u  = s.get_x_list(), s.get_u_list()          # (n, H+1), (m, H): one column per node, as Eigen returns them

Constraint objects are copied into each node at construction (new_bicycle_node.h:169-170), so multipliers are
per node.  All numerics run in libalilqr_hip.so; without it or without a GPU the calls raise.
``solve_batch`` is the addition the reference lacks: the same node list solved from B start states in one launch.
"""
import copy

import numpy as np

from . import batch, capi


def _vec(a, n, what):
    a = np.asarray(a, dtype=np.float64)
    if a.size != n:
        raise TypeError(f"{what}: expected {n} values, got shape {a.shape}")     # pybind's Eigen caster raises TypeError
    return a.reshape(n).copy()


def _mat(a, r, c, what):
    a = np.asarray(a, dtype=np.float64)
    if a.size != r * c or (a.ndim == 2 and a.shape != (r, c)):
        raise TypeError(f"{what}: expected a {r}x{c} matrix, got shape {a.shape}")
    return a.reshape(r, c).copy()


PARALLEL_NUM = 5        # columns of the parallel_* calls (constraints.h:73)


class _Constraints:
    """Constraints<n, m, c>: multipliers, penalty and the live-row index (constraints_bind.h:14-31)."""
    N = M = C = 0

    def __init__(self):
        self._lambda = np.zeros(self.C)
        self._mu = 1.0
        self._index = 0          # current_constraints_index_ (constraints.h:1191)

    # `lambda` is a Python keyword: reachable as getattr(obj, "lambda") like on the pybind object, and as lambda_
    @property
    def lambda_(self):
        return self._lambda

    @lambda_.setter
    def lambda_(self, v):
        self._lambda = _vec(v, self.C, "lambda")

    def __getattr__(self, name):
        if name == "lambda":
            return self._lambda
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name == "lambda":
            self.lambda_ = value
        else:
            object.__setattr__(self, name, value)

    @property
    def mu(self):
        return self._mu

    @mu.setter
    def mu(self, v):
        self._mu = float(v)

    def update_mu(self, v):
        self._mu = float(v)

    @classmethod
    def get_state_dim(cls):
        return cls.N

    @classmethod
    def get_control_dim(cls):
        return cls.M

    @classmethod
    def get_constraint_dim(cls):
        return cls.C

    def projection(self, x):
        """min(x, 0) per entry (constraints.h:365-367): a selection, no arithmetic."""
        return np.minimum(_vec(x, self.C, "x"), 0.0)

    def projection_jacobian(self, x):
        """diag(x < 0) (constraints.h:396-404)."""
        return np.diag((_vec(x, self.C, "x") < 0).astype(np.float64))

    # ---- the table calls run on the GPU (alilqr_constraints_batch), one point or the five columns of the ladder
    def _eval(self, X, U):
        """X (P, n), U (P, m) -> dict of cval / par_cval [P, c], jac_x [P, c, n], jac_u [P, c, m]."""
        P = X.shape[0]
        desc = capi.make_problem("lat_k" if self.N == 4 else "new_bicycle", "new_al", 1, P, 1.0, 0.1, c=self.C, cons_kind=self.KIND)
        x = np.zeros((P, 2, self.N)); x[:, 0] = X
        u = np.zeros((P, 2, self.M)); u[:, 0] = U
        out = batch.constraints_batch(desc, x, u, self.table())
        return {k: v.cpu().numpy()[:, 0] for k, v in out.items()}

    def constraints(self, x, u):
        """c(x, u)  (linear_constraints.h:176-181, quadratic_constraints.h:15-27)."""
        return self._eval(_vec(x, self.N, "x")[None], _vec(u, self.M, "u")[None])["cval"][0]

    def parallel_constraints(self, x, u):
        """c of PARALLEL_NUM = 5 points at once, one per column (linear_constraints.h:218-224,
        quadratic_constraints.h:29-45)."""
        X, U = _mat(x, self.N, PARALLEL_NUM, "x"), _mat(u, self.M, PARALLEL_NUM, "u")
        return np.ascontiguousarray(self._eval(X.T, U.T)["par_cval"].T)

    def constraints_jacobian(self, x=None, u=None):
        """(c_x, c_u) at (x, u)  (linear_constraints.h:260-265: A, B; quadratic_constraints.h:47-55: A + 2 x'Q_i)."""
        x = np.zeros(self.N) if x is None else x
        u = np.zeros(self.M) if u is None else u
        o = self._eval(_vec(x, self.N, "x")[None], _vec(u, self.M, "u")[None])
        return o["jac_x"][0], o["jac_u"][0]

    def constraints_hessian(self, x=None, u=None):
        """(h_xx[c], h_uu[c], h_xu[c]): zero for linear tables (linear_constraints.h:303-328)."""
        n, m = self.N, self.M
        return [np.zeros((n, n))] * self.C, [np.zeros((m, m))] * self.C, [np.zeros((n, m))] * self.C

    def set_current_constraints_index(self, i):
        self._index = int(i)

    def get_current_constraints_index(self):
        return self._index

    KIND = capi.CONS_LINEAR

    def table(self):
        """A | B | C packed the way the C ABI takes a linear table."""
        return capi.pack_linear(self.A, self.B, self.Cv)


class _LinearConstraints(_Constraints):
    """LinearConstraints<n, m, c>(A, B, C, is_equality=False)  (linear_constraints.h:105-373)."""

    def __init__(self, A, B, C, is_equality=False):
        super().__init__()
        if is_equality:
            self.KIND = capi.CONS_LINEAR | capi.CONS_EQUALITY
        self.A = _mat(A, self.C, self.N, "A")
        self.B = _mat(B, self.C, self.M, "B")
        self.Cv = _vec(C, self.C, "C")

    def constraints_jacobian(self, x=None, u=None):
        """(A, B): the stored matrices, whatever the point (linear_constraints.h:260-265); nothing is computed."""
        return self.A.copy(), self.B.copy()


class _BoxConstraints(_LinearConstraints):
    """BoxConstraints<n, m>(state_min, state_max, control_min, control_max): rows
    [I; -I; 0; 0] x + [0; 0; I; -I] u + [-x_max; x_min; -u_max; u_min]  (box_constraints.h:35-78)."""

    def __init__(self, state_min, state_max, control_min, control_max):
        n, m = self.N, self.M
        smin, smax = _vec(state_min, n, "state_min"), _vec(state_max, n, "state_max")
        cmin, cmax = _vec(control_min, m, "control_min"), _vec(control_max, m, "control_max")
        A = np.vstack([np.eye(n), -np.eye(n), np.zeros((2 * m, n))])
        B = np.vstack([np.zeros((2 * n, m)), np.eye(m), -np.eye(m)])
        C = np.concatenate([-smax, smin, -cmax, cmin])
        super().__init__(A, B, C)


class _QuadraticConstraints(_Constraints):
    """QuadraticConstraints<n, m, c>(Q, A, B, C, is_equality=False): c_i = x'Q_i x + A_i x + B_i u + C_i
    (quadratic_constraints.h:6-98)."""
    KIND = capi.CONS_QUADRATIC

    def __init__(self, Q, A, B, C, is_equality=False):
        super().__init__()
        if is_equality:
            self.KIND = capi.CONS_QUADRATIC | capi.CONS_EQUALITY
        Q = list(Q)
        if len(Q) != self.C:
            raise TypeError(f"Q: expected {self.C} matrices, got {len(Q)}")
        self.Qc = np.stack([_mat(q, self.N, self.N, "Q[i]") for q in Q])
        self.A = _mat(A, self.C, self.N, "A")
        self.B = _mat(B, self.C, self.M, "B")
        self.Cv = _vec(C, self.C, "C")

    def table(self):
        return capi.pack_quadratic(self.Qc, self.A, self.B, self.Cv)

    def constraints_hessian(self, x=None, u=None):
        """(2 Q_i, 0, 0)  (quadratic_constraints.h:66-90); doubling is exact."""
        n, m = self.N, self.M
        return [2.0 * q for q in self.Qc], [np.zeros((m, m))] * self.C, [np.zeros((n, m))] * self.C


def _quadratic(n, m, c):
    return type(f"QuadraticConstraints{n}_{m}_{c}", (_QuadraticConstraints,), dict(N=n, M=m, C=c))


def _linear(n, m, c):
    return type(f"LinearConstraints{n}_{m}_{c}", (_LinearConstraints,), dict(N=n, M=m, C=c))


def _box(n, m):
    return type(f"BoxConstraints{n}_{m}", (_BoxConstraints,), dict(N=n, M=m, C=2 * (n + m)))


def _base(n, m, c):
    return type(f"Constraints{n}_{m}_{c}", (_Constraints,), dict(N=n, M=m, C=c))


# the abstract bases the module also exports (ilqr_pybind.cc:39-47): multipliers, penalty, projection, row index
Constraints6_2_16, Constraints6_2_5 = _base(6, 2, 16), _base(6, 2, 5)
Constraints4_1_10, Constraints4_1_6, Constraints4_1_3 = _base(4, 1, 10), _base(4, 1, 6), _base(4, 1, 3)
LinearConstraints6_2_16 = _linear(6, 2, 16)
LinearConstraints4_1_10 = _linear(4, 1, 10)
LinearConstraints4_1_6 = _linear(4, 1, 6)
BoxConstraints6_2 = _box(6, 2)
BoxConstraints4_1 = _box(4, 1)
QuadraticConstraints6_2_5 = _quadratic(6, 2, 5)
QuadraticConstraints4_1_3 = _quadratic(4, 1, 3)


class _Node:
    """NewILQRNode<n, m> with a constraint object copied in (node_bind.h:15-93)."""
    MODEL = None
    N = M = 0
    CONSTRAINTS = None

    def _init(self, L, dt, k, v, goal, Q, R, constraints):
        if not isinstance(constraints, self.CONSTRAINTS):
            raise TypeError(f"{type(self).__name__} takes {self.CONSTRAINTS.__name__}")
        self.L, self.dt, self.k, self.v = float(L), float(dt), float(k), float(v)
        self._goal = _vec(goal, self.N, "goal")
        self.Q = _mat(Q, self.N, self.N, "Q")
        self.R = _mat(R, self.M, self.M, "R")
        self.constraints_obj = copy.deepcopy(constraints)

    @classmethod
    def get_state_dim(cls):
        return cls.N

    @classmethod
    def get_control_dim(cls):
        return cls.M

    def goal(self):
        return self._goal.copy()

    def normalize_angle(self, a):
        """Identity inside (-pi, pi), C fmod outside (new_ilqr_node.h:32-37)."""
        a = float(a)
        if abs(a) < np.pi:
            return a
        return float(np.fmod(a + np.pi, 2 * np.pi) - np.pi)

    def update_mu(self, v):
        self.constraints_obj.update_mu(v)

    def reset_mu(self):
        self.constraints_obj.update_mu(1.0)

    def reset_lambda(self):
        self.constraints_obj.lambda_ = np.zeros(self.constraints_obj.C)

    # ---- node maths: one node through the parallel-in-time kernel ----
    def _eval(self, state, control):
        c = self.constraints_obj
        desc = capi.make_problem(self.MODEL, "new_al", 1, 1, self.L, self.dt, v=self.v, k=self.k, c=c.C, cons_kind=c.KIND)
        x = np.zeros((1, 2, self.N)); x[0, 0] = _vec(state, self.N, "state")
        u = np.zeros((1, 2, self.M)); u[0, 0] = _vec(control, self.M, "control")
        lam = np.zeros((1, 2, c.C)); lam[0, 0] = c.lambda_
        out = batch.linearize_batch(desc, x, u, lam, np.array([c.mu]), np.tile(self._goal, (2, 1)),
                                    self.Q[None], self.R[None], c.table()[None])
        return {k: v[0, 0].cpu().numpy() for k, v in out.items()}

    def dynamics(self, state, control):
        return self._eval(state, control)["f"]

    def dynamics_jacobian(self, state, control):
        o = self._eval(state, control)
        return o["fx"], o["fu"]

    def cost(self, state, control):
        return float(self._eval(state, control)["l"])

    def cost_jacobian(self, state, control):
        o = self._eval(state, control)
        return o["lx"], o["lu"]

    def cost_hessian(self, state, control):
        o = self._eval(state, control)
        return o["lxx"], o["luu"]

    def _aux(self, state, control):
        c = self.constraints_obj
        desc = capi.make_problem(self.MODEL, "new_al", 1, 1, self.L, self.dt, v=self.v, k=self.k, c=c.C, cons_kind=c.KIND)
        x = np.zeros((1, 2, self.N)); x[0, 0] = _vec(state, self.N, "state")
        u = np.zeros((1, 2, self.M)); u[0, 0] = _vec(control, self.M, "control")
        lam = np.zeros((1, 2, c.C)); lam[0, 0] = c.lambda_
        out = batch.node_aux_batch(desc, x, u, lam, np.array([c.mu]), np.tile(self._goal, (2, 1)),
                                   self.Q[None], self.R[None], c.table()[None])
        return {k: v[0, 0].cpu().numpy() for k, v in out.items()}

    def parallel_cost(self, state, control):
        """One value per column of the (n x 5, m x 5) arguments, as NewILQRNode::parallel_cost returns them."""
        X, U = np.asarray(state, dtype=np.float64), np.asarray(control, dtype=np.float64)
        if X.ndim != 2 or U.ndim != 2 or X.shape[0] != self.N or U.shape[0] != self.M or X.shape[1] != U.shape[1]:
            raise TypeError("parallel_cost takes (n x k) states and (m x k) controls")
        return np.array([float(self._aux(X[:, j], U[:, j])["par_cost"]) for j in range(X.shape[1])])

    def dynamics_hessian_fxx(self, state, control):
        f = self._aux(state, control)["fxx"]
        return f[0], f[1], f[2]

    def max_constraints_violation(self, state, control):
        return float(self._aux(state, control)["viol"])

    def update_lambda(self, state, control):
        self.constraints_obj.lambda_ = self._aux(state, control)["lam_new"]

    def constraints(self, state, control):
        return self._aux(state, control)["cval"]


def _bicycle_node(cons_cls):
    def __init__(self, L, dt, k, goal, Q, R, constraints):
        self._init(L, dt, k, 0.0, goal, Q, R, constraints)
    return type(f"NewBicycleNode{cons_cls.__name__}", (_Node,),
                dict(MODEL="new_bicycle", N=6, M=2, CONSTRAINTS=cons_cls, __init__=__init__))


def _lat_node(cons_cls):
    def __init__(self, L, dt, k, v, goal, Q, R, constraints):
        self._init(L, dt, k, v, goal, Q, R, constraints)
    return type(f"NewLatBicycleNode{cons_cls.__name__}", (_Node,),
                dict(MODEL="lat_k", N=4, M=1, CONSTRAINTS=cons_cls, __init__=__init__))


# the abstract node bases the module also exports (ilqr_pybind.cc:63-65): no constructor, like the pybind classes
ILQRNode6_2 = type("ILQRNode6_2", (_Node,), dict(N=6, M=2))
ILQRNode4_1 = type("ILQRNode4_1", (_Node,), dict(N=4, M=1))
NewBicycleNodeBoxConstraints6_2 = _bicycle_node(BoxConstraints6_2)
NewBicycleNodeLinearConstraints6_2_16 = _bicycle_node(LinearConstraints6_2_16)
NewLatBicycleNodeLinearConstraints4_1_6 = _lat_node(LinearConstraints4_1_6)
NewLatBicycleNodeLinearConstraints4_1_10 = _lat_node(LinearConstraints4_1_10)
NewLatBicycleNodeBoxConstraints4_1 = _lat_node(BoxConstraints4_1)
NewBicycleNodeQuadraticConstraints6_2_5 = _bicycle_node(QuadraticConstraints6_2_5)
NewLatBicycleNodeQuadraticConstraints4_1_3 = _lat_node(QuadraticConstraints4_1_3)


class _NewALILQR:
    """NewALILQR<n, m>(ilqr_nodes, init_state[, left_obs, right_obs])  (new_al_ilqr.h:52-1220)."""
    N = M = 0

    def __init__(self, ilqr_nodes, init_state, left_obs=None, right_obs=None):
        self.left_obs = list(left_obs) if left_obs is not None else []
        self.right_obs = list(right_obs) if right_obs is not None else []
        capi.pack_rectangles(self.left_obs), capi.pack_rectangles(self.right_obs)      # shape check (TypeError)
        nodes = list(ilqr_nodes)
        if len(nodes) < 2:
            raise ValueError("a trajectory needs at least two nodes")
        kind = type(nodes[0])
        if any(type(nd) is not kind for nd in nodes) or kind.N != self.N or kind.M != self.M:
            raise TypeError(f"{type(self).__name__} takes one {self.N}/{self.M} node class for the whole trajectory")
        self.ilqr_nodes = nodes
        self.horizon = len(nodes) - 1
        self.init_state = _vec(init_state, self.N, "init_state")
        self._x = np.zeros((self.N, self.horizon + 1))
        self._u = np.zeros((self.M, self.horizon))
        self.last_status = None
        self._mu_solver = 1.0        # mu_ of the solver object: survives optimize() calls (new_al_ilqr.h:1125-1130)

    def _describe(self, B, max_outer_iter, max_inner_iter, max_violation):
        nodes = self.ilqr_nodes
        nd0 = nodes[0]
        consts = (nd0.L, nd0.dt, nd0.v, nd0.k)
        if any((nd.L, nd.dt, nd.v, nd.k) != consts for nd in nodes):
            raise ValueError("L, dt, v, k must be the same on every node of a trajectory")
        Qs, Rs = [nd.Q for nd in nodes], [nd.R for nd in nodes]
        Ts = [nd.constraints_obj.table() for nd in nodes]
        if any(nd.constraints_obj.KIND != nd0.constraints_obj.KIND for nd in nodes):
            raise ValueError("one constraint kind per trajectory")
        shared = all(np.array_equal(q, Qs[0]) for q in Qs) and all(np.array_equal(r, Rs[0]) for r in Rs) \
            and all(np.array_equal(t, Ts[0]) for t in Ts)
        tn = 1 if shared else len(nodes)
        Q, R, T = (np.stack(a[:tn]) for a in (Qs, Rs, Ts))
        prm = capi.default_params("new_al")
        prm.max_outer, prm.max_inner = int(max_outer_iter), int(max_inner_iter)
        prm.viol_stop, prm.viol_lambda = float(max_violation), 5.0 * float(max_violation)   # new_al_ilqr.h:1198-1211
        desc = capi.make_problem(type(nd0).MODEL, "new_al", self.horizon, B, nd0.L, nd0.dt, v=nd0.v, k=nd0.k,
                                 c=nd0.constraints_obj.C, table_nodes=tn, params=prm, cons_kind=nd0.constraints_obj.KIND)
        goals = np.stack([nd.goal() for nd in nodes])[None]
        return desc, goals, Q, R, T

    def solve_batch(self, x0, max_outer_iter, max_inner_iter, max_violation, device="cuda:0"):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        if x0.ndim != 2 or x0.shape[1] != self.N:
            raise TypeError("x0 must be [B, n]")
        desc, goals, Q, R, T = self._describe(x0.shape[0], max_outer_iter, max_inner_iter, max_violation)
        # every node starts from the row index its constraint object holds (set_current_constraints_index, or what
        # the rectangle activation of an earlier optimize() left there)
        rows = [nd.constraints_obj.get_current_constraints_index() for nd in self.ilqr_nodes]
        rect = bool(self.left_obs or self.right_obs)
        if not rect and any(r != rows[0] for r in rows):
            raise ValueError("the constraint row index must be the same on every node of a trajectory")
        return batch.solve_batch(desc, x0, goals, Q, R, T, device=device, left_obs=self.left_obs,
                                 right_obs=self.right_obs, first_row=rows[0], first_rows=rows if rect else None,
                                 overlay=rect, mu_solver=self._mu_solver)

    def optimize(self, max_outer_iter, max_inner_iter, max_violation):
        out = {k: v.cpu().numpy() for k, v in
               self.solve_batch(self.init_state[None], max_outer_iter, max_inner_iter, max_violation).items()}
        self._x = np.ascontiguousarray(out["x"][0].T)
        self._u = np.ascontiguousarray(out["u"][0].T)
        for t, nd in enumerate(self.ilqr_nodes):
            nd.constraints_obj.lambda_ = out["lam"][0, t]
            nd.constraints_obj.mu = out["mu"][0]
        self._mu_solver = float(out["mu_solver"][0])
        if "overlay" in out:
            # UpdateConstraints mutates the nodes' own tables and row indices, and they stay that way
            # (linear_constraints.h:360-373, quadratic_constraints.h:57-64)
            c = self.ilqr_nodes[0].constraints_obj.C
            for t, nd in enumerate(self.ilqr_nodes):
                ov = out["overlay"][0, t]
                for i in np.nonzero(ov[:c])[0]:
                    nd.constraints_obj.A[i, :] = 0.0
                    nd.constraints_obj.A[i, 1] = ov[i]
                    nd.constraints_obj.Cv[i] = ov[c + i]
                nd.constraints_obj.set_current_constraints_index(int(ov[2 * c]))
        self.last_status = dict(inner=int(out["status"][0, 0]), outer=int(out["status"][0, 1]),
                                flags=int(out["status"][0, 2]), rollouts=int(out["status"][0, 3]),
                                cost=float(out["cost"][0]), violation=float(out["viol"][0]))

    def get_x_list(self):
        return self._x.copy()

    def get_u_list(self):
        return self._u.copy()


NewALILQR6_2 = type("NewALILQR6_2", (_NewALILQR,), dict(N=6, M=2))
NewALILQR4_1 = type("NewALILQR4_1", (_NewALILQR,), dict(N=4, M=1))
