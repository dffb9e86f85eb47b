"""Node and constraint classes with the reference's names, signatures and error behaviour.

They hold the problem data a solver needs (model constants, goal, weights, bounds or a constraint object,
multipliers) and validate it exactly like the reference (``ValueError`` on wrong lengths).  Node maths
(``dynamics``, ``dynamics_jacobian``, ``cost`` ...) is evaluated by the HIP node kernel
(``alilqr_linearize_batch``); there is no CPU implementation of it in this package.

Reference: cilqr/ilqr_node.py:5-95, fast_ilqr_node.py:5-78, lat_bicycle_node.py:4-180,
full_bicycle_kinematic_node.py:4-194, full_bicycle_dynamic_node.py:4-197, fast_bicycle_node.py:7-212,
constraints.py:4-119, linear_constraints.py:4-29, box_constrains.py:5-32.
"""
from abc import ABC, abstractmethod

import numpy as np

from . import capi


# ---------------------------------------------------------------------------
# constraints (cilqr/constraints.py, linear_constraints.py, box_constrains.py)
# ---------------------------------------------------------------------------
class Constraints(ABC):
    """Inequality (or equality) constraints c(x,u) <= 0 with multipliers ``lambda_`` (<= 0) and penalty ``mu``
    (cilqr/constraints.py:4-119).

    The augmented-Lagrangian calls (cost / gradient / Hessian of one point, multiplier update, violation) are evaluated
    by the HIP node kernels, which take constraint TABLES: they exist for ``LinearConstraints`` / ``BoxConstraint``
    below.  A subclass with a Python ``constrains()`` of its own cannot be evaluated on the GPU; for it these calls
    raise, as the batched solve does.
    """

    def __init__(self, constraint_dim, is_equality=False):
        self.constraint_dim = constraint_dim
        self.c = np.zeros(constraint_dim)
        self.cx = self.cu = self.hx = self.hu = self.hxu = None
        self._lambda = np.zeros(constraint_dim)
        self._mu = 1.0
        self.is_equality = is_equality
        self.lambda_proj = np.zeros(constraint_dim)
        self.proj_jac = np.zeros((constraint_dim, constraint_dim))

    @property
    def lambda_(self):
        return self._lambda

    @lambda_.setter
    def lambda_(self, value):
        self._lambda = np.array(value)

    @property
    def mu(self):
        return self._mu

    @mu.setter
    def mu(self, value):
        self._mu = value

    @abstractmethod
    def constrains(self, x, u):
        pass

    @abstractmethod
    def constrains_jacobian(self, x, u):
        pass

    @abstractmethod
    def constrains_hessian(self, x, u):
        pass

    def projection(self, x):
        return np.minimum(x, 0)

    def projection_jacobian(self, x):
        return np.diag(np.where(np.asarray(x) > 0, 0.0, 1.0))

    def projection_hessian(self, x, b):
        return np.zeros((x.size, x.size))

    def _no_table(self, *a, **k):
        raise NotImplementedError("the augmented-Lagrangian terms are evaluated on the GPU from a constraint table: "
                                  "use LinearConstraints / BoxConstraint")

    augmented_lagrangian_cost = augmented_lagrangian_jacobian = augmented_lagrangian_hessian = _no_table
    update_lambda = max_violation = _no_table

    def update_mu(self, new_mu):
        self._mu = new_mu


class LinearConstraints(Constraints):
    """c(x,u) = A x + B u + C  (cilqr/linear_constraints.py:4-29)."""

    def __init__(self, A, B, C, is_equality=False):
        A = np.asarray(A)
        super().__init__(A.shape[0], is_equality)
        self.A, self.B, self.C = A, np.asarray(B), np.asarray(C)

    def constrains(self, x, u):
        self.c = self.A @ x + self.B @ u + self.C
        return self.c

    def constrains_jacobian(self, x, u):
        self.cx, self.cu = self.A, self.B
        return self.cx, self.cu

    def constrains_hessian(self, x, u):
        x, u = np.asarray(x), np.asarray(u)
        self.hx = np.zeros((self.constraint_dim, x.size, x.size))
        self.hu = np.zeros((self.constraint_dim, u.size, u.size))
        self.hxu = np.zeros((self.constraint_dim, x.size, u.size))
        return self.hx, self.hu, self.hxu

    # -- augmented-Lagrangian calls of cilqr/constraints.py:59-119 on the GPU ------------------------------------
    # The node kernel of the FAST generation evaluates  l = e'Qe + u'Ru + AL  and its derivatives for a 6-state /
    # 2-control table (the only shape FastILQR has a node for, fast_bicycle_node.py:7-16); with Q = 0, R = 0 what
    # it returns IS the AL value, gradient and Hessian blocks.
    def _gpu_point(self, x, u):
        from . import batch
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        u = np.asarray(u, dtype=np.float64).reshape(-1)
        c = self.constraint_dim
        if self.A.shape != (c, 6) or self.B.shape != (c, 2) or x.size != 6 or u.size != 2:
            raise NotImplementedError("the GPU node kernels of this generation take 6-state / 2-control tables")
        desc = capi.make_problem("dyn_mid", "fast", 1, 1, 1.0, 0.1, c=c,
                                 cons_kind=capi.CONS_EQUALITY if self.is_equality else capi.CONS_LINEAR)
        xs = np.zeros((1, 2, 6)); xs[0, 0] = x
        us = np.zeros((1, 2, 2)); us[0, 0] = u
        ls = np.zeros((1, 2, c)); ls[0, 0] = np.asarray(self._lambda, dtype=np.float64)
        args = (desc, xs, us, ls, np.array([float(self._mu)]), np.zeros((2, 6)), np.zeros((6, 6)), np.zeros((2, 2)),
                capi.pack_linear(self.A, self.B, self.C))
        return batch, args

    def augmented_lagrangian_cost(self, x, u):
        batch, args = self._gpu_point(x, u)
        return float(batch.linearize_batch(*args)["l"][0, 0])

    def augmented_lagrangian_jacobian(self, x, u):
        batch, args = self._gpu_point(x, u)
        out = batch.linearize_batch(*args)
        return out["lx"][0, 0].cpu().numpy(), out["lu"][0, 0].cpu().numpy()

    def augmented_lagrangian_hessian(self, x, u, full_newton=False):
        """(dxdx, dudu, dxdu).  The mixed block is used by no solver generation (fast_bicycle_node.py:209 drops it)
        and the node kernel does not form it: it is assembled on the device from the kernel's constraint values
        and Jacobians."""
        batch, args = self._gpu_point(x, u)
        out = batch.linearize_batch(*args)
        desc, xs, us, ls, mu = args[0], args[1], args[2], args[3], float(self._mu)
        cb = batch.constraints_batch(desc, xs, us, args[8])
        f = out["lx"].new_tensor(ls[0, 0]) - mu * cb["cval"][0, 0]
        on = (f <= 0).to(f.dtype) if not self.is_equality else f.new_ones(f.shape)
        dxdu = mu * (on[:, None] * cb["jac_x"][0, 0]).T @ (on[:, None] * cb["jac_u"][0, 0])
        return out["lxx"][0, 0].cpu().numpy(), out["luu"][0, 0].cpu().numpy(), dxdu.cpu().numpy()

    def _aux(self, x, u):
        batch, args = self._gpu_point(x, u)
        return batch.node_aux_batch(*args)

    def max_violation(self, x, u):
        out = self._aux(x, u)
        self.c = out["cval"][0, 0].cpu().numpy()
        self._last_point = (np.array(x, dtype=np.float64), np.array(u, dtype=np.float64))
        return float(out["viol"][0, 0])

    def update_lambda(self):
        """lambda <- projection(lambda - mu c) at the point of the preceding max_violation call (constraints.py:106-110
        uses the constraint values that call cached)."""
        if getattr(self, "_last_point", None) is None:
            raise RuntimeError("update_lambda follows a max_violation call (it uses the constraint values cached there)")
        self._lambda = self._aux(*self._last_point)["lam_new"][0, 0].cpu().numpy()


class BoxConstraint(LinearConstraints):
    """state_min <= x <= state_max, control_min <= u <= control_max as a linear table
    (cilqr/box_constrains.py:5-32): rows [I; -I; 0; 0] x + [0; 0; I; -I] u + [-xmax; xmin; -umax; umin]."""

    def __init__(self, state_min, state_max, control_min, control_max):
        smin, smax, cmin, cmax = (np.asarray(a, dtype=float) for a in (state_min, state_max, control_min, control_max))
        n, m = smin.size, cmin.size
        A = np.vstack((np.eye(n), -np.eye(n), np.zeros((2 * m, n))))
        B = np.vstack((np.zeros((2 * n, m)), np.eye(m), -np.eye(m)))
        C = np.hstack((-smax, smin, -cmax, cmin))
        super().__init__(A, B, C)


# ---------------------------------------------------------------------------
# node bases
# ---------------------------------------------------------------------------
class _NodeCommon(ABC):
    MODEL = None      # key of capi.MODELS
    MODE = None       # key of capi.MODES

    def _init_common(self, state_dim, control_dim, goal):
        self._state_dim = state_dim
        self._control_dim = control_dim
        self._state = np.zeros(state_dim)
        self._control = np.zeros(control_dim)
        self._goal = np.zeros(state_dim) if goal is None else np.array(goal)

    @property
    def state_dim(self):
        return self._state_dim

    @property
    def control_dim(self):
        return self._control_dim

    @property
    def state(self):
        return self._state

    @state.setter
    def state(self, value):
        if len(value) != self._state_dim:
            raise ValueError("Invalid state dimension")
        self._state = np.array(value)

    @property
    def control(self):
        return self._control

    @control.setter
    def control(self, value):
        if len(value) != self._control_dim:
            raise ValueError("Invalid control dimension")
        self._control = np.array(value)

    @property
    def goal(self):
        return self._goal

    @goal.setter
    def goal(self, value):
        if len(value) != self._state_dim:
            raise ValueError("Invalid goal dimension")
        self._goal = np.array(value)

    # -- data the batched solver reads -------------------------------------
    def model_constants(self):
        """(L, dt, v, k) of this node."""
        return (float(self.L), float(self.dt), float(getattr(self, "v", 0.0)), float(getattr(self, "k", 0.0)))

    @abstractmethod
    def constraint_table(self):
        """Packed constraint table of this node in the layout of include/alilqr.h."""

    @abstractmethod
    def multipliers(self):
        """(lambda array, mu)."""

    @abstractmethod
    def _set_multipliers(self, lam, mu):
        pass

    # -- node maths on the GPU (one node = a batch of one, horizon 1) -------
    def _eval(self, state=None, control=None):
        from . import batch
        x = np.array(self._state if state is None else state, dtype=np.float64).reshape(-1)
        u = np.array(self._control if control is None else control, dtype=np.float64).reshape(-1)
        lam, mu = self.multipliers()
        n, m = self._state_dim, self._control_dim
        L, dt, v, k = self.model_constants()
        desc = capi.make_problem(self.MODEL, self.MODE, 1, 1, L, dt, v=v, k=k, c=len(lam),
                                 cons_kind=self.cons_kind() if hasattr(self, "cons_kind") else capi.CONS_LINEAR)
        xs = np.zeros((1, 2, n)); xs[0, 0] = x
        us = np.zeros((1, 2, m)); us[0, 0] = u
        ls = np.zeros((1, 2, len(lam))); ls[0, 0] = lam
        out = batch.linearize_batch(desc, xs, us, ls, np.array([float(mu)]), np.tile(np.asarray(self._goal, dtype=np.float64), (2, 1)),
                                    np.asarray(self.Q, dtype=np.float64), np.asarray(self.R, dtype=np.float64),
                                    self.constraint_table())
        return {key: val[0, 0].cpu().numpy() for key, val in out.items()}

    def normalize_angle(self, angle):
        """(angle + pi) % (2 pi) - pi  (cilqr/lat_bicycle_node.py:23-25)."""
        return (angle + np.pi) % (2 * np.pi) - np.pi

    def normalize_state(self):
        self.state[2] = self.normalize_angle(self.state[2])
        self.state[3] = self.normalize_angle(self.state[3])

    def dynamics(self, state, control):
        return self._eval(state, control)["f"]

    def dynamics_jacobian(self, state=None, control=None):
        out = self._eval(state, control)
        return out["fx"], out["fu"]

    def cost(self):
        self.normalize_state()
        return float(self._eval()["l"])

    def cost_jacobian(self):
        self.normalize_state()
        out = self._eval()
        return out["lx"], out["lu"]

    def cost_hessian(self):
        out = self._eval()
        return out["lxx"], out["luu"]


class ILQRNode(_NodeCommon):
    """cilqr/ilqr_node.py:5-95 — node with built-in box constraints (legacy generation)."""
    MODE = "legacy"

    def __init__(self, state_dim, control_dim, constraint_dim, goal):
        self._init_common(state_dim, control_dim, goal)
        self._constraint_dim = constraint_dim
        self._constraints = np.zeros(constraint_dim)

    @property
    def constraint_dim(self):
        return self._constraint_dim


class _LegacyBoxNode(ILQRNode):
    def _init_box(self, state_bounds, control_bounds, goal, Q, R, n, m):
        super().__init__(n, m, 2 * (n + m), goal)
        self.state_max = np.array(state_bounds[1])
        self.state_min = np.array(state_bounds[0])
        self.control_max = np.array(control_bounds[1])
        self.control_min = np.array(control_bounds[0])
        for name, arr, want in (("state_bounds", self.state_min, n), ("state_bounds", self.state_max, n),
                                ("control_bounds", self.control_min, m), ("control_bounds", self.control_max, m)):
            if arr.size != want:
                raise ValueError(f"Invalid {name} dimension")
        self.Q = np.array(Q)
        self.R = np.array(R)
        self.lambda_ = np.zeros(2 * (n + m))
        self.mu = 1.0
        self.Imu = np.zeros((2 * (n + m), 2 * (n + m)))

    def constraint_table(self):
        return capi.pack_box(self.state_min, self.state_max, self.control_min, self.control_max)

    def multipliers(self):
        return np.asarray(self.lambda_, dtype=np.float64), float(self.mu)

    def _set_multipliers(self, lam, mu):
        self.lambda_ = np.array(lam)
        self.mu = float(mu)

    def constraints(self):
        """[x - x_max ; x_min - x ; u - u_max ; u_min - u]  (cilqr/lat_bicycle_node.py:127-133) — plain data access."""
        self.normalize_state()
        return np.hstack((self.state - self.state_max, self.state_min - self.state,
                          self.control - self.control_max, self.control_min - self.control))

    def constraint_jacobian(self):
        n, m = self._state_dim, self._control_dim
        Jx = np.zeros((self._constraint_dim, n)); Ju = np.zeros((self._constraint_dim, m))
        Jx[:n] = np.eye(n); Jx[n:2 * n] = -np.eye(n)
        Ju[2 * n:2 * n + m] = np.eye(m); Ju[2 * n + m:] = -np.eye(m)
        return Jx, Ju

    def get_lambda(self):
        return self.lambda_

    def set_lambda(self, new_lambda):
        self.lambda_ = new_lambda

    def get_mu(self):
        return self.mu

    def set_mu(self, new_mu):
        self.mu = new_mu

    def update_lambda(self):
        self.lambda_ = np.maximum(0, self.lambda_ + self.mu * self.constraints())

    def update_Imu(self):
        c = self.constraints()
        d = np.where((np.asarray(self.lambda_) == 0) & (c <= 0), 0.0, self.mu)
        self.Imu = np.diag(d)

    def is_state_within_constraints(self):
        self.normalize_state()
        return not (np.any((self.state > self.state_max) | (self.state < self.state_min))
                    or np.any((self.control > self.control_max) | (self.control < self.control_min)))


class LatBicycleKinematicNode(_LegacyBoxNode):
    """cilqr/lat_bicycle_node.py:4-21 — 4 states (x, y, theta, delta), 1 control, constant speed v."""
    MODEL = "lat"

    def __init__(self, L, dt, v, state_bounds, control_bounds, goal, Q, R):
        self._init_box(state_bounds, control_bounds, goal, Q, R, 4, 1)
        self.L, self.dt, self.v = L, dt, v


class FullBicycleKinematicNode(_LegacyBoxNode):
    """cilqr/full_bicycle_kinematic_node.py:4-20 — 6 states (x, y, theta, delta, v, a), 2 controls."""
    MODEL = "kin"

    def __init__(self, L, dt, state_bounds, control_bounds, goal, Q, R):
        self._init_box(state_bounds, control_bounds, goal, Q, R, 6, 2)
        self.L, self.dt = L, dt


class FullBicycleDynamicNode(_LegacyBoxNode):
    """cilqr/full_bicycle_dynamic_node.py:4-21 — kinematic model with the (1 + k v^2) understeer factor."""
    MODEL = "dyn"

    def __init__(self, L, dt, k, state_bounds, control_bounds, goal, Q, R):
        self._init_box(state_bounds, control_bounds, goal, Q, R, 6, 2)
        self.L, self.dt, self.k = L, dt, k


class LongitudinalNode(_LegacyBoxNode):
    """cilqr/lon_node.py:4-19 — 3 states (s, v, a), jerk control; linear dynamics, no angles."""
    MODEL = "lon"

    def __init__(self, dt, state_bounds, control_bounds, goal, Q, R):
        self._init_box(state_bounds, control_bounds, goal, Q, R, 3, 1)
        self.L, self.dt = 0.0, dt

    def normalize_state(self):
        pass


class LatBicycleInteriorNode(_LegacyBoxNode):
    """cilqr/lat_bicycle_interior_node.py:4-21 — lateral model whose steering rate is control_max * tanh(u).  Only
    the states are bounded (8 rows); the state cost weighs the state itself, not its distance to the goal (:96)."""
    MODEL = "lat_interior"

    def __init__(self, L, dt, v, state_bounds, control_bounds, goal, Q, R):
        self._init_box(state_bounds, control_bounds, goal, Q, R, 4, 1)
        self.L, self.dt, self.v = L, dt, v
        self._constraint_dim = 8
        self._constraints = np.zeros(8)
        self.lambda_ = np.zeros(8)
        self.Imu = np.zeros((8, 8))

    def model_constants(self):
        # the k slot of the descriptor carries control_max[0] for this model
        return (float(self.L), float(self.dt), float(self.v), float(self.control_max[0]))

    def constraints(self):
        self.normalize_state()
        return np.hstack((self.state - self.state_max, self.state_min - self.state))

    def constraint_jacobian(self):
        n = self._state_dim
        Jx = np.vstack((np.eye(n), -np.eye(n)))
        return Jx, np.zeros((self._constraint_dim, self._control_dim))

    def is_state_within_constraints(self):
        self.normalize_state()
        return not np.any((self.state > self.state_max) | (self.state < self.state_min))


class FastILQRNode(_NodeCommon):
    """cilqr/fast_ilqr_node.py:5-78 — node with a pluggable ``Constraints`` object."""
    MODE = "fast"

    def __init__(self, state_dim, control_dim, goal, constraints):
        self._init_common(state_dim, control_dim, goal)
        self.constraints_obj = constraints

    @property
    def constraints(self):
        return self.constraints_obj.constrains(self._state, self._control)

    def constraint_table(self):
        c = self.constraints_obj
        if not isinstance(c, LinearConstraints):
            raise NotImplementedError("the GPU solve path takes linear constraint tables (LinearConstraints / BoxConstraint)")
        if c.A.shape != (c.constraint_dim, self._state_dim) or c.B.shape != (c.constraint_dim, self._control_dim):
            raise ValueError("Invalid constraint table dimension")
        return capi.pack_linear(c.A, c.B, c.C)

    def cons_kind(self):
        return capi.CONS_EQUALITY if self.constraints_obj.is_equality else capi.CONS_LINEAR

    def multipliers(self):
        return np.asarray(self.constraints_obj.lambda_, dtype=np.float64), float(self.constraints_obj.mu)

    def _set_multipliers(self, lam, mu):
        self.constraints_obj.lambda_ = np.array(lam)
        self.constraints_obj.mu = float(mu)


class FastBicycleNode(FastILQRNode):
    """cilqr/fast_bicycle_node.py:7-16 — 6-state dynamic bicycle (f_x heading row at v_mid) + constraints object."""
    MODEL = "dyn_mid"

    def __init__(self, L, dt, k, goal, Q, R, constraints):
        super().__init__(6, 2, goal, constraints)
        self.L, self.dt, self.k = L, dt, k
        self.Q = np.array(Q)
        self.R = np.array(R)
