"""Drop-in solver classes: ``ILQR(nodes).optimize()`` / ``FastILQR(nodes).optimize()`` on the GPU.

Same call shape and return value as the reference (cilqr/ilqr.py:151-175, cilqr/fast_ilqr.py:136-160):
``optimize(max_iters=20, tol=1e-8) -> (x_init, u_init, x, u)``; after the call the nodes hold the final
``state`` / ``control`` and multipliers, as they do in the reference.  ``solve_batch`` is the extension the
reference lacks: the same node list solved from B start states at once.
"""
import numpy as np

from . import batch, capi


def _tables(nodes):
    """Collect (Q, R, constraint table) per node; collapse to one shared table when all nodes agree."""
    Qs = [np.asarray(nd.Q, dtype=np.float64) for nd in nodes]
    Rs = [np.asarray(nd.R, dtype=np.float64) for nd in nodes]
    Cs = [nd.constraint_table() for nd in nodes]
    n, m = nodes[0].state_dim, nodes[0].control_dim
    for q, r, c in zip(Qs, Rs, Cs):
        if q.shape != (n, n) or r.shape != (m, m) or c.shape != Cs[0].shape:
            raise ValueError("Invalid Q / R / constraint dimension")
    shared = all(np.array_equal(q, Qs[0]) for q in Qs) and all(np.array_equal(r, Rs[0]) for r in Rs) \
        and all(np.array_equal(c, Cs[0]) for c in Cs)
    if shared:
        return 1, Qs[0][None], Rs[0][None], Cs[0][None]
    return len(nodes), np.stack(Qs), np.stack(Rs), np.stack(Cs)


class _SolverBase:
    MODE = None

    def __init__(self, ilqr_nodes):
        self.ilqr_nodes = ilqr_nodes
        self.horizon = len(ilqr_nodes) - 1
        self.last_status = None

    def _describe(self, batch_size, max_iters, tol):
        nodes = self.ilqr_nodes
        if self.horizon < 1:
            raise ValueError("a trajectory needs at least two nodes")
        kind = type(nodes[0])
        if any(type(nd) is not kind for nd in nodes):
            raise TypeError("all nodes of a trajectory must be of the same node class")
        if nodes[0].MODE != self.MODE:
            raise TypeError(f"{type(self).__name__} takes {'ILQRNode' if self.MODE == 'legacy' else 'FastILQRNode'} nodes")
        if hasattr(nodes[0], "cons_kind") and any(nd.cons_kind() != nodes[0].cons_kind() for nd in nodes):
            raise ValueError("equality and inequality tables cannot be mixed along a trajectory")
        consts = nodes[0].model_constants()
        if any(nd.model_constants() != consts for nd in nodes):
            raise ValueError("L, dt, v, k must be the same on every node of a trajectory")
        table_nodes, Q, R, cons = _tables(nodes)
        prm = capi.default_params(self.MODE)
        prm.max_inner = int(max_iters)
        prm.inner_tol = float(tol)
        lam0, _ = nodes[0].multipliers()
        L, dt, v, k = consts
        desc = capi.make_problem(kind.MODEL, self.MODE, self.horizon, batch_size, L, dt, v=v, k=k, c=len(lam0),
                                 table_nodes=table_nodes, params=prm,
                                 cons_kind=nodes[0].cons_kind() if hasattr(nodes[0], "cons_kind") else capi.CONS_LINEAR)
        goals = np.stack([np.asarray(nd.goal, dtype=np.float64) for nd in nodes])[None]
        return desc, goals, Q, R, cons

    def solve_batch(self, x0, max_iters=20, tol=1e-8, device="cuda:0"):
        """Solve from every row of ``x0`` [B, n]; returns the dict of device tensors of ``batch.solve_batch``."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        if x0.ndim != 2 or x0.shape[1] != self.ilqr_nodes[0].state_dim:
            raise ValueError("Invalid state dimension")
        desc, goals, Q, R, cons = self._describe(x0.shape[0], max_iters, tol)
        return batch.solve_batch(desc, x0, goals, Q, R, cons, device=device)

    def optimize(self, max_iters=20, tol=1e-8):
        nodes = self.ilqr_nodes
        x0 = np.asarray(nodes[0].state, dtype=np.float64)[None]
        out = {k: v.cpu().numpy() for k, v in self.solve_batch(x0, max_iters, tol).items()}
        for t, nd in enumerate(nodes):
            nd.state = out["node_x"][0, t]
            if t < self.horizon:
                nd.control = out["node_u"][0, t]
            nd._set_multipliers(out["lam"][0, t], out["mu"][0])
        self.last_status = dict(inner=int(out["status"][0, 0]), outer=int(out["status"][0, 1]),
                                flags=int(out["status"][0, 2]), rollouts=int(out["status"][0, 3]),
                                cost=float(out["cost"][0]), violation=float(out["viol"][0]))
        return out["x_init"][0], out["u_init"][0], out["x"][0], out["u"][0]

    def compute_total_cost(self):
        """Sum of the nodes' costs at their current state / control / multipliers (cilqr/ilqr.py:52-56)."""
        total = 0
        for node in self.ilqr_nodes:
            total += node.cost()
        return total


class ILQR(_SolverBase):
    """cilqr/ilqr.py:6 — legacy generation (20 outer iterations, Quu + 1e-9 I, violation sum of 2-norms,
    lambda <- max(0, lambda + mu c), mu x 8)."""
    MODE = "legacy"


class FastILQR(_SolverBase):
    """cilqr/fast_ilqr.py:5 — projection-form AL over the nodes' ``constraints_obj`` tables."""
    MODE = "fast"
