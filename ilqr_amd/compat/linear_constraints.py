"""Flat module name of the reference (cilqr/linear_constraints.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.nodes import LinearConstraints, Constraints  # noqa: F401
