"""Flat module name of the reference (cilqr/fast_bicycle_node.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.nodes import FastBicycleNode, FastILQRNode, Constraints, LinearConstraints, BoxConstraint  # noqa: F401
