"""Flat module name of the reference (cilqr/fast_ilqr_node.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.nodes import FastILQRNode  # noqa: F401
