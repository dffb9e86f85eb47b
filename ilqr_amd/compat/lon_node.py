"""Flat module name of the reference (cilqr/lon_node.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.nodes import LongitudinalNode  # noqa: F401
