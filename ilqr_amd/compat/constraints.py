"""Flat module name of the reference (cilqr/constraints.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.nodes import Constraints  # noqa: F401
