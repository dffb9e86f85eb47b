"""Flat module name of the reference (cilqr/box_constrains.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.nodes import BoxConstraint, LinearConstraints, Constraints  # noqa: F401
