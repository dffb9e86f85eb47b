"""Flat module name of the reference (cilqr/ilqr.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.solvers import ILQR  # noqa: F401
