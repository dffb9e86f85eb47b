"""Flat module name of the reference (cilqr/fast_ilqr.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.solvers import FastILQR  # noqa: F401
