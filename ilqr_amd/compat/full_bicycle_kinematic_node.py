"""Flat module name of the reference (cilqr/full_bicycle_kinematic_node.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.nodes import FullBicycleKinematicNode  # noqa: F401
