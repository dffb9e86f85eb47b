"""Flat module name of the reference (cilqr/lat_bicycle_node.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.nodes import LatBicycleKinematicNode  # noqa: F401
