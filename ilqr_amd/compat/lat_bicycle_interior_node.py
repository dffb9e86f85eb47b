"""Flat module name of the reference (cilqr/lat_bicycle_interior_node.py): put this directory on sys.path to run its scripts unchanged."""
from ilqr_amd.nodes import LatBicycleInteriorNode  # noqa: F401
