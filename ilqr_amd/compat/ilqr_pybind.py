"""Module name of the reference's pybind11 extension (al_ilqr_cpp/ilqr_pybind.cc:37): put this directory on
sys.path and ``import ilqr_pybind`` resolves to the GPU implementation."""
from ilqr_amd.pybind_api import *  # noqa: F401,F403
from ilqr_amd.pybind_api import (BoxConstraints4_1, BoxConstraints6_2, LinearConstraints4_1_6,  # noqa: F401
                                 LinearConstraints4_1_10, LinearConstraints6_2_16, NewALILQR4_1, NewALILQR6_2,
                                 NewBicycleNodeBoxConstraints6_2, NewBicycleNodeLinearConstraints6_2_16,
                                 NewLatBicycleNodeBoxConstraints4_1, NewLatBicycleNodeLinearConstraints4_1_6,
                                 NewLatBicycleNodeLinearConstraints4_1_10, QuadraticConstraints4_1_3, QuadraticConstraints6_2_5,
                                 NewBicycleNodeQuadraticConstraints6_2_5, NewLatBicycleNodeQuadraticConstraints4_1_3,
                                 Constraints6_2_16, Constraints6_2_5, Constraints4_1_10, Constraints4_1_6, Constraints4_1_3,
                                 ILQRNode6_2, ILQRNode4_1)
