"""ilqr_amd — MI355X-native batched augmented-Lagrangian iLQR (solve path of liugang97/ilqr).

Importing the package is cheap (numpy only).  ``ilqr_amd.batch`` talks to the
HIP library through ctypes; torch is used there only to own device buffers.
"""
__version__ = "0.1.0"
