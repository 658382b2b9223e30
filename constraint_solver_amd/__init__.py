"""constraint_solver_amd -- MI355X-native XPBD rigid-body stepper.

The product is lib/libxpbd_hip.so (hand-written HIP kernels for gfx950 behind the
C ABI in include/xpbd.h) plus the C++ host mirror in host/.  This Python package
is test and benchmark glue over that ABI.
"""
from . import capi  # noqa: F401
from .capi import World, XpbdError, step_one  # noqa: F401
