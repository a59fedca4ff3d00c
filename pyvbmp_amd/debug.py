"""Opt-in run-time checks (they synchronise with the device, so they are off by default).

    with pyvbmp_amd.debug.check_spd():
        model.update(...)        # raises VbmpHipError as soon as K1 / K2 / K2a meets a non-positive pivot

The elimination kernels do not pivot: their operands are symmetric positive definite in exact arithmetic (precisions,
covariances, Schur complements).  A matrix that is not -- a degenerate statistic, an indefinite blend -- yields the
reference's own NaN / -inf from the log-determinant, silently, as the reference does; under `check_spd` the kernels' device
counter of matrices with a non-positive pivot (`nonspd` of the C-ABI) is read back after every such launch instead.
Not covered: the recursions of the LDS smoother (K9), which invert the reference's cross-covariance operand
(`models/LinearDynamicalSystems.py:372`, not symmetric by construction) in registers without a counter."""
import contextlib

from . import ops


@contextlib.contextmanager
def check_spd(enabled=True):
    old = ops.CHECK_SPD
    ops.CHECK_SPD = bool(enabled)
    try:
        yield
    finally:
        ops.CHECK_SPD = old
