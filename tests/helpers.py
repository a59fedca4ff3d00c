"""Shared test helpers: golden-fixture loading and the normwise parity metric."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerances: natural parameters within 1e-10 rel (fp64) / 1e-4 (fp32), measured
# normwise per tensor: max|a-b| / max|b|   (SURVEY.md section 7 "Parity metric").
TOL64 = 1e-10
TOL32 = 1e-4


def load_golden(group):
    """tests/golden/<group>.npz -> {case: {field: torch tensor}} (no pickle)."""
    z = np.load(os.path.join(GOLDEN, group + ".npz"), allow_pickle=False)
    out = {}
    for key in z.files:
        case, field = key.split("/", 1)
        a = z[key]
        out.setdefault(case, {})[field] = torch.from_numpy(np.array(a))
    return out


def relerr(a, b):
    """normwise relative error max|a-b| / max(|b|, tiny); shapes must match exactly."""
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert tuple(a.shape) == tuple(b.shape), f"shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if b.numel() == 0:
        return 0.0
    # the reference propagates NaN silently for non-SPD arguments (Tensor.logdet of a matrix with
    # negative determinant); parity then means "NaN in exactly the same places"
    nan_a, nan_b = torch.isnan(a), torch.isnan(b)
    if nan_a.any() or nan_b.any():
        if not torch.equal(nan_a, nan_b):
            return float("inf")
        a, b = a[~nan_a], b[~nan_b]
        if b.numel() == 0:
            return 0.0
    den = max(float(b.abs().max()), 1e-300)
    return float((a - b).abs().max()) / den


def assert_close(a, b, tol=TOL64, what=""):
    e = relerr(a, b)
    assert e <= tol, f"{what}: normwise rel err {e:.3e} > {tol:.1e}"
