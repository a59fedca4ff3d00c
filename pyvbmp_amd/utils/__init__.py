"""Helpers of the conjugate-update path: the block-matrix algebra (K1-backed) and the log-space utilities."""
from .matrix_utils import matrix_utils
from .torch_functions import log_mvgamma, logmatmulexp, mvdigamma, mvgammaln, stable_logsumexp, stable_softmax

__all__ = ["matrix_utils", "stable_logsumexp", "stable_softmax", "logmatmulexp", "log_mvgamma", "mvgammaln", "mvdigamma"]
