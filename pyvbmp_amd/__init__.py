"""pyvbmp_amd: MI355X-native conjugate-update hot path of pyVBMP (HIP kernels behind the reference's
method surface).  See DESIGN.md."""
from . import dists  # noqa: F401

__all__ = ["dists"]
