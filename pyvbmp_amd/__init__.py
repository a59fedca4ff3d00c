"""pyvbmp_amd: MI355X-native conjugate-update hot path of pyVBMP (HIP kernels behind the reference's
method surface).  See DESIGN.md."""
from . import dists, models, transforms, utils  # noqa: F401

__all__ = ["dists", "transforms", "utils", "models"]
