"""gprf_amd — MI355X-native GPRF block-local log-likelihood + gradient path (davmre/gprf's hot path)
behind the reference's GPRF object.  HIP kernels in csrc/, C ABI in include/gprf_hip.h."""
from .cov import GPCov  # noqa: F401
from .blocking import Blocker, grid_centers, pair_distances  # noqa: F401


def __getattr__(name):
    # GPRF needs the HIP library; keep `import gprf_amd` cheap and GPU-free for the host-only helpers.
    if name == "GPRF":
        from .gprf import GPRF
        return GPRF
    raise AttributeError(name)
