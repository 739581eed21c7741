"""toyslam_amd -- MI355X-native NDT scan matching behind ToySLAM's pclomp API.

The package holds only what the hot path needs: csrc/ (HIP kernels + C-ABI,
built in-tree into libndt_mi355.so), the Python mirror of
pclomp::NormalDistributionsTransform (ndt.py) and cloud plumbing (clouds.py).
"""
from ._lib import DIRECT1, DIRECT7, DIRECT26, KDTREE, NdtError  # noqa: F401
from .gicp import GeneralizedIterativeClosestPoint  # noqa: F401
from .ndt import NormalDistributionsTransform  # noqa: F401
