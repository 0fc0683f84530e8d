"""`diff_gaussian_rasterization_df` as the reference imports it (gaussian_renderer/__init__.py:15) -> ex4dgs_amd's HIP rasterizer."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)          # appended, not inserted: only the package name `ex4dgs_amd` is needed from there

from ex4dgs_amd import _C  # noqa: E402,F401
from ex4dgs_amd.diff_gaussian_rasterization_df import (  # noqa: E402,F401
    GaussianRasterizationSettings, GaussianRasterizer, SplitSH, _RasterizeGaussians, rasterize_gaussians)
