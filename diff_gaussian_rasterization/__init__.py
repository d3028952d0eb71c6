"""Import-compatible alias: ``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
(the import used by TS/renderer/diff_gaussian_rasterizer.py:6-9) resolves to the MI355X implementation in
``soar_amd.rasterizer``."""
from soar_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
    _C,
)
