"""threestudio-soar renderer plugin (``"gaussiansurfel-rasterizer"``) on the MI355X kernels."""
from . import registry  # noqa: F401
from .batch import GaussianBatchRenderer  # noqa: F401
from .cameras import Camera, get_cam_info_gaussian_cxcy, get_projection_matrix_gaussian, sample_camera  # noqa: F401
from .diff_gaussian import DiffGaussian, axis_permutation, transform_point_cloud  # noqa: F401
from .postops import depth2normal, fov2focal, normal2curv  # noqa: F401
