"""bevy_gaussian_splatting_amd — MI355X-native sort + rasterize path for planar Gaussian
clouds, behind the plugin surface of mosure/bevy_gaussian_splatting (src/lib.rs:7-29).

Only the hot path is implemented: cloud upload, per-view depth sort,
per-splat projection + SH colour, tile binning, tile rasterisation — all as hand-written
HIP kernels for gfx950 in `csrc/`, reached through the C ABI in `include/bgs.h`.
"""
from .camera import GaussianCamera, View, transform_from, rotation_y
from .gaussian import (
    Gaussian3d,
    PlanarGaussian3d,
    PlanarGaussian3dF16,
    SphericalHarmonicCoefficients,
    SH_COEFF_COUNT,
    compute_covariance_3d,
    covariance_3d_opacity,
    random_gaussians_3d,
    random_gaussians_3d_seeded,
    trained_like_gaussians_3d_seeded,
)
from .settings import (
    CloudSettings,
    DrawMode,
    GaussianColorSpace,
    GaussianMode,
    RadixSortDepthBits,
    RasterizeMode,
    ShaderDefines,
    SortMode,
    compute_aabb,
)
from .io_ply import parse_ply_3d, write_ply_3d
from .io_gcloud import decode_gcloud, encode_gcloud, read_gcloud, write_gcloud
from .sort_policy import SortConfig, SortTrigger, update_sort_trigger
from .plugin import (
    GaussianSplattingPlugin,
    PlanarGaussian3dHandle,
    SortedEntries,
    SORT_ENTRY_DTYPE,
)

__all__ = [
    "GaussianCamera", "View", "transform_from", "rotation_y",
    "Gaussian3d", "PlanarGaussian3d", "PlanarGaussian3dF16", "SphericalHarmonicCoefficients",
    "SH_COEFF_COUNT", "compute_covariance_3d", "covariance_3d_opacity", "random_gaussians_3d", "random_gaussians_3d_seeded", "trained_like_gaussians_3d_seeded",
    "CloudSettings", "DrawMode", "GaussianColorSpace", "GaussianMode", "RadixSortDepthBits",
    "RasterizeMode", "ShaderDefines", "SortMode", "compute_aabb",
    "parse_ply_3d", "write_ply_3d", "decode_gcloud", "encode_gcloud", "read_gcloud", "write_gcloud", "SortConfig", "SortTrigger", "update_sort_trigger",
    "GaussianSplattingPlugin", "PlanarGaussian3dHandle", "SortedEntries", "SORT_ENTRY_DTYPE",
]
