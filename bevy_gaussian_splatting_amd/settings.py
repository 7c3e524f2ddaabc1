"""Runtime settings: mirror of the reference's `CloudSettings` surface.

Reference: src/gaussian/settings.rs:6-132 (enums + CloudSettings + Default),
src/sort/mod.rs:46-74 (SortMode), src/render/mod.rs:698-764 (ShaderDefines).
Only the fields that select behaviour on the sort + rasterize hot path are honoured by
the native library; the rest are carried so reference code that sets them keeps working.
"""
from __future__ import annotations

import ctypes
import enum
from dataclasses import dataclass, field

import numpy as np


class GaussianMode(enum.IntEnum):
    """src/gaussian/settings.rs:17-22 (Gaussian4d is out of scope for the hot path)."""

    Gaussian2d = 0
    Gaussian3d = 1


class SortMode(enum.IntEnum):
    """src/sort/mod.rs:46-58."""

    NONE = 0
    Radix = 1
    Rayon = 2
    Std = 3


class RadixSortDepthBits(enum.IntEnum):
    """src/gaussian/settings.rs:52-77."""

    Bits16 = 16
    Bits24 = 24
    Bits32 = 32

    def bits(self) -> int:
        return int(self)


class GaussianColorSpace(enum.IntEnum):
    """src/gaussian/settings.rs:79-84; uniform value per src/render/mod.rs:1066-1069."""

    SrgbRec709Display = 0
    LinRec709Display = 1


class RasterizeMode(enum.IntEnum):
    """src/gaussian/settings.rs:36-47 (same discriminants). Color is the benchmarked mode;
    Classification / Depth / Normal / Position are colour-stage variants of the same pipeline
    (src/render/gaussian.wgsl:312-405); OpticalFlow reads the previous frame's clip_from_world and the
    frame time from the View. Velocity (4D clouds only) is outside the path and rejected by the library."""

    Classification = 0
    Color = 1
    Depth = 2
    Normal = 3
    OpticalFlow = 4
    Position = 5
    Velocity = 6


class DrawMode(enum.IntEnum):
    """src/gaussian/settings.rs:6-12; shader defs DRAW_SELECTED / HIGHLIGHT_SELECTED
    (src/render/mod.rs:889-893, src/render/gaussian.wgsl:203-205,423-427)."""

    All = 0
    Selected = 1
    HighlightSelected = 2


@dataclass(frozen=True)
class ShaderDefines:
    """src/render/mod.rs:698-764: radix-sort geometry derived from the depth-bit setting."""

    radix_bits_per_digit: int
    radix_digit_places: int
    radix_key_shift: int
    radix_base: int
    entries_per_invocation_a: int
    entries_per_invocation_c: int
    workgroup_invocations_a: int
    workgroup_invocations_c: int
    workgroup_entries_a: int
    workgroup_entries_c: int
    sorting_buffer_size: int

    @staticmethod
    def for_radix_depth_bits(bits: RadixSortDepthBits | int) -> "ShaderDefines":
        b = int(bits)
        if b not in (16, 24, 32):
            raise ValueError(f"unsupported radix depth bits: {b}")
        per_digit = 8
        places = b // per_digit
        base = 1 << per_digit
        inv_a = base * places
        return ShaderDefines(
            radix_bits_per_digit=per_digit,
            radix_digit_places=places,
            radix_key_shift=32 - b,
            radix_base=base,
            entries_per_invocation_a=4,
            entries_per_invocation_c=4,
            workgroup_invocations_a=inv_a,
            workgroup_invocations_c=base,
            workgroup_entries_a=inv_a * 4,
            workgroup_entries_c=base * 4,
            sorting_buffer_size=base * places * 4 + (5 + base) * 4,
        )

    def radix_initial_parity(self) -> int:
        return self.radix_digit_places % 2

    def max_tile_count(self, count: int) -> int:
        return -(-count // self.workgroup_entries_c)


class BgsSettings(ctypes.Structure):
    """ctypes image of `bgs_settings` (include/bgs.h)."""

    _fields_ = [
        ("transform", ctypes.c_float * 16),
        ("global_opacity", ctypes.c_float),
        ("global_scale", ctypes.c_float),
        ("gaussian_mode", ctypes.c_uint32),
        ("aabb", ctypes.c_uint32),
        ("opacity_adaptive_radius", ctypes.c_uint32),
        ("color_space", ctypes.c_uint32),
        ("radix_depth_bits", ctypes.c_uint32),
        ("sh_degree", ctypes.c_uint32),
        ("sort_mode", ctypes.c_uint32),
        ("rasterize_mode", ctypes.c_uint32),
        ("num_classes", ctypes.c_uint32),
        ("draw_mode", ctypes.c_uint32),
        ("visualize_bounding_box", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32 * 1),
        ("position_min", ctypes.c_float * 4),
        ("position_max", ctypes.c_float * 4),
    ]


def _identity4() -> np.ndarray:
    return np.eye(4, dtype=np.float32)


@dataclass
class CloudSettings:
    """src/gaussian/settings.rs:87-132. Defaults equal `CloudSettings::default()`.

    `transform` stands in for the entity's `GlobalTransform` (column-vector 4x4, i.e.
    `transform @ [x, y, z, 1]`), which the reference passes through `CloudUniform.transform`
    (src/render/mod.rs:1057). `sh_degree` mirrors the compile-time `sh0..sh3` features.
    """

    aabb: bool = False
    global_opacity: float = 1.0
    global_scale: float = 1.0
    opacity_adaptive_radius: bool = True
    visualize_bounding_box: bool = False  # the quads' frames drawn over the splats (src/render/gaussian.wgsl:486-495)
    sort_mode: SortMode = SortMode.Radix
    radix_sort_depth_bits: RadixSortDepthBits = RadixSortDepthBits.Bits32
    draw_mode: DrawMode = DrawMode.All
    gaussian_mode: GaussianMode = GaussianMode.Gaussian3d
    rasterize_mode: RasterizeMode = RasterizeMode.Color
    color_space: GaussianColorSpace = GaussianColorSpace.SrgbRec709Display
    num_classes: int = 1
    time: float = 0.0
    time_scale: float = 1.0
    time_start: float = 0.0
    time_stop: float = 1.0
    sh_degree: int = 3
    transform: np.ndarray = field(default_factory=_identity4)
    # CloudUniform.min / .max = the cloud entity's Aabb (src/render/mod.rs:1070-1071); only read by
    # RasterizeMode.Position. `compute_aabb(cloud)` gives what the reference would have attached.
    position_min: tuple = (0.0, 0.0, 0.0)
    position_max: tuple = (1.0, 1.0, 1.0)

    def to_native(self) -> BgsSettings:
        s = BgsSettings()
        m = np.asarray(self.transform, dtype=np.float32)
        if m.shape != (4, 4):
            raise ValueError("transform must be 4x4")
        # column-major, like glam's Mat4::to_cols_array
        s.transform[:] = m.T.reshape(16).tolist()
        s.global_opacity = float(self.global_opacity)
        s.global_scale = float(self.global_scale)
        s.gaussian_mode = int(self.gaussian_mode)
        s.aabb = 1 if self.aabb else 0
        s.opacity_adaptive_radius = 1 if self.opacity_adaptive_radius else 0
        s.color_space = int(self.color_space)
        s.radix_depth_bits = int(self.radix_sort_depth_bits)
        s.sh_degree = int(self.sh_degree)
        s.sort_mode = int(self.sort_mode)
        s.rasterize_mode = int(self.rasterize_mode)
        s.num_classes = int(self.num_classes)
        s.draw_mode = int(self.draw_mode)
        s.visualize_bounding_box = 1 if self.visualize_bounding_box else 0
        s.position_min[:] = [float(v) for v in self.position_min] + [1.0]
        s.position_max[:] = [float(v) for v in self.position_max] + [1.0]
        return s


def compute_aabb(cloud):
    """(min, max) the reference hands to the shaders for `cloud`: `compute_aabb`
    (src/gaussian/interface.rs:22-63: position -/+ 0.1 per splat) -> Bevy `Aabb {center,
    half_extents}` (src/gaussian/cloud.rs:56-59) -> `aabb.min()/max()` = center -/+ half_extents
    (src/render/mod.rs:1070-1071), all in f32. Returns None for an empty cloud."""
    if len(cloud) == 0:
        return None
    pos = cloud.position_visibility[:, :3]
    off = np.float32(0.1)
    mn = (pos - off).min(axis=0).astype(np.float32)
    mx = (pos + off).max(axis=0).astype(np.float32)
    center = ((mn + mx) / np.float32(2.0)).astype(np.float32)
    half = ((mx - mn) / np.float32(2.0)).astype(np.float32)
    return tuple((center - half).tolist()), tuple((center + half).tolist())
