"""Camera marker + the `View` uniform the shaders of the reference read.

Reference: src/camera.rs:6-9 (GaussianCamera), src/render/bindings.wgsl:3-9 (`view`),
examples/headless.rs:177-184 (camera of the headline configuration:
`Camera3d::default()` at (0, 1.5, 5), identity rotation).

Bevy's `Camera3d::default()` projection is `PerspectiveProjection { fov: pi/4, near: 0.1 }`
evaluated with glam's `Mat4::perspective_infinite_reverse_rh` (third-party, bevy 0.19 /
glam 0.32 — restated here, "parity unpinned": the native library takes every matrix
explicitly through `bgs_view`, so results never depend on this helper).
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass, field

from typing import Optional

import numpy as np


class BgsView(ctypes.Structure):
    """ctypes image of `bgs_view` (include/bgs.h)."""

    _fields_ = [
        ("world_from_view", ctypes.c_float * 16),
        ("view_from_world", ctypes.c_float * 16),
        ("clip_from_view", ctypes.c_float * 16),
        ("clip_from_world", ctypes.c_float * 16),
        ("viewport", ctypes.c_float * 4),
        ("clear_color", ctypes.c_float * 4),
        ("previous_clip_from_world", ctypes.c_float * 16),
        ("delta_time", ctypes.c_float),
        ("sample_count", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32 * 2),
        ("depth_device_ptr", ctypes.c_uint64),
        ("reserved_ptr", ctypes.c_uint64),
    ]


@dataclass
class GaussianCamera:
    """src/camera.rs:6-9. `order` is Bevy's `Camera.order` = index of this camera's sorted
    entries (src/sort/mod.rs:166-171)."""

    warmup: bool = False
    order: int = 0


def perspective_infinite_reverse_rh(fov_y: float, aspect: float, near: float) -> np.ndarray:
    f = np.float32(1.0) / np.float32(math.tan(0.5 * fov_y))
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, 0] = f / np.float32(aspect)
    m[1, 1] = f
    m[3, 2] = -1.0  # column 2 = (0, 0, 0, -1)
    m[2, 3] = near  # column 3 = (0, 0, near, 0)
    return m


def quat_to_mat3(q_xyzw) -> np.ndarray:
    x, y, z, w = (float(v) for v in q_xyzw)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ],
        dtype=np.float32,
    )


def transform_from(translation=(0.0, 0.0, 0.0), rotation_xyzw=(0.0, 0.0, 0.0, 1.0)) -> np.ndarray:
    """`Transform::from_translation(..).with_rotation(..)` as a 4x4 (column-vector) matrix."""
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = quat_to_mat3(rotation_xyzw)
    m[:3, 3] = np.asarray(translation, dtype=np.float32)
    return m


def rotation_y(angle: float):
    """Quaternion (x, y, z, w) of a rotation about +Y (glam `Quat::from_rotation_y`)."""
    return (0.0, math.sin(angle * 0.5), 0.0, math.cos(angle * 0.5))


@dataclass
class View:
    """The fields of Bevy's `View` uniform used on the hot path
    (src/render/helpers.wgsl:18-38, transform.wgsl:6, gaussian_2d.wgsl:104,
    src/sort/radix.wgsl:90). Matrices are numpy 4x4 in column-vector convention."""

    world_from_view: np.ndarray
    view_from_world: np.ndarray
    clip_from_view: np.ndarray
    clip_from_world: np.ndarray
    viewport: tuple  # x, y, w, h
    clear_color: tuple = (0.0, 0.0, 0.0, 1.0)  # examples/headless.rs:70
    # RasterizeMode.OpticalFlow: last frame's clip_from_world (None = camera did not move) and the
    # frame time in seconds (Bevy `globals.delta_time`)
    previous_clip_from_world: Optional[np.ndarray] = None
    delta_time: float = 1.0 / 60.0
    camera: GaussianCamera = field(default_factory=GaussianCamera)
    # The camera's `Msaa` component (Bevy: required component of Camera, default Msaa::Sample4) as the pipeline is
    # specialised on it: CloudPipelineKey.sample_count = msaa.samples() (src/render/mod.rs:357,412,422,975-979).
    # 1 = Msaa::Off, 2 / 4 / 8 = Msaa::Sample2 / Sample4 / Sample8 (0 = not set = 4). Nothing in the reference sets it, so
    # its cameras all run with 4.
    msaa_samples: int = 4
    # Device address of the view's depth attachment (Depth32Float, reverse-Z; [y][x][sample] floats) the draw is tested
    # against with GreaterEqual (src/render/mod.rs:959-974), or 0: no scene depth. GaussianSplattingPlugin.upload_depth
    # puts a host array there.
    depth_device_ptr: int = 0

    @property
    def width(self) -> int:
        return int(self.viewport[2])

    @property
    def height(self) -> int:
        return int(self.viewport[3])

    @property
    def world_position(self) -> np.ndarray:
        return np.asarray(self.world_from_view, dtype=np.float32)[:3, 3].copy()

    @staticmethod
    def perspective(
        world_from_view: np.ndarray,
        width: int,
        height: int,
        fov_y: float = math.pi / 4.0,
        near: float = 0.1,
        clear_color=(0.0, 0.0, 0.0, 1.0),
        order: int = 0,
        msaa_samples: int = 4,
    ) -> "View":
        wfv = np.asarray(world_from_view, dtype=np.float32)
        vfw = np.linalg.inv(wfv.astype(np.float64)).astype(np.float32)
        cfv = perspective_infinite_reverse_rh(fov_y, width / height, near)
        cfw = (cfv.astype(np.float32) @ vfw.astype(np.float32)).astype(np.float32)
        return View(
            world_from_view=wfv,
            view_from_world=vfw,
            clip_from_view=cfv,
            clip_from_world=cfw,
            viewport=(0.0, 0.0, float(width), float(height)),
            clear_color=tuple(float(c) for c in clear_color),
            camera=GaussianCamera(order=order),
            msaa_samples=int(msaa_samples),
        )

    @staticmethod
    def headless(width: int = 1920, height: int = 1080, yaw: float = 0.0, order: int = 0, msaa_samples: int = 4) -> "View":
        """examples/headless.rs:177-184 camera, optionally yawed about +Y in place
        (SURVEY 8(d) cfg 5: camera g = this camera yawed by g * 45 degrees). The example spawns `Camera3d::default()`
        without an `Msaa` component of its own, so it renders with Bevy's default, Msaa::Sample4."""
        wfv = transform_from((0.0, 1.5, 5.0), rotation_y(yaw))
        return View.perspective(wfv, width, height, order=order, msaa_samples=msaa_samples)

    def to_native(self) -> BgsView:
        v = BgsView()
        for name in ("world_from_view", "view_from_world", "clip_from_view", "clip_from_world"):
            m = np.asarray(getattr(self, name), dtype=np.float32)
            getattr(v, name)[:] = m.T.reshape(16).tolist()  # column-major
        v.viewport[:] = [float(c) for c in self.viewport]
        v.clear_color[:] = [float(c) for c in self.clear_color]
        prev = self.clip_from_world if self.previous_clip_from_world is None else self.previous_clip_from_world
        v.previous_clip_from_world[:] = np.asarray(prev, dtype=np.float32).T.reshape(16).tolist()
        v.delta_time = float(self.delta_time)
        v.sample_count = int(self.msaa_samples)
        v.depth_device_ptr = int(self.depth_device_ptr)
        return v
