"""ctypes binding of the CPU oracle (oracle/libbgs_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py — never by the product package. See bgs_oracle.h for the
arithmetic contract and the pinning status.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from bevy_gaussian_splatting_amd.camera import BgsView, View
from bevy_gaussian_splatting_amd.gaussian import PlanarGaussian3d, PlanarGaussian3dF16
from bevy_gaussian_splatting_amd.settings import BgsSettings, CloudSettings

_HERE = os.path.dirname(os.path.abspath(__file__))
# BGS_ORACLE_LIB: load another build of the same source instead (the sanitizer build of tests/test_oracle_sanitize.py)
LIB_PATH = os.environ.get("BGS_ORACLE_LIB") or os.path.join(_HERE, "libbgs_oracle.so")
SORT_ENTRY_DTYPE = np.dtype([("key", np.uint32), ("index", np.uint32)])


class _Entry(ctypes.Structure):
    _fields_ = [("key", ctypes.c_uint32), ("index", ctypes.c_uint32)]


class _Cloud(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint32),
        ("position_visibility", ctypes.POINTER(ctypes.c_float)),
        ("spherical_harmonic", ctypes.POINTER(ctypes.c_float)),
        ("rotation", ctypes.POINTER(ctypes.c_float)),
        ("scale_opacity", ctypes.POINTER(ctypes.c_float)),
    ]


class VsOut(ctypes.Structure):
    _fields_ = [
        ("discard", ctypes.c_int32),
        ("projected", ctypes.c_float * 4),
        ("bb", (ctypes.c_float * 4) * 4),
        ("color", ctypes.c_float * 4),
        ("cov2d", ctypes.c_float * 3),
        ("conic", ctypes.c_float * 3),
        ("cutoff", ctypes.c_float),
        ("local_to_pixel", ctypes.c_float * 9),
        ("mean_2d", ctypes.c_float * 2),
        ("extent", ctypes.c_float * 2),
        ("radius", ctypes.c_float * 2),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no-op if up to date)."""
    if os.environ.get("BGS_ORACLE_LIB"):
        return LIB_PATH
    if force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(LIB_PATH)
        for f in ("bgs_oracle.c", "bgs_oracle.h")
    ):
        subprocess.run(["make", "-C", _HERE, "libbgs_oracle.so"], check=True, capture_output=True)
    return LIB_PATH


def effective_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes show
    256 logical CPUs under a 16-CPU quota: 128 OpenMP threads there are throttled to a crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = ctypes.CDLL(LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        up = ctypes.POINTER(ctypes.c_uint32)
        ep = ctypes.POINTER(_Entry)
        vp, sp = ctypes.POINTER(BgsView), ctypes.POINTER(BgsSettings)
        u32, i32 = ctypes.c_uint32, ctypes.c_int32
        l.oracle_radix_defines.argtypes = [u32, up, up, up]
        l.oracle_radix_defines.restype = ctypes.c_int
        l.oracle_distance_squared.argtypes = [fp, fp]
        l.oracle_distance_squared.restype = ctypes.c_float
        l.oracle_ln_f32_array.argtypes = [fp, u32, fp]
        l.oracle_ln_f32_array.restype = None
        l.oracle_ln_f32_checksum.argtypes = [u32, u32]
        l.oracle_ln_f32_checksum.restype = ctypes.c_uint64
        l.oracle_radix_depth_key.argtypes = [ctypes.c_float, u32]
        l.oracle_radix_depth_key.restype = u32
        l.oracle_keygen.argtypes = [fp, u32, vp, sp, ep]
        l.oracle_keygen.restype = ctypes.c_int
        l.oracle_radix_sort.argtypes = [ep, u32, u32, ep]
        l.oracle_radix_sort.restype = None
        l.oracle_sort_descending_f32.argtypes = [ep, u32]
        l.oracle_sort_descending_f32.restype = None
        l.oracle_sort.argtypes = [fp, u32, vp, sp, ep]
        l.oracle_sort.restype = ctypes.c_int
        l.oracle_vs.argtypes = [ctypes.POINTER(_Cloud), _Entry, vp, sp, ctypes.POINTER(VsOut)]
        l.oracle_vs.restype = ctypes.c_int
        l.oracle_vs_sorted.argtypes = [ctypes.POINTER(_Cloud), ep, u32, u32, vp, sp, ctypes.POINTER(VsOut)]
        l.oracle_vs_sorted.restype = ctypes.c_int
        l.oracle_depth_range.argtypes = [ctypes.POINTER(_Cloud), ep, u32, vp, sp, fp]
        l.oracle_depth_range.restype = ctypes.c_int
        l.oracle_render.argtypes = [ctypes.POINTER(_Cloud), ep, u32, vp, sp, i32, i32, i32, i32, fp, fp]
        l.oracle_render.restype = ctypes.c_int
        l.oracle_render_depth.argtypes = [ctypes.POINTER(_Cloud), ep, u32, vp, sp, i32, i32, i32, i32, fp, fp, fp]
        l.oracle_render_depth.restype = ctypes.c_int
        l.oracle_render_target.argtypes = [ctypes.POINTER(_Cloud), ep, u32, vp, sp, i32, i32, i32, i32, fp, ctypes.c_int, fp, fp]
        l.oracle_render_target.restype = ctypes.c_int
        l.oracle_srgb8_codes.argtypes = [fp, u32, ctypes.POINTER(ctypes.c_uint8)]
        l.oracle_srgb8_codes.restype = None
        l.oracle_sample_positions.argtypes = [u32, fp]
        l.oracle_sample_positions.restype = ctypes.c_int
        l.oracle_decode_f16.argtypes = [u32, up, up, fp, fp, fp]
        l.oracle_decode_f16.restype = None
        l.oracle_encode_f16.argtypes = [u32, fp, fp, fp, up, up]
        l.oracle_encode_f16.restype = None
        l.oracle_instance_stats.argtypes = [
            ctypes.POINTER(_Cloud), ep, u32, vp, sp, up, ctypes.POINTER(ctypes.c_uint64)]
        l.oracle_instance_stats.restype = ctypes.c_int
        l.oracle_encode_srgb8.argtypes = [fp, u32, ctypes.POINTER(ctypes.c_uint8)]
        l.oracle_encode_srgb8.restype = None
        l.oracle_set_threads.argtypes = [ctypes.c_int]
        l.oracle_set_threads.restype = None
        l.oracle_max_threads.argtypes = []
        l.oracle_max_threads.restype = ctypes.c_int
        l.oracle_set_edge_band_px.argtypes = [ctypes.c_double]
        l.oracle_set_edge_band_px.restype = None
        l.oracle_edge_band_px.argtypes = []
        l.oracle_edge_band_px.restype = ctypes.c_double
        if os.environ.get("BGS_ORACLE_EDGE_BAND_PX"):
            l.oracle_set_edge_band_px(float(os.environ["BGS_ORACLE_EDGE_BAND_PX"]))
        if "OMP_NUM_THREADS" not in os.environ:
            l.oracle_set_threads(min(l.oracle_max_threads(), effective_cpus()))
        _lib = l
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _up(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))


def _ep(a):
    return a.ctypes.data_as(ctypes.POINTER(_Entry))


def _as_f32_cloud(cloud) -> PlanarGaussian3d:
    if isinstance(cloud, PlanarGaussian3dF16):
        return decode_f16(cloud)
    return cloud


def _cloud_struct(cloud: PlanarGaussian3d) -> _Cloud:
    return _Cloud(len(cloud), _fp(cloud.position_visibility), _fp(cloud.spherical_harmonic),
                  _fp(cloud.rotation), _fp(cloud.scale_opacity))


def radix_defines(bits: int):
    p, s, par = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    rc = lib().oracle_radix_defines(bits, ctypes.byref(p), ctypes.byref(s), ctypes.byref(par))
    if rc:
        raise ValueError(f"unsupported depth bits {bits}")
    return p.value, s.value, par.value


def distance_squared(position, camera) -> np.float32:
    p = np.asarray(position, np.float32)
    c = np.asarray(camera, np.float32)
    return np.float32(lib().oracle_distance_squared(_fp(p), _fp(c)))


def radix_depth_key(dist2, key_shift: int) -> int:
    return int(lib().oracle_radix_depth_key(ctypes.c_float(float(dist2)), key_shift))


def keygen(cloud, view: View, settings: CloudSettings) -> np.ndarray:
    cloud = _as_f32_cloud(cloud)
    out = np.empty(len(cloud), SORT_ENTRY_DTYPE)
    v, s = view.to_native(), settings.to_native()
    rc = lib().oracle_keygen(_fp(cloud.position_visibility), len(cloud), ctypes.byref(v), ctypes.byref(s), _ep(out))
    if rc:
        raise RuntimeError(f"oracle_keygen failed: {rc}")
    return out


def radix_sort(entries: np.ndarray, places: int) -> np.ndarray:
    e = np.ascontiguousarray(entries, dtype=SORT_ENTRY_DTYPE).copy()
    tmp = np.empty_like(e)
    lib().oracle_radix_sort(_ep(e), e.shape[0], places, _ep(tmp))
    return e


def sort(cloud, view: View, settings: CloudSettings) -> np.ndarray:
    cloud = _as_f32_cloud(cloud)
    out = np.empty(len(cloud), SORT_ENTRY_DTYPE)
    v, s = view.to_native(), settings.to_native()
    rc = lib().oracle_sort(_fp(cloud.position_visibility), len(cloud), ctypes.byref(v), ctypes.byref(s), _ep(out))
    if rc:
        raise RuntimeError(f"oracle_sort failed: {rc}")
    return out


def vs(cloud, entry, view: View, settings: CloudSettings) -> VsOut:
    cloud = _as_f32_cloud(cloud)
    c = _cloud_struct(cloud)
    out = VsOut()
    v, s = view.to_native(), settings.to_native()
    e = _Entry(int(entry[0]), int(entry[1]))
    rc = lib().oracle_vs(ctypes.byref(c), e, ctypes.byref(v), ctypes.byref(s), ctypes.byref(out))
    if rc:
        raise RuntimeError(f"oracle_vs failed: {rc} (RasterizeMode.Depth needs vs_sorted)")
    return out


def vs_sorted(cloud, entries: np.ndarray, instance: int, view: View, settings: CloudSettings) -> VsOut:
    """vs_points for instance `instance` of the sorted list (needed by RasterizeMode.Depth)."""
    cloud = _as_f32_cloud(cloud)
    c = _cloud_struct(cloud)
    e = np.ascontiguousarray(entries, dtype=SORT_ENTRY_DTYPE)
    out = VsOut()
    v, s = view.to_native(), settings.to_native()
    rc = lib().oracle_vs_sorted(ctypes.byref(c), _ep(e), e.shape[0], instance, ctypes.byref(v), ctypes.byref(s),
                                ctypes.byref(out))
    if rc:
        raise RuntimeError(f"oracle_vs_sorted failed: {rc}")
    return out


def depth_range(cloud, entries: np.ndarray, view: View, settings: CloudSettings):
    cloud = _as_f32_cloud(cloud)
    c = _cloud_struct(cloud)
    e = np.ascontiguousarray(entries, dtype=SORT_ENTRY_DTYPE)
    out = np.zeros(2, np.float32)
    v, s = view.to_native(), settings.to_native()
    rc = lib().oracle_depth_range(ctypes.byref(c), _ep(e), e.shape[0], ctypes.byref(v), ctypes.byref(s), _fp(out))
    if rc:
        raise RuntimeError(f"oracle_depth_range failed: {rc}")
    return float(out[0]), float(out[1])


TARGET_F32, TARGET_SRGB8, TARGET_RGBA16F = 0, 1, 2


def render(cloud, entries: np.ndarray, view: View, settings: CloudSettings, window=None,
           with_ambiguity: bool = False, depth=None, target_format: int = TARGET_F32):
    """Draw `entries` in order into a target with `view.msaa_samples` samples per pixel (coverage and depth test per
    sample, shading once per pixel, box resolve). window = (x0, y0, x1, y1) or None for the full viewport.
    depth = None or the view's scene depth as a HOST array [height, width, msaa_samples] float32 (reverse-Z; a fragment
    passes where its depth >= the stored one). target_format: TARGET_F32 (the ideal binary32 target the parity is stated
    against), TARGET_SRGB8 / TARGET_RGBA16F = the reference's colour attachment, every sample rounded at every blend
    (src/render/mod.rs:917-921,944-948). Returns rgba [h, w, 4] (and the ambiguity bound [h, w] if requested)."""
    cloud = _as_f32_cloud(cloud)
    c = _cloud_struct(cloud)
    x0, y0, x1, y1 = window if window is not None else (0, 0, view.width, view.height)
    e = np.ascontiguousarray(entries, dtype=SORT_ENTRY_DTYPE)
    out = np.empty((y1 - y0, x1 - x0, 4), np.float32)
    amb = np.empty((y1 - y0, x1 - x0), np.float32) if with_ambiguity else None
    v, s = view.to_native(), settings.to_native()
    d = None
    if depth is not None:
        d = np.ascontiguousarray(depth, dtype=np.float32)
        if d.shape != (view.height, view.width, view.msaa_samples):
            raise ValueError(f"depth must be [height, width, samples] = {(view.height, view.width, view.msaa_samples)}, got {d.shape}")
    rc = lib().oracle_render_target(ctypes.byref(c), _ep(e), e.shape[0], ctypes.byref(v), ctypes.byref(s),
                                    x0, y0, x1, y1, _fp(d) if d is not None else None, int(target_format), _fp(out),
                                    _fp(amb) if amb is not None else None)
    if rc:
        raise RuntimeError(f"oracle_render failed: {rc}")
    return (out, amb) if with_ambiguity else out


def srgb8_codes(rgba: np.ndarray) -> np.ndarray:
    """Rgba8UnormSrgb codes [..., 4] uint8 of RGBA values (exact rounding by threshold search, no pow per value)."""
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.empty(a.shape, np.uint8)
    lib().oracle_srgb8_codes(_fp(a), a.size // 4, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out


def sample_positions(sample_count: int) -> np.ndarray:
    """[sample_count, 2] sample positions inside a pixel (origin = its top-left corner, y down)."""
    out = np.empty((sample_count, 2), np.float32)
    if lib().oracle_sample_positions(sample_count, _fp(out)):
        raise ValueError(f"unsupported sample count {sample_count}")
    return out


def sort_and_render(cloud, view: View, settings: CloudSettings, window=None, with_ambiguity=False, depth=None):
    entries = sort(cloud, view, settings)
    return entries, render(cloud, entries, view, settings, window, with_ambiguity, depth)


def instance_stats(cloud, entries, view: View, settings: CloudSettings):
    cloud = _as_f32_cloud(cloud)
    c = _cloud_struct(cloud)
    e = np.ascontiguousarray(entries, dtype=SORT_ENTRY_DTYPE)
    vis, inst = ctypes.c_uint32(), ctypes.c_uint64()
    v, s = view.to_native(), settings.to_native()
    lib().oracle_instance_stats(ctypes.byref(c), _ep(e), e.shape[0], ctypes.byref(v), ctypes.byref(s),
                                ctypes.byref(vis), ctypes.byref(inst))
    return vis.value, inst.value


def decode_f16(cloud: PlanarGaussian3dF16) -> PlanarGaussian3d:
    n = len(cloud)
    sh = np.empty((n, 48), np.float32)
    rot = np.empty((n, 4), np.float32)
    so = np.empty((n, 4), np.float32)
    lib().oracle_decode_f16(n, _up(cloud.spherical_harmonic), _up(cloud.rotation_scale_opacity),
                            _fp(sh), _fp(rot), _fp(so))
    return PlanarGaussian3d(cloud.position_visibility, sh, rot, so)


def encode_f16(cloud: PlanarGaussian3d) -> PlanarGaussian3dF16:
    n = len(cloud)
    sh = np.empty((n, 24), np.uint32)
    rso = np.empty((n, 4), np.uint32)
    lib().oracle_encode_f16(n, _fp(cloud.spherical_harmonic), _fp(cloud.rotation),
                            _fp(cloud.scale_opacity), _up(sh), _up(rso))
    return PlanarGaussian3dF16(cloud.position_visibility, sh, rso)


def encode_srgb8(rgba: np.ndarray) -> np.ndarray:
    """[h, w, 4] float32 -> [h, w, 4] uint8 in the reference's Rgba8UnormSrgb target format."""
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.empty(a.shape, np.uint8)
    lib().oracle_encode_srgb8(_fp(a), a.size // 4, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def set_threads(n: int) -> None:
    lib().oracle_set_threads(int(n))


def ln_f32(x: np.ndarray) -> np.ndarray:
    """ln(x) correctly rounded to binary32 (x87 logl rounded once): the log of the adaptive cutoff."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().oracle_ln_f32_array(x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), x.size,
                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def ln_f32_checksum(first_bits: int, count: int) -> int:
    """The sum `bgs_selftest_ln_f32` forms on the device, computed from the oracle's log."""
    return int(lib().oracle_ln_f32_checksum(first_bits, count))
