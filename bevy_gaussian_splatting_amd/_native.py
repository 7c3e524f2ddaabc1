"""ctypes binding of libbgs.so (the C ABI in include/bgs.h).

The shared library is built in-tree by `__graft_entry__.build()` (or `make -C
bevy_gaussian_splatting_amd/csrc`). There is no CPU fallback: if the library is missing,
or no HIP device is usable, every entry point of the package raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

from . import _build_id
from .camera import BgsView
from .settings import BgsSettings

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbgs.so")

BGS_OK = 0
BGS_EINVAL = -1
BGS_ENOMEM = -2
BGS_EHIP = -3
BGS_ECAPACITY = -4
BGS_EINTERNAL = -5

STAGE_NAMES = ("keygen", "depth_sort", "project", "tile_sort", "ranges", "raster")

# every symbol include/bgs.h declares (tests check the library exports each one)
EXPORTED_SYMBOLS = (
    "bgs_create",
    "bgs_destroy",
    "bgs_last_error",
    "bgs_version",
    "bgs_build_id",
    "bgs_set_queue_holders",
    "bgs_set_tile_trace",
    "bgs_selftest_ln_f32",
    "bgs_settings_default",
    "bgs_view_perspective",
    "bgs_cloud_upload_f32",
    "bgs_cloud_upload_f16",
    "bgs_cloud_free",
    "bgs_cloud_len",
    "bgs_sort",
    "bgs_render",
    "bgs_framebuffer_device_ptr",
    "bgs_sorted_entries_device_ptr",
    "bgs_set_output_srgb8",
    "bgs_framebuffer_srgb8_device_ptr",
    "bgs_set_srgb8_target",
    "bgs_set_pipeline_depth",
    "bgs_pipeline_pop",
    "bgs_frames_in_flight",
    "bgs_synchronize",
    "bgs_set_async",
    "bgs_stream",
    "bgs_set_profiling",
    "bgs_set_profiling_stride",
    "bgs_set_binning",
    "bgs_set_debug_flags",
    "bgs_get_stats",
    "bgs_radix_sort_pairs",
    "bgs_hbm_probe",
    "bgs_download",
    "bgs_set_pipeline_streams",
    "bgs_set_graphs",
    "bgs_graph_counters",
    "bgs_tile_order_counters",
    "bgs_selftest_tile_order",
    "bgs_reset_adaptive_state",
    "bgs_cloud_upload_cov3d_f32",
    "bgs_adaptive_counters",
    "bgs_set_output_rgba16f",
    "bgs_framebuffer_rgba16f_device_ptr",
    "bgs_set_packed_only",
    "bgs_device_alloc",
    "bgs_device_free",
    "bgs_upload",
    "bgs_abi_check",
    "bgs_learning_counters",
    "bgs_comm_unique_id",
    "bgs_comm_create",
    "bgs_comm_gather",
    "bgs_comm_gather_after",
    "bgs_comm_wait",
    "bgs_comm_stream",
    "bgs_comm_destroy",
)

COMM_ID_BYTES = 128
# what this binding was written against (include/bgs.h BGS_VERSION_*): load() hands it to bgs_abi_check together with
# the sizes of its ctypes structs
ABI_VERSION = (0 << 16) | 4


class BgsSortEntry(ctypes.Structure):
    _fields_ = [("key", ctypes.c_uint32), ("index", ctypes.c_uint32)]


class BgsStats(ctypes.Structure):
    _fields_ = [
        ("stage_ms", ctypes.c_float * 6),
        ("total_ms", ctypes.c_float),
        ("splat_count", ctypes.c_uint32),
        ("visible_count", ctypes.c_uint32),
        ("draw_count", ctypes.c_uint32),
        ("sort_path", ctypes.c_uint32),
        ("instance_count", ctypes.c_uint64),
        ("instance_capacity", ctypes.c_uint64),
        ("tiles_x", ctypes.c_uint32),
        ("tiles_y", ctypes.c_uint32),
        ("depth_passes", ctypes.c_uint32),
        ("tile_passes", ctypes.c_uint32),
        ("algorithmic_bytes", ctypes.c_uint64),
        ("regrow_count", ctypes.c_uint32),
        ("binning_mode", ctypes.c_uint32),
        ("frames_averaged", ctypes.c_uint32),
        ("list_capacity", ctypes.c_uint32),
        ("list_entries_allocated", ctypes.c_uint64),
        ("strip_tiles", ctypes.c_uint32),
        ("tile_saturation", ctypes.c_uint32),
    ]


class BgsError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libbgs error {status}: {message}")
        self.status = status


_lib: Optional[ctypes.CDLL] = None


def rebuild() -> str:
    """`make -C csrc` (hipcc cross-compiles gfx950 without a GPU). Returns the build log; raises on failure."""
    import subprocess
    p = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4", "ARCH=gfx950"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise ImportError(f"building libbgs.so failed:\n{p.stdout}")
    return p.stdout


def ensure_current() -> str:
    """The library on disk must have been compiled from THIS tree's kernel sources (`bgs_build_id()` = SHA-256 of
    csrc/*.hip + csrc/*.h, `_build_id.py`): a prebuilt libbgs.so that is missing or stale is rebuilt (unless
    BGS_NO_AUTOBUILD=1), and anything that still does not match is refused. Returns the id."""
    want = _build_id.kernel_source_sha256()
    have = _build_id.library_build_id(LIB_PATH) if os.path.exists(LIB_PATH) else None
    if have != want and os.environ.get("BGS_NO_AUTOBUILD", "0") != "1":
        # One builder at a time: bench.py's ranks and pytest-xdist workers import the package concurrently, and N
        # `make` processes in one directory corrupt each other's objects. The id is looked at again under the lock —
        # whoever waited finds the library its predecessor built.
        import fcntl
        with open(os.path.join(_HERE, "csrc", ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                have = _build_id.library_build_id(LIB_PATH) if os.path.exists(LIB_PATH) else None
                if have != want:
                    rebuild()
                    have = _build_id.library_build_id(LIB_PATH) if os.path.exists(LIB_PATH) else None
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    if have is None:
        raise ImportError(
            f"{LIB_PATH} not found (or it carries no build id): build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    if have != want:
        raise ImportError(f"{LIB_PATH} was built from kernel sources {have[:12]}, this tree is {want[:12]}: rebuild it "
                          "(make -C bevy_gaussian_splatting_amd/csrc)")
    return want


def build_id() -> str:
    """`bgs_build_id()` of the loaded library (= the tree's kernel-source hash, load() checked it)."""
    return load().bgs_build_id().decode()


def load() -> ctypes.CDLL:
    """Load libbgs.so once and declare prototypes. Raises if it is not built from this tree's sources and cannot
    be rebuilt."""
    global _lib
    if _lib is not None:
        return _lib
    override = os.environ.get("BGS_LIB_OVERRIDE")
    if override:
        # EXPERIMENTS ONLY (scripts/ab_variants.sh: same-box A/B of library variants built from modified sources): the
        # named library is loaded as it is, whatever it was built from. Said loudly; bench.py and the tests refuse it.
        import sys
        print(f"bevy_gaussian_splatting_amd: BGS_LIB_OVERRIDE={override} — NOT the library of this tree "
              f"(build id {_build_id.library_build_id(override)})", file=sys.stderr)
        lib = ctypes.CDLL(override, mode=ctypes.RTLD_GLOBAL)
        lib.bgs_build_id.argtypes = []
        lib.bgs_build_id.restype = ctypes.c_char_p
    else:
        want = ensure_current()
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        lib.bgs_build_id.argtypes = []
        lib.bgs_build_id.restype = ctypes.c_char_p
        if lib.bgs_build_id().decode() != want:
            raise ImportError(f"{LIB_PATH}: bgs_build_id() disagrees with the id in the file's bytes")
    vp = ctypes.c_void_p
    u32 = ctypes.c_uint32
    fp = ctypes.POINTER(ctypes.c_float)
    up = ctypes.POINTER(ctypes.c_uint32)

    lib.bgs_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    lib.bgs_create.restype = ctypes.c_int
    lib.bgs_destroy.argtypes = [vp]
    lib.bgs_destroy.restype = None
    lib.bgs_last_error.argtypes = [vp]
    lib.bgs_last_error.restype = ctypes.c_char_p
    lib.bgs_version.argtypes = []
    lib.bgs_version.restype = u32
    lib.bgs_build_id.argtypes = []
    lib.bgs_build_id.restype = ctypes.c_char_p
    lib.bgs_set_tile_trace.argtypes = [vp, vp]
    lib.bgs_set_tile_trace.restype = ctypes.c_int
    lib.bgs_set_queue_holders.argtypes = [ctypes.c_int]
    lib.bgs_set_queue_holders.restype = ctypes.c_int
    lib.bgs_selftest_ln_f32.argtypes = [vp, u32, u32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint64)]
    lib.bgs_selftest_ln_f32.restype = ctypes.c_int
    lib.bgs_settings_default.argtypes = [ctypes.POINTER(BgsSettings)]
    lib.bgs_settings_default.restype = None
    lib.bgs_view_perspective.argtypes = [
        fp, ctypes.c_float, ctypes.c_float, u32, u32, ctypes.POINTER(BgsView)]
    lib.bgs_view_perspective.restype = None
    lib.bgs_cloud_upload_f32.argtypes = [vp, u32, fp, fp, fp, fp, ctypes.POINTER(vp)]
    lib.bgs_cloud_upload_f32.restype = ctypes.c_int
    lib.bgs_cloud_upload_f16.argtypes = [vp, u32, fp, up, up, ctypes.POINTER(vp)]
    lib.bgs_cloud_upload_f16.restype = ctypes.c_int
    lib.bgs_cloud_free.argtypes = [vp, vp]
    lib.bgs_cloud_free.restype = None
    lib.bgs_cloud_len.argtypes = [vp]
    lib.bgs_cloud_len.restype = u32
    lib.bgs_sort.argtypes = [
        vp, vp, ctypes.POINTER(BgsView), ctypes.POINTER(BgsSettings), ctypes.POINTER(BgsSortEntry)]
    lib.bgs_sort.restype = ctypes.c_int
    lib.bgs_render.argtypes = [
        vp, vp, ctypes.POINTER(BgsView), ctypes.POINTER(BgsSettings), fp]
    lib.bgs_render.restype = ctypes.c_int
    lib.bgs_framebuffer_device_ptr.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint64)]
    lib.bgs_framebuffer_device_ptr.restype = ctypes.c_int
    lib.bgs_sorted_entries_device_ptr.argtypes = [vp, ctypes.POINTER(vp), up]
    lib.bgs_sorted_entries_device_ptr.restype = ctypes.c_int
    lib.bgs_synchronize.argtypes = [vp]
    lib.bgs_synchronize.restype = ctypes.c_int
    lib.bgs_set_output_srgb8.argtypes = [vp, ctypes.c_int]
    lib.bgs_set_output_srgb8.restype = ctypes.c_int
    lib.bgs_framebuffer_srgb8_device_ptr.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint64)]
    lib.bgs_framebuffer_srgb8_device_ptr.restype = ctypes.c_int
    lib.bgs_set_srgb8_target.argtypes = [vp, vp]
    lib.bgs_set_srgb8_target.restype = ctypes.c_int
    lib.bgs_set_pipeline_depth.argtypes = [vp, u32]
    lib.bgs_set_pipeline_depth.restype = ctypes.c_int
    lib.bgs_pipeline_pop.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    lib.bgs_pipeline_pop.restype = ctypes.c_int
    lib.bgs_frames_in_flight.argtypes = [vp, up]
    lib.bgs_frames_in_flight.restype = ctypes.c_int
    lib.bgs_set_async.argtypes = [vp, ctypes.c_int]
    lib.bgs_set_async.restype = ctypes.c_int
    lib.bgs_stream.argtypes = [vp, ctypes.POINTER(vp)]
    lib.bgs_stream.restype = ctypes.c_int
    lib.bgs_set_profiling.argtypes = [vp, ctypes.c_int]
    lib.bgs_set_profiling.restype = ctypes.c_int
    lib.bgs_set_profiling_stride.argtypes = [vp, u32]
    lib.bgs_set_profiling_stride.restype = ctypes.c_int
    lib.bgs_set_binning.argtypes = [vp, u32]
    lib.bgs_set_binning.restype = ctypes.c_int
    lib.bgs_set_debug_flags.argtypes = [vp, u32]
    lib.bgs_set_debug_flags.restype = ctypes.c_int
    lib.bgs_get_stats.argtypes = [vp, ctypes.POINTER(BgsStats)]
    lib.bgs_get_stats.restype = ctypes.c_int
    lib.bgs_radix_sort_pairs.argtypes = [vp, ctypes.POINTER(BgsSortEntry), u32, u32]
    lib.bgs_radix_sort_pairs.restype = ctypes.c_int
    lib.bgs_hbm_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32,
                                  ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    lib.bgs_hbm_probe.restype = ctypes.c_int
    lib.bgs_download.argtypes = [vp, vp, vp, ctypes.c_uint64]
    lib.bgs_download.restype = ctypes.c_int
    lib.bgs_set_pipeline_streams.argtypes = [vp, u32]
    lib.bgs_set_pipeline_streams.restype = ctypes.c_int
    lib.bgs_set_graphs.argtypes = [vp, ctypes.c_int]
    lib.bgs_set_graphs.restype = ctypes.c_int
    lib.bgs_graph_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.bgs_graph_counters.restype = ctypes.c_int
    lib.bgs_tile_order_counters.argtypes = [vp] + [ctypes.POINTER(ctypes.c_uint64)] * 3
    lib.bgs_tile_order_counters.restype = ctypes.c_int
    lib.bgs_selftest_tile_order.argtypes = [vp, vp, u32, u32, vp, vp]
    lib.bgs_selftest_tile_order.restype = ctypes.c_int
    lib.bgs_cloud_upload_cov3d_f32.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                               ctypes.POINTER(ctypes.c_float), ctypes.POINTER(vp)]
    lib.bgs_cloud_upload_cov3d_f32.restype = ctypes.c_int
    lib.bgs_set_output_rgba16f.argtypes = [vp, ctypes.c_int]
    lib.bgs_set_output_rgba16f.restype = ctypes.c_int
    lib.bgs_set_packed_only.argtypes = [vp, ctypes.c_int]
    lib.bgs_set_packed_only.restype = ctypes.c_int
    lib.bgs_framebuffer_rgba16f_device_ptr.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint64)]
    lib.bgs_framebuffer_rgba16f_device_ptr.restype = ctypes.c_int
    lib.bgs_device_alloc.argtypes = [vp, ctypes.c_uint64, ctypes.POINTER(vp)]
    lib.bgs_device_alloc.restype = ctypes.c_int
    lib.bgs_device_free.argtypes = [vp, vp]
    lib.bgs_device_free.restype = ctypes.c_int
    lib.bgs_upload.argtypes = [vp, vp, vp, ctypes.c_uint64]
    lib.bgs_upload.restype = ctypes.c_int
    lib.bgs_adaptive_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.bgs_adaptive_counters.restype = ctypes.c_int
    lib.bgs_reset_adaptive_state.argtypes = [vp]
    lib.bgs_reset_adaptive_state.restype = ctypes.c_int
    lib.bgs_learning_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.bgs_learning_counters.restype = ctypes.c_int
    lib.bgs_abi_check.argtypes = [u32, u32, u32, u32]
    lib.bgs_abi_check.restype = ctypes.c_int
    lib.bgs_comm_unique_id.argtypes = [ctypes.c_char_p]
    lib.bgs_comm_unique_id.restype = ctypes.c_int
    lib.bgs_comm_create.argtypes = [vp, ctypes.c_char_p, u32, u32, ctypes.POINTER(vp)]
    lib.bgs_comm_create.restype = ctypes.c_int
    lib.bgs_comm_gather.argtypes = [vp, vp, u32, vp, ctypes.c_uint64, vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.bgs_comm_gather.restype = ctypes.c_int
    lib.bgs_comm_gather_after.argtypes = [vp, vp, u32, vp, ctypes.c_uint64, vp, vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.bgs_comm_gather_after.restype = ctypes.c_int
    lib.bgs_comm_wait.argtypes = [vp, vp, ctypes.c_uint64]
    lib.bgs_comm_wait.restype = ctypes.c_int
    lib.bgs_comm_stream.argtypes = [vp, vp, ctypes.POINTER(vp)]
    lib.bgs_comm_stream.restype = ctypes.c_int
    lib.bgs_comm_destroy.argtypes = [vp, vp]
    lib.bgs_comm_destroy.restype = None
    # the handshake a binding owes the library (a stale struct layout is refused here, not read past)
    rc = lib.bgs_abi_check(ABI_VERSION, ctypes.sizeof(BgsView), ctypes.sizeof(BgsSettings), ctypes.sizeof(BgsStats))
    if rc != BGS_OK:
        raise ImportError("libbgs.so refuses this binding: " + (lib.bgs_last_error(None) or b"").decode("utf-8", "replace"))
    _lib = lib
    return lib


def check(lib: ctypes.CDLL, ctx, status: int) -> None:
    if status != BGS_OK:
        msg = lib.bgs_last_error(ctx)
        raise BgsError(status, msg.decode("utf-8", "replace") if msg else "")
