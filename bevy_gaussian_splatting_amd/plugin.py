"""Host-side mirror of the reference's plugin surface for the sort + rasterize path.

Reference call sites this stands behind:
  - cloud upload: RenderAsset upload of the planar cloud (src/lib.rs:65-68,
    src/render/mod.rs:279-313)                               -> `upload`
  - sort: `run_radix_sort` (src/sort/radix.rs:616-756) / `rayon_sort`
    (src/sort/rayon.rs:27-130), output `SortedEntries` (src/sort/mod.rs:331-393) -> `sort`
  - draw: `DrawGaussianInstanced::render` (src/render/mod.rs:1513-1569)   -> `render`

All compute happens in libbgs.so (hand-written HIP for gfx950). This module only marshals
arguments; it has no CPU implementation of any stage.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Union

import numpy as np

from . import _native
from .camera import View
from .gaussian import PlanarGaussian3d, PlanarGaussian3dF16
from .settings import CloudSettings

SORT_ENTRY_DTYPE = np.dtype([("key", np.uint32), ("index", np.uint32)])


@dataclass
class SortedEntries:
    """src/sort/mod.rs:331-393: `camera_count * entry_count` (key, index) pairs, where the creator
    passes `entry_count = cloud.len_sqrt_ceil()**2` (src/sort/mod.rs:259-262). Camera `c` owns the
    chunk `sorted[c * gaussians : (c + 1) * gaussians]` with `gaussians = cloud.len()` — the stride is
    the CLOUD length, not entry_count, both where the sort writes (`chunks_mut(gaussians).nth(
    camera_index)`, src/sort/rayon.rs:82-84) and where the draw binds (dynamic offset
    `camera_index * 8 * cloud.len()`, src/render/mod.rs:1548-1554); the square-padding tail is unused."""

    camera_count: int
    entry_count: int
    sorted: np.ndarray  # structured SORT_ENTRY_DTYPE

    @staticmethod
    def new(camera_count: int, entry_count: int) -> "SortedEntries":
        s = np.empty(camera_count * entry_count, dtype=SORT_ENTRY_DTYPE)
        s["key"] = 1  # src/sort/mod.rs:349-352
        s["index"] = np.tile(np.arange(entry_count, dtype=np.uint32), camera_count)
        return SortedEntries(camera_count, entry_count, s)

    @staticmethod
    def for_cloud(camera_count: int, cloud_len: int) -> "SortedEntries":
        """`auto_insert_sorted_entries` (src/sort/mod.rs:218-268)."""
        side = int(np.ceil(np.sqrt(np.float32(cloud_len))))
        return SortedEntries.new(camera_count, side * side)

    def chunk(self, camera_index: int, gaussians: Optional[int] = None) -> np.ndarray:
        g = self.entry_count if gaussians is None else int(gaussians)
        if camera_index < 0 or (camera_index + 1) * g > self.sorted.shape[0]:
            raise IndexError("camera chunk out of range")  # `.nth(camera_index).unwrap()` panics
        return self.sorted[camera_index * g : (camera_index + 1) * g]

    def resized(self, camera_count: int) -> "SortedEntries":
        """`update_sorted_entries_sizes` (src/sort/mod.rs:270-296): a camera-count change re-creates
        the asset (all chunks back to key 1 / identity order)."""
        return self if camera_count == self.camera_count else SortedEntries.new(camera_count, self.entry_count)


@dataclass
class PreparedView:
    """C-struct images of one (View, CloudSettings) pair (GaussianSplattingPlugin.prepare)."""

    view: object
    settings: object
    width: int
    height: int


class PlanarGaussian3dHandle:
    """Device-resident cloud (the reference's `PlanarGaussian3dHandle` +
    `GpuPlanarStorage` rolled into one opaque handle)."""

    def __init__(self, plugin: "GaussianSplattingPlugin", ptr, n: int, fmt: str, nbytes: int):
        self._plugin = plugin
        self._ptr = ptr
        self.n = n
        self.format = fmt
        self.nbytes = nbytes

    def __len__(self) -> int:
        return self.n

    def free(self) -> None:
        if self._ptr is not None and self._plugin._ctx is not None:
            self._plugin._lib.bgs_cloud_free(self._plugin._ctx, self._ptr)
        self._ptr = None


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _uptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))


class GaussianSplattingPlugin:
    """One instance = one HIP device + one stream (`bgs_ctx`). Not re-entrant, like a Bevy
    render world. Use one instance per GPU / per process rank."""

    def __init__(self, device: int = 0):
        self._lib = _native.load()
        ctx = ctypes.c_void_p()
        st = self._lib.bgs_create(int(device), ctypes.byref(ctx))
        if st != _native.BGS_OK:
            msg = self._lib.bgs_last_error(None)
            raise _native.BgsError(st, msg.decode("utf-8", "replace") if msg else "")
        self._ctx = ctx
        self.device = device

    # -- lifecycle -------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None:
            self._lib.bgs_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status: int) -> None:
        _native.check(self._lib, self._ctx, status)

    # -- cloud upload ----------------------------------------------------------------
    def upload(self, cloud: Union[PlanarGaussian3d, PlanarGaussian3dF16],
               precompute_covariance_3d: bool = False) -> PlanarGaussian3dHandle:
        """Make the cloud resident in HBM (the reference's asset upload). `precompute_covariance_3d`: store
        the `Covariance3dOpacity` plane (src/gaussian/f32.rs:218-251) instead of rotation and scale, as the
        reference's feature of that name does; the vertex stage then skips compute_cov3d."""
        out = ctypes.c_void_p()
        n = len(cloud)
        if precompute_covariance_3d:
            if not isinstance(cloud, PlanarGaussian3d):
                raise TypeError("precompute_covariance_3d needs an f32 PlanarGaussian3d")
            from .gaussian import covariance_3d_opacity
            cov = np.ascontiguousarray(covariance_3d_opacity(cloud), np.float32)
            self._check(
                self._lib.bgs_cloud_upload_cov3d_f32(
                    self._ctx, n, _fptr(cloud.position_visibility), _fptr(cloud.spherical_harmonic),
                    _fptr(cov), ctypes.byref(out)))
            return PlanarGaussian3dHandle(self, out, n, "cov3d", n * (16 + 192 + 32))
        if isinstance(cloud, PlanarGaussian3dF16):
            self._check(
                self._lib.bgs_cloud_upload_f16(
                    self._ctx, n, _fptr(cloud.position_visibility),
                    _uptr(cloud.spherical_harmonic), _uptr(cloud.rotation_scale_opacity),
                    ctypes.byref(out)))
            fmt = "f16"
        elif isinstance(cloud, PlanarGaussian3d):
            self._check(
                self._lib.bgs_cloud_upload_f32(
                    self._ctx, n, _fptr(cloud.position_visibility),
                    _fptr(cloud.spherical_harmonic), _fptr(cloud.rotation),
                    _fptr(cloud.scale_opacity), ctypes.byref(out)))
            fmt = "f32"
        else:
            raise TypeError("cloud must be PlanarGaussian3d or PlanarGaussian3dF16")
        return PlanarGaussian3dHandle(self, out, n, fmt, cloud.nbytes())

    # -- hot path --------------------------------------------------------------------
    def sort(self, handle: PlanarGaussian3dHandle, view: View, settings: CloudSettings,
             download: bool = True) -> Optional[np.ndarray]:
        """Depth sort for one camera. Returns the (key, index) entries of this camera's
        chunk in draw order (structured array), or None if `download` is False."""
        v, s = view.to_native(), settings.to_native()
        if download:
            out = np.empty(handle.n, dtype=SORT_ENTRY_DTYPE)
            ptr = out.ctypes.data_as(ctypes.POINTER(_native.BgsSortEntry))
        else:
            out, ptr = None, None
        self._check(self._lib.bgs_sort(self._ctx, handle._ptr, ctypes.byref(v), ctypes.byref(s), ptr))
        return out

    def sort_cameras(self, handle: PlanarGaussian3dHandle, views, settings: CloudSettings,
                     sorted_entries: Optional[SortedEntries] = None, camera_indices=None) -> SortedEntries:
        """Multi-camera entry layout (src/sort/mod.rs:331-393): depth-sort the cloud for each view and
        store the result in that camera's chunk of one `SortedEntries`. `views[i]` belongs to camera
        index `camera_indices[i]` (default i = `Camera.order`, src/sort/mod.rs:171-176). Unlike the
        reference's GPU radix path, which only ever sorts into chunk 0 (TODO at src/sort/mod.rs:427),
        every camera gets its own order here. Cameras whose trigger does not ask for a sort are simply
        left out of `views`; their chunks keep their previous content."""
        views = list(views)
        idx = list(range(len(views))) if camera_indices is None else [int(i) for i in camera_indices]
        if len(idx) != len(views):
            raise ValueError("camera_indices must match views")
        count = (max(idx) + 1) if idx else 0
        if sorted_entries is None:
            sorted_entries = SortedEntries.for_cloud(count, handle.n)
        if count > sorted_entries.camera_count:
            raise ValueError("sorted_entries has fewer camera chunks than the camera indices need")
        for view, ci in zip(views, idx):
            chunk = sorted_entries.chunk(ci, handle.n)
            v, s = view.to_native(), settings.to_native()
            tmp = np.empty(handle.n, dtype=SORT_ENTRY_DTYPE)
            self._check(self._lib.bgs_sort(self._ctx, handle._ptr, ctypes.byref(v), ctypes.byref(s),
                                           tmp.ctypes.data_as(ctypes.POINTER(_native.BgsSortEntry))))
            chunk[:] = tmp
        return sorted_entries

    def prepare(self, view: View, settings: CloudSettings) -> "PreparedView":
        """Marshal a (view, settings) pair once (the per-call conversion to the C structs costs ~20 us
        of host time, a fifth of a pipelined frame). Pass the result as `view` with `settings=None`."""
        return PreparedView(view.to_native(), settings.to_native(), view.width, view.height)

    def render(self, handle: PlanarGaussian3dHandle, view, settings: Optional[CloudSettings] = None,
               download: bool = True) -> Optional[np.ndarray]:
        """Sort + project + bin + rasterize one view. Returns [H, W, 4] float32
        (premultiplied linear RGBA, unclamped, row 0 = top) or None if not downloaded.
        `view` is a View (with `settings`) or a PreparedView from `prepare()`."""
        if isinstance(view, PreparedView):
            v, s = view.view, view.settings
        else:
            v, s = view.to_native(), settings.to_native()
        if download:
            out = np.empty((view.height, view.width, 4), dtype=np.float32)
            ptr = _fptr(out)
        else:
            out, ptr = None, None
        self._check(self._lib.bgs_render(self._ctx, handle._ptr, ctypes.byref(v), ctypes.byref(s), ptr))
        return out

    def radix_sort_pairs(self, keys: np.ndarray, passes: int = 4) -> np.ndarray:
        """Run the device Onesweep kernel on arbitrary (key, index=i) pairs (test hook)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        e = np.empty(keys.shape[0], dtype=SORT_ENTRY_DTYPE)
        e["key"] = keys
        e["index"] = np.arange(keys.shape[0], dtype=np.uint32)
        self._check(
            self._lib.bgs_radix_sort_pairs(
                self._ctx, e.ctypes.data_as(ctypes.POINTER(_native.BgsSortEntry)),
                keys.shape[0], int(passes)))
        return e

    def hbm_probe(self, nbytes: int = 1 << 29, iters: int = 20):
        """Measured HBM ceiling: (DtoD-copy GB/s counting read+write, triad GB/s) — for the roofline."""
        c, t = ctypes.c_float(), ctypes.c_float()
        self._check(self._lib.bgs_hbm_probe(self._ctx, int(nbytes), int(iters), ctypes.byref(c), ctypes.byref(t)))
        return float(c.value), float(t.value)

    def selftest_ln(self, first_bits: int, count: int, download: Optional[bool] = None):
        """`bgs_selftest_ln_f32`: the correctly rounded ln of the adaptive cutoff evaluated ON THE DEVICE for the
        binary32 bit patterns first_bits .. first_bits + count - 1. Returns (results or None, checksum); the results
        are downloaded for count <= 2^24 unless `download` says otherwise."""
        if download is None:
            download = count <= (1 << 24)
        out = np.empty(count, np.float32) if download else None
        chk = ctypes.c_uint64()
        self._check(self._lib.bgs_selftest_ln_f32(
            self._ctx, int(first_bits), int(count),
            out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if download else None, ctypes.byref(chk)))
        return out, int(chk.value)

    def set_pipeline_streams(self, streams: int) -> None:
        """HIP streams the lanes are multiplexed onto (0 = one per lane); see bgs_set_pipeline_streams."""
        self._check(self._lib.bgs_set_pipeline_streams(self._ctx, int(streams)))

    def set_graphs(self, enabled: bool) -> None:
        """Replay steady-state async frames from a captured hipGraph (default off; opt-in for CPU-bound hosts; bgs_set_graphs)."""
        self._check(self._lib.bgs_set_graphs(self._ctx, 1 if enabled else 0))

    def graph_counters(self) -> tuple:
        """(frames captured into a graph, frames replayed from one) since the plugin was created."""
        c, r = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.bgs_graph_counters(self._ctx, ctypes.byref(c), ctypes.byref(r)))
        return int(c.value), int(r.value)

    def tile_order_counters(self) -> tuple:
        """(frames that left per-tile costs, frames whose raster workgroups ran in cost order, times that order was made
        anew) since the plugin was created."""
        c, r, n = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.bgs_tile_order_counters(self._ctx, ctypes.byref(c), ctypes.byref(r), ctypes.byref(n)))
        return int(c.value), int(r.value), int(n.value)

    def selftest_tile_order(self, cost: np.ndarray, runs: int = 1, sums: bool = False):
        """`bgs_selftest_tile_order`: the raster workgroups' order tile_order_kernel makes of per-tile costs (uint16,
        one per tile): uint16[(tiles + 3) // 4] — with `sums` also (work of all tiles, work of the tiles with bit 15 set)."""
        c = np.ascontiguousarray(cost, dtype=np.uint16)
        out = np.empty((c.size + 3) // 4, np.uint16)
        two = np.zeros(2, np.uint32)
        self._check(self._lib.bgs_selftest_tile_order(self._ctx, c.ctypes.data_as(ctypes.c_void_p), c.size, runs,
                                                      out.ctypes.data_as(ctypes.c_void_p), two.ctypes.data_as(ctypes.c_void_p)))
        return (out, (int(two[0]), int(two[1]))) if sums else out

    # -- interop / introspection -----------------------------------------------------
    def synchronize(self) -> None:
        self._check(self._lib.bgs_synchronize(self._ctx))

    def set_profiling(self, level: int) -> None:
        """0 = no HIP events, 1 = frame start/end only, 2 = every stage (default)."""
        self._check(self._lib.bgs_set_profiling(self._ctx, int(level)))

    def set_profiling_stride(self, every_nth_frame: int) -> None:
        """Record HIP events only on every Nth frame (each record costs ~4 us of GPU timeline)."""
        self._check(self._lib.bgs_set_profiling_stride(self._ctx, int(every_nth_frame)))

    def set_pipeline_depth(self, lanes: int) -> None:
        """Frames in flight (1..8 lanes = per-frame buffer sets, multiplexed onto set_pipeline_streams() HIP
        streams); see bgs_set_pipeline_depth."""
        self._check(self._lib.bgs_set_pipeline_depth(self._ctx, int(lanes)))

    def set_output_srgb8(self, enabled: bool) -> None:
        """Also produce every frame as Rgba8UnormSrgb (the reference's target format)."""
        self._check(self._lib.bgs_set_output_srgb8(self._ctx, 1 if enabled else 0))

    def set_srgb8_target(self, device_ptr: Optional[int]) -> None:
        """The next `render` writes its Rgba8UnormSrgb image to this device address (one-shot)."""
        self._check(self._lib.bgs_set_srgb8_target(self._ctx, ctypes.c_void_p(device_ptr or 0)))

    def set_output_rgba16f(self, enabled: bool) -> None:
        """Also produce every frame as Rgba16Float (the reference's hdr colour attachment); exclusive with sRGB8."""
        self._check(self._lib.bgs_set_output_rgba16f(self._ctx, 1 if enabled else 0))

    def set_packed_only(self, enabled: bool) -> None:
        """Frames that write a packed image (sRGB8 / Rgba16Float) skip the f32 target."""
        self._check(self._lib.bgs_set_packed_only(self._ctx, 1 if enabled else 0))

    def framebuffer_rgba16f_device_ptr(self):
        p = ctypes.c_void_p()
        nbytes = ctypes.c_uint64()
        self._check(self._lib.bgs_framebuffer_rgba16f_device_ptr(self._ctx, ctypes.byref(p), ctypes.byref(nbytes)))
        return p.value, nbytes.value

    def framebuffer_srgb8_device_ptr(self):
        p = ctypes.c_void_p()
        nbytes = ctypes.c_uint64()
        self._check(self._lib.bgs_framebuffer_srgb8_device_ptr(self._ctx, ctypes.byref(p), ctypes.byref(nbytes)))
        return p.value, nbytes.value

    def pipeline_pop(self):
        """Complete the oldest frame in flight; returns (f32 framebuffer ptr, srgb8 ptr or None)."""
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.bgs_pipeline_pop(self._ctx, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def frames_in_flight(self) -> int:
        c = ctypes.c_uint32()
        self._check(self._lib.bgs_frames_in_flight(self._ctx, ctypes.byref(c)))
        return int(c.value)

    def set_async(self, enabled: bool) -> None:
        """Async frames: render(download=False) only enqueues (scan binning); see bgs_set_async."""
        self._check(self._lib.bgs_set_async(self._ctx, 1 if enabled else 0))

    def set_binning(self, mode: str) -> None:
        """'scan' (default): ordered coarse lists + lazy per-tile scan; 'sort': (tile, splat)
        instances + stable radix sort on the tile id. Images are bit-identical."""
        self._check(self._lib.bgs_set_binning(self._ctx, {"scan": 0, "sort": 1}[mode]))

    def set_debug_flags(self, flags: int) -> None:
        """Kernel-ablation switches for experiments only (non-zero => wrong images)."""
        self._check(self._lib.bgs_set_debug_flags(self._ctx, int(flags)))

    def draw_list(self) -> np.ndarray:
        """The sorted entries the last call left on the device (`bgs_sorted_entries_device_ptr` + `bgs_download`): after
        `sort` all n entries, after `render` the drawable prefix (what reaches the vertex stage, back to front)."""
        p, n = ctypes.c_void_p(), ctypes.c_uint32()
        self._check(self._lib.bgs_sorted_entries_device_ptr(self._ctx, ctypes.byref(p), ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.dtype([("key", np.uint32), ("index", np.uint32)]))
        if n.value:
            self._check(self._lib.bgs_download(self._ctx, p, out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
        return out

    def device_alloc(self, nbytes: int) -> int:
        """`bgs_device_alloc`: device memory for what the caller hands over by device pointer (a depth buffer, a packed
        image target). Returns the address."""
        p = ctypes.c_void_p()
        self._check(self._lib.bgs_device_alloc(self._ctx, int(nbytes), ctypes.byref(p)))
        return int(p.value or 0)

    def device_free(self, device_ptr: int) -> None:
        self._check(self._lib.bgs_device_free(self._ctx, ctypes.c_void_p(device_ptr)))

    def upload_bytes(self, device_ptr: int, host: np.ndarray) -> None:
        """`bgs_upload`: blocking host-to-device copy of a contiguous array."""
        a = np.ascontiguousarray(host)
        self._check(self._lib.bgs_upload(self._ctx, ctypes.c_void_p(device_ptr), a.ctypes.data_as(ctypes.c_void_p), a.nbytes))

    def download(self, device_ptr: int, host_out: np.ndarray) -> np.ndarray:
        """`bgs_download`: blocking device-to-host copy into a contiguous array (of a COMPLETED frame's memory)."""
        if not host_out.flags["C_CONTIGUOUS"] or not host_out.flags["WRITEABLE"]:
            raise ValueError("host_out must be a writeable C-contiguous array")
        self._check(self._lib.bgs_download(self._ctx, ctypes.c_void_p(device_ptr), host_out.ctypes.data_as(ctypes.c_void_p),
                                           host_out.nbytes))
        return host_out

    def upload_depth(self, depth: np.ndarray) -> int:
        """Put a view's scene depth on the device: `depth` is [height, width, samples] float32 (reverse-Z, what Bevy's
        opaque passes left in the view's Depth32Float attachment; src/render/mod.rs:959-974). Returns the device
        address for `View.depth_device_ptr`; release it with `device_free` once the frames that use it are complete."""
        d = np.ascontiguousarray(depth, dtype=np.float32)
        if d.ndim != 3 or d.shape[2] not in (1, 2, 4, 8):
            raise ValueError("depth must be [height, width, samples] with 1, 2, 4 or 8 samples")
        p = self.device_alloc(d.nbytes)
        self.upload_bytes(p, d)
        return p

    def set_tile_trace(self, device_ptr: Optional[int]) -> None:
        """`bgs_set_tile_trace`: per-tile timing / placement trace of the rasteriser into a caller-owned device buffer
        (tiles_x * tiles_y * 32 bytes); None switches it off."""
        self._check(self._lib.bgs_set_tile_trace(self._ctx, ctypes.c_void_p(device_ptr or 0)))

    def build_id(self) -> str:
        """`bgs_build_id()`: SHA-256 of the kernel sources the loaded library was compiled from."""
        return self._lib.bgs_build_id().decode()

    def adaptive_counters(self) -> dict:
        """Cumulative counters of the adaptive machinery (bgs_adaptive_counters)."""
        out = (ctypes.c_uint64 * 8)()
        self._check(self._lib.bgs_adaptive_counters(self._ctx, out))
        names = ("bucket_frames", "onesweep_frames", "reruns_sort", "reruns_lists", "reruns_instances",
                 "level_changes", "supertile_level", "list_capacity_hint")
        return {k: int(v) for k, v in zip(names, out)}

    def learning_counters(self) -> dict:
        """Async frames completed inside their render call because their kind of frame was new to the context, and the
        kinds it has settled on (bgs_learning_counters)."""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.bgs_learning_counters(self._ctx, ctypes.byref(a), ctypes.byref(b)))
        return {"early_frames": a.value, "kinds_settled": b.value}

    # -- the multi-GPU frame gather (bgs_comm_*: RCCL behind the C ABI) ---------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128 bytes rank 0 ships to every rank (by any means) before `comm_create`."""
        buf = ctypes.create_string_buffer(_native.COMM_ID_BYTES)
        lib = _native.load()
        _native.check(lib, None, lib.bgs_comm_unique_id(buf))
        return buf.raw

    def comm_create(self, unique_id: bytes, world_size: int, rank: int) -> int:
        """Collective: returns the communicator handle once every rank has called it."""
        if len(unique_id) != _native.COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        out = ctypes.c_void_p()
        self._check(self._lib.bgs_comm_create(self._ctx, unique_id, int(world_size), int(rank), ctypes.byref(out)))
        return out.value

    def comm_gather(self, comm: int, root: int, send_ptr: int, nbytes: int, recv_ptr: Optional[int]) -> int:
        """Enqueue one gather of `nbytes` per rank (device pointers; `recv_ptr` on the root only). Returns its ticket."""
        ticket = ctypes.c_uint64()
        self._check(self._lib.bgs_comm_gather(self._ctx, comm, int(root), ctypes.c_void_p(send_ptr), int(nbytes),
                                              ctypes.c_void_p(recv_ptr or 0), ctypes.byref(ticket)))
        return ticket.value

    def comm_gather_after(self, comm: int, root: int, send_ptr: int, nbytes: int, recv_ptr: Optional[int],
                          hip_stream: Optional[int] = None) -> int:
        """`bgs_comm_gather_after`: the gather ordered ON THE DEVICE behind everything enqueued so far on `hip_stream` (None:
        behind every frame this context still has in flight) — no pop / synchronise before the batch goes out. Returns the
        ticket. A frame re-run later (adaptive_counters: reruns_*) went out in the state of its first attempt."""
        t = ctypes.c_uint64(0)
        self._check(self._lib.bgs_comm_gather_after(self._ctx, comm, int(root), ctypes.c_void_p(send_ptr), int(nbytes),
                                                    ctypes.c_void_p(recv_ptr or 0), ctypes.c_void_p(hip_stream or 0), ctypes.byref(t)))
        return int(t.value)

    def comm_wait(self, comm: int, ticket: int = 0) -> None:
        """Block until the gather with `ticket` (and every earlier one) has completed; 0 = all of them."""
        self._check(self._lib.bgs_comm_wait(self._ctx, comm, int(ticket)))

    def comm_destroy(self, comm: int) -> None:
        self._lib.bgs_comm_destroy(self._ctx, comm)

    def reset_adaptive_state(self) -> None:
        """Forget the hints completed frames left in the context (draw count, key range, list capacity,
        supertile rule): the next frames behave like the first frames of a fresh context."""
        self._check(self._lib.bgs_reset_adaptive_state(self._ctx))

    def framebuffer_device_ptr(self):
        p = ctypes.c_void_p()
        nbytes = ctypes.c_uint64()
        self._check(self._lib.bgs_framebuffer_device_ptr(self._ctx, ctypes.byref(p), ctypes.byref(nbytes)))
        return p.value, nbytes.value

    def stream_handle(self) -> int:
        p = ctypes.c_void_p()
        self._check(self._lib.bgs_stream(self._ctx, ctypes.byref(p)))
        return p.value or 0

    def stats(self) -> dict:
        st = _native.BgsStats()
        self._check(self._lib.bgs_get_stats(self._ctx, ctypes.byref(st)))
        d = {
            "total_ms": float(st.total_ms),
            "splat_count": int(st.splat_count),
            "visible_count": int(st.visible_count),
            "draw_count": int(st.draw_count),
            "instance_count": int(st.instance_count),
            "instance_capacity": int(st.instance_capacity),
            "list_entries_allocated": int(st.list_entries_allocated),
            "strip_tiles": int(st.strip_tiles),
            "tile_saturation": {"known": bool((int(st.tile_saturation) >> 16) & 1), "work_share": (int(st.tile_saturation) & 0x7FFF) / 0x7FFF,
                                "midround_exit": bool(int(st.tile_saturation) >> 31)},
            "tiles": (int(st.tiles_x), int(st.tiles_y)),
            "depth_passes": int(st.depth_passes),
            "tile_passes": int(st.tile_passes),
            "algorithmic_bytes": int(st.algorithmic_bytes),
            "regrow_count": int(st.regrow_count),
            "sort_path": "bucket" if st.sort_path else "onesweep",
            "list_capacity": int(st.list_capacity),
            "binning": "sort" if st.binning_mode else "scan",
            "frames_averaged": int(st.frames_averaged),
            "stage_ms": {n: float(st.stage_ms[i]) for i, n in enumerate(_native.STAGE_NAMES)},
        }
        return d
