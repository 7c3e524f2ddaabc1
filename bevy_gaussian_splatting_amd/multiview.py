"""Multi-GPU: independent views (cameras) sharded one per GPU, one gather of framebuffers.

The reference keys its sort state by camera (`SortTrigger.camera_index`, src/sort/mod.rs:143-150;
per-camera chunk of the entry buffer, src/render/mod.rs:1548-1554) and has no collective of any
kind. Views are independent units, so the path shards with NO data-path collective: rank g owns
camera g and a full replica of the cloud; the only exchange is the final gather of the
framebuffers to rank 0 (RCCL over xGMI on GPUs: every non-root rank sends on its own link; gloo
in the CPU tests).
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np

from .camera import View


def assign_views(num_views: int, world_size: int) -> List[List[int]]:
    """View g goes to rank g % world_size (one view per GPU when num_views == world_size)."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    out: List[List[int]] = [[] for _ in range(world_size)]
    for g in range(num_views):
        out[g % world_size].append(g)
    return out


def headless_view(g: int, width: int = 1920, height: int = 1080) -> View:
    """SURVEY 8(d) cfg 5: camera g = the examples/headless.rs camera yawed by g * 45 degrees about
    +Y at the same position; `order` = g is the camera's index into the sorted entries."""
    return View.headless(width, height, yaw=g * math.pi / 4.0, order=g)


def gather_framebuffers(local, dst: int = 0, group=None):
    """Gather every rank's framebuffer tensor ([k, H, W, 4] float32, same shape on all ranks)
    to `dst`. Returns the list of per-rank tensors on `dst`, None elsewhere. Works on any
    torch.distributed backend (nccl == RCCL on ROCm, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local]
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    local = local.contiguous()
    if rank == dst:
        bufs = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, gather_list=bufs, dst=dst, group=group)
        return bufs
    dist.gather(local, gather_list=None, dst=dst, group=group)
    return None


class BatchedFrameGather:
    """Gather frames to rank `dst` in batches: one collective per `batch` frames, asynchronous and
    double-buffered, so that neither the collective's latency (~100 us per call) nor its transfer time
    (8.3 MB per 1080p Rgba8UnormSrgb frame per xGMI link) stalls the renderer. `push(frame)` copies the
    frame into the current staging batch (the caller may reuse the frame's memory as soon as push
    returns); a full batch is sent with `gather(..., async_op=True)` while the other staging buffer
    fills. `flush()` sends a partial batch and waits for everything. On `dst`, `on_batch(list of
    per-rank [count, ...] tensors)` is called for every completed batch (default: count frames).
    Works on any backend (nccl == RCCL on ROCm; gloo in the CPU tests)."""

    def __init__(self, shape, dtype, device, batch: int = 8, dst: int = 0, group=None, on_batch=None):
        import torch
        import torch.distributed as dist

        self.dist, self.torch = dist, torch
        self.group, self.dst = group, dst
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.world = dist.get_world_size(group) if self.active else 1
        self.batch = max(1, int(batch))
        self.stage = [torch.empty((self.batch, *shape), dtype=dtype, device=device) for _ in range(2)]
        self.recv = [[torch.empty_like(self.stage[0]) for _ in range(self.world)] if self.rank == dst else None
                     for _ in range(2)]
        self.work = [None, None]
        self.count_in_flight = [0, 0]
        self.pushed = 0   # frames complete in their slot
        self.issued = 0   # slots handed out by next_target()
        self.frames_received = 0  # on dst: frames of all ranks that have arrived
        self.on_batch = on_batch
        self._is_cuda = torch.device(device).type == "cuda"

    def _complete(self, s: int) -> None:
        if self.work[s] is None:
            return
        self.work[s].wait()
        if self._is_cuda:
            self.torch.cuda.current_stream().synchronize()
        self.work[s] = None
        if self.rank == self.dst:
            k = self.count_in_flight[s]
            self.frames_received += k * self.world
            if self.on_batch is not None:
                self.on_batch([t[:k] for t in self.recv[s]])

    def _send(self, s: int, count: int) -> None:
        self.count_in_flight[s] = count
        if not self.active:
            self.frames_received += count
            if self.on_batch is not None:
                self.on_batch([self.stage[s][:count]])
            return
        self.work[s] = self.dist.gather(self.stage[s], gather_list=self.recv[s], dst=self.dst, group=self.group,
                                        async_op=True)

    # -- zero-copy interface: the producer writes each frame straight into its slot -------------
    def next_target(self):
        """Slot ([*shape] tensor view) the NEXT frame to be enqueued must be written to (e.g. via
        `plugin.set_srgb8_target(slot.data_ptr())`). Frames may be enqueued up to `batch` ahead of
        `frame_completed()` calls."""
        s, slot = (self.issued // self.batch) % 2, self.issued % self.batch
        if slot == 0:
            self._complete(s)  # the previous gather out of this staging buffer must have finished
        self.issued += 1
        return self.stage[s][slot]

    def frame_completed(self) -> None:
        """The oldest outstanding frame (in `next_target` order) is now complete in its slot."""
        self.pushed += 1
        if self.pushed % self.batch == 0:
            self._send((self.pushed // self.batch - 1) % 2, self.batch)

    def push(self, frame) -> None:
        s, slot = (self.pushed // self.batch) % 2, self.pushed % self.batch
        if slot == 0:
            self._complete(s)  # the staging buffer is about to be overwritten
        self.stage[s][slot].copy_(frame)
        if self._is_cuda:
            self.torch.cuda.current_stream().synchronize()  # the caller's frame memory may now be reused
        self.pushed += 1
        self.issued = self.pushed
        if slot == self.batch - 1:
            self._send(s, self.batch)

    def flush(self) -> None:
        s, slot = (self.pushed // self.batch) % 2, self.pushed % self.batch
        if slot:
            self._send(s, slot)
            self.pushed += self.batch - slot  # keep the batch phase of every rank aligned
            self.issued = self.pushed
            self._complete(1 - s)  # completion in send order: the other buffer went out first
            self._complete(s)
        else:
            self._complete(s)      # buffer s would be filled next, so it holds the older batch
            self._complete(1 - s)


class _DeviceArray:
    """Zero-copy view of a raw device pointer for torch (CUDA array interface v2)."""

    def __init__(self, ptr: int, shape, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_ptr_as_tensor(ptr: int, shape, typestr: str, device: str):
    """Zero-copy torch view of raw device memory (e.g. a popped pipeline frame)."""
    import torch

    return torch.as_tensor(_DeviceArray(ptr, shape, typestr), device=device)


def framebuffer_as_tensor(plugin, height: int, width: int, device: Optional[str] = None):
    """Wrap the device framebuffer of the last `render` as a torch tensor [H, W, 4] without a
    copy (so RCCL can send it straight from where the rasteriser wrote it)."""
    import torch

    ptr, nbytes = plugin.framebuffer_device_ptr()
    assert nbytes == height * width * 16
    return torch.as_tensor(_DeviceArray(ptr, (height, width, 4)), device=device or f"cuda:{plugin.device}")


class NativeFrameGather:
    """`BatchedFrameGather` without torch: the same double-buffered, batched gather of packed frames to rank `dst`,
    through the C ABI (`bgs_comm_*`: RCCL's ncclGather on a stream the communicator owns) — what a Rust host binding
    include/bgs.h runs for BASELINE configs[4]. `unique_id` = `GaussianSplattingPlugin.comm_unique_id()` of rank 0,
    shipped to every rank by the launcher (bench.py: one torch.distributed broadcast at start-up; the tests: a file).
    The producer writes each frame straight into its slot: `next_target()` is the device pointer for the NEXT frame to
    be enqueued (`plugin.set_srgb8_target`), `frame_completed()` says the oldest outstanding frame is complete in its
    slot (after `plugin.pipeline_pop()`), `flush()` sends a partial batch and waits for everything. On `dst`,
    `on_batch(recv_ptr, count)` is called for every completed batch: rank r's frames are at
    recv_ptr + (r * batch + k) * frame_bytes, k < count."""

    def __init__(self, plugin, frame_bytes: int, world: int, rank: int, unique_id: bytes, batch: int = 8, dst: int = 0,
                 on_batch=None, device_wait: bool = False):
        self.plugin, self.frame_bytes, self.world, self.rank, self.dst = plugin, int(frame_bytes), int(world), int(rank), int(dst)
        self.batch = max(1, int(batch))
        # device_wait (round 6, `bgs_comm_gather_after`): a batch goes out as soon as its last frame is ENQUEUED — the
        # gather waits on the device for the context's frames in flight — instead of once its frames were popped; the
        # producer calls frame_enqueued() after every render and pops frames whenever it likes (frame_completed() then only
        # counts). A frame the library re-ran after its batch went out is counted in `stale_frames` (watch
        # adaptive_counters' reruns_* around the pops and call note_rerun()).
        self.device_wait = bool(device_wait)
        self.enqueued = 0
        self.stale_frames = 0
        self.comm = plugin.comm_create(unique_id, world, rank)
        self.stage = [plugin.device_alloc(self.batch * self.frame_bytes) for _ in range(2)]
        self.recv = [plugin.device_alloc(self.world * self.batch * self.frame_bytes) if rank == dst else None for _ in range(2)]
        self.ticket = [0, 0]
        self.count_in_flight = [0, 0]
        self.pushed = 0
        self.issued = 0
        self.frames_received = 0
        self.gathers = 0
        self.on_batch = on_batch

    def _complete(self, s: int) -> None:
        if not self.ticket[s]:
            return
        self.plugin.comm_wait(self.comm, self.ticket[s])
        self.ticket[s] = 0
        if self.rank == self.dst:
            k = self.count_in_flight[s]
            self.frames_received += k * self.world
            if self.on_batch is not None:
                self.on_batch(self.recv[s], k)

    def _send(self, s: int, count: int) -> None:
        # every rank sends the whole staging batch (a partial batch only at the very end): one message size per
        # collective on every rank, whatever `count` is
        self.count_in_flight[s] = count
        gather = self.plugin.comm_gather_after if self.device_wait else self.plugin.comm_gather
        self.ticket[s] = gather(self.comm, self.dst, self.stage[s], self.batch * self.frame_bytes, self.recv[s])
        self.gathers += 1

    def next_target(self) -> int:
        s, slot = (self.issued // self.batch) % 2, self.issued % self.batch
        if slot == 0:
            # The buffer is about to be overwritten: every frame of its previous use must have gone out. That is the case
            # when at most `batch` frames are outstanding — then they all sit in the OTHER buffer. (Round 5's advisor:
            # with a pipeline deeper than `batch` the previous batch of this buffer had not been sent yet, its ticket
            # was 0, and next_target handed the slot out over frames nobody had gathered.)
            sent = self.enqueued if self.device_wait else self.pushed
            if self.issued - sent > self.batch:
                raise RuntimeError(f"NativeFrameGather: {self.issued - sent} frames outstanding with batch {self.batch}: the staging "
                                   "ring holds two batches — complete frames before asking for more slots, or gather larger batches "
                                   "(batch >= the pipeline depth)")
            self._complete(s)  # the previous gather out of this staging buffer must have finished
        self.issued += 1
        return self.stage[s] + slot * self.frame_bytes

    def frame_enqueued(self) -> None:
        """device_wait mode: the frame that was given the last next_target() slot has been enqueued (bgs_render returned)."""
        self.enqueued += 1
        if self.device_wait and self.enqueued % self.batch == 0:
            self._send((self.enqueued // self.batch - 1) % 2, self.batch)

    def note_rerun(self, frames: int = 1) -> None:
        """device_wait mode: the library re-ran `frames` frames whose batch may already have gone out."""
        self.stale_frames += int(frames)

    def frame_completed(self) -> None:
        self.pushed += 1
        if not self.device_wait and self.pushed % self.batch == 0:
            self._send((self.pushed // self.batch - 1) % 2, self.batch)

    def flush(self) -> None:
        if self.device_wait:
            self.pushed = self.enqueued   # (the batch phase follows the frames ENQUEUED; the caller has synchronised / popped them)
        s, slot = (self.pushed // self.batch) % 2, self.pushed % self.batch
        if slot:
            self._send(s, slot)
            self.pushed += self.batch - slot  # keep the batch phase of every rank aligned
            self.issued = self.enqueued = self.pushed
            self._complete(1 - s)
            self._complete(s)
        else:
            self._complete(s)
            self._complete(1 - s)

    def close(self) -> None:
        if self.comm is None:
            return
        self.plugin.comm_wait(self.comm, 0)
        self.plugin.comm_destroy(self.comm)
        self.comm = None
        for p in self.stage + [r for r in self.recv if r]:
            self.plugin.device_free(p)
