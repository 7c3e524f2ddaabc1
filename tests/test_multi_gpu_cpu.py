"""N > 1 path on CPU: 2 processes, gloo backend. Each rank renders ITS camera (with the oracle,
standing in for the GPU renderer — this test covers the sharding + gather logic bench.py uses,
not the kernels) and rank 0 gathers the framebuffers."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_views, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from bevy_gaussian_splatting_amd import CloudSettings, random_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.multiview import assign_views, gather_framebuffers, headless_view

    cloud = random_gaussians_3d_seeded(800, 21)  # replicated on every rank
    s = CloudSettings(global_scale=0.3)
    mine = assign_views(num_views, world)[rank]
    frames = []
    for g in mine:
        v = headless_view(g, 48, 32)
        frames.append(oracle.render(cloud, oracle.sort(cloud, v, s), v, s))
    local = torch.from_numpy(np.stack(frames))
    dist.barrier()
    got = gather_framebuffers(local, dst=0)
    if rank == 0:
        q.put([t.numpy() for t in got])
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_views_shard_one_per_rank_and_gather_to_rank0():
    world, num_views = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_views, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from bevy_gaussian_splatting_amd import CloudSettings, random_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.multiview import assign_views, headless_view

    cloud = random_gaussians_3d_seeded(800, 21)
    s = CloudSettings(global_scale=0.3)
    plan = assign_views(num_views, world)
    assert len(gathered) == world
    for r in range(world):
        assert gathered[r].shape == (len(plan[r]), 32, 48, 4)
        for k, g in enumerate(plan[r]):
            v = headless_view(g, 48, 32)
            ref = oracle.render(cloud, oracle.sort(cloud, v, s), v, s)
            assert np.array_equal(gathered[r][k], ref)
    # different cameras really produce different images
    assert not np.array_equal(gathered[0][0], gathered[1][0])


def _batched_worker(rank, world, port, frames, batch, zero_copy, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bevy_gaussian_splatting_amd.multiview import BatchedFrameGather

    got = []
    g = BatchedFrameGather((4, 6, 4), torch.uint8, "cpu", batch=batch,
                           on_batch=lambda per_rank: got.append([t.clone() for t in per_rank]))
    if zero_copy:
        # producer writes straight into the slot (bgs_set_srgb8_target), up to 3 frames ahead of completion
        pending = []
        for i in range(frames):
            slot = g.next_target()
            pending.append((slot, (17 * rank + i) % 251))
            if len(pending) == 3:
                t, val = pending.pop(0)
                t.fill_(val)
                g.frame_completed()
        for t, val in pending:
            t.fill_(val)
            g.frame_completed()
    else:
        frame = torch.empty((4, 6, 4), dtype=torch.uint8)
        for i in range(frames):
            frame.fill_((17 * rank + i) % 251)   # the SAME buffer is overwritten every frame, like a lane's
            g.push(frame)
    g.flush()
    if rank == 0:
        q.put((g.frames_received, [[t.numpy() for t in b] for b in got]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("zero_copy", [False, True])
@pytest.mark.parametrize("frames,batch", [(11, 4), (8, 4), (3, 8)])
def test_batched_asynchronous_frame_gather(frames, batch, zero_copy):
    """bench.py's gather for N > 1: one collective per `batch` frames, double-buffered; every frame of
    every rank arrives once, in order, including the partial last batch."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batched_worker, args=(r, world, port, frames, batch, zero_copy, q)) for r in range(world)]
    for p in procs:
        p.start()
    received, batches = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert received == frames * world
    for r in range(world):
        seq = np.concatenate([b[r] for b in batches])
        assert seq.shape == (frames, 4, 6, 4)
        assert [int(f[0, 0, 0]) for f in seq] == [(17 * r + i) % 251 for i in range(frames)]


class _FakePlugin:
    """What bench.measure needs from the plugin, without a GPU: frames are 'rendered' into the staging slot the
    gather handed out; `delay` makes one rank slower than the other."""

    def __init__(self, delay):
        self.delay, self.in_flight, self.target, self.rendered = delay, [], None, 0

    def prepare(self, view, settings):
        return (view, settings)

    def set_target(self, t):
        self.target = t

    def render(self, handle, prepared, download=False):
        import time
        time.sleep(self.delay)
        self.target.fill_(self.rendered % 251)
        self.rendered += 1
        self.in_flight.append(self.target)

    def frames_in_flight(self):
        return len(self.in_flight)

    def pipeline_pop(self):
        self.in_flight.pop(0)
        return None, None

    def synchronize(self):
        pass

    def stats(self):
        return {"stage_ms": {}}


def _measure_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from bevy_gaussian_splatting_amd.multiview import BatchedFrameGather

    plugin = _FakePlugin(delay=0.0004 if rank == 0 else 0.0015)   # rank 1 is ~4x slower
    batcher = BatchedFrameGather((2, 3, 4), torch.uint8, "cpu", batch=4)

    def gather(f32_ptr, srgb8_ptr):
        batcher.frame_completed()

    gather.flush = batcher.flush
    gather.before_render = lambda: plugin.set_target(batcher.next_target())
    dt, _, _, dts = bench.measure(plugin, None, None, None, steps=5, warmup=3, gather=gather, barrier=dist.barrier,
                                  depth=2, trials=3, busy_warm_frames=17)
    t = torch.tensor(dts, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((plugin.rendered, batcher.frames_received, len(dts), t.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bench_measure_issues_the_same_collectives_on_every_rank():
    """bench.measure with N > 1: warm-up, extra warm-up frames and the timed trials are all frame COUNTS, so a
    slow and a fast rank run the same number of frames and gathers (a time-based warm-up would leave the
    ranks with different numbers of collectives: a hang). Every frame of every rank arrives on rank 0."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_measure_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rendered, received, ntrials, dts = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # depth + warmup + extra warm-up (17 frames in chunks of max(steps, 4 * depth) = 8 -> 3 chunks) + 3 trials
    expected = 2 + 3 + 3 * 8 + 3 * 5
    assert rendered == expected and received == expected * world and ntrials == 3
    assert all(d > 5 * 0.0015 * 0.5 for d in dts)   # a trial lasts as long as its slowest rank


class _FakeCommPlugin:
    """The plugin surface NativeFrameGather uses (device_alloc / comm_*), on host memory: ranks of one process share a
    mailbox, a gather copies every rank's staging batch into the root's receive buffer when the last rank has called."""

    mailbox = {}

    def __init__(self, rank, world):
        self.rank, self.world, self.mem, self.next_ptr, self.tickets, self.waited = rank, world, {}, 0x1000, 0, []

    def device_alloc(self, nbytes):
        p = self.next_ptr
        self.mem[p] = np.zeros(nbytes, np.uint8)
        self.next_ptr += (nbytes + 0xFFF) & ~0xFFF
        return p

    def device_free(self, p):
        self.mem.pop(p)

    def view(self, p, n):
        base = max(b for b in self.mem if b <= p)
        return self.mem[base][p - base:p - base + n]

    def comm_create(self, uid, world, rank):
        assert len(uid) == 128 and world == self.world and rank == self.rank
        return 7

    def comm_gather(self, comm, root, send, nbytes, recv):
        self.tickets += 1
        box = _FakeCommPlugin.mailbox.setdefault(self.tickets, {})
        box[self.rank] = (self.view(send, nbytes).copy(), recv, self)
        if len(box) == self.world:
            data_root, recv_root, plug_root = box[root]
            assert recv_root is not None
            for r in range(self.world):
                plug_root.view(recv_root + r * nbytes, nbytes)[:] = box[r][0]
        return self.tickets

    def comm_gather_after(self, comm, root, send, nbytes, recv, hip_stream=None):
        self.after_calls = getattr(self, "after_calls", 0) + 1   # (the device-side wait is the library's: nothing to fake)
        return self.comm_gather(comm, root, send, nbytes, recv)

    def comm_wait(self, comm, ticket=0):
        self.waited.append(ticket)

    def comm_destroy(self, comm):
        pass


def test_native_frame_gather_batches_and_flushes_like_the_torch_one():
    """NativeFrameGather (the bgs_comm_* loop bench.py runs for N > 1) on a fake two-rank communicator: zero-copy slots,
    one gather per `batch` completed frames, double buffering (the ticket of the gather out of a staging buffer is waited
    for before that buffer is handed out again), a partial batch at flush, every frame of every rank on the root in order."""
    from bevy_gaussian_splatting_amd.multiview import NativeFrameGather
    _FakeCommPlugin.mailbox = {}
    frame_bytes, batch, frames, world = 64, 4, 11, 2
    got = []
    plugs = [_FakeCommPlugin(r, world) for r in range(world)]

    def on_batch(recv_ptr, count):
        blk = plugs[0].view(recv_ptr, world * batch * frame_bytes).reshape(world, batch, frame_bytes)
        got.append(blk[:, :count].copy())

    gs = [NativeFrameGather(plugs[r], frame_bytes, world, r, bytes([1]) * 128, batch=batch, on_batch=on_batch if r == 0 else None)
          for r in range(world)]
    for i in range(frames):
        for r in reversed(range(world)):  # (two ranks in lock step, the root last: every rank issues the same collectives)
            slot = gs[r].next_target()
            plugs[r].view(slot, frame_bytes)[:] = (17 * r + i) % 251
            gs[r].frame_completed()
    for r in reversed(range(world)):     # (the fake's wait does not block: the root flushes last, when every rank has posted)
        gs[r].flush()
    assert gs[0].frames_received == world * frames and gs[0].gathers == gs[1].gathers == 3
    allf = np.concatenate(got, axis=1)
    assert allf.shape == (world, frames, frame_bytes)
    for r in range(world):
        for i in range(frames):
            assert (allf[r, i] == (17 * r + i) % 251).all(), (r, i)
    # buffer 0 was handed out again for frames 8..10 only after the gather out of it (ticket 1) had been waited for
    assert plugs[0].waited[0] == 1 and set(plugs[0].waited) == {1, 2, 3}
    for g in gs:
        g.close()


def test_native_frame_gather_world_8_partial_last_batch_offsets():
    """BASELINE configs[4]'s shape on the fake communicator: eight ranks, batches of 8 frames, 19 frames per rank — two full
    batches and a partial one of 3. On the root, rank r's frame k of a batch lies at (r * batch + k) * frame_bytes of the
    receive buffer whatever the batch's count is (every rank always sends the whole staging batch), `on_batch` is told the
    count, and every frame of every rank arrives exactly once, in order."""
    from bevy_gaussian_splatting_amd.multiview import NativeFrameGather
    _FakeCommPlugin.mailbox = {}
    frame_bytes, batch, frames, world = 48, 8, 19, 8
    seen = []
    plugs = [_FakeCommPlugin(r, world) for r in range(world)]

    def on_batch(recv_ptr, count):
        raw = plugs[0].view(recv_ptr, world * batch * frame_bytes)
        for r in range(world):
            for k in range(count):
                at = (r * batch + k) * frame_bytes
                seen.append((r, int(raw[at]), bytes(raw[at:at + frame_bytes])))

    gs = [NativeFrameGather(plugs[r], frame_bytes, world, r, bytes([2]) * 128, batch=batch, on_batch=on_batch if r == 0 else None)
          for r in range(world)]
    for i in range(frames):
        for r in reversed(range(world)):
            slot = gs[r].next_target()
            plugs[r].view(slot, frame_bytes)[:] = (31 * r + i) % 251
            gs[r].frame_completed()
    for r in reversed(range(world)):
        gs[r].flush()
    assert all(g.gathers == 3 for g in gs) and gs[0].frames_received == world * frames
    assert len(seen) == world * frames
    for r in range(world):
        mine = [v for (rr, v, _) in seen if rr == r]
        assert mine == [(31 * r + i) % 251 for i in range(frames)], r          # in order, once each
    assert all(blob == bytes([v]) * frame_bytes for (_, v, blob) in seen)       # whole frames, not shifted ones
    for g in gs:
        g.close()


def test_native_frame_gather_device_wait_sends_at_enqueue_and_guards_its_ring():
    """device_wait (bgs_comm_gather_after): a batch goes out when its last frame has been ENQUEUED — the gather waits on the
    device — so frames may be popped long after; and the staging ring refuses to hand out a slot over frames that have not
    gone out (round 5's advisor: a pipeline deeper than the batch overwrote them silently)."""
    from bevy_gaussian_splatting_amd.multiview import NativeFrameGather
    _FakeCommPlugin.mailbox = {}
    frame_bytes, batch, frames, world = 32, 4, 10, 2
    got = []
    plugs = [_FakeCommPlugin(r, world) for r in range(world)]

    def on_batch(recv_ptr, count):
        got.append(plugs[0].view(recv_ptr, world * batch * frame_bytes).reshape(world, batch, frame_bytes)[:, :count].copy())

    gs = [NativeFrameGather(plugs[r], frame_bytes, world, r, bytes([3]) * 128, batch=batch, on_batch=on_batch if r == 0 else None,
                            device_wait=True) for r in range(world)]
    for i in range(frames):
        for r in reversed(range(world)):
            slot = gs[r].next_target()
            plugs[r].view(slot, frame_bytes)[:] = (5 * r + i) % 251
            gs[r].frame_enqueued()          # no pop, no frame_completed: the batch leaves at its 4th enqueue
    assert gs[1].gathers == 2 and plugs[1].after_calls == 2
    for r in reversed(range(world)):
        gs[r].flush()
    allf = np.concatenate(got, axis=1)
    assert allf.shape == (world, frames, frame_bytes) and gs[0].frames_received == world * frames
    for r in range(world):
        assert [int(allf[r, i, 0]) for i in range(frames)] == [(5 * r + i) % 251 for i in range(frames)]
    assert gs[0].stale_frames == 0
    gs[0].note_rerun(2)
    assert gs[0].stale_frames == 2
    for g in gs:
        g.close()
    # the guard, in the default mode: nine slots handed out, none completed, batch 4 -> the ninth would overwrite frame 0
    _FakeCommPlugin.mailbox = {}
    one = _FakeCommPlugin(0, 1)
    g = NativeFrameGather(one, frame_bytes, 1, 0, bytes([4]) * 128, batch=4)
    for _ in range(8):
        g.next_target()
    with pytest.raises(RuntimeError, match="outstanding"):
        g.next_target()
