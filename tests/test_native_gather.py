"""The multi-GPU frame gather BEHIND THE C ABI (include/bgs.h bgs_comm_*: RCCL's ncclGather on a stream the communicator
owns) — SURVEY 8(e): "grouped ncclSend / ncclRecv or RCCL ncclGather". No torch.distributed anywhere in these tests: the
128-byte unique id travels through a file, as a Rust host's launcher would ship it. The per-camera seam this serves:
/root/reference/src/sort/mod.rs:143-150 (camera index), src/render/mod.rs:1548-1554 (per-camera entry offset)."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, FRAMES, BATCH = 640, 360, 11, 4


def _worker(rank, world, outdir):
    os.environ["BGS_QUEUE_HOLDERS"] = "0"
    from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.multiview import NativeFrameGather, headless_view

    cloud = random_gaussians_3d_seeded(100_000, 2)
    idfile = os.path.join(outdir, "unique_id.bin")
    with GaussianSplattingPlugin(rank) as p:
        if rank == 0:
            uid = p.comm_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(uid)
            os.replace(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile):
                assert time.time() - t0 < 300, "rank 0 never wrote the unique id"
                time.sleep(0.05)
            uid = open(idfile, "rb").read()
        assert len(uid) == 128
        h = p.upload(cloud)
        v, s = headless_view(rank, W, H), CloudSettings()
        p.set_output_srgb8(True)
        p.render(h, v, s)                                    # this rank's own frame, blocking: the reference bytes
        ptr, nbytes = p.framebuffer_srgb8_device_ptr()
        assert nbytes == W * H * 4
        own = np.empty((H, W, 4), np.uint8)
        p.download(ptr, own)
        np.save(os.path.join(outdir, f"own_{rank}.npy"), own)
        got = []

        def on_batch(recv_ptr, count):
            blk = np.empty((world, BATCH, H, W, 4), np.uint8)
            p.download(recv_ptr, blk)
            got.append(blk[:, :count].copy())

        device_wait = os.environ.get("BGS_TEST_GATHER_AFTER") == "1"   # bgs_comm_gather_after: batches leave at enqueue
        bg = NativeFrameGather(p, W * H * 4, world, rank, uid, batch=BATCH, on_batch=on_batch, device_wait=device_wait)
        p.set_async(True)
        p.set_pipeline_depth(4)
        p.set_packed_only(True)
        reruns0 = sum(p.adaptive_counters()[k] for k in ("reruns_sort", "reruns_lists"))
        for _ in range(FRAMES):
            p.set_srgb8_target(bg.next_target())
            p.render(h, v, s, download=False)
            bg.frame_enqueued()
            if p.frames_in_flight() >= 4:
                p.pipeline_pop()
                bg.frame_completed()
        while p.frames_in_flight():
            p.pipeline_pop()
            bg.frame_completed()
        if device_wait:   # a frame re-run after its batch went out would have been sent stale: none on a settled view
            bg.note_rerun(sum(p.adaptive_counters()[k] for k in ("reruns_sort", "reruns_lists")) - reruns0)
            assert bg.stale_frames == 0
        bg.flush()
        if rank == 0:
            np.save(os.path.join(outdir, "received.npy"), np.array([bg.frames_received, bg.gathers]))
            allf = np.concatenate(got, axis=1)               # [world, frames, H, W, 4]
            for r in range(world):
                np.save(os.path.join(outdir, f"gathered_{r}.npy"), allf[r])
        # a ticket that was never handed out is refused; ticket 0 waits for everything
        with pytest.raises(Exception):
            p.comm_wait(bg.comm, 10_000)
        p.comm_wait(bg.comm, 0)
        p.set_async(False)
        p.set_packed_only(False)
        bg.close()
        h.free()


def _run(world, tmp_path):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
        assert p.exitcode == 0
    received, gathers = np.load(tmp_path / "received.npy")
    assert int(received) == world * FRAMES and int(gathers) == -(-FRAMES // BATCH)
    for r in range(world):
        own, seq = np.load(tmp_path / f"own_{r}.npy"), np.load(tmp_path / f"gathered_{r}.npy")
        assert seq.shape == (FRAMES, H, W, 4) and own.any()
        for f in seq:
            assert np.array_equal(f, own), f"rank {r}: a gathered frame differs from the frame the rank rendered"


@pytest.mark.timeout(600)
def test_native_gather_with_one_rank(tmp_path):
    """World size 1 on the one device every box has: communicator, staged batches, tickets, flush — every frame arrives
    and holds exactly the bytes of the rank's own blocking frame."""
    _run(1, tmp_path)


@pytest.mark.timeout(600)
def test_native_gather_after_with_one_rank(tmp_path, monkeypatch):
    """The same through bgs_comm_gather_after (round 6): a batch is handed to the communicator when its last frame has been
    ENQUEUED, the gather waits on the device for the context's frames in flight — and the bytes on the root are still each
    frame's own."""
    monkeypatch.setenv("BGS_TEST_GATHER_AFTER", "1")
    _run(1, tmp_path)


@pytest.mark.timeout(600)
def test_native_gather_with_two_ranks(tmp_path):
    """Two processes, one GPU and one camera each (skipped on a one-GPU box): rank 0 holds, for both ranks, the bytes each
    rendered on its own; the two cameras' frames differ."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (the driver's multi-GPU box)")
    _run(2, tmp_path)
    assert not np.array_equal(np.load(tmp_path / "own_0.npy"), np.load(tmp_path / "own_1.npy"))


def test_comm_arguments_are_validated(plugin):
    uid = plugin.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with pytest.raises(Exception) as ei:
        plugin.comm_create(uid, 2, 2)              # rank out of range: refused before any collective call
    assert "rank" in str(ei.value)
    with pytest.raises(ValueError):
        plugin.comm_create(uid[:64], 1, 0)
    comm = plugin.comm_create(uid, 1, 0)
    buf = plugin.device_alloc(4096)
    with pytest.raises(Exception):
        plugin.comm_gather(comm, 1, buf, 1024, buf)     # root outside the communicator
    with pytest.raises(Exception):
        plugin.comm_gather(comm, 0, buf, 1024, None)    # the root needs a receive buffer
    plugin.upload_bytes(buf, np.arange(1024, dtype=np.uint8).astype(np.uint8))
    t1 = plugin.comm_gather(comm, 0, buf, 1024, buf + 2048)
    t2 = plugin.comm_gather(comm, 0, buf, 0, buf + 2048)   # nothing to move: no ticket
    assert t1 == 1 and t2 == 0
    plugin.comm_wait(comm, t1)
    back = np.empty(1024, np.uint8)
    plugin.download(buf + 2048, back)
    assert np.array_equal(back, np.arange(1024, dtype=np.uint8).astype(np.uint8))
    plugin.comm_destroy(comm)
    plugin.device_free(buf)
