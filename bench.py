#!/usr/bin/env python
"""bench.py — headline benchmark of the sort + rasterize hot path (contract in the task brief).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one view: keygen (+ stable cull partition) -> depth
radix sort -> project + ordered supertile binning -> tile raster of a 1920x1080 frame, with the cloud
already resident in HBM (uploaded once before the timed region, like the reference's asset
upload). Workload = BASELINE.json configs[1]: 1M random 3DGS splats (the reference's own
`random_gaussians_3d` distributions), f32 planar cloud, SH degree 3, `CloudSettings::default()`,
examples/headless.rs camera — which carries no `Msaa` component of its own, so the reference draws it with Bevy's
default, Msaa::Sample4 (CloudPipelineKey.sample_count, src/render/mod.rs:357-424): the headline renders 4 samples per
pixel (coverage per sample, shading per pixel, resolved frame); `msaa_off` has the single-sampled rate of rounds 1-3. With N > 1 ranks every rank renders ITS camera (camera g = the
headless camera yawed g*45 degrees) of the replicated cloud and rank 0 gathers every rank's frames
over RCCL, asynchronously, 8 frames per collective (weak scaling: per-GPU work fixed).

Prints ONE JSON line on rank 0. `value` = whole-job frames/s. Extra objects:
  roofline      dominant kernel of the hot path, algorithmic bytes per launch / average launch duration
                measured live with HIP events on the library's own stream (bgs_get_stats) on frames
                that are not overlapped with other frames; `in_flight` has the same for the pipelined
                timed region, where an event interval also contains the wait for the other lanes'
                kernels; `frame` has the whole-frame effective rate (algorithmic bytes x frames/s).
                `measured_peak` = this device's DtoD-copy / triad ceiling (bgs_hbm_probe).
  cpu_baseline  the oracle ("port": C restatement, OpenMP) timed on this host on a bounded sample
  stages        per-stage ms / algorithmic GB/s / %peak, V, I
  sort_msplats_per_s   "Msplats/s sorted" (keygen + depth sort only, bgs_sort)
  scene_like    the same workload with global_scale = 0.05
  msaa_off      the same workload on a camera with Msaa::Off (one sample per pixel)
  latency       first_frame_ms (a context that has learnt nothing yet: digit passes, first-guess capacities, the re-run
                they may cost) and after_cut_ms (a 137-degree camera cut in a warmed-up context), blocking frames
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import math

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
N_SPLATS = 1_000_000
SEED = 2
WIDTH, HEIGHT = 1920, 1080
GATHER_BATCH = 8  # frames per framebuffer gather (N > 1)


def kernel_source_sha256() -> str:
    """Identity of the kernels a counter file was measured on — and of the library this run loads: SHA-256 over the
    HIP sources and headers of libbgs (`bgs_build_id()`; not the commit id, which also changes with every
    documentation commit)."""
    from bevy_gaussian_splatting_amd import _build_id
    return _build_id.kernel_source_sha256()


def load_pmc():
    """profiles/pmc_traffic.json (rocprofv3 --pmc passes, scripts/gpu_pmc.sh + scripts/make_pmc_traffic.py) if
    it was measured on THESE kernel sources, else (None, why): a counter value from another revision of the
    kernels must not be printed into this run's line."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
    except Exception as e:  # noqa: BLE001
        return None, f"profiles/pmc_traffic.json unreadable: {e}"
    have, want = d.get("kernel_source_sha256"), kernel_source_sha256()
    if have != want:
        return None, f"profiles/pmc_traffic.json was measured on kernel sources {str(have)[:12]}, this run is {want[:12]}"
    return d, f"rocprofv3 PMC passes on kernel sources {want[:12]} ({d.get('measured', 'profiles/')})"


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks ourselves, one per GPU, exactly as the
    driver would (torch.distributed.run, rendezvous on 127.0.0.1). Fails loudly when the box has fewer
    than N devices."""
    import socket

    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        print(f"bench.py: --gpus {n} needs {n} HIP devices, {have} visible on this box "
              "(there is no CPU fallback and no oversubscription of one GPU by several ranks)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def stage_table(stats: dict, cloud_bytes_per_splat: int, rec_bytes: int = 48) -> dict:
    """Algorithmic bytes per stage (SURVEY 8(d) terms) and the launches each stage comprises."""
    N, V, I = stats["splat_count"], stats["visible_count"], stats["instance_count"]
    D = stats["draw_count"]  # pairs that actually go through the radix passes (culled ones are partitioned off)
    k, kt = stats["depth_passes"], stats["tile_passes"]
    P = WIDTH * HEIGHT
    B = cloud_bytes_per_splat
    bucket = stats.get("sort_path") == "bucket"
    # bucket sort: keygen also writes the D drawable pairs into their buckets (8 B each, instead of the
    # index-ordered list), ONE launch reads them and writes the sorted list (16 B per pair)
    sort_stage = {"bytes": D * 16, "launches": 1} if bucket else {"bytes": k * D * 16, "launches": max(k, 1)}
    if stats.get("binning") == "scan":
        # I = coarse (supertile) list entries (rank + tile rect, 8 B): written once by bin_kernel, read
        # once by the rasteriser, which also reads each visible record at least once
        return {
            # keygen reads N positions and writes the D drawable pairs (the culled tail has no reader in a Color frame)
            "keygen": {"bytes": N * 16 + D * 8, "launches": 1},
            "depth_sort": sort_stage,
            # two launches: project_kernel (entry + cloud record in, projected record + 4-byte rect out) and bin_kernel
            # (rects in, list entries out); timed together between two HIP events
            "project": {"bytes": V * 8 + V * (B - 16) + V * rec_bytes + V * 8 + I * 8, "launches": 2},
            "raster": {"bytes": I * 8 + V * rec_bytes + P * 16, "launches": 1},
        }
    return {
        "keygen": {"bytes": N * 16 + N * 8, "launches": 1},
        "depth_sort": sort_stage,
        "project": {"bytes": V * (B - 16) + V * rec_bytes + I * 8, "launches": 1},
        "tile_sort": {"bytes": kt * I * 16, "launches": max(kt, 1)},
        "ranges": {"bytes": I * 8, "launches": 1},  # not in SURVEY's bytes_frame (pure overhead pass)
        "raster": {"bytes": I * (4 + rec_bytes) + P * 16, "launches": 1},
    }


FRAMES_ISSUED = [0]  # every frame any measure() call enqueued (the gather path checks it against what rank 0 received)


def quad_size_histogram(cloud, view, settings, sample=200_000, seed=0):
    """How large the quads of a frame are on screen, from a numpy float64 evaluation of the vertex stage's footprint
    (EWA: cov2d = J W Sigma W^T J^T + 0.3 I in half-pixel units, quad sides = cutoff sqrt(lambda) pixels; src/render/
    helpers.wgsl:8-120) on a random subsample of the cloud: the share of VISIBLE quads by the length of their longer
    side, the medians of both sides, the share no wider than a third of a tile. A report, not part of any path."""
    import numpy as np
    rng = np.random.default_rng(seed)
    n = len(cloud)
    idx = rng.choice(n, size=min(sample, n), replace=False)
    pos = cloud.position_visibility[idx, :3].astype(np.float64)
    rot = cloud.rotation[idx].astype(np.float64)
    so = cloud.scale_opacity[idx].astype(np.float64)
    M = np.asarray(settings.transform, np.float64).reshape(4, 4)
    V = np.asarray(view.view_from_world, np.float64).reshape(4, 4)
    P = np.asarray(view.clip_from_view, np.float64).reshape(4, 4)
    pw = pos @ M[:3, :3].T + M[:3, 3]
    t = pw @ V[:3, :3].T + V[:3, 3]
    clip = np.concatenate([t, np.ones((len(t), 1))], 1) @ P.T
    w = clip[:, 3] + 1e-9
    ndc = clip[:, :3] / w[:, None]
    vis = (np.abs(ndc[:, 0]) < 1.1) & (np.abs(ndc[:, 1]) < 1.1) & (np.abs(ndc[:, 2] - 0.5) < 0.5)
    t, rot, so = t[vis], rot[vis], so[vis]
    r, x, y, z = rot.T
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y)], 1),
                  np.stack([2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x)], 1),
                  np.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], 1)], 1)   # rows = local axes
    S = so[:, :3] * float(settings.global_scale)
    Mm = S[:, :, None] * R
    Sigma = np.einsum("nki,nkj->nij", Mm, Mm)
    T3 = M[:3, :3]
    Sigma = np.einsum("ab,nbc,dc->nad", T3, Sigma, T3)
    fx, fy = P[0, 0] * view.width, P[1, 1] * view.height
    J = np.zeros((len(t), 2, 3))
    J[:, 0, 0] = fx / t[:, 2]; J[:, 0, 2] = -fx * t[:, 0] / t[:, 2] ** 2
    J[:, 1, 1] = -fy / t[:, 2]; J[:, 1, 2] = fy * t[:, 1] / t[:, 2] ** 2
    JW = J @ V[:3, :3]
    cov = np.einsum("nab,nbc,ndc->nad", JW, Sigma, JW)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    mid, rad = 0.5 * (a + c), np.sqrt(np.maximum(0.25 * (a - c) ** 2 + b * b, 0.0))
    op = so[:, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        cutoff = np.sqrt(np.maximum(9.0 + 2.0 * np.log(op), 1e-6)) if settings.opacity_adaptive_radius else np.full(len(op), 3.0)
    major = cutoff * np.sqrt(np.maximum(mid + rad, 0.0))     # pixels: 2 x (cutoff sqrt(lambda) half-pixels)
    minor = cutoff * np.sqrt(np.maximum(mid - rad, 0.0))
    ok = np.isfinite(major) & np.isfinite(minor)
    major, minor = major[ok], minor[ok]
    edges = [0, 2, 4, 6, 8, 16, 32, 64, 128, 1e9]
    hist = np.histogram(major, bins=edges)[0] / max(len(major), 1)
    return {"sampled_splats": int(len(idx)), "visible_in_sample": int(vis.sum()),
            "longer_side_px_share": {f"{int(lo)}-{'inf' if hi > 1e8 else int(hi)}": round(float(h), 4) for lo, hi, h in zip(edges[:-1], edges[1:], hist)},
            "longer_side_px_median": round(float(np.median(major)), 2) if len(major) else None,
            "shorter_side_px_median": round(float(np.median(minor)), 2) if len(minor) else None,
            "share_under_6px": round(float((major < 6.0).mean()), 4) if len(major) else None,
            "tiles_per_quad_estimate_mean": round(float(((major / 16.0 + 1.0) * (minor / 16.0 + 1.0)).mean()), 2) if len(major) else None}


def measure(plugin, handle, view, settings, steps, warmup, gather=None, barrier=None, depth=1, trials=1,
            busy_warm_frames=0, views=None):
    """W untimed + K timed steps. A step ENQUEUES one frame: the scan pipeline needs no host round
    trip, and the context keeps `depth` frames in flight (lanes, multiplexed onto a few HIP streams). With a
    `gather` callback (N > 1 ranks) the oldest frame is popped and handed to it as soon as `depth`
    frames are in flight, so gathers overlap the following frames. The closing synchronize waits for
    everything, so dt covers exactly K complete frames (and their gathers). With `trials` > 1 the timed
    region (barrier, K steps, barrier) is repeated and the list of durations returned as well: K = 20 frames
    are 1.5 ms of GPU time, which one clock ramp or one late lane moves by 10 %, so the headline reports the
    median trial. `busy_warm_frames`: that many more untimed frames before the first trial (clocks, adaptive
    state) — a frame COUNT, not a duration, so that every rank of a multi-GPU run issues the same collectives.
    Returns (seconds of the median trial, per-stage ms averaged by the library over the timed frames' HIP
    events, stats[, every trial's seconds])."""
    prepared = plugin.prepare(view, settings)  # marshal the C structs once, like a caller's per-view cache
    # `views`: a list of prepared views, one per frame, cycled (a moving camera; marshalled outside the timed region
    # like `prepared`, so the host cost per frame is the same as for the static view)
    cursor = [0]

    def run(k):
        for _ in range(k):
            if gather is not None and hasattr(gather, "before_render"):
                gather.before_render()
            if views is not None:
                plugin.render(handle, views[cursor[0] % len(views)], download=False)
                cursor[0] += 1
            else:
                plugin.render(handle, prepared, download=False)
            FRAMES_ISSUED[0] += 1
            if gather is not None and plugin.frames_in_flight() >= depth:
                gather(*plugin.pipeline_pop())
        if gather is not None:
            while plugin.frames_in_flight():
                gather(*plugin.pipeline_pop())
            if hasattr(gather, "flush"):
                gather.flush()  # the last (partial) batch and every outstanding collective
        plugin.synchronize()  # also checks the device watchdog word of every frame

    # every lane allocates its buffers on its first frame: done before the W warm-up steps, so that a short
    # --warmup (fewer steps than lanes) does not leave allocations inside the timed region
    run(depth)
    run(warmup)
    chunk = max(steps, 4 * depth)
    for _ in range((busy_warm_frames + chunk - 1) // chunk):
        run(chunk)
    dts = []
    for _ in range(max(1, trials)):
        if barrier:
            barrier()
        t0 = time.perf_counter()
        run(steps)
        if barrier:
            barrier()
        dts.append(time.perf_counter() - t0)
    st = plugin.stats()
    if trials > 1:
        return statistics.median(dts), dict(st["stage_ms"]), st, dts
    return dts[0], dict(st["stage_ms"]), st


def whole_frame_parity(oracle, cloud, entries, view, settings, ref, got):
    """The GPU frame against the oracle's whole frame (which `cpu_baseline` renders anyway), all W x H pixels, at the
    tolerance of the parity tests: |gpu - oracle| <= 1e-3 + 1e-4 |oracle| per channel, plus the oracle's per-pixel
    ambiguity bound where a coverage decision lies within rounding distance of a quad edge. The timed oracle frame
    carries no ambiguity map (that would change what is timed), so the map is computed afterwards and only for the
    48 x 48 cells that hold a value beyond the strict bound. OUTSIDE every timed region."""
    err = np.abs(got.astype(np.float64) - ref)
    strict = err <= 1e-3 + 1e-4 * np.abs(ref)
    bad = np.argwhere(~strict.all(axis=-1))
    cells = sorted({(int(y) // 48, int(x) // 48) for y, x in bad})
    out = {"checked": f"whole {ref.shape[1]}x{ref.shape[0]} frame vs the oracle (every pixel, every channel)",
           "pixels": int(ref.shape[0] * ref.shape[1]), "max_abs_err": float(err.max()),
           "tolerance": "1e-3 + 1e-4*|ref| per channel (+ the oracle's ambiguity bound on quad-edge pixels)",
           "values_beyond_strict_tolerance": int((~strict).sum()), "values_on_ambiguity_slack": 0, "ok": True}
    if len(cells) > 256:
        out["ok"] = False
        out["why"] = f"{len(cells)} cells of 48x48 px hold values beyond the strict tolerance"
        return out
    on_slack = 0
    for cy, cx in cells:
        x0, y0 = cx * 48, cy * 48
        x1, y1 = min(x0 + 48, ref.shape[1]), min(y0 + 48, ref.shape[0])
        r, amb = oracle.render(cloud, entries, view, settings, window=(x0, y0, x1, y1), with_ambiguity=True)
        g = got[y0:y1, x0:x1]
        e = np.abs(g.astype(np.float64) - r)
        lim = 1e-3 + 1e-4 * np.abs(r)
        on_slack += int((e > lim).sum())
        if not (e <= lim + amb[..., None]).all():
            out["ok"] = False
            out["why"] = f"cell ({x0},{y0}): max |err| {float(e.max()):.3e} beyond tolerance + ambiguity bound"
    out["values_on_ambiguity_slack"] = on_slack
    if on_slack > 0.0005 * strict.size:
        out["ok"] = False
        out["why"] = "the ambiguity slack is used by more than 0.05 % of the values"
    return out


def cpu_baseline(cloud, view, settings, gpu_frame=None, gpu_entries=None):
    """The oracle (C restatement, OpenMP) on a bounded sample of the SAME workload:
    all host cores this process is granted (affinity capped by the cgroup quota): the full 1M-splat sort (both reference sorts: the radix semantics and the
    rayon/std descending-f32 semantics) + the vertex stage for every splat + the raster of the
    WHOLE 1920x1080 frame (no scaling); and the same pinned to ONE core with the centred 480x270
    window (1/16 of the frame, time scaled x16). About 15-20 s of CPU work in total."""
    # The checker library travels to this box prebuilt and portable (-O2, no -march); the BASELINE is the same
    # source built here, for this host: -O3 -march=native. Arithmetic is unchanged
    # (-ffp-contract=off, no fast-math).
    flags = "-O2 (portable checker build; the native build failed)"
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libbgs_oracle_native.so"], capture_output=True, text=True)
    if r.returncode == 0 and "oracle.oracle" not in sys.modules:
        os.environ["BGS_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "libbgs_oracle_native.so")
        flags = "-O3 -march=native -fopenmp -ffp-contract=off, built on this host"
    from oracle import oracle
    from bevy_gaussian_splatting_amd import CloudSettings, SortMode

    oracle.build()
    all_cores = oracle.max_threads()  # already capped to the CPUs the cgroup grants this process (oracle.effective_cpus)

    kept = {}

    def frame(window_w, window_h):
        t0 = time.perf_counter()
        entries = oracle.sort(cloud, view, settings)
        t_sort = time.perf_counter() - t0
        t0 = time.perf_counter()
        oracle.render(cloud, entries, view, settings, window=(0, 0, 1, 1))
        t_vs = time.perf_counter() - t0
        x0, y0 = (WIDTH - window_w) // 2, (HEIGHT - window_h) // 2
        t0 = time.perf_counter()
        img = oracle.render(cloud, entries, view, settings, window=(x0, y0, x0 + window_w, y0 + window_h))
        t_win = max(time.perf_counter() - t0 - t_vs, 1e-6)
        scale = (WIDTH * HEIGHT) / float(window_w * window_h)
        if window_w == WIDTH and window_h == HEIGHT:
            kept["entries"], kept["image"] = entries, img
        return t_sort, t_vs, t_win, scale

    t_sort, t_vs, t_win, scale = frame(WIDTH, HEIGHT)
    parity = None
    if gpu_frame is not None:
        # the whole oracle frame the baseline just paid for is also the checker of the frame this run benchmarked
        parity = whole_frame_parity(oracle, cloud, kept["entries"], view, settings, kept["image"], gpu_frame)
        if gpu_entries is not None:
            parity["sort_entries_bit_exact"] = bool(np.array_equal(gpu_entries["key"], kept["entries"]["key"]) and
                                                    np.array_equal(gpu_entries["index"], kept["entries"]["index"]))
            parity["ok"] = parity["ok"] and parity["sort_entries_bit_exact"]
    kept.clear()
    t0 = time.perf_counter()
    oracle.sort(cloud, view, CloudSettings(sort_mode=SortMode.Rayon))
    t_sort_std = time.perf_counter() - t0
    t_frame = t_sort + t_vs + scale * t_win

    oracle.set_threads(1)
    try:
        s1, v1, w1, sc1 = frame(480, 270)
    finally:
        oracle.set_threads(all_cores)
    t_frame1 = s1 + v1 + sc1 * w1
    return {
        "parity": parity,
        "value": 1.0 / t_frame,
        "unit": "frames/s",
        "cores": all_cores,
        "kind": "port",
        "sample": (f"oracle/bgs_oracle.c (gcc {flags}): full 1M-splat keygen+LSD radix sort "
                   f"({t_sort:.3f}s) + vertex stage of all splats ({t_vs:.3f}s) + raster of the "
                   f"whole {WIDTH}x{HEIGHT} frame ({t_win:.2f}s): the reference's algorithm as it stands, every quad "
                   "rasterised in full, back to front, no early termination"),
        "sort_msplats_per_s": len(cloud) / t_sort / 1e6,
        "sort_std_msplats_per_s": len(cloud) / t_sort_std / 1e6,
        "sort_std_note": ("rayon semantics (src/sort/rayon.rs:86-104): parallel keygen + parallel descending-f32 "
                          "comparison sort (one qsort run per thread, parallel pairwise merges)"),
        "one_core": {"value": 1.0 / t_frame1, "unit": "frames/s", "cores": 1,
                     "sort_msplats_per_s": len(cloud) / s1 / 1e6,
                     "sample": (f"same, OMP threads = 1: sort {s1:.3f}s + vertex stage {v1:.2f}s + centred 480x270 window "
                                f"({w1:.2f}s, scaled x{sc1:.0f})")},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--splats", type=int, default=N_SPLATS)
    ap.add_argument("--depth", type=int, default=8, help="frames in flight (pipeline lanes, 1..8)")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the lanes are multiplexed onto (0 = one per lane): frames competing for the chip")
    ap.add_argument("--trials", type=int, default=0,
                    help="timed regions of --steps frames; the median is reported (default: 5, 3 from 2000 steps up)")
    args = ap.parse_args()

    if os.environ.get("BGS_LIB_OVERRIDE"):
        raise SystemExit("bench.py measures the library built from this tree's sources: unset BGS_LIB_OVERRIDE")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("BGS_BENCH_FORCE_DIST") != "1":
        raise SystemExit(launch_ranks(args.gpus))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and (world > 1 or os.environ.get("BGS_BENCH_FORCE_DIST") != "1"):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} HIP devices visible")
    torch.cuda.set_device(local_rank)
    dist = None
    # BGS_BENCH_FORCE_DIST=1: run the N > 1 code path (process group, sRGB8 output, popped frames,
    # batched gather, max-over-ranks) with a single rank, to test it on a one-GPU box
    if world > 1 or os.environ.get("BGS_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device(f"cuda:{local_rank}"))
        # RCCL's own (idle) streams already hold the HIP runtime's four hardware queues, which is what the
        # library's queue-holder streams are for in a process without them (csrc/bgs_frame.hip, assign_streams)
        os.environ.setdefault("BGS_QUEUE_HOLDERS", "0")
        from bevy_gaussian_splatting_amd import _native
        _native.load().bgs_set_queue_holders(0)   # the same switch through the API (process-global)

    from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor, gather_framebuffers, headless_view

    cloud = random_gaussians_3d_seeded(args.splats, SEED)  # replicated on every rank
    plugin = GaussianSplattingPlugin(local_rank)
    handle = plugin.upload(cloud)
    view = headless_view(rank, WIDTH, HEIGHT)  # rank g owns camera g
    settings = CloudSettings()
    DEPTH = max(1, min(8, args.depth))  # frames in flight (lanes); 1 = a single stream
    plugin.set_async(True)
    plugin.set_pipeline_depth(DEPTH)
    # lane i runs on stream i % STREAMS: with 8 lanes on 4 streams (one stream per hardware queue of the HIP
    # runtime, include/bgs.h) every stream already holds its next frame while one executes
    plugin.set_pipeline_streams(max(0, min(8, args.streams)))
    # every kernel of every Nth frame is bracketed by HIP events (a record costs ~4 us of GPU time, so
    # timing every frame would cost ~20 % of the frame rate being measured)
    STRIDE = 16 if args.steps >= 64 else max(1, min(8, args.steps // 4))
    plugin.set_profiling_stride(STRIDE)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gather_ms = [0.0]
    gather = None
    batcher = None
    native_gather, native_err = False, None
    if dist is not None:
        # the gathered frame is the reference's colour-attachment format (Rgba8UnormSrgb, 8.3 MB at
        # 1080p); the f32 target stays on its GPU. One collective per GATHER_BATCH frames, asynchronous
        # and double-buffered (BatchedFrameGather): a per-frame gather would be bound by the collective's
        # latency at these frame rates.
        # The gather itself runs behind the C ABI (bgs_comm_*: RCCL's ncclGather on the communicator's own stream,
        # multiview.NativeFrameGather) — what a host without torch would run; torch.distributed is the launcher's
        # rendezvous here (rank / world size, the barrier, shipping rank 0's 128-byte unique id) and the fallback
        # (BGS_BENCH_GATHER=torch, or a box whose librccl cannot be opened: said in the line, "gather_backend").
        from bevy_gaussian_splatting_amd.multiview import BatchedFrameGather, NativeFrameGather, device_ptr_as_tensor
        frame_bytes8 = WIDTH * HEIGHT * 4
        native_err = None
        if os.environ.get("BGS_BENCH_GATHER", "native") != "torch":
            try:
                uid = plugin.comm_unique_id()        # every rank: proves librccl opens here (only rank 0's id is used)
            except Exception as e:  # noqa: BLE001
                uid, native_err = None, str(e)
            flag = torch.tensor([0 if uid is None else 1], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                box = [uid if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                batcher = NativeFrameGather(plugin, frame_bytes8, world, rank, box[0], batch=GATHER_BATCH)
            elif native_err is None:
                native_err = "another rank could not open librccl"
        else:
            native_err = "BGS_BENCH_GATHER=torch"
        native_gather = batcher is not None
        if batcher is None:
            batcher = BatchedFrameGather((HEIGHT, WIDTH, 4), torch.uint8, f"cuda:{local_rank}", batch=GATHER_BATCH)

        def gather(f32_ptr, srgb8_ptr):
            # the frame was rendered straight into its slot of the staging batch (bgs_set_srgb8_target)
            t0 = time.perf_counter()
            batcher.frame_completed()
            gather_ms[0] += (time.perf_counter() - t0) * 1e3

        gather.flush = batcher.flush
        gather.before_render = ((lambda: plugin.set_srgb8_target(batcher.next_target())) if native_gather else
                                (lambda: plugin.set_srgb8_target(batcher.next_target().data_ptr())))
        # the gathered frame IS the product on this path (the reference's Rgba8UnormSrgb colour attachment):
        # the rasteriser writes it straight into the staging batch and skips the f32 target nobody reads
        # (33 MB of writes per frame); the N = 1 headline keeps the f32 target
        plugin.set_packed_only(True)

    # ---- headline: reference distribution, CloudSettings::default() -------------------------
    # with a consumer popping frames the host waits for the oldest frame while the others run: 8 lanes on
    # 4 streams keep the GPU fed meanwhile
    lanes, streams = (8, 4) if gather is not None else (DEPTH, args.streams)
    if os.environ.get("BGS_BENCH_LANES"):
        lanes = max(1, min(8, int(os.environ["BGS_BENCH_LANES"])))  # experiment override
    if os.environ.get("BGS_BENCH_STREAMS"):
        streams = max(0, min(8, int(os.environ["BGS_BENCH_STREAMS"])))
    plugin.set_pipeline_depth(lanes)
    plugin.set_pipeline_streams(streams)
    # the headline region runs without stage events (each record is a packet on the stream; timing every
    # 16th frame cost ~5 % of the rate being measured); the per-stage numbers come from separate passes
    plugin.set_profiling(0)
    # Trials: at least 5 (3 from 2000 steps up), and as many as it takes to put >= 0.5 s of GPU time inside timed
    # regions (20 steps are ~1 ms: the driver's utilisation sampling cannot see five of those). The count comes from
    # a pilot region and is agreed between the ranks (max over ranks), so every rank issues the same collectives.
    trials = args.trials if args.trials > 0 else (5 if args.steps < 2000 else 3)
    if args.trials <= 0:
        pilot, _, _ = measure(plugin, handle, view, settings, args.steps, args.warmup, gather, barrier, lanes,
                              busy_warm_frames=3000)
        if dist is not None:
            t = torch.tensor([pilot], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pilot = float(t.item())
        trials = int(max(trials, min(1000, math.ceil(0.5 / max(pilot, 1e-6)))))
    _, _, _, dts = measure(plugin, handle, view, settings, args.steps, args.warmup, gather, barrier, lanes,
                           trials=max(trials, 2), busy_warm_frames=0 if args.trials <= 0 else 3000)
    issued_main = FRAMES_ISSUED[0]
    plugin.set_profiling(2)
    plugin.set_packed_only(False)
    per_rank = None
    if dist is not None:  # a trial lasts as long as its slowest rank
        t = torch.tensor(dts, dtype=torch.float64, device="cuda")
        mine = torch.tensor([statistics.median(dts)], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dts = [float(x) for x in t.tolist()]
        frame_bytes8 = WIDTH * HEIGHT * 4
        per_rank = [{"rank": r, "frames_per_s": round(args.steps / float(e.item()), 2),
                     "gather_bytes_per_trial": args.steps * frame_bytes8 if r != 0 else 0,
                     # what the rank's link carried while it rendered: its frames' bytes over ITS OWN median region
                     "link_GBps": round(args.steps * frame_bytes8 / float(e.item()) / 1e9, 2) if r != 0 else None}
                    for r, e in enumerate(every)]
        if rank == 0 and batcher is not None:
            # every frame every rank issued on this path has arrived on rank 0 (flush() waits for all collectives)
            want = world * issued_main
            if batcher.frames_received != want:
                raise SystemExit(f"gather check failed: rank 0 received {batcher.frames_received} frames, "
                                 f"{world} ranks issued {want}")
    gather_selfcheck = None
    if dist is not None and batcher is not None and native_gather:
        # outside the timed region: one more native gather of a known pattern — rank r's batch holds the byte r + 1 —
        # checked on rank 0 block by block (the frame COUNT above says every collective completed, this says the bytes
        # landed where bgs.h says they land)
        nb = GATHER_BATCH * frame_bytes8
        import numpy as _np
        plugin.upload_bytes(batcher.stage[0], _np.full(nb, rank + 1, dtype=_np.uint8))
        tk = plugin.comm_gather(batcher.comm, 0, batcher.stage[0], nb, batcher.recv[0])
        plugin.comm_wait(batcher.comm, tk)
        if rank == 0:
            got = device_ptr_as_tensor(batcher.recv[0], (world, nb), "|u1", f"cuda:{local_rank}")
            want = torch.arange(1, world + 1, dtype=torch.uint8, device=got.device)[:, None]
            gather_selfcheck = bool((got == want).all().item())
            if not gather_selfcheck:
                raise SystemExit("native gather self-check failed: a rank's block does not hold that rank's bytes")
    dt = statistics.median(dts)
    fps = world * args.steps / dt

    # ---- side measurements on rank 0 (outside the timed region) -----------------------------
    out = None
    if rank == 0:
        # per-stage times with the frames overlapped as in the headline region (every STRIDE-th frame timed)
        _, stage_ms, st = measure(plugin, handle, view, settings, max(args.steps // 2, 4 * STRIDE), 4, depth=lanes)
        table = stage_table(st, 240)
        stages = {}
        for name, info in table.items():
            ms = stage_ms.get(name, 0.0)
            gbs = info["bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            stages[name] = {"ms": round(ms, 4), "algorithmic_MB": round(info["bytes"] / 1e6, 2),
                            "GBps": round(gbs, 1), "pct_hbm_peak": round(100 * gbs / HBM_PEAK_GBS, 2)}
        dom = max(stages, key=lambda k: stages[k]["ms"])
        launches = table[dom]["launches"]
        per_launch_bytes = table[dom]["bytes"] / launches
        per_launch_s = stage_ms[dom] * 1e-3 / launches if stage_ms.get(dom, 0) > 0 else float("inf")
        achieved = per_launch_bytes / per_launch_s / 1e9
        scan_mode = st.get("binning") == "scan"
        kernel_names = {"keygen": "keygen_kernel",
                        "depth_sort": "bucket_sort_kernel" if st.get("sort_path") == "bucket" else "onesweep_kernel",
                        "project": "project_kernel" if scan_mode else "project_emit_kernel",
                        "tile_sort": "onesweep_kernel", "ranges": "tile_ranges_kernel",
                        "raster": "raster_scan_kernel" if scan_mode else "raster_kernel"}
        # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE
        # doubled per the gfx950 correction + WRITE_SIZE), for the same dense workload; null unless that
        # file was measured on exactly these kernel sources
        pmc, pmc_note = load_pmc()
        pmc_kernels = pmc["kernels"] if pmc else {}
        traffic = pmc_kernels.get(kernel_names[dom], {}).get("hbm_bytes_per_launch")
        roofline = {"bound": "hbm", "kernel": kernel_names[dom], "stage": dom,
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": int(per_launch_bytes), "launch_ms": round(per_launch_s * 1e3, 4)}
        frame_bytes = st["algorithmic_bytes"]
        frame_ms = sum(stage_ms.values())
        frame_gbs = frame_bytes / (frame_ms * 1e-3) / 1e9 if frame_ms > 0 else 0.0
        eff_gbs = frame_bytes * (fps / world) / 1e9  # per GPU: bytes one frame must move x frames/s
        try:
            copy_gbs, triad_gbs = plugin.hbm_probe(1 << 29, 10)
        except Exception:
            copy_gbs, triad_gbs = None, None
        measured = max(copy_gbs or 0.0, triad_gbs or 0.0) or None
        roofline["measured_peak"] = {"copy_GBps": round(copy_gbs, 1) if copy_gbs else None,
                                     "triad_GBps": round(triad_gbs, 1) if triad_gbs else None}
        roofline["frac_of_measured"] = round(achieved / measured, 4) if measured else None

        # the same frames on ONE stream (pipeline depth 1): per-kernel times without overlap
        plugin.set_pipeline_depth(1)
        toc0 = plugin.tile_order_counters()
        dt1, stage1, st1 = measure(plugin, handle, view, settings, args.steps, args.warmup)
        toc1 = plugin.tile_order_counters()
        single = {"value": round(args.steps / dt1, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dt1 / args.steps, 4),
                  "stage_ms": {k: round(v, 4) for k, v in stage1.items() if v},
                  # raster workgroups drawn in the order of a completed frame's per-tile work (kernels.h TileCost)
                  "tile_order": dict(zip(("frames_leaving_costs", "frames_in_cost_order", "orders_made"),
                                         (b - a for a, b in zip(toc0, toc1))))}
        dom1 = max(stage1, key=lambda k: stage1[k] / table[k]["launches"] if k in table else 0.0)
        b1 = table[dom1]["bytes"] / table[dom1]["launches"]
        t1 = max(stage1[dom1] * 1e-3 / table[dom1]["launches"], 1e-12)
        single["roofline"] = {"bound": "hbm", "kernel": kernel_names[dom1], "achieved": round(b1 / t1 / 1e9, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b1 / t1 / 1e9 / HBM_PEAK_GBS, 4),
                              "launch_ms": round(t1 * 1e3, 4), "bytes_per_launch": int(b1)}
        # The roofline object: the dominant kernel of the hot path, its duration measured live with HIP
        # events on frames that are NOT overlapped with other frames (this agrees with the rocprofv3
        # kernel durations under profiles/). Inside the pipelined timed region an event interval also
        # contains the time a kernel waits for the other lanes' kernels, which is reported as `in_flight`.
        roofline_main = dict(single["roofline"])
        roofline_main["traffic"] = pmc_kernels.get(roofline_main["kernel"], {}).get("hbm_bytes_per_launch")
        roofline_main["traffic_source"] = pmc_note
        roofline_main["measured_peak"] = roofline["measured_peak"]
        roofline_main["frac_of_measured"] = round(roofline_main["achieved"] / measured, 4) if measured else None
        roofline_main["measured_on"] = "single-stream frames (HIP events, every kernel of every Nth frame)"
        # What binds the kernel is vector-instruction ISSUE, not HBM. The floor: every wave-level vector instruction
        # the launch executes (SQ_INSTS_VALU of the committed PMC pass), priced per class at the rates of
        # MI355X_MICROARCH.md — a wave64 instruction issues over 2 clocks on the SIMD-32 (157.3 TFLOP/s fp32 peak),
        # transcendentals (v_exp / v_log / v_rcp / v_sqrt ...) at a quarter of that (8 clocks), fp64 at half (4) —
        # on 1024 SIMDs at the 2.4 GHz peak clock. (Round 2 priced every instruction at 4 clocks, which is the
        # ACHIEVED rate of this instruction mix — SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU — not a bound.)
        # Issue cost per wave64 instruction MEASURED on this chip (scripts/micro/valu_issue.hip, profiles/r4_micro/
        # valu_issue.txt; 8 waves per SIMD, wall clock): v_fma / v_mul / v_add / v_mov / v_and / v_add_u32 1.06-1.28 ns
        # (= 2 clocks at the ~1.9 GHz the chip sustains under vector load), v_max / v_min / v_cmp / v_cndmask / v_cvt /
        # shifts / every v_pk_* 1.8-2.0 ns (4 clocks), v_exp / v_rcp / v_log / v_sqrt 3.44 ns (8 clocks). The PMC passes
        # count the f32 FMA / ADD / MUL and the transcendental classes; what is left is a mix of both simple classes, so
        # the floor is given as a band: `low` prices the rest at 2 clocks, `high` at 4.
        def issue_floor_ms(k, rest_clocks=4.0):
            pk = pmc_kernels.get(k, {})
            wi = pk.get("valu_wave_instructions")
            if not wi:
                return None, None
            cls = pk.get("valu_classes") or {}
            trans = pk.get("valu_trans_wave_instructions") or 0.0
            f64 = pk.get("valu_f64_wave_instructions") or 0.0
            fast = (cls.get("fma_f32") or 0.0) + (cls.get("add_f32") or 0.0) + (cls.get("mul_f32") or 0.0)
            rest = max(wi - trans - f64 - fast, 0.0)
            clocks = 2.0 * fast + rest_clocks * rest + 8.0 * trans + 4.0 * f64
            return clocks / (4 * 256) / 2.4e9 * 1e3, {"wave_instructions": int(wi), "transcendental": int(trans), "fp64": int(f64),
                                                       "fma_add_mul_f32": int(fast), "rest": int(rest),
                                                       "classes_measured": pk.get("valu_trans_wave_instructions") is not None}
        floor_ms, mix = issue_floor_ms(roofline_main["kernel"])
        if floor_ms:
            floor_lo = issue_floor_ms(roofline_main["kernel"], 2.0)[0]
            guide_ms = (2.0 * (mix["wave_instructions"] - mix["transcendental"] - mix["fp64"]) + 8.0 * mix["transcendental"] +
                        4.0 * mix["fp64"]) / (4 * 256) / 2.4e9 * 1e3
            roofline_main["valu"] = {"wave_instructions_per_launch": mix["wave_instructions"],
                                     "fma_add_mul_f32_per_launch": mix["fma_add_mul_f32"],
                                     "transcendental_per_launch": mix["transcendental"], "fp64_per_launch": mix["fp64"],
                                     "other_per_launch": mix["rest"],
                                     "per_class_counters": mix["classes_measured"],
                                     "clocks_per_instruction": ("MEASURED (profiles/r4_micro/valu_issue.txt): f32 fma / add / mul 2, "
                                                                "min / max / compare / select / convert / packed 4, transcendental 8 "
                                                                "(fp64 priced 4); 1024 SIMDs x 2.4 GHz peak clock. The counters "
                                                                "do not split `other` into its 2-clock (moves, logic, integer add) and "
                                                                "4-clock members: issue_floor_ms prices it at 4, issue_floor_ms_low at 2"),
                                     "issue_floor_ms": round(floor_ms, 4), "issue_floor_ms_low": round(floor_lo, 4),
                                     "frac_of_issue_peak": round(floor_ms / roofline_main["launch_ms"], 3),
                                     "frac_of_issue_peak_low": round(floor_lo / roofline_main["launch_ms"], 3),
                                     "guide_pricing": {"note": "every non-transcendental instruction at 2 clocks (MI355X_MICROARCH.md)",
                                                       "issue_floor_ms": round(guide_ms, 4),
                                                       "frac_of_issue_peak": round(guide_ms / roofline_main["launch_ms"], 3)}}
            # the whole frame against the same floor: every kernel's vector instructions x its launches per
            # frame, at the frame rate of the timed region — the chip-wide VALU-issue utilisation
            per_frame = {"keygen_kernel": 1, kernel_names["depth_sort"]: table["depth_sort"]["launches"],
                         "project_kernel": 1, "bin_kernel": 1, "raster_scan_kernel": 1}
            floors = {k: issue_floor_ms(k)[0] for k in per_frame}
            if scan_mode and all(floors.values()):
                frame_issue_ms = sum(floors[k] * m for k, m in per_frame.items())
                roofline_main["valu"]["frame"] = {
                    "wave_instructions_per_frame": int(sum(pmc_kernels[k]["valu_wave_instructions"] * m for k, m in per_frame.items())),
                    "issue_floor_ms": round(frame_issue_ms, 4),
                    "ms_per_frame": round(1e3 * dt / args.steps, 4),
                    "frac_of_issue_peak": round(frame_issue_ms / (1e3 * dt / args.steps), 3)}
        # `bound` names what binds: the HBM figures stay what the contract defines (algorithmic bytes of the launch /
        # its duration against the 8 TB/s peak), `valu` is the floor the kernel is actually nearest to
        hbm_frac = roofline_main["frac"]
        valu_frac = roofline_main.get("valu", {}).get("frac_of_issue_peak")
        roofline_main["bound"] = "valu" if (valu_frac is None or valu_frac >= hbm_frac) else "hbm"
        roofline_main["bound_note"] = ("achieved / peak / frac are the HBM figures of the contract (algorithmic bytes per launch / launch "
                                       "duration / 8 TB/s); what binds the kernel is vector-instruction issue (`valu`, priced at the "
                                       "rates measured on this chip): neither HBM nor MFMA binds this algorithm")
        roofline_main["in_flight"] = roofline

        # "Msplats/s sorted": keygen + depth sort only (blocking calls, every one timed)
        plugin.set_profiling_stride(1)
        for _ in range(3):
            plugin.sort(handle, view, settings, download=False)
        t0 = time.perf_counter()
        reps = max(args.steps, 10)
        sort_dev_ms = 0.0
        for _ in range(reps):
            plugin.sort(handle, view, settings, download=False)
            sort_dev_ms += plugin.stats()["total_ms"]
        sort_wall = (time.perf_counter() - t0) / reps
        sort_dev_ms /= reps
        sort_bytes = plugin.stats()["algorithmic_bytes"]

        # The same stages with EVERY splat drawable (D = N): `sort.device_ms` above is the headline camera's sort, whose
        # list is 88 % culled sentinels that never go through a digit pass. SortMode::Rayon / Std never cull
        # (src/sort/rayon.rs:86-104) and a SortMode::Radix camera that sees the whole cloud keys every splat
        # (src/sort/radix.wgsl:109-279): four onesweep passes over N pairs (a frame with more than 786 k drawable pairs
        # does not take the bucket path). GB/s on SURVEY 8(d)'s 88 B per splat (16 positions + 8 pairs + 4 x 16).
        def sort_rate(h, n, vw, st_, reps_):
            for _ in range(3):
                plugin.sort(h, vw, st_, download=False)
            ms, kg, ds = 0.0, 0.0, 0.0
            for _ in range(reps_):
                plugin.sort(h, vw, st_, download=False)
                stt = plugin.stats()
                ms += stt["total_ms"]; kg += stt["stage_ms"]["keygen"]; ds += stt["stage_ms"]["depth_sort"]
            ms, kg, ds = ms / reps_, kg / reps_, ds / reps_
            stt = plugin.stats()
            launches = 1 if stt["sort_path"] == "bucket" else stt["depth_passes"]   # one bucket-sort launch, or a launch per digit place
            moved = 40 if stt["sort_path"] == "bucket" else 16 + 8 + 16 * stt["depth_passes"]
            return {"device_ms": round(ms, 4), "keygen_ms": round(kg, 4), "depth_sort_ms": round(ds, 4),
                    "depth_sort_launches": launches, "ms_per_launch": round(ds / max(launches, 1), 4),
                    "drawable": stt["draw_count"], "splats": n, "sort_path": stt["sort_path"],
                    "Msplats_per_s": round(n / (ms * 1e-3) / 1e6, 1) if ms > 0 else None,
                    "GBps_on_88B_per_splat": round(88.0 * n / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                    "pct_hbm_peak": round(100 * 88.0 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 2) if ms > 0 else None,
                    # what the path actually MOVES (round 5's verdict: the 88-byte figure is the contract's four passes; the
                    # bucket path reads 16 B of position and writes, reads and writes the 8-byte pair once each: 40 B per splat)
                    "bytes_moved_per_splat": moved,
                    "GBps_moved": round(moved * n / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                    "pct_hbm_peak_moved": round(100 * moved * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 2) if ms > 0 else None}
        from bevy_gaussian_splatting_amd import SortMode, View, transform_from
        far_view = View.perspective(transform_from((0.0, 0.0, 120.0), (0.0, 0.0, 0.0, 1.0)), WIDTH, HEIGHT)  # sees all of U(-20, 20)^3
        sort_all = {"note": "keygen + depth sort with D = N drawable pairs (nothing culled): blocking bgs_sort calls, device time "
                            "by HIP events; 88 B per splat = SURVEY 8(d) bytes_sort at 4 digit places",
                    "1m_rayon": sort_rate(handle, args.splats, view, CloudSettings(sort_mode=SortMode.Rayon), max(args.steps, 10)),
                    "1m_radix_whole_cloud_in_view": sort_rate(handle, args.splats, far_view, settings, max(args.steps, 10))}
        if os.environ.get("BGS_BENCH_SKIP_5M") != "1":   # (BASELINE configs[2]'s cloud: ~10 s of host time to make and upload)
            c5 = random_gaussians_3d_seeded(5_000_000, 3)
            h5 = plugin.upload(c5)
            sort_all["5m_rayon"] = sort_rate(h5, 5_000_000, view, CloudSettings(sort_mode=SortMode.Rayon), 10)
            sort_all["5m_radix_whole_cloud_in_view"] = sort_rate(h5, 5_000_000, far_view, settings, 10)
            h5.free()
            del c5
        # ... and whole FRAMES of such a view (the camera outside the cloud, every splat keyed, projected and drawn: what an
        # object-centric scene is; small on screen, so the quads are small): the frame rate that the 768-bucket sort serves
        plugin.reset_adaptive_state()
        plugin.set_profiling(0)
        plugin.set_pipeline_depth(lanes)
        plugin.set_pipeline_streams(streams)
        _, _, st_far, dts_far = measure(plugin, handle, far_view, settings, args.steps, args.warmup, depth=lanes, trials=5)
        plugin.set_pipeline_depth(1)
        _, _, _, dts_far1 = measure(plugin, handle, far_view, settings, args.steps, args.warmup, depth=1, trials=3)
        plugin.set_profiling(2)
        sort_all["frames_1m_whole_cloud_in_view"] = {
            "value": round(args.steps / statistics.median(dts_far), 2), "unit": "frames/s",
            "single_stream_value": round(args.steps / statistics.median(dts_far1), 2),
            "visible_splats": st_far["visible_count"], "drawable": st_far["draw_count"], "sort_path": st_far.get("sort_path"),
            "camera": "(0, 0, 120) looking at the cloud, fov pi/4: all 1 M splats inside the frustum"}
        plugin.reset_adaptive_state()

        # scene-like variant: global_scale = 0.05
        plugin.set_profiling_stride(STRIDE)
        plugin.set_pipeline_depth(DEPTH)
        plugin.set_pipeline_streams(max(0, min(8, args.streams)))
        s2 = CloudSettings(global_scale=0.05)
        # measured like the headline: several timed regions of --steps frames, the median reported (one 20-frame region
        # pays the fill and drain of the lanes once per region either way, but a single one is at the mercy of one late lane)
        side_trials = max(5, min(trials, 40))
        plugin.set_profiling(0)
        dt2, _, _, dts2 = measure(plugin, handle, view, s2, args.steps, args.warmup, depth=DEPTH, trials=side_trials,
                                  busy_warm_frames=400)
        plugin.set_profiling(2)
        _, stage2, st2 = measure(plugin, handle, view, s2, max(args.steps // 2, 4 * STRIDE), 4, depth=DEPTH)
        ms2 = sum(stage2.values())
        plugin.set_pipeline_depth(1)
        dt2s, _, _, _ = measure(plugin, handle, view, s2, args.steps, args.warmup, trials=5)

        # ---- a cloud with TRAINED-ASSET statistics (round 6; gaussian.py trained_like_gaussians_3d_seeded): surfaces, flat
        # log-normal splats, bimodal opacity, DC-dominated SH with colours in [0, 1] — the workload the reference is used
        # on (it demos trained scenes), which its own random generator does not resemble. Same size, camera and settings
        # as the headline; measured like the scene-like leg; the quad-size histogram says which regime it is.
        from bevy_gaussian_splatting_amd import trained_like_gaussians_3d_seeded
        plugin.set_pipeline_depth(DEPTH)
        plugin.set_pipeline_streams(max(0, min(8, args.streams)))
        cloud_t = trained_like_gaussians_3d_seeded(args.splats, SEED + 5)
        handle_t = plugin.upload(cloud_t)
        plugin.reset_adaptive_state()
        plugin.set_profiling(0)
        dt_t, _, _, dts_t = measure(plugin, handle_t, view, settings, args.steps, args.warmup, depth=DEPTH, trials=side_trials,
                                    busy_warm_frames=400)
        plugin.set_profiling(2)
        _, stage_t, st_t = measure(plugin, handle_t, view, settings, max(args.steps // 2, 4 * STRIDE), 4, depth=DEPTH)
        plugin.set_pipeline_depth(1)
        dt_ts, stage_ts, _, _ = measure(plugin, handle_t, view, settings, args.steps, args.warmup, trials=5)
        trained_like = {"value": round(args.steps / dt_t, 2), "unit": "frames/s", "trials": len(dts_t),
                        "single_stream_value": round(args.steps / dt_ts, 2),
                        "single_stream_stage_ms": {k: round(x, 4) for k, x in stage_ts.items() if x},
                        "visible_splats": st_t["visible_count"], "coarse_entries": st_t["instance_count"],
                        "sort_path": st_t.get("sort_path"),
                        "cloud": f"trained_like_gaussians_3d_seeded({args.splats}, {SEED + 5}): 96 surface patches, log-normal flat "
                                 "splats, 60 / 40 % opaque / faint, colours in [0, 1]; CloudSettings::default(), the headline's camera",
                        "quads": quad_size_histogram(cloud_t, view, settings),
                        "quads_dense_headline": quad_size_histogram(cloud, view, settings),
                        "quads_scene_like": quad_size_histogram(cloud, view, s2)}
        handle_t.free()
        del cloud_t
        plugin.reset_adaptive_state()

        # ---- the same workload on a camera with Msaa::Off: one sample per pixel (what rounds 1-3 reported) -------
        from bevy_gaussian_splatting_amd import View as _View
        view_off = _View.headless(WIDTH, HEIGHT, yaw=rank * math.pi / 4.0, order=rank, msaa_samples=1)
        plugin.set_pipeline_depth(lanes)
        plugin.set_pipeline_streams(streams)
        plugin.set_profiling(0)
        dt_off, _, _, dts_off = measure(plugin, handle, view_off, settings, args.steps, args.warmup, depth=lanes,
                                        trials=side_trials, busy_warm_frames=400)
        dt_off2, _, _, _ = measure(plugin, handle, view_off, s2, args.steps, args.warmup, depth=lanes, trials=side_trials,
                                   busy_warm_frames=400)
        plugin.set_profiling(2)
        plugin.set_pipeline_depth(1)
        dt_off1, stage_off1, _, _ = measure(plugin, handle, view_off, settings, args.steps, args.warmup, trials=5)
        msaa_off = {"value": round(args.steps / dt_off, 2), "unit": "frames/s", "sample_count": 1,
                    "trials": len(dts_off), "scene_like_value": round(args.steps / dt_off2, 2),
                    "single_stream_value": round(args.steps / dt_off1, 2),
                    "single_stream_stage_ms": {k: round(x, 4) for k, x in stage_off1.items() if x},
                    "note": "Camera with Msaa::Off (bgs_view.sample_count = 1): the configuration rounds 1-3 reported as the headline"}

        # ---- latency of the frames that cannot lean on what completed frames taught the context -------------------
        plugin.set_pipeline_streams(max(0, min(8, args.streams)))

        def blocking_ms(v):
            t0 = time.perf_counter()
            plugin.render(handle, v, settings, download=False)
            plugin.synchronize()
            return 1e3 * (time.perf_counter() - t0)
        plugin.set_async(False)
        firsts, cuts, steady = [], [], []
        cut_view = _View.headless(WIDTH, HEIGHT, yaw=rank * math.pi / 4.0 + math.radians(137.0), order=rank)
        for _ in range(5):
            plugin.reset_adaptive_state()
            firsts.append(blocking_ms(view))          # nothing learnt: digit passes, first-guess capacities (+ a re-run)
            for _ in range(6):
                blocking_ms(view)
            steady.append(blocking_ms(view))
            cuts.append(blocking_ms(cut_view))        # warmed up on another view: stale splitters / hints
        latency = {"first_frame_ms": round(statistics.median(firsts), 4), "after_cut_ms": round(statistics.median(cuts), 4),
                   "steady_blocking_frame_ms": round(statistics.median(steady), 4),
                   "note": "blocking frames (host wall clock incl. launch + synchronize), median of 5; first = after "
                           "bgs_reset_adaptive_state, cut = a 137-degree yaw in a context warmed up on the headline view"}
        plugin.set_async(True)
        plugin.reset_adaptive_state()

        # ---- moving camera (the reference's use case: an interactive camera re-sorts whenever it moves,
        # src/sort/mod.rs:153-194). The static view above is the best case for everything the context learns from
        # completed frames (bucket-sort splitters, draw-count hint, list capacity, supertile level); here the camera
        # yaws 0.25 degrees per frame about the headless pose and makes a hard cut of 137 degrees every 500 frames.
        # Same lanes / streams as the headline; views marshalled outside the timed region.
        from bevy_gaussian_splatting_amd import View
        base_yaw = rank * math.pi / 4.0
        orbit_frames = 2000
        orbit_views = []
        for i in range(orbit_frames):
            deg = 0.25 * i + 137.0 * (i // 500)
            orbit_views.append(plugin.prepare(View.headless(WIDTH, HEIGHT, yaw=base_yaw + math.radians(deg)), settings))
        plugin.set_pipeline_depth(lanes)
        plugin.set_pipeline_streams(streams)
        plugin.set_profiling(0)
        k_orbit = max(args.steps, 250)
        ac0 = plugin.adaptive_counters()
        _, _, st_o, dts_o = measure(plugin, handle, view, settings, k_orbit, min(args.warmup, 20), depth=lanes, trials=8,
                                    views=orbit_views)
        ac1 = plugin.adaptive_counters()
        plugin.set_profiling(2)
        dt_o = statistics.median(dts_o)
        orbit = {"value": round(k_orbit / dt_o, 2), "unit": "frames/s", "vs_static": round((k_orbit / dt_o) / (fps / world), 3),
                 "camera": "yaw += 0.25 deg per frame about the headless pose, a 137-deg cut every 500 frames",
                 "frames_per_trial": k_orbit, "trials_ms": [round(1e3 * x, 3) for x in dts_o],
                 "lanes": lanes, "streams": streams,
                 "adaptive_counters_delta": {k: ac1[k] - ac0[k] for k in ("bucket_frames", "onesweep_frames", "reruns_sort",
                                                                         "reruns_lists", "reruns_instances", "level_changes")},
                 "supertile_level_at_end": ac1["supertile_level"], "visible_splats_last_frame": st_o["visible_count"]}
        # ---- one SUSTAINED region: >= 3 s of the orbit camera, back to back, no barriers inside. The headline is the
        # median of many 20-frame regions (~1 ms of GPU work each): a monitor that samples the device a few times per
        # second sees an idle GPU during such a run. This leg is what it can see, and the rate a long-running host gets.
        plugin.set_profiling(0)
        orbit_prepared = orbit_views
        t_s0 = time.perf_counter()
        frames_s = 0
        while True:
            for pv_o in orbit_prepared[:500]:
                plugin.render(handle, pv_o, download=False)
            frames_s += 500
            if time.perf_counter() - t_s0 >= 3.0:
                break
        plugin.synchronize()
        dt_s = time.perf_counter() - t_s0
        plugin.set_profiling(2)
        sustained = {"value": round(frames_s / dt_s, 2), "unit": "frames/s", "seconds": round(dt_s, 3), "frames": frames_s,
                     "camera": "the orbit leg's (0.25 deg per frame, the first 500 poses cycled)", "lanes": lanes, "streams": streams,
                     "vs_headline": round((frames_s / dt_s) / (fps / world), 3)}
        plugin.reset_adaptive_state()

        # one blocking frame of the benchmarked view for the whole-frame parity check (download outside any timing)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        gpu_frame = plugin.render(handle, view, settings)
        gpu_entries = plugin.sort(handle, view, settings)
        plugin.set_async(True)

        # the named instance-sort pipeline (tile-major|depth radix sort) on the same workload
        plugin.set_binning("sort")
        plugin.set_profiling_stride(1)  # blocking frames: time every one
        dt3, stage3, st3, _ = measure(plugin, handle, view, settings, max(args.steps // 3, 3), 2, trials=3)
        dt4, stage4, st4, _ = measure(plugin, handle, view, s2, max(args.steps // 3, 3), 2, trials=3)
        plugin.set_binning("scan")

        out = {
            "metric": "frames/sec @1080p, 1M-splat 3DGS (sort + rasterize every frame)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "timing": {"trials": len(dts), "trials_ms_min_median_max": [round(1e3 * min(dts), 4), round(1e3 * dt, 4), round(1e3 * max(dts), 4)],
                       "trials_ms_first_10": [round(1e3 * x, 4) for x in dts[:10]], "reported": "median trial",
                       "gpu_seconds_in_timed_regions": round(sum(dts), 3),
                       "note": "every trial is a barrier + sync, --steps frames, barrier + sync; 3000 untimed frames (~0.2 s) and a "
                               "pilot region precede the first one; the number of trials puts >= 0.5 s inside timed regions"},
            "config": {"workload": f"{args.splats}-splat 3DGS f32 planar cloud (seed {SEED}, reference random_gaussians_3d "
                                   "distributions), 1920x1080, SH degree 3, CloudSettings::default(), "
                                   "examples/headless.rs camera incl. its Msaa (Bevy's default Sample4: 4 samples per pixel, "
                                   "resolved frame); one camera per GPU",
                       "sample_count": view.msaa_samples,
                       "parallelism": f"views{world}", "sort": "radix32", "global_scale": 1.0,
                       "frames_in_flight": lanes, "lanes": lanes, "streams": streams},
            "build_id": plugin.build_id(),
            "single_stream": single,
            "msaa_off": msaa_off,
            "latency": latency,
            "orbit": orbit,
            "sustained": sustained,
            "trained_like": trained_like,
            "per_rank": per_rank,
            "roofline": roofline_main,
            "frame": {"device_ms": round(frame_ms, 4), "algorithmic_GB": round(frame_bytes / 1e9, 4),
                      "GBps": round(frame_gbs, 1), "pct_hbm_peak": round(100 * frame_gbs / HBM_PEAK_GBS, 2),
                      "effective_GBps": round(eff_gbs, 1),
                      "effective_pct_hbm_peak": round(100 * eff_gbs / HBM_PEAK_GBS, 2),
                      "effective_pct_measured_peak": round(100 * eff_gbs / measured, 2) if measured else None,
                      "visible_splats": st["visible_count"], "coarse_entries": st["instance_count"],
                      "tile_instances": st3["instance_count"],
                      "tile_instances_note": ("V visible splats overlap I (tile, splat) pairs (counted by the instance-sort pipeline "
                                              "below); the default pipeline bins them into `coarse_entries` (rank, tile-rect) list "
                                              "entries and expands lazily, only as far as a tile composites"),
                      "gather_ms_per_step": round(gather_ms[0] / max(args.steps + args.warmup, 1), 4),
                      "gathered_format": "Rgba8UnormSrgb (packed-only frames: no f32 target)" if dist is not None else None,
                      "gather_batch_frames": GATHER_BATCH if dist is not None else None,
                      "gather_backend": (None if dist is None else ("bgs_comm_gather (RCCL ncclGather behind the C ABI)" if native_gather
                                                                    else f"torch.distributed.gather (fallback: {native_err})")),
                      "gather_selfcheck": gather_selfcheck,
                      "frames_gathered_on_rank0": batcher.frames_received if batcher is not None else None,
                      "frames_issued_per_rank": issued_main},
            "stages": stages,
            "sort_msplats_per_s": round(args.splats / (sort_dev_ms * 1e-3) / 1e6, 1) if sort_dev_ms > 0 else None,
            "sort": {"device_ms": round(sort_dev_ms, 4), "wall_ms": round(sort_wall * 1e3, 4),
                     "GBps": round(sort_bytes / (sort_dev_ms * 1e-3) / 1e9, 1) if sort_dev_ms > 0 else None},
            "sort_all_visible": sort_all,
            "scene_like": {"global_scale": 0.05, "value": round(args.steps / dt2, 2), "unit": "frames/s", "trials": len(dts2),
                           "single_stream_value": round(args.steps / dt2s, 2),
                           "device_ms": round(ms2, 4), "visible_splats": st2["visible_count"],
                           "coarse_entries": st2["instance_count"], "tile_instances": st4["instance_count"],
                           "GBps": round(st2["algorithmic_bytes"] / (ms2 * 1e-3) / 1e9, 1) if ms2 > 0 else None},
            "binning": st["binning"], "sort_path": st.get("sort_path"),
            "instance_sort_pipeline": {
                "note": "bgs_set_binning(SORT): (tile,splat) instances + stable radix sort on tile ids + ranges",
                "value": round(max(args.steps // 3, 3) / dt3, 2), "unit": "frames/s",
                "visible_splats": st3["visible_count"], "tile_instances": st3["instance_count"],
                "stage_ms": {k: round(v, 4) for k, v in stage3.items()},
                "GBps": round(st3["algorithmic_bytes"] / (max(sum(stage3.values()), 1e-9) * 1e-3) / 1e9, 1),
                "scene_like_value": round(max(args.steps // 3, 3) / dt4, 2),
                "scene_like_stage_ms": {k: round(v, 4) for k, v in stage4.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cloud, view, settings, gpu_frame, gpu_entries)
            out["parity"] = out["cpu_baseline"].pop("parity")
            out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 5)
            cb = out["cpu_baseline"]
            cb["sort_msplats_per_s"] = round(cb["sort_msplats_per_s"], 2)
            cb["sort_std_msplats_per_s"] = round(cb["sort_std_msplats_per_s"], 2)
            cb["one_core"]["value"] = round(cb["one_core"]["value"], 6)
            cb["one_core"]["sort_msplats_per_s"] = round(cb["one_core"]["sort_msplats_per_s"], 2)
        else:
            out["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
    handle.free()
    plugin.close()
    if dist is not None:
        dist.destroy_process_group()
    # RCCL prints its version banner through C stdio, which is block-buffered on a pipe and would come out at
    # process exit, i.e. AFTER the result: flush it first so that the JSON line is the last line on stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)
        if out.get("parity") and not out["parity"]["ok"]:
            raise SystemExit(f"bench.py: the benchmarked frame FAILED the whole-frame parity check: {out['parity']}")


if __name__ == "__main__":
    main()
