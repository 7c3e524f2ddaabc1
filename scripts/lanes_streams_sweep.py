"""Frames/s against the number of frame lanes and the streams they are multiplexed onto (bgs_set_pipeline_depth / _streams):
long regions (300 frames, best of 4) and the driver's short ones (20 frames between synchronisations, median of 60).
python scripts/lanes_streams_sweep.py ["dense scene trained"]"""
import os, statistics, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevy_gaussian_splatting_amd import (CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded,
                                         trained_like_gaussians_3d_seeded)
want = (sys.argv[1] if len(sys.argv) > 1 else "dense scene trained").split()
p = GaussianSplattingPlugin(0)
v = View.headless(1920, 1080)
clouds = {}
for name in want:
    key = "trained" if name == "trained" else "random"
    if key not in clouds:
        clouds[key] = p.upload(trained_like_gaussians_3d_seeded(1_000_000, 7) if key == "trained" else random_gaussians_3d_seeded(1_000_000, 2))
    h, s = clouds[key], (CloudSettings(global_scale=0.05) if name == "scene" else CloudSettings())
    for depth, streams in ((8, 4), (4, 4), (8, 8), (6, 6), (5, 4), (4, 2), (3, 3), (6, 4), (8, 4)):
        p.reset_adaptive_state(); p.set_async(True); p.set_pipeline_streams(streams); p.set_pipeline_depth(depth); p.set_profiling(0)
        for _ in range(80): p.render(h, v, s, download=False)
        p.synchronize()
        best = 0.0
        for _ in range(4):
            t0 = time.perf_counter()
            for _ in range(300): p.render(h, v, s, download=False)
            p.synchronize()
            best = max(best, 300 / (time.perf_counter() - t0))
        short = []
        for _ in range(60):
            t0 = time.perf_counter()
            for _ in range(20): p.render(h, v, s, download=False)
            p.synchronize()
            short.append(20 / (time.perf_counter() - t0))
        print(f"{name:8s} lanes {depth} streams {streams}: {best:9.1f} fps in 300-frame regions   {statistics.median(short):9.1f} in 20-frame regions", flush=True)
        p.set_async(False)
