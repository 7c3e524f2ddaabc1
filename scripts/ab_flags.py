"""Same-box A/B of debug-flag variants on several workloads, one process (each cloud uploaded once):
python scripts/ab_flags.py "<configs>" "<flags>"    configs: dense scene surfel surfel_scene 5m_dense 5m_scene 2d_obb
                                                     flags: comma-separated debug flag words (0 = the product)
Prints, per (config, flags): frames/s with 8 lanes on 4 streams, single-stream frames/s and its stage times."""
import dataclasses
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import (CloudSettings, GaussianMode, GaussianSplattingPlugin, View,  # noqa: E402
                                         random_gaussians_3d_seeded)

configs = (sys.argv[1] if len(sys.argv) > 1 else "dense scene").split()
flags = [int(f, 0) for f in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2


def run(p, h, v, s, steps, warm=40, depth=8):
    p.set_async(True); p.set_pipeline_depth(depth); p.set_profiling(0)
    for _ in range(warm): p.render(h, v, s, download=False)
    p.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): p.render(h, v, s, download=False)
        p.synchronize()
        best = max(best, steps / (time.perf_counter() - t0))
    p.set_pipeline_depth(1); p.set_profiling(2); p.set_profiling_stride(1)
    for _ in range(warm // 2): p.render(h, v, s, download=False)
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): p.render(h, v, s, download=False)
    p.synchronize()
    dt1 = time.perf_counter() - t0
    st1 = p.stats()
    p.set_async(False)
    stages = {k: round(1e3 * x, 1) for k, x in st1["stage_ms"].items() if x}
    if st1.get("strip_tiles"):
        stages["feedback_tiles"] = st1["strip_tiles"]
    if st1.get("tile_saturation", {}).get("known"):
        ts = st1["tile_saturation"]
        stages["saturated_work"] = "%.2f%s" % (ts["work_share"], " midround" if ts["midround_exit"] else "")
    return best, steps / dt1, stages


CLOUDS = {}


def cloud(kind):
    if kind not in CLOUDS:
        if kind == "trained":
            from bevy_gaussian_splatting_amd import trained_like_gaussians_3d_seeded
            CLOUDS[kind] = trained_like_gaussians_3d_seeded(1_000_000, 7)
        else:
            CLOUDS[kind] = (random_gaussians_3d_seeded(5_000_000, 3).to_f16() if kind == "5m" else
                            random_gaussians_3d_seeded(1_000_000, 4 if kind == "2d" else 2))
    return CLOUDS[kind]


SPEC = {"trained": ("trained", CloudSettings(), 120), "dense": ("1m", CloudSettings(), 300), "scene": ("1m", CloudSettings(global_scale=0.05), 300),
        "surfel": ("2d", CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True), 60),
        "surfel_scene": ("2d", CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, global_scale=0.05), 200),
        "2d_obb": ("2d", CloudSettings(gaussian_mode=GaussianMode.Gaussian2d), 300),
        "5m_dense": ("5m", CloudSettings(), 120), "5m_scene": ("5m", CloudSettings(global_scale=0.05), 80)}
p = GaussianSplattingPlugin(0)
# BGS_AB_MSAA="1,4": samples per pixel of the views (default 4 = Msaa::Sample4, Bevy's default)
msaas = [int(x) for x in os.environ.get("BGS_AB_MSAA", "4").split(",")]
bbox = os.environ.get("BGS_AB_BBOX", "0") != "0"   # BGS_AB_BBOX=1: CloudSettings::visualize_bounding_box on every workload
handles = {}
for _ in range(rounds):   # every variant is measured `rounds` times, interleaved (clock / thermal drift shows up as spread)
    for c in configs:
        kind, s, steps = SPEC[c]
        if bbox:
            s = dataclasses.replace(s, visualize_bounding_box=True)
        if kind not in handles:
            handles[kind] = p.upload(cloud(kind))
        for fl in flags:
          for m in msaas:
            v = View.headless(1920, 1080, msaa_samples=m)
            p.set_debug_flags(fl)
            p.reset_adaptive_state()
            fps, fps1, stages = run(p, handles[kind], v, s, steps)
            print(f"{c + ('+bbox' if bbox else ''):13s} x{m} flags {fl:#10x}: {fps:9.1f} fps (8 lanes / 4 streams)  {fps1:9.1f} single stream  {json.dumps(stages)}", flush=True)
p.set_debug_flags(0)
