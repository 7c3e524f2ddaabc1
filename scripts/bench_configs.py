"""Throughput of the other BASELINE.json configs (parity for these is in tests/test_gpu_parity.py):
configs[0] 10k/256x256, configs[2] 5M f16 1080p, configs[3] 1M 2DGS (surfel AABB path and OBB)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import (CloudSettings, GaussianMode, GaussianSplattingPlugin, View,
                                         random_gaussians_3d_seeded)

def run(p, h, v, s, steps=60, warm=30, depth=8):
    # throughput with `depth` lanes (on the default 4 streams), no stage events in the timed loop
    p.set_async(True); p.set_pipeline_depth(depth); p.set_profiling(0)
    for _ in range(warm): p.render(h, v, s, download=False)
    p.synchronize()
    dts = []
    for _ in range(5):   # five timed regions, the median reported (like bench.py's legs)
        t0 = time.perf_counter()
        for _ in range(steps): p.render(h, v, s, download=False)
        p.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[2]
    p.set_profiling(2); p.set_profiling_stride(4)
    for _ in range(8): p.render(h, v, s, download=False)
    p.synchronize()
    st = p.stats()
    p.set_pipeline_depth(1)
    for _ in range(warm): p.render(h, v, s, download=False)
    p.synchronize()
    dts1 = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): p.render(h, v, s, download=False)
        p.synchronize()
        dts1.append(time.perf_counter() - t0)
    dt1 = sorted(dts1)[1]
    st1 = p.stats()
    p.set_async(False)
    return {"frames_per_s": round(steps / dt, 1), "single_stream_frames_per_s": round(steps / dt1, 1),
            "single_stream_stage_ms": {k: round(x, 4) for k, x in st1["stage_ms"].items() if x},
            "visible_splats": st["visible_count"], "coarse_entries": st["instance_count"], "sort_path": st["sort_path"],
            "list_capacity": st["list_capacity"], "reruns": st["regrow_count"]}

out = {}
p = GaussianSplattingPlugin(0)
SAMPLES = ((4, ""), (1, "_msaa_off"))
c = random_gaussians_3d_seeded(10_000, 1); h = p.upload(c)
for m, tag in SAMPLES:
    out["cfg0_10k_256x256" + tag] = run(p, h, View.headless(256, 256, msaa_samples=m), CloudSettings())
h.free()
c = random_gaussians_3d_seeded(5_000_000, 3).to_f16(); h = p.upload(c)
for gs in (1.0, 0.05):
    for m, tag in SAMPLES:
        out[f"cfg2_5M_f16_1080p_gs{gs}{tag}"] = run(p, h, View.headless(1920, 1080, msaa_samples=m), CloudSettings(global_scale=gs), steps=40)
h.free()
c = random_gaussians_3d_seeded(1_000_000, 4); h = p.upload(c)
for name, kw in (("surfel_aabb", {"aabb": True}), ("obb", {})):
    for gs in (1.0, 0.05):
        s = CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, global_scale=gs, **kw)
        for m, tag in SAMPLES:
            out[f"cfg3_1M_2dgs_{name}_gs{gs}{tag}"] = run(p, h, View.headless(1920, 1080, msaa_samples=m), s, steps=40)
h.free()
print(json.dumps(out, indent=1))
