"""Liveness stress: many chained-scan kernels of several lanes sharing the chip (5 M splats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded
v = View.headless(1920, 1080)
big = random_gaussians_3d_seeded(5_000_000, 3)
fails = 0
for depth in (3, 4, 5, 6, 8, 3, 4, 3):
    p = GaussianSplattingPlugin(0)
    h = p.upload(big)
    p.set_async(True); p.set_pipeline_depth(depth)
    try:
        for i in range(2 * depth):  # warm-up: every lane allocates its buffers (hipMalloc of GBs)
            p.render(h, v, CloudSettings(), download=False)
        p.synchronize()
        t0 = time.perf_counter()
        for gs in (1.0, 0.05):
            for i in range(24):
                p.render(h, v, CloudSettings(global_scale=gs), download=False)
        p.synchronize()
        print(f"depth {depth}: OK {48 / (time.perf_counter() - t0):.0f} fps", flush=True)
    except Exception as e:
        fails += 1
        print(f"depth {depth}: FAIL {str(e)[:90]}", flush=True)
    h.free(); p.close()
print("fails", fails)
