"""Plain enqueue loop vs enqueue + pop-the-oldest-when-full, f32 target vs packed-only sRGB8, by lanes / streams:
python scripts/packed_loop.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(1_000_000, 2))
v = headless_view(0); s = CloudSettings()
p.set_async(True); p.set_profiling(0)
def loop(n, pop, lanes):
    for _ in range(n):
        p.render(h, pv, download=False)
        if pop and p.frames_in_flight() >= lanes:
            p.pipeline_pop()
    if pop:
        while p.frames_in_flight():
            p.pipeline_pop()
    p.synchronize()
for packed in (0, 1):
    p.set_output_srgb8(bool(packed)); p.set_packed_only(bool(packed))
    for lanes, streams in ((6, 3), (8, 4), (8, 8), (6, 6)):
        p.set_pipeline_depth(lanes); p.set_pipeline_streams(streams)
        pv = p.prepare(v, s)
        for pop in (0, 1):
            loop(60, pop, lanes)
            t0 = time.perf_counter(); loop(800, pop, lanes); dt = time.perf_counter() - t0
            print(f"packed_only {packed} lanes {lanes} streams {streams} pop {pop}: {800/dt:.0f} fps", flush=True)
