"""Experiment: does the GPU overlap frames from two independent contexts (two HIP streams)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
cloud = random_gaussians_3d_seeded(1_000_000, 2)
v = headless_view(0)
for gs in (1.0, 0.05):
    s = CloudSettings(global_scale=gs)
    for nctx in (1, 2, 3):
        ps = [GaussianSplattingPlugin(0) for _ in range(nctx)]
        hs = [p.upload(cloud) for p in ps]
        for p in ps:
            p.set_async(True); p.set_profiling(0)
        for _ in range(5):
            for p, h in zip(ps, hs): p.render(h, v, s, download=False)
        for p in ps: p.synchronize()
        K = 60
        t0 = time.perf_counter()
        for _ in range(K):
            for p, h in zip(ps, hs): p.render(h, v, s, download=False)
        for p in ps: p.synchronize()
        dt = time.perf_counter() - t0
        print(f"gs={gs} contexts={nctx}: {nctx*K/dt:8.1f} frames/s aggregate ({1e6*dt/(nctx*K):.1f} us/frame)")
        for p, h in zip(ps, hs):
            h.free(); p.close()
