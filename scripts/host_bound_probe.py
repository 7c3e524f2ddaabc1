"""Is the pipelined path host-bound? Drive T threads x C contexts (ctypes releases the GIL)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
cloud = random_gaussians_3d_seeded(1_000_000, 2)
v = headless_view(0); s = CloudSettings()
def worker(p, h, K, depth):
    for _ in range(K):
        p.render(h, v, s, download=False)
    p.synchronize()
for threads, depth in ((1, 3), (2, 2), (2, 3), (3, 2), (4, 1), (4, 2)):
    ps = [GaussianSplattingPlugin(0) for _ in range(threads)]
    hs = [p.upload(cloud) for p in ps]
    for p in ps:
        p.set_async(True); p.set_profiling(0); p.set_pipeline_depth(depth)
    for p, h in zip(ps, hs): worker(p, h, 6, depth)
    K = 80
    ts = [threading.Thread(target=worker, args=(p, h, K, depth)) for p, h in zip(ps, hs)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print(f"threads={threads} depth={depth} streams={threads*depth}: {threads*K/dt:8.1f} frames/s")
    for p, h in zip(ps, hs): h.free(); p.close()
