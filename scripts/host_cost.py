"""Host cost of enqueueing one frame: with 8 free lanes, time bursts of 8 render calls (nothing to
wait for inside the call), then the sync. python scripts/host_cost.py [profiling level] [nograph]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(1_000_000, 2))
v = headless_view(0)
s = CloudSettings()
p.set_async(True)
p.set_pipeline_depth(8)
p.set_profiling(int(sys.argv[1]) if len(sys.argv) > 1 else 0)   # 0: frames replay a hipGraph; 2: direct launches
p.set_graphs(not (len(sys.argv) > 2 and sys.argv[2] == "nograph"))
pv = p.prepare(v, s)
for _ in range(32):
    p.render(h, pv, download=False)
p.synchronize()
enq, tot = [], []
for rep in range(20):
    t0 = time.perf_counter()
    for _ in range(8):
        p.render(h, pv, download=False)
    t1 = time.perf_counter()
    p.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) / 8 * 1e6)
    tot.append((t2 - t0) / 8 * 1e6)
enq.sort(); tot.sort()
print(f"enqueue per frame: median {enq[10]:.1f} us (min {enq[0]:.1f}); burst of 8 incl. sync: {tot[10]:.1f} us/frame")
