"""Time bgs_sort alone (keygen with the ordered partition + the depth sort): python scripts/time_sort.py [n] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(n, 2))
v, s = View.headless(1920, 1080), CloudSettings()
for _ in range(30):
    p.sort(h, v, s, download=False)
p.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    p.sort(h, v, s, download=False)
p.synchronize()
dt = (time.perf_counter() - t0) / reps
st = p.stats()
print(f"n {n}: {dt * 1e6:.1f} us per sort = {n / dt / 1e9:.2f} Gsplats/s; stages {({k: round(1e3 * x, 1) for k, x in st['stage_ms'].items() if x})} path {st.get('sort_path')}")
