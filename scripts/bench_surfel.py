"""2DGS surfel (aabb) stage times, optionally with debug / ablation flags and forced supertile levels:
python scripts/bench_surfel.py [flags ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianMode, GaussianSplattingPlugin, View, random_gaussians_3d_seeded
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(1_000_000, 4))
v = View.headless(1920, 1080)
flags = [int(a, 0) for a in sys.argv[1:]] or [0]
for gs in (1.0, 0.05):
    s = CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, global_scale=gs)
    for fl in flags:
        p.set_debug_flags(fl)
        for i in range(12):
            p.render(h, v, s, download=False)
        st = p.stats()
        print(gs, hex(fl), {k: round(x * 1e3, 1) for k, x in st["stage_ms"].items() if x}, st["visible_count"], st["instance_count"], st["list_capacity"])
