"""Per-stage kernel times (HIP events, blocking frames) of the 1 M f32 headline frame and the 5 M f16 frame:
python scripts/stage_times.py [n5m=5000000]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.gaussian import PlanarGaussian3dF16
p = GaussianSplattingPlugin(0)
v = View.headless(1920, 1080)
GS = [float(a) for a in os.environ.get("GS", "1.0,0.05").split(",")]   # global_scale values
FLAGS = [int(a, 0) for a in os.environ.get("FLAGS", "0").split(",")]   # debug / ablation flags, one run per value
def run(tag, h, gs):
    for fl in FLAGS:
        p.set_debug_flags(fl)
        run1(f"{tag} flags {fl:#x}", h, gs)
    p.set_debug_flags(0)
def run1(tag, h, gs):
    s = CloudSettings(global_scale=gs)
    acc = {}
    for i in range(40):
        p.render(h, v, s, download=False)
        if i >= 20:
            for k, x in p.stats()["stage_ms"].items():
                acc[k] = acc.get(k, 0.0) + x / 20.0
    st = p.stats()
    print(tag, gs, {k: round(x * 1e3, 1) for k, x in acc.items() if x}, "sum", round(sum(acc.values()) * 1e3, 1), st["sort_path"], flush=True)
if not os.environ.get("SKIP_1M"):
    h = p.upload(random_gaussians_3d_seeded(1_000_000, 2))
    for gs in GS:
        run("1M f32", h, gs)
    h.free()
n5 = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
if n5:
    c = random_gaussians_3d_seeded(n5, 3)
    h = p.upload(PlanarGaussian3dF16.from_f32(c))
    for gs in GS:
        run("5M f16", h, gs)
