"""Kernel ablation on the headline workload: time stages with parts of a kernel switched off
(bgs_set_debug_flags). Images are WRONG under non-zero flags; this only measures."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view

flag_sets = [int(a, 0) for a in sys.argv[1:]] or [0]
cloud = random_gaussians_3d_seeded(1_000_000, 2)
p = GaussianSplattingPlugin(0)
h = p.upload(cloud)
v = headless_view(0)
for gs in (1.0, 0.05, 1.0, 0.05):
    s = CloudSettings(global_scale=gs)
    for flags in flag_sets:
        p.set_debug_flags(flags)
        acc = None
        for i in range(25):
            p.render(h, v, s, download=False)
            if i >= 5:
                ms = p.stats()["stage_ms"]
                acc = ms if acc is None else {k: acc[k] + ms[k] for k in ms}
        print(f"gs={gs} flags={flags:#x}", {k: round(1e3 * x / 20, 1) for k, x in acc.items() if x}, "us")
p.set_debug_flags(0)
