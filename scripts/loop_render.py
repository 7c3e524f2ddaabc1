"""Render the headline workload in a loop (for rocprofv3): python scripts/loop_render.py <global_scale> [frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
gs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(1_000_000, 2))
v = headless_view(0)
s = CloudSettings(global_scale=gs)
for _ in range(frames):
    p.render(h, v, s, download=False)
print(p.stats())
