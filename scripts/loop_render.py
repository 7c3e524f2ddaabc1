"""Render one workload in a loop of blocking frames (for rocprofv3): python scripts/loop_render.py <what> [frames]
<what>: a global_scale for the 1 M-splat headline cloud (1.0 = dense, 0.05 = scene-like), or one of
surfel (1 M 2DGS, aabb) | 5m_scene | 5m_dense (5 M f16 cloud)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianMode, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
what = sys.argv[1] if len(sys.argv) > 1 else "1.0"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = GaussianSplattingPlugin(0)
if what == "surfel":
    cloud, s = random_gaussians_3d_seeded(1_000_000, 2), CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True)
elif what in ("5m_scene", "5m_dense"):
    cloud, s = random_gaussians_3d_seeded(5_000_000, 3).to_f16(), CloudSettings(global_scale=0.05 if what == "5m_scene" else 1.0)
else:
    cloud, s = random_gaussians_3d_seeded(1_000_000, 2), CloudSettings(global_scale=float(what))
h = p.upload(cloud)
v = headless_view(0)
for _ in range(frames):
    p.render(h, v, s, download=False)
print(p.stats())
