"""Reproduce the blocking-render sequence of test_frame_graphs_follow_changing_inputs step by step."""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import *
from bevy_gaussian_splatting_amd.multiview import headless_view
p = GaussianSplattingPlugin(0)
clouds = [random_gaussians_3d_seeded(40_000, 61), random_gaussians_3d_seeded(25_000, 62)]
handles = [p.upload(c) for c in clouds]
def view(i, w=320, h=192):
    v = headless_view(i % 8, w, h)
    v.clear_color = (0.1 * (i % 3), 0.05, 0.2, 0.5 if i % 2 else 0.0)
    return v
steps = []
for i in range(6):
    steps.append((0, view(i), CloudSettings(global_scale=0.5)))
for i in range(4):
    steps.append((0, view(i), CloudSettings(global_scale=0.3 + 0.1 * i, global_opacity=0.7, sh_degree=i % 4)))
steps.append((0, view(1, 256, 144), CloudSettings(global_scale=0.5)))
steps.append((0, view(2, 256, 144), CloudSettings(global_scale=0.5)))
steps.append((0, view(3), CloudSettings(global_scale=0.5, aabb=True)))
steps.append((0, view(4), CloudSettings(global_scale=0.5, rasterize_mode=RasterizeMode.Normal)))
steps.append((0, view(5), CloudSettings(global_scale=0.5, radix_sort_depth_bits=RadixSortDepthBits.Bits16)))
steps.append((1, view(6), CloudSettings(global_scale=0.5)))
steps.append((1, view(7), CloudSettings(global_scale=0.5, sort_mode=SortMode.Rayon)))
for k, (ci, v, s) in enumerate(steps):
    print("step", k, flush=True)
    p.render(handles[ci], v, s)
    print("   ", p.stats()["sort_path"], p.stats()["draw_count"], p.stats()["regrow_count"], flush=True)
print("done")
