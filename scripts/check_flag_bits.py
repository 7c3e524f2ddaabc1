"""Does a debug-flag variant of the rasteriser draw the same frame? python scripts/check_flag_bits.py "<configs>" "<flags>"
Prints, per (config, flags), how many of the frame's f32 values differ from the product's (flags 0) and the largest
difference. Experiments only (the flags are debug switches)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import (CloudSettings, GaussianMode, GaussianSplattingPlugin, View,  # noqa: E402
                                         random_gaussians_3d_seeded)

configs = (sys.argv[1] if len(sys.argv) > 1 else "dense").split()
flags = [int(f, 0) for f in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
SPEC = {"dense": (1_000_000, 2, False, CloudSettings()), "scene": (1_000_000, 2, False, CloudSettings(global_scale=0.05)),
        "surfel": (1_000_000, 4, False, CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True)),
        "surfel_scene": (1_000_000, 4, False, CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, global_scale=0.05)),
        "2d_obb": (1_000_000, 4, False, CloudSettings(gaussian_mode=GaussianMode.Gaussian2d)),
        "aabb3d": (1_000_000, 2, False, CloudSettings(aabb=True)),
        "5m_scene": (5_000_000, 3, True, CloudSettings(global_scale=0.05))}
p = GaussianSplattingPlugin(0)
v = View.headless(1920, 1080)
for c in configs:
    n, seed, f16, s = SPEC[c]
    cl = random_gaussians_3d_seeded(n, seed)
    h = p.upload(cl.to_f16() if f16 else cl)
    p.set_debug_flags(0)
    for _ in range(4):
        ref = np.array(p.render(h, v, s), copy=True)
    for fl in flags:
        p.set_debug_flags(fl)
        for _ in range(3):
            img = np.array(p.render(h, v, s), copy=True)
        d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
        print(f"{c:13s} flags {fl:#10x}: {int((img.view(np.uint32) != ref.view(np.uint32)).sum())} of {img.size} values differ, max |diff| {d.max():.3e}, strip_tiles {p.stats().get('strip_tiles')}", flush=True)
    p.set_debug_flags(0)
    h.free()
