"""SHA-256 of four 1080p frames (dense / scene-like 1 M-splat frame at 4 and 1 samples per pixel) of the library in use —
run once under the tree's library and once under BGS_LIB_OVERRIDE=<variant> to see that an experiment variant draws the
same bits (scripts/build_patched_variant.sh).   python scripts/image_hash.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded  # noqa: E402

p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(1_000_000, 2))
for gs in (1.0, 0.05):
    for m in (4, 1):
        v = View.headless(1920, 1080, msaa_samples=m)
        s = CloudSettings(global_scale=gs)
        for _ in range(4):            # (the supertile level settles; the image does not depend on it)
            img = p.render(h, v, s)
        print(f"gs {gs} x{m}: {hashlib.sha256(img.tobytes()).hexdigest()[:20]}", flush=True)
