"""Generates the committed golden fixtures. Run from the repo root:  python tests/golden/make_golden.py

1. radix_keys.json — the known-answer cases of the reference's own tests/radix.rs:9-106
   (positions (-/+0.02,0,1), cameras (-/+0.01,0,0); the (bits -> places, shift, parity) table;
   the 16-bit collision). Key values are computed HERE with numpy float32 scalar arithmetic,
   independently of the C oracle, following tests/radix.rs:96-106 literally.
2. render_*.npz — small images produced by the oracle (oracle/bgs_oracle.c) for scenes of the
   reference's tests/tools, so a drift of the oracle itself is caught, and so GPU parity can be
   checked against committed data: `rgba` / `amb` at 4 samples per pixel (Msaa::Sample4, Bevy's default), `rgba_msaa1` /
   `amb_msaa1` at one (Msaa::Off); amb = the oracle's per-pixel ambiguity bound. The reference cannot run here (no cargo/wgpu/GPU); these are
   oracle outputs, not reference outputs (see oracle/bgs_oracle.h "PINNING STATUS").
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.dirname(os.path.abspath(__file__))


def radix_golden():
    f32 = np.float32
    positions = [(-0.02, 0.0, 1.0), (0.02, 0.0, 1.0)]
    cameras = [(-0.01, 0.0, 0.0), (0.01, 0.0, 0.0)]
    table = {16: (2, 16, 0), 24: (3, 8, 1), 32: (4, 0, 0)}  # tests/radix.rs:43-47

    def dist2(p, c):  # tests/radix.rs:96-101
        dx, dy, dz = f32(p[0]) - f32(c[0]), f32(p[1]) - f32(c[1]), f32(p[2]) - f32(c[2])
        return f32(f32(f32(dx * dx) + f32(dy * dy)) + f32(dz * dz))

    cases = []
    for bits, (places, shift, parity) in table.items():
        for cam in cameras:
            entries = []
            for p in positions:
                d2 = dist2(p, cam)
                bits_u = int(np.array(d2, np.float32).view(np.uint32))
                key = ((0xFFFFFFFF - bits_u) & 0xFFFFFFFF) >> shift  # tests/radix.rs:103-106
                entries.append({"position": p, "dist2_bits": bits_u, "key": key})
            cases.append({"depth_bits": bits, "camera": cam, "entries": entries})
    return {
        "source": "reference tests/radix.rs:9-106",
        "table": {str(b): {"digit_places": v[0], "key_shift": v[1], "initial_parity": v[2]} for b, v in table.items()},
        "cases": cases,
    }


def render_goldens():
    from oracle import oracle
    import helpers as H
    from bevy_gaussian_splatting_amd import (
        CloudSettings, GaussianMode, SortMode, View, random_gaussians_3d_seeded, transform_from)

    def save(name, cloud, view, settings):
        # every image for both sample counts the path is built for (View.msaa_samples: 4 = Bevy's default, what the
        # reference's cameras render with; 1 = Msaa::Off), each with the oracle's ambiguity bound, so that a consumer of
        # the fixture can hold a renderer to the standard tolerance without re-running the oracle
        entries = oracle.sort(cloud, view, settings)
        arrays = {"keys": entries["key"], "index": entries["index"]}
        for samples, tag in ((4, ""), (1, "_msaa1")):
            view.msaa_samples = samples
            img, amb = oracle.render(cloud, entries, view, settings, with_ambiguity=True)
            arrays["rgba" + tag] = img
            arrays["amb" + tag] = amb
        view.msaa_samples = 4
        np.savez_compressed(os.path.join(OUT, name), **arrays)
        print(name, arrays["rgba"].shape, float(np.abs(arrays["rgba"]).max()),
              "max |4x - 1x|", float(np.abs(arrays["rgba"] - arrays["rgba_msaa1"]).max()))

    save("render_visibility_128.npz", H.visibility_test_cloud(),
         View.perspective(transform_from((0, 0, 5)), 128, 128),
         CloudSettings(sort_mode=SortMode.NONE, global_opacity=2.0, opacity_adaptive_radius=False))
    c = random_gaussians_3d_seeded(2000, 1)
    v = View.headless(96, 64)
    save("render_random2k_obb.npz", c, v, CloudSettings())
    save("render_random2k_aabb.npz", c, v, CloudSettings(aabb=True))
    save("render_random2k_2d_aabb.npz", c, v, CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True))
    sp = H.surfel_plane_cloud()
    save("render_surfel_plane.npz", sp, View.perspective(transform_from((0, 1.5, 20)), 128, 72),
         CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, transform=transform_from((5.0, 5.0, 0.0))))


if __name__ == "__main__":
    with open(os.path.join(OUT, "radix_keys.json"), "w") as f:
        json.dump(radix_golden(), f, indent=1)
    render_goldens()
