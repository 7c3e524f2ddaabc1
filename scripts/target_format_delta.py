#!/usr/bin/env python3
"""How far is ONE conversion of the resolved f32 pixel (what libbgs's packed targets hold: bgs_set_output_srgb8 /
_rgba16f) from the reference's REAL colour attachment — Rgba8UnormSrgb or Rgba16Float, multisampled, every covered
sample read, blended and stored rounded at every draw (/root/reference/src/render/mod.rs:917-921,944-948,
examples/headless.rs:120-123)? CPU only: both images come from the oracle (oracle_render_target: target_format 0 = the
ideal binary32 samples, 1 / 2 = the packed attachment with per-blend, per-sample quantisation), on a window of the six
whole-frame parity configurations (tests/test_gpu_parity.py::test_whole_frame_parity_*), at the reference's 4 samples
per pixel. Also an in-gamut control (SH degree 0 with colours in [0, 1]): the synthetic clouds' SH ~ U(-1, 1) give
colours far outside [0, 1], which a fixed-point attachment clamps at every blend — a second effect next to the rounding.

Writes profiles/r5/target_format_delta.json (round 4's verdict, "Next round" item 4).
usage: python scripts/target_format_delta.py [--window 640x360] [--configs a,b,...]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from bevy_gaussian_splatting_amd import CloudSettings, GaussianMode, View, random_gaussians_3d_seeded  # noqa: E402
from oracle import oracle  # noqa: E402


def in_gamut(cloud):
    """the same geometry with SH degree 0 colours in [0, 1] (what a trained asset's splats mostly are)"""
    import copy
    c = copy.deepcopy(cloud)
    sh = np.zeros_like(c.spherical_harmonic)
    rng = np.random.default_rng(11)
    # rgb = 0.5 + 0.2820948 * sh0 in [0.05, 0.95]
    sh[:, 0:3] = (rng.uniform(0.05, 0.95, size=(len(c), 3)).astype(np.float32) - 0.5) / 0.2820948
    c.spherical_harmonic = sh
    return c


def stats(a, b, unit):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return {"unit": unit, "max": float(d.max()), "mean": float(d.mean()), "p99": float(np.percentile(d, 99)),
            "frac_above_1": float((d > 1.0).mean()), "frac_above_2": float((d > 2.0).mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", default="640x360")
    ap.add_argument("--configs", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r5", "target_format_delta.json"))
    args = ap.parse_args()
    ww, wh = (int(t) for t in args.window.split("x"))
    W, H = 1920, 1080
    win = ((W - ww) // 2, (H - wh) // 2, (W + ww) // 2, (H + wh) // 2)
    c1 = random_gaussians_3d_seeded(1_000_000, 2)
    configs = {
        "1m_3dgs_dense": (lambda: c1, CloudSettings()),
        "1m_3dgs_scene_like": (lambda: c1, CloudSettings(global_scale=0.05)),
        "1m_2dgs_obb": (lambda: c1, CloudSettings(gaussian_mode=GaussianMode.Gaussian2d)),
        "1m_2dgs_surfel_aabb": (lambda: c1, CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True)),
        "5m_f16_dense": (lambda: oracle.decode_f16(random_gaussians_3d_seeded(5_000_000, 3).to_f16()), CloudSettings()),
        "5m_f16_scene_like": (lambda: oracle.decode_f16(random_gaussians_3d_seeded(5_000_000, 3).to_f16()), CloudSettings(global_scale=0.05)),
        "1m_3dgs_scene_like_in_gamut": (lambda: in_gamut(c1), CloudSettings(global_scale=0.05)),
        "1m_3dgs_dense_in_gamut": (lambda: in_gamut(c1), CloudSettings()),
    }
    want = [t for t in args.configs.split(",") if t] or list(configs)
    out = {"window": list(win), "viewport": [W, H], "sample_count": 4, "cpus": oracle.lib().oracle_max_threads(), "configs": {}}
    if os.path.exists(args.out):
        out["configs"] = json.load(open(args.out)).get("configs", {})
    v = View.headless(W, H, msaa_samples=4)
    for name in want:
        make, s = configs[name]
        cloud = make()
        t0 = time.time()
        e = oracle.sort(cloud, v, s)
        ideal = oracle.render(cloud, e, v, s, window=win)
        t_ideal = time.time() - t0
        t0 = time.time()
        q8 = oracle.render(cloud, e, v, s, window=win, target_format=oracle.TARGET_SRGB8)
        t8 = time.time() - t0
        t0 = time.time()
        q16 = oracle.render(cloud, e, v, s, window=win, target_format=oracle.TARGET_RGBA16F)
        t16 = time.time() - t0
        one8, ref8 = oracle.srgb8_codes(ideal), oracle.srgb8_codes(q8)
        one16 = ideal.astype(np.float16)
        ref16 = q16.astype(np.float16)   # exact: q16's values are binary16 values
        # binary16 distance in units in the last place of the LARGER value (codes are not uniform)
        ulp = np.spacing(np.maximum(np.abs(one16), np.abs(ref16)).astype(np.float16)).astype(np.float64)
        d16 = np.abs(one16.astype(np.float64) - ref16.astype(np.float64)) / np.maximum(ulp, 2.0 ** -24)
        r = {"srgb8_LSB": stats(one8, ref8, "LSB of Rgba8UnormSrgb (per channel)"),
             "srgb8_LSB_rgb_only": stats(one8[..., :3], ref8[..., :3], "LSB, colour channels"),
             "rgba16f_ulp": {"unit": "binary16 ulp", "max": float(d16.max()), "mean": float(d16.mean()),
                             "p99": float(np.percentile(d16, 99)), "frac_above_1": float((d16 > 1.0).mean())},
             "rgba16f_abs": {"max": float(np.abs(q16 - ideal).max()), "mean": float(np.abs(q16 - ideal).mean()),
                             "frac_above_1e-3": float((np.abs(q16 - ideal) > 1e-3).mean())},
             "srgb8_abs_linear": {"max": float(np.abs(q8 - ideal).max()), "mean": float(np.abs(q8 - ideal).mean()),
                                  "frac_above_1e-3": float((np.abs(q8 - ideal) > 1e-3).mean())},
             "ideal_out_of_gamut_frac": float(((ideal[..., :3] < 0) | (ideal[..., :3] > 1)).mean()),
             "seconds": {"ideal": round(t_ideal, 1), "srgb8": round(t8, 1), "rgba16f": round(t16, 1)}}
        out["configs"][name] = r
        print(name, json.dumps(r), flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
