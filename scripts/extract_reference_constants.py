#!/usr/bin/env python3
"""BUILD-CONTAINER ONLY: parse the numeric constants of the reference's render / sort shaders and loaders out of
/root/reference into tests/golden/reference_constants.json (values + file:line provenance; no source text is copied —
numbers only). tests/test_reference_constants.py then holds the oracle (oracle/bgs_oracle.c) and the device arithmetic
(csrc/splat_math.h, csrc/render_kernels.hip) to that file on any box: the one reference-HELD pin the render half of the
oracle can get (the reference has no golden images and cannot be built or run here). Round 4's verdict, item 7."""
import json
import os
import re
import sys

REF = "/root/reference"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

# name, file, regex with one group per value (or None for the table parser), how many values
SPECS = [
    ("srgb_to_linear.threshold", "src/material/spherical_harmonics.wgsl", r"srgb_color\[i\] <= ([0-9.]+)"),
    ("srgb_to_linear.linear_divisor", "src/material/spherical_harmonics.wgsl", r"srgb_color\[i\] / ([0-9.]+)"),
    ("srgb_to_linear.offset_scale_exponent", "src/material/spherical_harmonics.wgsl",
     r"pow\(\(srgb_color\[i\] \+ ([0-9.]+)\) / ([0-9.]+), ([0-9.]+)\)"),
    ("cov2d.low_pass", "src/render/helpers.wgsl", r"cov\[0\]\[0\] \+= ([0-9.]+)f?;"),
    ("world_to_clip.w_epsilon", "src/render/transform.wgsl", r"homogenous_pos\.w \+ ([0-9.]+)\)"),
    ("in_frustum.xy_limit", "src/render/transform.wgsl", r"abs\(clip_space_pos\.x\) < ([0-9.]+)"),
    ("in_frustum.z_centre_halfwidth", "src/render/transform.wgsl", r"abs\(clip_space_pos\.z - ([0-9.]+)\) < ([0-9.]+)"),
    ("cutoff.adaptive", "src/render/gaussian.wgsl", r"sqrt\(max\(([0-9.]+) \+ ([0-9.]+) \* log\(opacity\), ([0-9.]+)\)\)"),
    ("cutoff.fixed", "src/render/gaussian.wgsl", r"let cutoff = ([0-9.]+);"),
    ("fs_main.alpha_clamp", "src/render/gaussian.wgsl", r"min\(exp\(power\) \* input\.color\.a, ([0-9.]+)\)"),
    ("fs_main.obb_sigma_inverse", "src/render/gaussian.wgsl", r"let sigma = 1\.0 / ([0-9.]+);"),
    ("fs_main.bounding_box_edge_width", "src/render/gaussian.wgsl", r"let edge_width = ([0-9.]+);"),
    ("highlight_colour", "src/render/gaussian.wgsl", r"return vec4<f32>\(([0-9.]+), ([0-9.]+), ([0-9.]+), ([0-9.]+)\);"),
    ("surfel.filter_size", "src/render/gaussian_2d.wgsl", r"let filter_size = ([0-9.]+);"),
    ("ply.max_size_variance", "src/io/ply.rs", r"MAX_SIZE_VARIANCE: f32 = ([0-9.]+);"),
]


def main():
    out = {"_comment": "numbers parsed from the reference tree by scripts/extract_reference_constants.py (build container only)",
           "constants": {}}
    # the SH constant table: array<f32, 16>( ... )
    rel = "src/material/spherical_harmonics.wgsl"
    text = open(os.path.join(REF, rel)).read()
    m = re.search(r"const shc = array<f32, 16>\((.*?)\);", text, re.S)
    vals = [float(t) for t in re.findall(r"-?[0-9]+\.[0-9]+", m.group(1))]
    assert len(vals) == 16
    line = text[:m.start()].count("\n") + 1
    out["constants"]["spherical_harmonics.shc"] = {"values": vals, "source": f"{rel}:{line}-{line + 17}"}
    for name, rel, rx in SPECS:
        text = open(os.path.join(REF, rel)).read()
        m = re.search(rx, text)
        if not m:
            sys.exit(f"{name}: pattern not found in {rel}")
        line = text[:m.start()].count("\n") + 1
        out["constants"][name] = {"values": [float(g) for g in m.groups()], "source": f"{rel}:{line}"}
    path = os.path.join(ROOT, "tests", "golden", "reference_constants.json")
    json.dump(out, open(path, "w"), indent=1)
    print(f"wrote {path}: {len(out['constants'])} constants")


if __name__ == "__main__":
    main()
