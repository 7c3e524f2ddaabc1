"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile target
libbgs_oracle_asan.so): the checker itself must not read out of bounds or rely on undefined behaviour on
the inputs the parity tests feed it, including the edge cases (empty cloud, one splat, windows on the
target's border, NaN / inf positions, every sort mode and raster variant, f16 codec).

Runs in a child process (the sanitizer runtime has to be the first DSO: LD_PRELOAD) with
BGS_ORACLE_LIB pointing the ctypes binding at the instrumented library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ORACLE_DIR = os.path.join(ROOT, "oracle")

CHILD = r'''
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from bevy_gaussian_splatting_amd import (CloudSettings, GaussianMode, RadixSortDepthBits, RasterizeMode, SortMode,
                                         View, DrawMode, random_gaussians_3d_seeded)
from oracle import oracle
assert oracle.LIB_PATH.endswith("libbgs_oracle_asan.so"), oracle.LIB_PATH
v = View.headless(97, 61)
for n in (0, 1, 2, 63, 1500):
    c = random_gaussians_3d_seeded(n, 30 + n)
    if n >= 63:
        c.position_visibility[:3, :3] = [[np.nan, 0, 0], [np.inf, 1, -2], [0, 1.5, 5]]
        c.scale_opacity[5] = [0, 0, 0, 0.5]
        c.rotation[6] = 0
    for kw in ({}, {"aabb": True}, {"gaussian_mode": GaussianMode.Gaussian2d}, {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True},
               {"sort_mode": SortMode.Rayon}, {"sort_mode": SortMode.NONE}, {"radix_sort_depth_bits": RadixSortDepthBits.Bits16},
               {"rasterize_mode": RasterizeMode.Depth}, {"rasterize_mode": RasterizeMode.Normal},
               {"rasterize_mode": RasterizeMode.Classification, "num_classes": 3}, {"draw_mode": DrawMode.HighlightSelected},
               {"sh_degree": 0}, {"global_scale": 3.0}):
        s = CloudSettings(**kw)
        e = oracle.sort(c, v, s)
        assert len(e) == n
        img, amb = oracle.render(c, e, v, s, with_ambiguity=True)
        assert img.shape == (61, 97, 4)
        for win in ((0, 0, 1, 1), (96, 60, 97, 61), (40, 0, 97, 7)):
            w = oracle.render(c, e, v, s, window=win)
            assert np.array_equal(w, img[win[1]:win[3], win[0]:win[2]], equal_nan=True)
        oracle.instance_stats(c, e, v, s)
    if n:
        f16 = oracle.encode_f16(c)
        oracle.decode_f16(f16)
oracle.encode_srgb8(np.random.default_rng(0).uniform(-1, 2, (8, 8, 4)).astype(np.float32))
print("SANITIZED-OK")
'''


def test_oracle_is_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", ORACLE_DIR, "libbgs_oracle_asan.so"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan runtime next to this gcc")
    env = dict(os.environ)
    env["LD_PRELOAD"] = libasan + (":" + libubsan if os.path.isabs(libubsan) and os.path.exists(libubsan) else "")
    # python itself "leaks" by design at exit; the oracle's own allocations are all freed per call and a
    # leak there would show up as growth, not as a report worth failing on
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["BGS_ORACLE_LIB"] = os.path.join(ORACLE_DIR, "libbgs_oracle_asan.so")
    env["OMP_NUM_THREADS"] = "4"
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, env=env, timeout=900)
    out = r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, out[-4000:]
