"""GPU parity tests proper: the HIP path (through the C ABI in libbgs.so) against the oracle on
the same seeded inputs, against the committed goldens, and — at BASELINE.json's full sizes —
through size-independent properties plus oracle-checked crops.

Bars (north_star): sort entries BIT-EXACT; RGBA within 1e-3 per channel (abs, plus 1e-4 relative
for the unclamped HDR values, plus the oracle's ambiguity bound on the rare pixels whose coverage
decision sits within rounding distance of a quad edge)."""
import os

import numpy as np
import pytest

import helpers as H
from bevy_gaussian_splatting_amd import (
    CloudSettings, GaussianColorSpace, GaussianMode, PlanarGaussian3d, RadixSortDepthBits, SortMode,
    View, random_gaussians_3d_seeded, transform_from, rotation_y)
from bevy_gaussian_splatting_amd import RasterizeMode, compute_aabb
from bevy_gaussian_splatting_amd.multiview import headless_view

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


SLACK_USED = {"values": 0, "checked": 0}  # how often the oracle's ambiguity slack was needed (printed at the end)


class _scene_depth:
    """`with _scene_depth(plugin, view):` puts the view's host depth buffer (helpers.random_case: view.depth_host) on the
    device for the frames inside (bgs_view.depth_device_ptr) and releases it afterwards."""

    def __init__(self, plugin, view):
        self.plugin, self.view, self.ptr = plugin, view, 0

    def __enter__(self):
        d = getattr(self.view, "depth_host", None)
        if d is not None:
            self.ptr = self.plugin.upload_depth(d)
            self.view.depth_device_ptr = self.ptr
        return self.view

    def __exit__(self, *exc):
        if self.ptr:
            self.view.depth_device_ptr = 0
            self.plugin.device_free(self.ptr)


def _assert_image(ref, got, amb, frac_slack=0.002, what="", overlay=False):
    assert got.shape == ref.shape and np.isfinite(got).all()
    ok, err = H.tolerance_mask(ref, got, amb)
    assert ok.all(), f"{what}: {(~ok).sum()} values out of tolerance, max err {err.max():.3e}"
    rec = H.account(ref, got, amb, what, overlay=overlay)
    SLACK_USED["values"] += rec["beyond_strict"]
    SLACK_USED["checked"] += rec["values"]
    if rec["beyond_strict"]:
        print(f"[tolerance: {what}] {rec['beyond_strict']} of {rec['values']} values beyond 1e-3 + 1e-4 |ref| "
              f"(accepted through the oracle's ambiguity bound), largest excess {rec['max_excess']:.2e}")
    assert rec["beyond_strict"] <= frac_slack * rec["values"], f"{what}: ambiguity slack used by too many pixels"


# ---------------------------------------------------------------------------------------------
# sort: bit-exact
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 2047, 2048, 2049, 4097, 100_000])
def test_sort_bit_exact_ragged_sizes(plugin, oracle, n):
    c = random_gaussians_3d_seeded(n, 100 + n)
    v = View.headless(640, 360)
    s = CloudSettings()
    h = plugin.upload(c)
    got = plugin.sort(h, v, s)
    ref = oracle.sort(c, v, s)
    assert np.array_equal(got["key"], ref["key"])
    assert np.array_equal(got["index"], ref["index"])
    h.free()


@pytest.mark.parametrize("bits", [16, 24, 32])
@pytest.mark.parametrize("mode", [SortMode.Radix, SortMode.Rayon, SortMode.Std, SortMode.NONE])
def test_sort_bit_exact_modes_and_depth_bits(plugin, oracle, mode, bits):
    c = random_gaussians_3d_seeded(50_000, 7)
    c.position_visibility[:4, :3] = [[0, 1.5, 5], [np.nan, 0, 0], [np.inf, 0, 0], [0, 1.5, 4.9]]
    v = View.headless(640, 360, yaw=0.3)
    s = CloudSettings(sort_mode=mode, radix_sort_depth_bits=RadixSortDepthBits(bits),
                      transform=transform_from((0.5, -0.25, 1.0), rotation_y(0.2)))
    h = plugin.upload(c)
    got = plugin.sort(h, v, s)
    ref = oracle.sort(c, v, s)
    assert np.array_equal(got["key"], ref["key"])
    assert np.array_equal(got["index"], ref["index"])
    h.free()


@pytest.mark.parametrize("yaw", [0.0, 0.37])
def test_frustum_verdict_without_divisions_is_the_verdict_with_them(plugin, oracle, yaw):
    """keygen decides in_frustum(world_to_clip(p)) from h * rcp(h.w) wherever that is clear of the thresholds by 2^-20
    and takes the reference's three IEEE divisions otherwise (splat_math.h in_frustum_of_world). ~1.1 M points within
    a few ulp of every threshold and of every guard band, on both sides: keys and order bit-exact with the oracle,
    which always divides."""
    v = View.headless(1920, 1080, yaw=yaw)
    c = H.frustum_boundary_cloud(v, 60_000, 17 + int(100 * yaw))
    s = CloudSettings()
    h = plugin.upload(c)
    got = plugin.sort(h, v, s)
    ref = oracle.sort(c, v, s)
    assert np.array_equal(got["key"], ref["key"]) and np.array_equal(got["index"], ref["index"])
    drawn = int((ref["key"] != 0xFFFFFFFF).sum())
    assert 0.2 * len(c) < drawn < 0.8 * len(c), (drawn, len(c))      # the points really straddle the thresholds
    # the render path's keygen (bucket placement, no culled tail) takes the same verdicts
    plugin.render(h, v, s, download=False)
    plugin.render(h, v, s, download=False)
    assert plugin.stats()["draw_count"] == drawn
    h.free()


def test_sort_collisions_ties_by_index(plugin, oracle):
    """Duplicated positions (exact key collisions) and the 16-bit collapse of tests/radix.rs:82-94:
    equal keys must stay in ascending splat index (stable LSD)."""
    base = random_gaussians_3d_seeded(3000, 5)
    pv = np.tile(base.position_visibility, (8, 1))
    c = PlanarGaussian3d(pv, np.tile(base.spherical_harmonic, (8, 1)), np.tile(base.rotation, (8, 1)),
                         np.tile(base.scale_opacity, (8, 1)))
    v = View.headless(320, 180)
    for bits in (16, 32):
        s = CloudSettings(radix_sort_depth_bits=RadixSortDepthBits(bits))
        h = plugin.upload(c)
        got = plugin.sort(h, v, s)
        ref = oracle.sort(c, v, s)
        assert np.array_equal(got["key"], ref["key"]) and np.array_equal(got["index"], ref["index"])
        same = np.diff(got["key"].astype(np.int64)) == 0
        assert same.sum() > 1000 and np.all(np.diff(got["index"].astype(np.int64))[same] > 0)
        h.free()


@pytest.mark.parametrize("n,passes", [(0, 4), (1, 1), (255, 2), (2048, 4), (2049, 3), (300_001, 4),
                                      (5_000_001, 2), (5_000_001, 4), (67_108_867, 4)])
def test_onesweep_kernel_on_arbitrary_keys(plugin, n, passes):
    """The radix kernel itself (both tile sizes: > 4M pairs uses 4096-pair tiles; 2^26 + 3 pairs =
    16 385 tiles per pass) on adversarial keys: random with many ties, all-equal, already sorted, reversed."""
    rng = np.random.default_rng(n + passes)
    mask = np.uint32((1 << (8 * passes)) - 1) if passes < 4 else np.uint32(0xFFFFFFFF)
    variants = [rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32) & mask]
    if n <= 300_001:
        variants += [
            (rng.integers(0, 7, size=n, dtype=np.uint64).astype(np.uint32) * np.uint32(0x01010101)) & mask,
            np.full(n, 0x00ABCDEF, np.uint32) & mask,
            (np.arange(n, dtype=np.uint32) * np.uint32(3)) & mask,
            (np.arange(n, dtype=np.uint32)[::-1].copy()) & mask,
        ]
    for keys in variants:
        out = plugin.radix_sort_pairs(keys, passes)
        order = np.argsort(keys, kind="stable").astype(np.uint32)
        assert np.array_equal(out["index"], order)
        assert np.array_equal(out["key"], keys[order])


# ---------------------------------------------------------------------------------------------
# render: <= 1e-3 per channel
# ---------------------------------------------------------------------------------------------
VARIANTS = {
    "obb3d": {},
    "aabb3d": {"aabb": True},
    "obb3d_fixed_radius_linear": {"opacity_adaptive_radius": False,
                                  "color_space": GaussianColorSpace.LinRec709Display, "global_scale": 0.3},
    "obb3d_sh0": {"sh_degree": 0},
    "obb3d_sh2_opacity2": {"sh_degree": 2, "global_opacity": 2.0},
    "obb3d_24bit": {"radix_sort_depth_bits": RadixSortDepthBits.Bits24},
    "obb3d_rayon": {"sort_mode": SortMode.Rayon},
    "obb3d_unsorted": {"sort_mode": SortMode.NONE},
    "obb2d": {"gaussian_mode": GaussianMode.Gaussian2d},
    "aabb2d": {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True},
}


@pytest.fixture(params=["scan", "sort"])
def binning(request, plugin):
    """Both tile-binning modes of the library (bgs_set_binning); restored to the default after."""
    plugin.set_binning(request.param)
    yield request.param
    plugin.set_binning("scan")


@pytest.mark.parametrize("name", sorted(VARIANTS))
@pytest.mark.parametrize("size", [(160, 96), (250, 130)])
def test_render_parity_small(plugin, oracle, binning, name, size):
    c = random_gaussians_3d_seeded(6000, 11)
    v = View.headless(*size)
    s = CloudSettings(**VARIANTS[name])
    h = plugin.upload(c)
    got = plugin.render(h, v, s)
    e = oracle.sort(c, v, s)
    ref, amb = oracle.render(c, e, v, s, with_ambiguity=True)
    _assert_image(ref, got, amb, what=name)
    st = plugin.stats()
    vis, inst = oracle.instance_stats(c, e, v, s)
    assert st["visible_count"] == vis and st["binning"] == binning
    if binning == "sort":
        assert st["instance_count"] >= inst * 0.5 and st["instance_count"] <= inst * 2 + 64
    h.free()


def test_binning_modes_give_bit_identical_images(plugin):
    """Same records, same per-tile front-to-back order => the two binning strategies must agree
    bitwise, including on a target whose size is not a multiple of the tile / supertile."""
    c = random_gaussians_3d_seeded(40_000, 17)
    h = plugin.upload(c)
    for size, kw in (((1000, 600), {}), ((333, 777), {"aabb": True}), ((640, 360), {"global_scale": 0.1})):
        v = View.headless(*size)
        s = CloudSettings(**kw)
        plugin.set_binning("sort")
        a = plugin.render(h, v, s)
        plugin.set_binning("scan")
        b = plugin.render(h, v, s)
        assert np.array_equal(a, b)
    h.free()


def test_surfel_tile_cull_is_invisible_and_the_same_in_both_rasterisers(plugin):
    """2DGS surfel records whose ellipse cannot reach a tile (<= 2^-23 of their opacity on every pixel) are
    dropped at staging time (surfel_negligible_in_tile). Against the same frame with the cull switched off
    (debug flag 0x40: keep every record the tile rectangle admits) the image moves by far less than the
    tolerance, and both rasterisers take the same decisions."""
    c = random_gaussians_3d_seeded(60_000, 23)
    h = plugin.upload(c)
    v = View.headless(800, 448)
    try:
        for gs in (1.0, 0.2):
            s = CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, global_scale=gs)
            plugin.set_binning("scan")
            a = plugin.render(h, v, s)
            plugin.set_debug_flags(0x40)
            full = plugin.render(h, v, s)
            plugin.set_debug_flags(0)
            plugin.set_binning("sort")
            b = plugin.render(h, v, s)
            assert np.array_equal(a, b)
            assert np.isfinite(full).all() and float(np.abs(a - full).max()) <= 1e-4
    finally:
        plugin.set_debug_flags(0)
        plugin.set_binning("scan")
    h.free()


def test_render_10k_config0(plugin, oracle, binning):
    """BASELINE.json configs[0]: 10k random splats, 256x256, single camera."""
    c = random_gaussians_3d_seeded(10_000, 1)
    v = View.headless(256, 256)
    s = CloudSettings()
    h = plugin.upload(c)
    got = plugin.render(h, v, s)
    e = oracle.sort(c, v, s)
    ref, amb = oracle.render(c, e, v, s, with_ambiguity=True)
    _assert_image(ref, got, amb, what="cfg0")
    got_sort = plugin.sort(h, v, s)
    assert np.array_equal(got_sort["key"], e["key"]) and np.array_equal(got_sort["index"], e["index"])
    h.free()


def test_bright_splats_behind_a_nearly_opaque_stack(plugin, oracle, binning):
    """The transmittance cut-off scales with the frame's largest colour (frame_t_eps): SH colours are not
    clamped, so what a fixed cut-off drops behind a nearly opaque stack is only bounded for colours up to 1.
    Here one splat in a hundred is orders of magnitude brighter than the rest and global_opacity pushes most alphas to
    the 0.999 clamp (the configuration the 200-seed sweep failed on with a fixed 2^-13)."""
    c = random_gaussians_3d_seeded(30_000, 91)
    c.spherical_harmonic[::97] *= 60.0
    v = View.headless(320, 200)
    s = CloudSettings(global_opacity=1.7, global_scale=0.5, opacity_adaptive_radius=False)
    h = plugin.upload(c)
    got = plugin.render(h, v, s)
    e = oracle.sort(c, v, s)
    ref, amb = oracle.render(c, e, v, s, with_ambiguity=True)
    assert float(np.abs(ref[..., :3]).max()) > 30.0          # the frame really has an HDR range
    _assert_image(ref, got, amb, what="bright splats")
    h.free()


def test_render_f16_cloud(plugin, oracle, binning):
    c = random_gaussians_3d_seeded(8000, 3)
    c16 = c.to_f16()
    v = View.headless(192, 108)
    for kw in ({}, {"aabb": True}):
        s = CloudSettings(**kw)
        h = plugin.upload(c16)
        got = plugin.render(h, v, s)
        c_dec = oracle.decode_f16(c16)
        e = oracle.sort(c_dec, v, s)
        ref, amb = oracle.render(c_dec, e, v, s, with_ambiguity=True)
        _assert_image(ref, got, amb, what="f16")
        gs = plugin.sort(h, v, s)
        assert np.array_equal(gs["index"], e["index"])
        h.free()


def test_reference_tool_scenes(plugin, oracle):
    cases = [
        ("visibility", H.visibility_test_cloud(), View.perspective(transform_from((0, 0, 5)), 128, 128),
         CloudSettings(sort_mode=SortMode.NONE, global_opacity=2.0, opacity_adaptive_radius=False)),
        ("surfel_plane", H.surfel_plane_cloud(), View.perspective(transform_from((0, 1.5, 20)), 128, 72),
         CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, transform=transform_from((5.0, 5.0, 0.0)))),
        ("pair_aabb", H.aabb_obb_pair_cloud(), View.headless(192, 108), CloudSettings(aabb=True)),
        ("pair_obb", H.aabb_obb_pair_cloud(), View.headless(192, 108), CloudSettings()),
    ]
    for name, c, v, s in cases:
        h = plugin.upload(c)
        got = plugin.render(h, v, s)
        ref, amb = oracle.render(c, oracle.sort(c, v, s), v, s, with_ambiguity=True)
        _assert_image(ref, got, amb, what=name)
        h.free()
    # the reference's own visibility thresholds (tests/visibility_render.rs:245-252) on the GPU image
    name, c, v, s = cases[0]
    h = plugin.upload(c)
    lin = np.clip(plugin.render(h, v, s)[..., :3], 0, 1)
    u8 = np.round(np.where(lin <= 0.0031308, lin * 12.92, 1.055 * lin ** (1 / 2.4) - 0.055) * 255)
    assert (u8.max(-1) > 8).sum() >= 64 and u8.max() > 32
    h.free()


@pytest.mark.parametrize("name,kw,cloud,view", [
    ("render_random2k_obb.npz", {}, "r2k", (96, 64)),
    ("render_random2k_aabb.npz", {"aabb": True}, "r2k", (96, 64)),
    ("render_random2k_2d_aabb.npz", {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True}, "r2k", (96, 64)),
])
def test_render_against_committed_goldens(plugin, name, kw, cloud, view):
    g = np.load(os.path.join(GOLDEN, name))
    c = random_gaussians_3d_seeded(2000, 1)
    v = View.headless(*view)
    s = CloudSettings(**kw)
    h = plugin.upload(c)
    es = plugin.sort(h, v, s)
    assert np.array_equal(es["key"], g["keys"]) and np.array_equal(es["index"], g["index"])
    # the goldens carry the oracle's ambiguity map: the standard tolerance, at both sample counts
    for samples, tag in ((4, ""), (1, "_msaa1")):
        v.msaa_samples = samples
        got = plugin.render(h, v, s)
        ok, err = H.tolerance_mask(g["rgba" + tag], got, g["amb" + tag])
        strict, _ = H.tolerance_mask(g["rgba" + tag], got, None)
        print(f"[golden {name} x{samples}] max |err| {err.max():.2e}, values on the ambiguity slack {int((~strict).sum())}/{ok.size}")
        assert ok.all(), f"x{samples}: max err {err.max():.3e}"
        assert (~strict).sum() <= 0.002 * ok.size
    h.free()


def test_edge_cases(plugin, oracle, binning):
    v = View.headless(100, 60)
    s = CloudSettings()
    clear = np.array(v.clear_color, np.float32)
    # empty cloud
    c0 = random_gaussians_3d_seeded(0, 1)
    h = plugin.upload(c0)
    img = plugin.render(h, v, s)
    assert img.shape == (60, 100, 4) and np.all(img == clear)
    assert plugin.sort(h, v, s).shape == (0,)
    h.free()
    # everything culled (cloud behind the camera)
    c = random_gaussians_3d_seeded(5000, 2)
    c.position_visibility[:, 2] = np.abs(c.position_visibility[:, 2]) + 10.0
    h = plugin.upload(c)
    img = plugin.render(h, v, s)
    assert np.all(img == clear)
    assert plugin.stats()["visible_count"] == 0 and plugin.stats()["instance_count"] == 0
    h.free()
    # one splat, non-black transparent clear colour, non-multiple-of-16 target
    v2 = View.perspective(transform_from((0, 0, 0)), 37, 21, clear_color=(0.25, 0.5, 0.75, 0.0))
    c1 = PlanarGaussian3d(np.array([[0.2, 0.1, -3.0, 1.0]], np.float32), np.full((1, 48), 0.3, np.float32),
                          np.array([[0.9, 0.1, 0.2, 0.3]], np.float32), np.array([[0.3, 0.1, 0.2, 0.7]], np.float32))
    h = plugin.upload(c1)
    got = plugin.render(h, v2, s)
    ref, amb = oracle.render(c1, oracle.sort(c1, v2, s), v2, s, with_ambiguity=True)
    _assert_image(ref, got, amb, frac_slack=0.02, what="single")
    h.free()


def test_invalid_arguments_are_reported_not_fatal(plugin):
    from bevy_gaussian_splatting_amd import _native
    c = random_gaussians_3d_seeded(10, 1)
    h = plugin.upload(c)
    bad = CloudSettings(sh_degree=7)
    with pytest.raises(_native.BgsError) as ei:
        plugin.render(h, View.headless(64, 64), bad)
    assert ei.value.status == _native.BGS_EINVAL and "sh_degree" in str(ei.value)
    with pytest.raises(_native.BgsError):
        plugin.render(h, View.headless(8192, 64), CloudSettings())
    # the context is still usable afterwards
    assert plugin.render(h, View.headless(64, 64), CloudSettings()).shape == (64, 64, 4)
    h.free()


def test_render_is_deterministic_and_views_are_independent(plugin, oracle):
    c = random_gaussians_3d_seeded(20_000, 9)
    s = CloudSettings(global_scale=0.5)
    h = plugin.upload(c)
    v0, v1 = headless_view(0, 192, 108), headless_view(3, 192, 108)
    a0 = plugin.render(h, v0, s)
    a1 = plugin.render(h, v1, s)
    b0 = plugin.render(h, v0, s)
    assert np.array_equal(a0, b0)  # bitwise: no atomics-order dependence anywhere in the frame
    assert not np.array_equal(a0, a1)
    ref, amb = oracle.render(c, oracle.sort(c, v1, s), v1, s, with_ambiguity=True)
    _assert_image(ref, a1, amb, what="yawed view")
    h.free()


# ---------------------------------------------------------------------------------------------
# BASELINE.json full sizes
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cloud_1m():
    return random_gaussians_3d_seeded(1_000_000, 2)


def test_full_size_sort_1m(plugin, oracle, cloud_1m):
    """configs[1] sort leg: bit-exact against the oracle AND the size-independent properties."""
    v = View.headless(1920, 1080)
    s = CloudSettings()
    h = plugin.upload(cloud_1m)
    got = plugin.sort(h, v, s)
    k = got["key"].astype(np.int64)
    assert np.all(np.diff(k) >= 0)                                   # sortedness
    same = np.diff(k) == 0
    assert np.all(np.diff(got["index"].astype(np.int64))[same] > 0)  # stability
    assert np.array_equal(np.sort(got["index"]), np.arange(len(cloud_1m), dtype=np.uint32))  # permutation
    ref = oracle.sort(cloud_1m, v, s)
    assert np.array_equal(got["key"], ref["key"]) and np.array_equal(got["index"], ref["index"])
    assert plugin.stats()["visible_count"] == int((ref["key"] != 0xFFFFFFFF).sum())
    h.free()


@pytest.mark.parametrize("global_scale", [1.0, 0.05])
def test_full_size_render_1m_1080p(plugin, oracle, binning, cloud_1m, global_scale):
    """configs[1]: 1M splats, 1920x1080, sh3. The oracle renders three 48x48 crops of the SAME
    frame (all 1M splats, every overlapping quad) for the parity check; the whole frame is checked
    through properties (finite, alpha == 1 over an opaque clear colour, bitwise repeatable)."""
    v = View.headless(1920, 1080)
    s = CloudSettings(global_scale=global_scale)
    h = plugin.upload(cloud_1m)
    got = plugin.render(h, v, s)
    st = plugin.stats()
    assert got.shape == (1080, 1920, 4) and np.isfinite(got).all()
    assert np.allclose(got[..., 3], 1.0, atol=1e-5)
    again = plugin.render(h, v, s)
    assert np.array_equal(got, again)
    e = oracle.sort(cloud_1m, v, s)
    vis, inst = oracle.instance_stats(cloud_1m, e, v, s)
    assert st["visible_count"] == vis
    if binning == "sort":
        assert 0.5 * inst <= st["instance_count"] <= 2 * inst + 1024
    for (x0, y0) in ((936, 516), (40, 30), (1800, 1000)):
        win = (x0, y0, x0 + 48, y0 + 48)
        ref, amb = oracle.render(cloud_1m, e, v, s, window=win, with_ambiguity=True)
        _assert_image(ref, got[y0:y0 + 48, x0:x0 + 48], amb, frac_slack=0.01, what=f"crop {win} gs={global_scale}")
    h.free()


def test_full_size_f16_5m_sort_and_crop(plugin, oracle):
    """configs[2]: 5M-splat f16 cloud at 1080p (scene-like scale to keep the oracle crop cheap)."""
    c = random_gaussians_3d_seeded(5_000_000, 3).to_f16()
    v = View.headless(1920, 1080)
    s = CloudSettings(global_scale=0.05)
    h = plugin.upload(c)
    got = plugin.render(h, v, s)
    assert np.isfinite(got).all()
    es = plugin.sort(h, v, s)
    dec = oracle.decode_f16(c)
    e = oracle.sort(dec, v, s)
    assert np.array_equal(es["key"], e["key"]) and np.array_equal(es["index"], e["index"])
    win = (936, 516, 984, 564)
    ref, amb = oracle.render(dec, e, v, s, window=win, with_ambiguity=True)
    _assert_image(ref, got[516:564, 936:984], amb, frac_slack=0.01, what="5M f16 crop")

    # Liveness with many frames in flight: 6 lanes x (1221-tile keygen, 512-block project + bin looping over
    # ~2900 tiles) oversubscribe the chip and thousands of threads wait in look-back at once. Without a
    # spin back-off the polling loads starved the blocks everybody waited for and the device watchdog
    # tripped (seen at depth 3); the pipelined frames must complete and equal the blocking one.
    plugin.set_async(True)
    plugin.set_pipeline_streams(0)  # one stream per lane: all `depth` frames compete for the chip
    for depth in (3, 6):
        plugin.set_pipeline_depth(depth)
        for _ in range(3 * depth):
            plugin.render(h, v, s, download=False)
        plugin.synchronize()
        from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor
        assert np.array_equal(framebuffer_as_tensor(plugin, 1080, 1920).cpu().numpy(), got)
    plugin.set_pipeline_streams(3)  # fewer streams than lanes: 6 lanes on 3 streams
    for _ in range(12):
        plugin.render(h, v, s, download=False)
    plugin.synchronize()
    assert np.array_equal(framebuffer_as_tensor(plugin, 1080, 1920).cpu().numpy(), got)
    plugin.set_async(False)
    plugin.set_pipeline_depth(1)
    h.free()


def test_cloud_past_2pow24_splats_f16(plugin, oracle):
    """Beyond BASELINE's largest config: 16.8 M + 4133 splats (f16 planes, 2.1 GB) — splat indices no
    longer fit an f32 mantissa, keygen runs 4097 tiles, every chained scan carries > 2^24 counts.
    Sort bit-exact, crop within tolerance, pipelined frames identical."""
    n = (1 << 24) + 4133
    rng = np.random.default_rng(77)
    pv = np.empty((n, 4), np.float32)
    pv[:, :3] = rng.uniform(-20.0, 20.0, size=(n, 3)).astype(np.float32)
    pv[:, 3] = 1.0
    small = random_gaussians_3d_seeded(1 << 16, 5).to_f16()
    reps = (n + (1 << 16) - 1) >> 16
    sh = np.tile(small.spherical_harmonic, (reps, 1))[:n]
    rso = np.tile(small.rotation_scale_opacity, (reps, 1))[:n]
    from bevy_gaussian_splatting_amd.gaussian import PlanarGaussian3dF16
    c = PlanarGaussian3dF16(pv, sh, rso)
    v = View.headless(1920, 1080)
    s = CloudSettings(global_scale=0.03)
    h = plugin.upload(c)
    got = plugin.render(h, v, s)
    assert np.isfinite(got).all()
    es = plugin.sort(h, v, s)
    dec = oracle.decode_f16(c)
    e = oracle.sort(dec, v, s)
    assert np.array_equal(es["key"], e["key"]) and np.array_equal(es["index"], e["index"])
    assert int(es["index"].max()) == n - 1
    win = (944, 524, 976, 556)
    ref, amb = oracle.render(dec, e, v, s, window=win, with_ambiguity=True)
    _assert_image(ref, got[524:556, 944:976], amb, frac_slack=0.01, what="16.8M f16 crop")
    plugin.set_async(True)
    plugin.set_pipeline_depth(3)
    for _ in range(6):
        plugin.render(h, v, s, download=False)
    plugin.synchronize()
    from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor
    assert np.array_equal(framebuffer_as_tensor(plugin, 1080, 1920).cpu().numpy(), got)
    plugin.set_async(False)
    plugin.set_pipeline_depth(1)
    h.free()


def test_full_size_2dgs_1m_crop(plugin, oracle, cloud_1m):
    """configs[3]: 1M-splat 2DGS surfel cloud, both the true surfel path (aabb) and default OBB."""
    v = View.headless(1920, 1080)
    for kw in ({"aabb": True}, {}):
        s = CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, global_scale=0.25, **kw)
        h = plugin.upload(cloud_1m)
        got = plugin.render(h, v, s)
        e = oracle.sort(cloud_1m, v, s)
        win = (936, 516, 984, 564)
        ref, amb = oracle.render(cloud_1m, e, v, s, window=win, with_ambiguity=True)
        _assert_image(ref, got[516:564, 936:984], amb, frac_slack=0.01, what=f"2dgs {kw}")
        h.free()


def _crop_parity(plugin, oracle, dec, h, v, s, what, crops=((936, 516), (40, 30), (1800, 1000))):
    """Oracle-checked 48x48 crops of one full frame; prints how many values needed the ambiguity slack."""
    got = plugin.render(h, v, s)
    assert got.shape == (v.height, v.width, 4) and np.isfinite(got).all()
    e = oracle.sort(dec, v, s)
    used = total = 0
    for (x0, y0) in crops:
        win = (x0, y0, x0 + 48, y0 + 48)
        ref, amb = oracle.render(dec, e, v, s, window=win, with_ambiguity=True)
        crop = got[y0:y0 + 48, x0:x0 + 48]
        _assert_image(ref, crop, amb, frac_slack=0.01, what=f"{what} crop {win}")
        strict, err = H.tolerance_mask(ref, crop, None)
        used += int((~strict).sum())
        total += strict.size
        print(f"[{what}] crop {win}: max |err| {err.max():.2e}, values on ambiguity slack {int((~strict).sum())}/{strict.size}")
    print(f"[{what}] ambiguity slack used by {used} of {total} values")
    return got


def test_full_size_f16_5m_default_scale_crops(plugin, oracle):
    """configs[2] AT THE BENCHMARKED SETTING (global_scale = 1.0, the dense frame: ~600 k visible splats,
    ~9 M coarse list entries): three oracle-checked crops of the same frame."""
    c = random_gaussians_3d_seeded(5_000_000, 3).to_f16()
    v = View.headless(1920, 1080)
    s = CloudSettings()
    h = plugin.upload(c)
    dec = oracle.decode_f16(c)
    got = _crop_parity(plugin, oracle, dec, h, v, s, "5M f16 gs=1.0")
    assert np.allclose(got[..., 3], 1.0, atol=1e-5)
    # supertile lists are sized from the longest list seen, not for the worst case (5 M entries x 240 lists =
    # 9.6 GB per lane in round 1): on a FRESH context (the shared one keeps the buffers earlier tests grew) eight
    # lanes' worth must stay under 8 GB, dense and scene-like
    from bevy_gaussian_splatting_amd import GaussianSplattingPlugin
    with GaussianSplattingPlugin(0) as fresh:
        h2 = fresh.upload(c)
        fresh.set_async(True)
        fresh.set_pipeline_depth(8)
        for gs in (1.0, 0.05):
            s2 = CloudSettings(global_scale=gs)
            for _ in range(24):
                fresh.render(h2, v, s2, download=False)
            fresh.synchronize()
            st, level = fresh.stats(), fresh.adaptive_counters()["supertile_level"]
            lane_bytes = st["list_entries_allocated"] * 8     # list entries allocated for one lane
            print(f"[5M f16 gs={gs}] supertile level {level}: {lane_bytes / 2**20:.0f} MiB of lists per lane "
                  f"({st['instance_count']} entries in use, capacity {st['list_capacity']} per list)")
            assert 8 * lane_bytes < 8 * 2**30
        h2.free()
    es = plugin.sort(h, v, s)
    e = oracle.sort(dec, v, s)
    assert np.array_equal(es["key"], e["key"]) and np.array_equal(es["index"], e["index"])
    h.free()


@pytest.mark.parametrize("aabb", [True, False])
def test_full_size_2dgs_1m_default_scale_crops(plugin, oracle, cloud_1m, aabb):
    """configs[3] AT THE BENCHMARKED SETTING (global_scale = 1.0): the true surfel path (aabb) and the
    default OBB quad, three oracle-checked crops each."""
    v = View.headless(1920, 1080)
    s = CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=aabb)
    h = plugin.upload(cloud_1m)
    _crop_parity(plugin, oracle, cloud_1m, h, v, s, f"1M 2dgs aabb={aabb} gs=1.0")
    h.free()


# ---------------------------------------------------------------------------------------------
# depth sort paths: bucket sort (one launch) vs onesweep digit passes; re-run on overflow
# ---------------------------------------------------------------------------------------------
def _sort_equal(got, ref):
    return np.array_equal(got["key"], ref["key"]) and np.array_equal(got["index"], ref["index"])


@pytest.mark.parametrize("mode", [SortMode.Radix, SortMode.Rayon, SortMode.Std])
@pytest.mark.parametrize("n", [1, 65, 2049, 50_000, 300_000, 1_000_000])
def test_bucket_sort_is_bit_exact(plugin, oracle, mode, n):
    """The first frame of a context has no splitter table and runs the onesweep passes; the frames behind
    it use the bucket sort (stats say which). Both must give the reference's stable order, bit for bit —
    also with a guessed table (debug flag 0x200000: equal steps over the 32-bit range, i.e. badly balanced
    buckets; when a bucket overflows the frame is re-run with the passes)."""
    c = random_gaussians_3d_seeded(n, 900 + n % 97)
    v = View.headless(960, 540, yaw=0.1)
    s = CloudSettings(sort_mode=mode)
    ref = oracle.sort(c, v, s)
    h = plugin.upload(c)
    plugin.reset_adaptive_state()
    first = plugin.sort(h, v, s)
    assert plugin.stats()["sort_path"] == "onesweep" and _sort_equal(first, ref)
    second = plugin.sort(h, v, s)
    st = plugin.stats()
    assert _sort_equal(second, ref)
    if 256 <= st["draw_count"] <= 500_000:
        assert st["sort_path"] == "bucket" and st["regrow_count"] == 0, st
    # a stale range from another camera: order may not depend on it
    v2 = View.headless(960, 540, yaw=2.0)
    third = plugin.sort(h, v2, s)
    assert _sort_equal(third, oracle.sort(c, v2, s))
    plugin.reset_adaptive_state()
    plugin.set_debug_flags(0x200000)
    try:
        guessed = plugin.sort(h, v, s)
        assert _sort_equal(guessed, ref)
    finally:
        plugin.set_debug_flags(0)
        plugin.reset_adaptive_state()
    h.free()


@pytest.mark.parametrize("wide", [0, 0x800], ids=["narrow", "wide"])
def test_bucket_sort_overflow_reruns_with_onesweep(plugin, oracle, wide):
    """(wide: the 16 384-pair buckets of the long lists, debug flag 0x800, on the same keys.)
    Keys the buckets cannot split: (a) 2000 splats at exactly one distance (one key value beyond the
    tie limit of the in-bucket ranking), (b) 40 000 splats inside a key range of a few ulps (one bucket over
    capacity, whatever the splitters). Forced onto the bucket path with a guessed table (debug flag 0x200000)
    and then with the table their own sorted list yields, the frame is re-run with the digit passes before
    anyone sees it: order stays bit-exact, the image is right, and the context backs off."""
    v = View.headless(640, 360)
    s = CloudSettings()
    for case in ("ties", "cluster"):
        c = random_gaussians_3d_seeded(60_000, 42)
        if case == "ties":
            c.position_visibility[:2000, :3] = np.float32([0.25, 1.0, -3.0])
        else:
            rng = np.random.default_rng(5)
            c.position_visibility[:40_000, :3] = (np.float32([0.0, 1.5, -4.0]) +
                                                   rng.uniform(-2e-6, 2e-6, (40_000, 3)).astype(np.float32))
        ref = oracle.sort(c, v, s)
        h = plugin.upload(c)
        plugin.reset_adaptive_state()
        plugin.set_debug_flags(0x200000 | wide)
        try:
            got = plugin.sort(h, v, s)
            st = plugin.stats()
            assert _sort_equal(got, ref), case
            assert st["regrow_count"] >= 1 and st["sort_path"] == "onesweep", (case, st)
            # the same in a RENDER frame: the kernels behind the sort must not touch the void list
            plugin.reset_adaptive_state()
            img = plugin.render(h, v, s)
            st = plugin.stats()
            assert st["regrow_count"] >= 1 and st["sort_path"] == "onesweep", (case, st)
            refimg, amb = oracle.render(c, ref, v, s, with_ambiguity=True)
            _assert_image(refimg, img, amb, frac_slack=0.01, what=f"render with bucket overflow ({case})")
        finally:
            plugin.set_debug_flags(wide)
        # without the flag: the table this view's own sorted list yields cannot split these keys either
        plugin.reset_adaptive_state()
        before = plugin.adaptive_counters()["reruns_sort"]
        for _ in range(6):
            assert _sort_equal(plugin.sort(h, v, s), ref), case
        after = plugin.adaptive_counters()
        assert after["reruns_sort"] > before               # it was tried, it failed, the frame was re-run ...
        assert plugin.stats()["sort_path"] == "onesweep"   # ... and the context has backed off by now
        plugin.set_debug_flags(0)
        h.free()
    plugin.reset_adaptive_state()


def test_splitter_tables_are_kept_per_view(plugin, oracle):
    """A context that alternates between cameras (the reference's multi_camera example) or clouds must not
    hand a frame the splitter table of another view: tables live in slots keyed by cloud, model transform and
    camera pose. After one frame per view every later frame runs the bucket sort without a single re-run."""
    ca, cb = random_gaussians_3d_seeded(300_000, 51), random_gaussians_3d_seeded(200_000, 52)
    ha, hb = plugin.upload(ca), plugin.upload(cb)
    views = [View.headless(640, 360, yaw=y) for y in (0.0, 1.6, 3.1, 4.7)]
    s = CloudSettings()
    plugin.reset_adaptive_state()
    jobs = [(ha, ca, v) for v in views] + [(hb, cb, views[0]), (hb, cb, views[2])]
    refs = [oracle.sort(c, v, s) for _, c, v in jobs]
    for (h, _, v), ref in zip(jobs, refs):   # one frame per (cloud, view): onesweep, leaves a table
        assert _sort_equal(plugin.sort(h, v, s), ref)
    before = plugin.adaptive_counters()
    for rnd in range(3):
        for (h, _, v), ref in zip(jobs, refs):
            assert _sort_equal(plugin.sort(h, v, s), ref)
            assert plugin.stats()["sort_path"] == "bucket", (rnd, plugin.stats())
    after = plugin.adaptive_counters()
    assert after["reruns_sort"] == before["reruns_sort"]
    assert after["bucket_frames"] - before["bucket_frames"] == 3 * len(jobs)
    # a small camera move reuses the table (that is what it is for); a model transform change does not
    near = View.perspective(transform_from((0.05, 1.5, 5.02)), 640, 360)
    assert _sort_equal(plugin.sort(ha, near, s), oracle.sort(ca, near, s)) and plugin.stats()["sort_path"] == "bucket"
    moved = CloudSettings(transform=transform_from((3.0, 0.0, 0.0)))
    assert _sort_equal(plugin.sort(ha, views[0], moved), oracle.sort(ca, views[0], moved))
    assert plugin.stats()["sort_path"] == "onesweep"
    ha.free()
    hb.free()
    plugin.reset_adaptive_state()


def test_bucket_and_onesweep_frames_are_bit_identical(plugin):
    """Pipelined frames (6 lanes on 3 streams, with and without frame graphs) on either sort path give the
    same bits as a blocking frame."""
    from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor
    c = random_gaussians_3d_seeded(400_000, 77)
    v = View.headless(1280, 720)
    s = CloudSettings(global_scale=0.5)
    h = plugin.upload(c)
    plugin.reset_adaptive_state()
    plugin.set_debug_flags(0x80000)
    ref = plugin.render(h, v, s)
    assert plugin.stats()["sort_path"] == "onesweep"
    plugin.set_debug_flags(0)
    plugin.render(h, v, s)
    got = plugin.render(h, v, s)
    assert plugin.stats()["sort_path"] == "bucket"
    assert np.array_equal(got, ref)
    plugin.set_async(True)
    plugin.set_pipeline_depth(6)
    for graphs in (False, True):
        plugin.set_graphs(graphs)
        plugin.set_profiling(0 if graphs else 2)
        for _ in range(24):
            plugin.render(h, v, s, download=False)
        plugin.synchronize()
        assert np.array_equal(framebuffer_as_tensor(plugin, 720, 1280).cpu().numpy(), ref)
        assert plugin.stats()["sort_path"] == "bucket"
    plugin.set_graphs(False)
    plugin.set_profiling(2)
    plugin.set_async(False)
    plugin.set_pipeline_depth(1)
    h.free()


def test_supertile_list_overflow_reruns_the_frame(plugin, oracle):
    """The supertile lists are sized from the longest list seen, not for the worst case (n entries in each
    of up to 256 lists). Started at 64 entries (debug flag 0x100000) every list of this frame overflows:
    the frame is re-run with larger lists, blocking and pipelined, and the image is the one a context with
    ample lists produces. Also: a 5 M-splat cloud on 8 lanes allocates < 8 GB of lists."""
    from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor
    c = random_gaussians_3d_seeded(200_000, 9)
    v = View.headless(1280, 720)
    s = CloudSettings()
    h = plugin.upload(c)
    plugin.reset_adaptive_state()
    ref = plugin.render(h, v, s)
    ample = plugin.stats()
    plugin.reset_adaptive_state()
    plugin.set_debug_flags(0x100000)
    try:
        got = plugin.render(h, v, s)
        st = plugin.stats()
        assert st["regrow_count"] >= 1 and np.array_equal(got, ref)
        assert st["list_capacity"] < 200_000 and st["instance_count"] == ample["instance_count"]
        # pipelined: the overflow is discovered when a lane is completed; every popped frame is correct
        plugin.reset_adaptive_state()
        plugin.set_async(True)
        plugin.set_pipeline_depth(4)
        for _ in range(10):
            plugin.render(h, v, s, download=False)
        plugin.synchronize()
        assert np.array_equal(framebuffer_as_tensor(plugin, 720, 1280).cpu().numpy(), ref)
    finally:
        plugin.set_debug_flags(0)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.reset_adaptive_state()
    e = oracle.sort(c, v, s)
    win = (616, 336, 664, 384)
    refo, amb = oracle.render(c, e, v, s, window=win, with_ambiguity=True)
    _assert_image(refo, got[336:384, 616:664], amb, frac_slack=0.01, what="crop after list regrow")
    h.free()


def test_precomputed_covariance_cloud(plugin, oracle):
    """The `precompute_covariance_3d` storage variant (Covariance3dOpacity plane instead of rotation + scale,
    src/gaussian/f32.rs:218-251, gaussian_3d.wgsl:77-88): with an identity model transform and global_scale 1
    (the two things compute_cov3d folds in and the precomputed path therefore cannot see) it must draw what
    the rotation / scale path draws — sort bit-exact, images equal to 1e-6 relative (the covariance entries
    are formed on the host by the CPU twin of the shader's arithmetic), and agree with the oracle."""
    c = random_gaussians_3d_seeded(30_000, 91)
    v = View.headless(640, 360, yaw=0.2)
    for kw in ({}, {"aabb": True}, {"sh_degree": 1, "opacity_adaptive_radius": False}):
        s = CloudSettings(**kw)
        h0, h1 = plugin.upload(c), plugin.upload(c, precompute_covariance_3d=True)
        a, b = plugin.render(h0, v, s), plugin.render(h1, v, s)
        e0, e1 = plugin.sort(h0, v, s), plugin.sort(h1, v, s)
        assert np.array_equal(e0["key"], e1["key"]) and np.array_equal(e0["index"], e1["index"])
        assert np.allclose(a, b, rtol=1e-6, atol=2e-6), np.abs(a - b).max()
        ref, amb = oracle.render(c, oracle.sort(c, v, s), v, s, with_ambiguity=True)
        _assert_image(ref, b, amb, frac_slack=0.01, what=f"cov3d cloud {kw}")
        h0.free()
    # what the variant cannot do is refused, not guessed
    for bad in (CloudSettings(gaussian_mode=GaussianMode.Gaussian2d), CloudSettings(rasterize_mode=RasterizeMode.Normal)):
        with pytest.raises(RuntimeError):
            plugin.render(h1, v, bad)
    # and it ignores global_scale / the transform's linear part, as the reference's shader variant does
    s2 = CloudSettings(global_scale=0.5)
    assert np.array_equal(plugin.render(h1, v, s2), plugin.render(h1, v, CloudSettings()))
    h1.free()


def test_async_frames_match_synchronous_frames(plugin, oracle):
    """bgs_set_async: frames are only enqueued; results and the watchdog check arrive at the next
    blocking call. Images must be bit-identical to the synchronous path."""
    c = random_gaussians_3d_seeded(30_000, 23)
    h = plugin.upload(c)
    s = CloudSettings(global_scale=0.5)
    views = [headless_view(g, 320, 180) for g in range(4)]
    sync_imgs = [plugin.render(h, v, s) for v in views]
    plugin.set_async(True)
    try:
        for v in views[:-1]:
            assert plugin.render(h, v, s, download=False) is None
        last = plugin.render(h, views[-1], s)  # download => completes the queue
        assert np.array_equal(last, sync_imgs[-1])
        plugin.stats()  # reading the stats resets the timing average
        for _ in range(5):
            plugin.render(h, views[1], s, download=False)
        plugin.synchronize()
        st = plugin.stats()
        assert st["frames_averaged"] == 5 and st["total_ms"] > 0
        again = plugin.render(h, views[1], s)
        assert np.array_equal(again, sync_imgs[1])
        es = plugin.sort(h, views[0], s)  # a blocking call right after async frames
        ref = oracle.sort(c, views[0], s)
        assert np.array_equal(es["index"], ref["index"])
    finally:
        plugin.set_async(False)
    h.free()


@pytest.mark.parametrize("size", [(1920, 1080), (1000, 600), (250, 130), (4096, 2304)])
def test_supertile_edge_does_not_change_the_image(plugin, size):
    """The coarse-bin (supertile) edge is a performance choice made from the completed frames' list
    statistics — 6, 8, 16 or 32 tiles at 1080p, 13, 16 or 32 at 4096x2304 (division by reciprocal multiply):
    the frames forced to every level and the automatic ones must be bit-identical, for splats smaller and
    larger than a supertile."""
    c = random_gaussians_3d_seeded(60_000, 71)
    h = plugin.upload(c)
    v = View.headless(*size)
    try:
        for gs in (0.05, 1.0):
            s = CloudSettings(global_scale=gs)
            imgs = []
            # levels 1, 0, 2, 3, then automatic (four times: the rule steps one level per completed frame)
            for flags in (0x8000, 0x10000, 0x400000, 0x800000, 0, 0, 0, 0):
                plugin.set_debug_flags(flags)
                imgs.append(plugin.render(h, v, s))
            for k in range(1, len(imgs)):
                assert np.array_equal(imgs[0], imgs[k]), (size, gs, k)
    finally:
        plugin.set_debug_flags(0)
    h.free()


def test_frame_graphs_follow_changing_inputs(plugin):
    """With bgs_set_graphs, async frames without stage timing replay a captured hipGraph:
    only keygen's node is updated per frame. Every frame of a sequence in which the view, the
    settings, the viewport, the pipeline variant, the cloud and the clear colour change must be
    bit-identical to the same frame launched directly (blocking call), and steady stretches must
    actually be replays."""
    from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor
    clouds = [random_gaussians_3d_seeded(40_000, 61), random_gaussians_3d_seeded(25_000, 62)]
    handles = [plugin.upload(c) for c in clouds]

    def view(i, w=320, h=192):
        v = headless_view(i % 8, w, h)
        v.clear_color = (0.1 * (i % 3), 0.05, 0.2, 0.5 if i % 2 else 0.0)
        return v

    steps = []  # (cloud index, view, settings)
    for i in range(6):   # moving camera, same everything else: replays
        steps.append((0, view(i), CloudSettings(global_scale=0.5)))
    for i in range(4):   # FrameParams-only settings changes: still replays
        steps.append((0, view(i), CloudSettings(global_scale=0.3 + 0.1 * i, global_opacity=0.7, sh_degree=i % 4)))
    steps.append((0, view(1, 256, 144), CloudSettings(global_scale=0.5)))              # viewport: recapture
    steps.append((0, view(2, 256, 144), CloudSettings(global_scale=0.5)))
    steps.append((0, view(3), CloudSettings(global_scale=0.5, aabb=True)))              # raster variant
    steps.append((0, view(4), CloudSettings(global_scale=0.5, rasterize_mode=RasterizeMode.Normal)))  # any_mode
    steps.append((0, view(5), CloudSettings(global_scale=0.5, radix_sort_depth_bits=RadixSortDepthBits.Bits16)))
    steps.append((1, view(6), CloudSettings(global_scale=0.5)))                        # another cloud
    steps.append((1, view(7), CloudSettings(global_scale=0.5, sort_mode=SortMode.Rayon)))
    for i in range(4):
        steps.append((0, view(i + 3), CloudSettings(global_scale=0.5)))

    direct = [plugin.render(handles[ci], v, s) for ci, v, s in steps]  # blocking: launched directly
    plugin.set_profiling(0)
    plugin.set_async(True)
    plugin.set_graphs(True)
    try:
        for depth in (1, 2, 3):
            plugin.set_pipeline_depth(depth)
            c0, r0 = plugin.graph_counters()
            for k, (ci, v, s) in enumerate(steps):
                plugin.render(handles[ci], v, s, download=False)
                plugin.synchronize()
                got = framebuffer_as_tensor(plugin, v.height, v.width).cpu().numpy()
                assert np.array_equal(got, direct[k]), f"depth {depth} step {k}"
            c1, r1 = plugin.graph_counters()
            assert c1 > c0 and r1 - r0 >= 2, (depth, c1 - c0, r1 - r0)  # (the supertile rule may re-capture too)
        # graphs off: same images, nothing captured or replayed
        plugin.set_graphs(False)
        c0, r0 = plugin.graph_counters()
        for k in (0, 5, 12):
            ci, v, s = steps[k]
            plugin.render(handles[ci], v, s, download=False)
            plugin.synchronize()
            assert np.array_equal(framebuffer_as_tensor(plugin, v.height, v.width).cpu().numpy(), direct[k])
        assert plugin.graph_counters() == (c0, r0)
    finally:
        plugin.set_graphs(False)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.set_profiling(2)
    for h in handles:
        h.free()


@pytest.mark.parametrize("wide", [0, 0x800], ids=["narrow", "wide"])
def test_frame_graphs_are_captured_anew_when_the_splitter_table_grows(plugin, wide):
    """(wide: the same under debug flag 0x800 — the wide buckets' 1024-thread sort launch, whose 128 KB of dynamic LDS need
    a function attribute, inside a stream capture.)
    Round 5's advisor (medium): the number of quantile keys a frame's clean-up leaves (FrameCleanup::split_sub, 256 * sub
    - 1 keys) is a rasteriser argument baked into a captured frame graph, but the graph's key did not hold it — when the
    draw-count hint crossed a 524 288-pair step under bgs_set_graphs, replays kept writing the CAPTURED sub's table under
    the new sub's label: a badly balanced (yet ascending) table, a bucket overflow and an onesweep re-run on every frame,
    silently. A camera that sees 60 k pairs, then one that sees all 700 k (sub 1 -> 2): the step must re-capture, and the
    frames behind it must sort on the bucket path without a single re-run."""
    from bevy_gaussian_splatting_amd import transform_from
    c = random_gaussians_3d_seeded(700_000, 23)
    h = plugin.upload(c)
    near = View.headless(480, 270)
    far = View.perspective(transform_from((0.0, 0.0, 120.0)), 480, 270)   # the whole cloud inside the frustum
    s = CloudSettings(global_scale=0.05)
    direct_far = plugin.render(h, far, s)
    assert plugin.stats()["draw_count"] > 600_000
    plugin.reset_adaptive_state()
    plugin.set_debug_flags(wide)
    plugin.set_profiling(0)
    plugin.set_async(True)
    plugin.set_graphs(True)
    plugin.set_pipeline_depth(2)
    from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor
    try:
        for _ in range(12):
            plugin.render(h, near, s, download=False)
        plugin.synchronize()
        c0, r0 = plugin.graph_counters()
        assert r0 > 0
        for _ in range(6):    # the hint jumps: frames here may re-run while the context learns the longer list
            plugin.render(h, far, s, download=False)
        plugin.synchronize()
        c1, _ = plugin.graph_counters()
        assert c1 > c0                                          # captured anew, not replayed with the old arguments
        a0 = plugin.adaptive_counters()
        for _ in range(24):
            plugin.render(h, far, s, download=False)
        plugin.synchronize()
        a1 = plugin.adaptive_counters()
        assert a1["reruns_sort"] == a0["reruns_sort"]           # the table under the label sub = 2 IS a sub = 2 table (wide: sub stays 1)
        assert a1["bucket_frames"] - a0["bucket_frames"] >= 20
        assert np.array_equal(framebuffer_as_tensor(plugin, 270, 480).cpu().numpy(), direct_far)
    finally:
        plugin.set_graphs(False)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.set_profiling(2)
        plugin.set_debug_flags(0)
        plugin.reset_adaptive_state()
    h.free()


def test_framebuffer_zero_copy_tensor_and_rccl_gather_single_rank(plugin):
    """The multi-GPU leg of bench.py: the device framebuffer wrapped zero-copy as a torch tensor
    (what RCCL sends) and gathered with the nccl backend (world_size 1 here; world_size 2 is
    covered on CPU with gloo in tests/test_multi_gpu_cpu.py)."""
    import os
    import torch
    import torch.distributed as dist
    from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor, gather_framebuffers

    c = random_gaussians_3d_seeded(20_000, 31)
    h = plugin.upload(c)
    v = headless_view(1, 320, 180)
    img = plugin.render(h, v, CloudSettings(global_scale=0.5))
    t = framebuffer_as_tensor(plugin, 180, 320)
    assert t.is_cuda and tuple(t.shape) == (180, 320, 4)
    assert np.array_equal(t.cpu().numpy(), img)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    try:
        bufs = [torch.empty_like(t.unsqueeze(0))]
        dist.gather(t.unsqueeze(0).contiguous(), gather_list=bufs, dst=0)
        torch.cuda.synchronize()
        assert np.array_equal(bufs[0][0].cpu().numpy(), img)
        assert gather_framebuffers(t.unsqueeze(0))[0].shape == (1, 180, 320, 4)
    finally:
        if created:
            dist.destroy_process_group()
    h.free()


def test_pipelined_frames_and_srgb8_output(plugin, oracle):
    """bgs_set_pipeline_depth: frames in flight on separate streams (lanes) must give exactly the
    images of the blocking path, in FIFO order from bgs_pipeline_pop; the Rgba8UnormSrgb copy must
    match the oracle's format conversion of the same f32 frame (+-1 LSB: pow vs exp2/log2)."""
    import torch
    from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor

    c = random_gaussians_3d_seeded(30_000, 41)
    h = plugin.upload(c)
    s = CloudSettings(global_scale=0.5)
    views = [headless_view(g, 320, 180) for g in range(7)]
    ref = [plugin.render(h, v, s) for v in views]
    plugin.set_output_srgb8(True)
    plugin.set_async(True)
    try:
        for depth in (1, 2, 3, 4):
            plugin.set_pipeline_depth(depth)
            got = []
            for v in views:
                plugin.render(h, v, s, download=False)
                if plugin.frames_in_flight() >= depth:
                    f32, u8 = plugin.pipeline_pop()
                    got.append((device_ptr_as_tensor(f32, (180, 320, 4), "<f4", "cuda:0").cpu().numpy(),
                                device_ptr_as_tensor(u8, (180, 320, 4), "|u1", "cuda:0").cpu().numpy()))
            while plugin.frames_in_flight():
                f32, u8 = plugin.pipeline_pop()
                got.append((device_ptr_as_tensor(f32, (180, 320, 4), "<f4", "cuda:0").cpu().numpy(),
                            device_ptr_as_tensor(u8, (180, 320, 4), "|u1", "cuda:0").cpu().numpy()))
            assert len(got) == len(views)
            for (f, u), r in zip(got, ref):
                assert np.array_equal(f, r)
                exp = oracle.encode_srgb8(r)
                assert np.abs(u.astype(np.int16) - exp.astype(np.int16)).max() <= 1
        # bench.py's N > 1 consumer: popped frames staged into batches (here on one rank, no collective)
        from bevy_gaussian_splatting_amd.multiview import BatchedFrameGather
        batches = []
        bg = BatchedFrameGather((180, 320, 4), torch.uint8, "cuda:0", batch=3,
                                on_batch=lambda per_rank: batches.append(per_rank[0].cpu().numpy().copy()))
        plugin.set_pipeline_depth(3)
        for v in views:
            plugin.render(h, v, s, download=False)
            if plugin.frames_in_flight() >= 3:
                bg.push(device_ptr_as_tensor(plugin.pipeline_pop()[1], (180, 320, 4), "|u1", "cuda:0"))
        while plugin.frames_in_flight():
            bg.push(device_ptr_as_tensor(plugin.pipeline_pop()[1], (180, 320, 4), "|u1", "cuda:0"))
        bg.flush()
        staged = np.concatenate(batches)
        assert bg.frames_received == len(views) and staged.shape[0] == len(views)
        for u, (_, u_ref) in zip(staged, got):
            assert np.array_equal(u, u_ref)
        # ... and the zero-copy variant: every frame is rendered straight into its slot of the batch
        # (bgs_set_srgb8_target), nothing is copied on pop
        batches2 = []
        bg2 = BatchedFrameGather((180, 320, 4), torch.uint8, "cuda:0", batch=3,
                                 on_batch=lambda per_rank: batches2.append(per_rank[0].cpu().numpy().copy()))
        plugin.set_output_srgb8(False)
        for v in views:
            plugin.set_srgb8_target(bg2.next_target().data_ptr())
            plugin.render(h, v, s, download=False)
            if plugin.frames_in_flight() >= 3:
                plugin.pipeline_pop()
                bg2.frame_completed()
        while plugin.frames_in_flight():
            plugin.pipeline_pop()
            bg2.frame_completed()
        bg2.flush()
        assert np.array_equal(np.concatenate(batches2), staged)
        plugin.set_output_srgb8(True)
        # more frames than lanes without popping: older frames are completed when their lane is reused
        plugin.set_pipeline_depth(3)
        for v in views:
            plugin.render(h, v, s, download=False)
        plugin.synchronize()
        assert plugin.frames_in_flight() == 0
        last = plugin.render(h, views[-1], s)  # blocking call after async ones
        assert np.array_equal(last, ref[-1])
    finally:
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.set_output_srgb8(False)
    h.free()


def test_rgba16f_target_and_packed_only_frames(plugin):
    """The reference's hdr colour attachment (TextureFormat::Rgba16Float, src/render/mod.rs:917-921): the
    binary16 image must be the f32 target rounded to nearest even (numpy float16), in both binning modes
    (fused into the rasteriser / separate encode pass); packed-only frames skip the f32 target, give the
    same packed bytes, and refuse an f32 read-back instead of handing out a stale buffer."""
    from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor
    c = random_gaussians_3d_seeded(40_000, 33)
    c.spherical_harmonic *= 4.0   # some channels beyond [0, 1] and a few past the f16 range after the pow
    v = View.headless(640, 360)
    v.clear_color = (0.25, 0.5, 0.125, 0.75)
    s = CloudSettings()
    h = plugin.upload(c)

    def packed(ptr, typestr):
        return device_ptr_as_tensor(ptr, (360, 640, 4), typestr, "cuda:0").cpu().numpy()
    try:
        for binning in ("scan", "sort"):
            plugin.set_binning(binning)
            plugin.set_output_rgba16f(True)
            img = plugin.render(h, v, s)
            ptr, nbytes = plugin.framebuffer_rgba16f_device_ptr()
            assert nbytes == 640 * 360 * 8
            half = packed(ptr, "<f2")
            with np.errstate(over="ignore"):
                assert np.array_equal(half.view(np.uint16), img.astype(np.float16).view(np.uint16)), binning
            with pytest.raises(RuntimeError):
                plugin.framebuffer_srgb8_device_ptr()
        plugin.set_binning("scan")
        plugin.set_packed_only(True)
        plugin.render(h, v, s, download=False)
        ptr, _ = plugin.framebuffer_rgba16f_device_ptr()
        assert np.array_equal(packed(ptr, "<f2").view(np.uint16), half.view(np.uint16))
        with pytest.raises(RuntimeError):
            plugin.framebuffer_device_ptr()
        with pytest.raises(RuntimeError):
            plugin.render(h, v, s)   # asks for a host copy of the f32 target
        plugin.set_output_srgb8(True)   # switches the packed format back
        plugin.render(h, v, s, download=False)
        p8, n8 = plugin.framebuffer_srgb8_device_ptr()
        assert n8 == 640 * 360 * 4
        only = packed(p8, "|u1")
        plugin.set_packed_only(False)
        ref = plugin.render(h, v, s)
        p8b, _ = plugin.framebuffer_srgb8_device_ptr()
        assert np.array_equal(only, packed(p8b, "|u1")) and np.isfinite(ref).all()
    finally:
        plugin.set_packed_only(False)
        plugin.set_output_srgb8(False)
        plugin.set_output_rgba16f(False)
        plugin.set_binning("scan")
    h.free()


# ---------------------------------------------------------------------------------------------
# a cloud that arrives through the INRIA .ply loader, padding splats included
# ---------------------------------------------------------------------------------------------
def test_ply_loaded_cloud_renders_like_the_oracle(plugin, oracle, binning, tmp_path):
    from bevy_gaussian_splatting_amd import parse_ply_3d, write_ply_3d
    c = random_gaussians_3d_seeded(5000, 11)
    c.scale_opacity[:, :3] = c.scale_opacity[:, :3] * 0.2 + 0.02
    c.scale_opacity[:, 3] = c.scale_opacity[:, 3] * 0.9 + 0.05
    path = os.path.join(tmp_path, "cloud.ply")
    write_ply_3d(c, path)
    loaded = parse_ply_3d(path)
    assert len(loaded) == 5024  # padded with Gaussian3d::default(): zero rotation / scale / opacity at the origin
    v = View.headless(160, 90)
    for kw in ({}, {"aabb": True}, {"gaussian_mode": GaussianMode.Gaussian2d}):
        s = CloudSettings(**kw)
        h = plugin.upload(loaded)
        got = plugin.render(h, v, s)
        e = oracle.sort(loaded, v, s)
        gs = plugin.sort(h, v, s)
        assert np.array_equal(gs["key"], e["key"]) and np.array_equal(gs["index"], e["index"])
        ref, amb = oracle.render(loaded, e, v, s, with_ambiguity=True)
        _assert_image(ref, got, amb, what=f"ply {kw}")
        h.free()


def test_gcloud_loaded_cloud_renders_like_the_oracle(plugin, oracle, tmp_path):
    """The `.gcloud` container on the hot path (src/io/gcloud/flexbuffers.rs:9-22, loader dispatch
    src/io/loader.rs:22-61): encode -> file -> decode (the vectorised path, and the general reader on the same
    bytes) -> upload -> sort + render, held to the oracle run on the ORIGINAL cloud — the codec is bit-exact
    for f32 planes, so the round trip may not move a pixel. Parity of the byte format itself stays unpinned
    (no reference-written file exists here)."""
    from bevy_gaussian_splatting_amd.io_gcloud import decode_gcloud, read_gcloud, write_gcloud
    c = random_gaussians_3d_seeded(6000, 21)
    c.position_visibility[:, 3] = np.arange(len(c)) % 4
    path = os.path.join(tmp_path, "cloud.gcloud")
    write_gcloud(c, path)
    loaded = read_gcloud(path)
    with open(path, "rb") as f:
        general = decode_gcloud(f.read(), fast=False)
    for a, b, d in zip((c.position_visibility, c.spherical_harmonic, c.rotation, c.scale_opacity),
                       (loaded.position_visibility, loaded.spherical_harmonic, loaded.rotation, loaded.scale_opacity),
                       (general.position_visibility, general.spherical_harmonic, general.rotation, general.scale_opacity)):
        assert np.array_equal(a, b) and np.array_equal(a, d)
    v = View.headless(320, 180)
    for kw in ({}, {"aabb": True, "rasterize_mode": RasterizeMode.Classification, "num_classes": 4}):
        s = CloudSettings(**kw)
        h = plugin.upload(loaded)
        got = plugin.render(h, v, s)
        e = oracle.sort(c, v, s)
        gs = plugin.sort(h, v, s)
        assert np.array_equal(gs["key"], e["key"]) and np.array_equal(gs["index"], e["index"])
        ref, amb = oracle.render(c, e, v, s, with_ambiguity=True)
        _assert_image(ref, got, amb, what=f"gcloud {kw}")
        h.free()


# ---------------------------------------------------------------------------------------------
# RasterizeMode colour variants (src/render/gaussian.wgsl:312-405)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [RasterizeMode.Classification, RasterizeMode.Depth, RasterizeMode.Normal,
                                  RasterizeMode.Position, RasterizeMode.OpticalFlow])
def test_rasterize_modes(plugin, oracle, binning, mode):
    c = random_gaussians_3d_seeded(6000, 23)
    c.position_visibility[:, 3] = (np.arange(len(c)) % 8).astype(np.float32)   # classes for Classification
    mn, mx = compute_aabb(c)
    v = View.headless(160, 90)
    v.previous_clip_from_world = View.headless(160, 90, yaw=0.002).clip_from_world   # the camera turned a little
    v.delta_time = 1.0 / 144.0
    tr = transform_from((0.5, -0.25, 0.0), rotation_y(0.3))
    for kw in ({}, {"aabb": True}, {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True},
               {"sort_mode": SortMode.Rayon}, {"radix_sort_depth_bits": RadixSortDepthBits.Bits16}):
        s = CloudSettings(rasterize_mode=mode, position_min=mn, position_max=mx, num_classes=5, transform=tr, **kw)
        for cloud in ((c, c.to_f16()) if not kw else (c,)):
            h = plugin.upload(cloud)
            got = plugin.render(h, v, s)
            cd = oracle.decode_f16(cloud) if cloud is not c else c
            e = oracle.sort(cd, v, s)
            if s.sort_mode == SortMode.Radix:
                assert (e["key"] == (0xFFFFFFFF >> (32 - int(s.radix_sort_depth_bits)))).any()  # Depth reads the culled tail
                gs = plugin.sort(h, v, s)
                assert np.array_equal(gs["index"], e["index"])
            else:
                e = plugin.sort(h, v, s)   # unstable-sort contract: draw in the device's (valid) order
            ref, amb = oracle.render(cd, e, v, s, with_ambiguity=True)
            _assert_image(ref, got, amb, what=f"{mode.name} {kw}")
            assert np.abs(got[..., :3]).max() > 0.05
            h.free()


def test_rasterize_mode_edge_cases(plugin, oracle):
    v = View.headless(64, 64)
    one = PlanarGaussian3d(np.array([[0.3, 1.2, 0, 1]], np.float32), np.zeros((1, 48), np.float32),
                           np.array([[1, 0, 0, 0]], np.float32), np.array([[0.5, 0.5, 0.5, 0.9]], np.float32))
    for mode in (RasterizeMode.Depth, RasterizeMode.Normal, RasterizeMode.Position, RasterizeMode.Classification):
        s = CloudSettings(rasterize_mode=mode, position_min=(-1, -1, -1), position_max=(1, 2, 1))
        h = plugin.upload(one)
        got = plugin.render(h, v, s)     # Depth with one splat: min == max -> 0/0 -> NaN colour, like the oracle
        e = oracle.sort(one, v, s)
        ref, amb = oracle.render(one, e, v, s, with_ambiguity=True)
        assert np.array_equal(np.isnan(ref), np.isnan(got))
        m = ~np.isnan(ref)
        ok, err = H.tolerance_mask(np.where(m, ref, 0), np.where(m, got, 0), amb)
        assert ok.all(), (mode, err.max())
        h.free()
    for bad in (RasterizeMode.Velocity,):
        h = plugin.upload(one)
        with pytest.raises(Exception, match="rasterize_mode"):
            plugin.render(h, v, CloudSettings(rasterize_mode=bad))
        h.free()


# ---------------------------------------------------------------------------------------------
# multi-camera entry layout (src/sort/mod.rs:331-393)
# ---------------------------------------------------------------------------------------------
def test_multi_camera_sorted_entries_layout(plugin, oracle):
    n = 5003   # not a perfect square: the asset is padded to 71*71 = 5041 entries per camera
    c = random_gaussians_3d_seeded(n, 31)
    views = [headless_view(g, 160, 90) for g in range(3)]
    s = CloudSettings()
    h = plugin.upload(c)
    se = plugin.sort_cameras(h, views, s)
    assert se.camera_count == 3 and se.entry_count == 71 * 71 and se.sorted.shape == (3 * 71 * 71,)
    for g, v in enumerate(views):
        ref = oracle.sort(c, v, s)
        chunk = se.sorted[g * n:(g + 1) * n]     # stride = cloud length, like the reference's draw offset
        assert np.array_equal(chunk["key"], ref["key"]) and np.array_equal(chunk["index"], ref["index"])
    assert np.all(se.sorted["key"][3 * n:] == 1)  # padding tail untouched
    # only camera 1's trigger fires: the other chunks keep their content
    before = se.sorted.copy()
    moved = headless_view(5, 160, 90)
    plugin.sort_cameras(h, [moved], s, sorted_entries=se, camera_indices=[1])
    assert np.array_equal(se.sorted[:n], before[:n]) and np.array_equal(se.sorted[2 * n:], before[2 * n:])
    assert np.array_equal(se.sorted[n:2 * n]["index"], oracle.sort(c, moved, s)["index"])
    with pytest.raises(ValueError):
        plugin.sort_cameras(h, [moved], s, sorted_entries=se, camera_indices=[3])
    h.free()



# ---------------------------------------------------------------------------------------------
# the correctly rounded ln of the adaptive cutoff, device build
# ---------------------------------------------------------------------------------------------
def test_exact_log_on_the_device_every_positive_input(plugin, oracle):
    """ln_f32_cr (csrc/exact_log.h) ON THE DEVICE against the oracle's x87 logl (and the host build of the same header)
    for EVERY positive finite binary32 input, subnormals included: 2 139 095 039 values compared through wrap-around
    checksums per 2^28-pattern chunk (bgs_selftest_ln_f32), plus a direct element-wise comparison of the patterns an
    opacity can take near the known ill-conditioned case, the special values and the hard cases of the exhaustive run."""
    first, last = 1, 0x7F7FFFFF
    chunk = 1 << 28
    total = 0
    b = first
    while b <= last:
        count = min(chunk, last - b + 1)
        dev = plugin.selftest_ln(b, count)[1]
        assert dev == oracle.ln_f32_checksum(b, count), f"device ln differs from the oracle in patterns {b:#x}..{b + count - 1:#x}"
        assert dev == int(H.shim().shim_ln_f32_checksum(b, count)), f"device ln differs from the host build in {b:#x}.."
        total += count
        b += count
    assert total == 2_139_095_039
    # element-wise: [0.0078, 1.0001] consecutively (opacities), specials and the nearest-to-a-boundary inputs
    lo = int(np.array([0.0078125], np.float32).view(np.uint32)[0])
    hi = int(np.array([1.0001], np.float32).view(np.uint32)[0])
    got = plugin.selftest_ln(lo, hi - lo + 1, download=True)[0]
    x = np.arange(lo, hi + 1, dtype=np.uint32).view(np.float32)
    assert np.array_equal(got.view(np.uint32), oracle.ln_f32(x).view(np.uint32))
    for pattern in (0x65d890d3, 0x4c5d65a5, 0x4d604ebe, 0x41178feb, 0x3c413d3a, 0x6f31a8ec, 1, 0x007FFFFF, 0x00800000, 0x3F800000):
        g = plugin.selftest_ln(pattern, 1)[0]
        with np.errstate(all="ignore"):
            assert g.view(np.uint32)[0] == oracle.ln_f32(np.array([pattern], np.uint32).view(np.float32)).view(np.uint32)[0]
    for pattern, want in ((0, -np.inf), (0x80000000, -np.inf), (0x7F800000, np.inf)):
        assert plugin.selftest_ln(pattern, 1)[0][0] == want
    for pattern in (0xBF800000, 0x7FC00000, 0xFF800000):
        assert np.isnan(plugin.selftest_ln(pattern, 1)[0][0])


# ---------------------------------------------------------------------------------------------
# WHOLE-FRAME parity at every BASELINE.json config (all 2 073 600 pixels against the oracle)
# ---------------------------------------------------------------------------------------------
def _whole_frame_parity(plugin, oracle, dec, handle, v, s, what, whole_frame_samples=None):
    """Every pixel of the 1920x1080 frame against the oracle's frame of the same inputs (the crops of the tests above
    stay as the fast path). The oracle rasterises every quad in full, back to front: seconds to minutes of CPU.
    whole_frame_samples = 1: the two frames whose multisampled oracle frame takes minutes (5 M dense splats, 1 M dense
    surfels) are compared whole on a camera with Msaa::Off, and at the view's own sample count on a 480 x 270 window in
    the middle of the frame (a sixteenth of it: 129 600 pixels, 510 tiles; round 6 — three 64 x 64 crops before) plus the
    three 64 x 64 corners that are not the window's (the heavy bottom-right one, top-left, top-right)."""
    import time
    if whole_frame_samples is not None and whole_frame_samples != v.msaa_samples:
        got4 = plugin.render(handle, v, s)
        e4 = oracle.sort(dec, v, s)
        t0 = time.time()
        for (x0, y0, w, hh) in ((720, 405, 480, 270), (1856, 1016, 64, 64), (0, 0, 64, 64), (1856, 0, 64, 64)):
            win = (x0, y0, x0 + w, y0 + hh)
            ref, amb = oracle.render(dec, e4, v, s, window=win, with_ambiguity=True)
            _assert_image(ref, got4[y0:y0 + hh, x0:x0 + w], amb, frac_slack=0.01 if w == 64 else 0.002, what=f"{what} x{v.msaa_samples} {win}")
        print(f"[windows at x{v.msaa_samples}: {what}] 480 x 270 + three 64 x 64 corners, oracle {time.time() - t0:.0f} s")
        v = View(v.world_from_view, v.view_from_world, v.clip_from_view, v.clip_from_world, v.viewport, v.clear_color,
                 msaa_samples=whole_frame_samples)
        what = f"{what} (whole frame at x{whole_frame_samples}, crops at the view's own sample count)"
    got = plugin.render(handle, v, s)
    st = plugin.stats()
    t0 = time.time()
    e = oracle.sort(dec, v, s)
    ref, amb = oracle.render(dec, e, v, s, with_ambiguity=True)
    t_oracle = time.time() - t0
    assert got.shape == ref.shape == (v.height, v.width, 4)
    strict, err = H.tolerance_mask(ref, got, None)
    _assert_image(ref, got, amb, frac_slack=0.0005, what=what)
    print(f"[whole frame: {what}] {ref.shape[0] * ref.shape[1]} pixels, {st['visible_count']} visible splats, max |err| "
          f"{err.max():.2e}, values on ambiguity slack {int((~strict).sum())} of {strict.size}, oracle {t_oracle:.0f} s")
    return got


@pytest.mark.parametrize("global_scale", [1.0, 0.05])
def test_whole_frame_parity_1m_3dgs(plugin, oracle, cloud_1m, global_scale):
    """configs[1] (the headline): 1 M f32 splats, 1920x1080, SH3, CloudSettings::default(); dense and scene-like."""
    h = plugin.upload(cloud_1m)
    _whole_frame_parity(plugin, oracle, cloud_1m, h, View.headless(1920, 1080), CloudSettings(global_scale=global_scale),
                        f"cfg1 1M 3DGS f32 gs={global_scale}")
    h.free()


def test_whole_frame_parity_1m_trained_like(plugin, oracle):
    """A cloud with trained-asset statistics (round 6: surfaces, flat log-normal splats, bimodal opacity, DC-dominated SH
    with colours in [0, 1]; gaussian.py trained_like_gaussians_3d_seeded) at the headline's size and camera: every pixel
    of the 1080p frame at the reference's sample count."""
    from bevy_gaussian_splatting_amd import trained_like_gaussians_3d_seeded
    c = trained_like_gaussians_3d_seeded(1_000_000, 7)
    h = plugin.upload(c)
    _whole_frame_parity(plugin, oracle, c, h, View.headless(1920, 1080), CloudSettings(), "trained-like 1M 3DGS f32")
    h.free()


@pytest.mark.parametrize("global_scale", [1.0, 0.05])
def test_whole_frame_parity_5m_f16(plugin, oracle, global_scale):
    """configs[2]: 5 M-splat f16 planar cloud at 1080p; dense and scene-like."""
    c = random_gaussians_3d_seeded(5_000_000, 3).to_f16()
    dec = oracle.decode_f16(c)
    h = plugin.upload(c)
    _whole_frame_parity(plugin, oracle, dec, h, View.headless(1920, 1080), CloudSettings(global_scale=global_scale),
                        f"cfg2 5M f16 gs={global_scale}", whole_frame_samples=1 if global_scale == 1.0 else None)
    h.free()


@pytest.mark.parametrize("aabb", [True, False])
def test_whole_frame_parity_1m_2dgs(plugin, oracle, cloud_1m, aabb):
    """configs[3]: 1 M-splat 2DGS cloud at the benchmarked setting (global_scale 1): the true surfel path (aabb) and
    the default OBB quad."""
    h = plugin.upload(cloud_1m)
    _whole_frame_parity(plugin, oracle, cloud_1m, h, View.headless(1920, 1080),
                        CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=aabb), f"cfg3 1M 2DGS aabb={aabb}",
                        whole_frame_samples=1 if aabb else None)
    h.free()



@pytest.mark.parametrize("what", ["seed_900850", "obb_x4", "aabb3d_x4", "obb_x1"])
def test_heavy_tiles_under_a_depth_buffer_give_the_same_bits(plugin, cloud_1m, what):
    """Round 6's last exploration found it (medium seed 900850 in a sequence of frames on one context: 48 k large splats, AABB
    quads, 4 samples, a depth buffer): with a depth buffer bound, a heavy tile drawn by four strip waves came out 1 ulp off
    on ~270 pixels — a strip wave took the depth range of its own strip for the pre-test that picks the all-samples or the
    per-sample update (they differ in the last bit), a tile wave the whole tile's. Both take the tile's now. That
    configuration at the supertile level the sequence had brought it to (debug flag 0x400000: level 2, strips from the
    second frame on), and dense 1 M frames under a jittered, tilted depth plane: every repetition equals the frame without
    strip workgroups (0x2000000)."""
    if what == "seed_900850":
        c, v, s = H.random_case(1000 + 900850, medium=True)
        assert v.depth_host is not None and v.msaa_samples == 4 and s.aabb
        level = 0x400000
    else:
        c = cloud_1m
        v = View.headless(1920, 1080, msaa_samples=1 if what.endswith("x1") else 4)
        s = CloudSettings(aabb=what.startswith("aabb3d"))
        v.depth_host = H.random_depth_buffer(c, v, s, np.random.default_rng(17))
        level = 0
    h = plugin.upload(c)
    plugin.reset_adaptive_state()
    try:
        with _scene_depth(plugin, v):
            plugin.set_debug_flags(level | 0x2000000)
            for _ in range(4):
                plain = plugin.render(h, v, s)
            assert plugin.stats()["strip_tiles"] == 0
            plugin.set_debug_flags(level)
            plugin.reset_adaptive_state()
            strips = []
            for k in range(8):
                img = plugin.render(h, v, s)
                strips.append(plugin.stats()["strip_tiles"])
                assert np.array_equal(img.view(np.uint32), plain.view(np.uint32)), (what, k, strips)
            assert strips[-1] > 0, strips
    finally:
        plugin.set_debug_flags(0)
        plugin.reset_adaptive_state()
        h.free()


@pytest.mark.parametrize("what", ["1m_f32", "2dgs_obb", "aabb3d"])
def test_heavy_tiles_drawn_by_strip_waves_give_the_same_bits(plugin, oracle, cloud_1m, what):
    """Dense frames (supertile level >= 2) run the rasteriser's mid-round-exit instantiation, and the tiles a completed
    frame found heavy (more than one staging round) are drawn by a workgroup of four strip waves in the frames behind it
    while their regular wave steps aside (`stats()["strip_tiles"]`). Whichever shape draws a pixel, its arithmetic is the
    same: every frame of the sequence — first frame (level 1, nothing of this), the frames while the level climbs, the
    frames with strips, the same frames pipelined on 8 lanes (strips off by default there, forced on as well) — is
    bit-identical to the frame with both switched off
    (debug flags 0x1000000 | 0x2000000), and the frame agrees with the oracle on crops in the heavy corner."""
    from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor
    kw = {"1m_f32": {}, "2dgs_obb": {"gaussian_mode": GaussianMode.Gaussian2d}, "aabb3d": {"aabb": True}}[what]
    v = View.headless(1920, 1080)
    s = CloudSettings(**kw)
    h = plugin.upload(cloud_1m)
    plugin.reset_adaptive_state()
    try:
        plugin.set_debug_flags(0x1000000 | 0x2000000 | 0x10000000)   # (and no cost-ordered workgroups: the test below)
        for _ in range(4):
            plain = plugin.render(h, v, s)
        assert plugin.stats()["strip_tiles"] == 0
        plugin.set_debug_flags(0)
        plugin.reset_adaptive_state()
        strips = []
        for k in range(8):
            img = plugin.render(h, v, s)
            strips.append(plugin.stats()["strip_tiles"])
            assert np.array_equal(img, plain), (what, k, strips)
        assert strips[0] == 0 and strips[-1] > 0, strips          # the feedback kicks in after the level has climbed
        assert strips[-1] < 2000                                     # ... for a few per cent of the 8160 tiles
        print(f"[heavy tiles {what}] strip tiles per frame: {strips}")
        # with several frames in flight the strips are off by default (the tail is filled by the other lanes' kernels):
        # forced on (debug flag 0x4000000) they must still give the same bits, whichever lane's list a frame reads
        plugin.set_async(True)
        plugin.set_pipeline_depth(8)
        for flags, want_strips in ((0, False), (0x4000000, True)):
            plugin.set_debug_flags(flags)
            for _ in range(40):
                plugin.render(h, v, s, download=False)
            plugin.synchronize()
            assert (plugin.stats()["strip_tiles"] > 0) == want_strips, (flags, plugin.stats()["strip_tiles"])
            assert np.array_equal(framebuffer_as_tensor(plugin, 1080, 1920).cpu().numpy(), plain)
        plugin.set_debug_flags(0)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        # a moving camera: every frame reads the list its predecessor — ANOTHER view — left: balance, never pixels.
        # The references first (a frame without feedback clears the lane's list), then the moving sequence in one go.
        views = [View.headless(1920, 1080, yaw=0.02 * (k + 1)) for k in range(6)]
        plugin.set_debug_flags(0x1000000 | 0x2000000)
        refs = [plugin.render(h, vk, s) for vk in views]
        plugin.set_debug_flags(0)
        plugin.render(h, v, s)                                   # leaves the static view's list behind
        stale_used = 0
        for k, vk in enumerate(views):
            a = plugin.render(h, vk, s)
            used = plugin.stats()["strip_tiles"]
            stale_used += used
            assert np.array_equal(a, refs[k]), (what, k, used)
        assert stale_used > 0 or what != "1m_f32", what          # stale lists of the previous views were consumed
        # a frame that is RE-RUN (a capacity turned out too small: here forced, debug flag 0x8000000) while strips are
        # active writes the feedback buffer of its failed attempt again and still reads its predecessor's list — never
        # the buffer it writes (round 3's advisor finding: the two used to alias, and flagged tiles went undrawn)
        plugin.render(h, v, s)
        plugin.set_debug_flags(0x8000000)
        for k in range(3):
            a = plugin.render(h, v, s)
            st = plugin.stats()
            assert st["regrow_count"] >= 1 and (st["strip_tiles"] > 0 or what != "1m_f32"), (what, k, st["strip_tiles"], st["regrow_count"])
            assert np.array_equal(a, plain), (what, k)
        plugin.set_debug_flags(0)
    finally:
        plugin.set_debug_flags(0)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
    e = oracle.sort(cloud_1m, v, s)
    for (x0, y0) in ((1860, 1020), (1780, 940)):
        win = (x0, y0, x0 + 48, y0 + 48)
        ref, amb = oracle.render(cloud_1m, e, v, s, window=win, with_ambiguity=True)
        _assert_image(ref, plain[y0:y0 + 48, x0:x0 + 48], amb, frac_slack=0.01, what=f"heavy corner {what} {win}")
    h.free()
    plugin.reset_adaptive_state()

@pytest.mark.parametrize("runs", [1, 2, 4])
def test_tile_order_is_a_permutation_whatever_the_costs_hold(plugin, runs):
    """tile_order_kernel on arbitrary per-tile costs (bgs_selftest_tile_order): for every grid size — fewer workgroups
    than XCDs, counts that are no multiple of 8 or 4, the largest grid — and whatever the cost words hold (zeros, all
    equal, all 0xFFFF, random, a completed frame's shape) the order is a permutation of the workgroups (a workgroup
    nobody draws would be a hole in the image, one drawn twice a race), XCD b % 8 keeps exactly the workgroups the
    static order gives it (splat_math.h xcd_runs_item, the host build of the same function), and inside a share the
    workgroups' heaviest tiles are non-increasing with ties in the share's own order. Round 6: bit 15 of a cost word says the
    tile ended saturated; the kernel's two sums (all work, work of those tiles) are what the host picks the mid-round-exit
    rasteriser by."""
    import ctypes
    l = H.shim()
    rng = np.random.default_rng(1234 + runs)
    for ntiles in [1, 3, 4, 5, 29, 32, 33, 61, 255, 1021, 1024, 5292, 8160, 8161, 14400, 65535]:
        nb = (ntiles + 3) // 4
        static = np.empty(nb, np.uint32)
        l.shim_xcd_runs_items(nb, runs, static.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
        for kind in ("zeros", "equal", "max", "random", "few_levels", "ramp"):
            cost = {"zeros": np.zeros(ntiles, np.uint16), "equal": np.full(ntiles, 777, np.uint16),
                    "max": np.full(ntiles, 0xFFFF, np.uint16),
                    "random": rng.integers(0, 65536, ntiles).astype(np.uint16),
                    "few_levels": (rng.integers(0, 4, ntiles) * 100).astype(np.uint16),
                    "ramp": (np.arange(ntiles) % 65536).astype(np.uint16)}[kind]
            order, sums = plugin.selftest_tile_order(cost, runs, sums=True)
            order = order.astype(np.int64)
            assert np.array_equal(np.sort(order), np.arange(nb)), (ntiles, runs, kind)
            work = cost.astype(np.int64) & 0x7FFF                      # (bit 15: the tile ended saturated — no part of its cost)
            assert sums == (int(work.sum()), int(work[cost >= 0x8000].sum())), (ntiles, runs, kind)
            group = np.zeros(nb * 4, np.int64)
            group[:ntiles] = work
            group = group.reshape(nb, 4).max(axis=1)                  # a workgroup costs what its heaviest tile costs
            for x in range(min(8, nb)):
                mine, share = order[x::8], static[x::8].astype(np.int64)
                assert np.array_equal(np.sort(mine), np.sort(share)), (ntiles, runs, kind, x)
                c = group[mine]
                assert np.all(np.diff(c) <= 0), (ntiles, runs, kind, x)
                pos = {g: i for i, g in enumerate(share)}              # ties: the share's own (spatial) order
                p = np.array([pos[g] for g in mine])
                same = np.diff(c) == 0
                assert np.all(np.diff(p)[same] > 0), (ntiles, runs, kind, x)


@pytest.mark.parametrize("what", ["dense", "scene_like", "surfel", "aabb3d_depth"])
def test_cost_ordered_raster_workgroups_give_the_same_bits(plugin, cloud_1m, what):
    """A frame with more tile waves than the chip holds at once (4 samples per pixel: 5 waves per SIMD, 5120 of a 1080p
    frame's 8160 tiles) draws its raster workgroups in the order made of the per-tile costs a completed frame left
    (tile_order_kernel: heaviest first inside every XCD's share). The order is a permutation of the
    workgroups whatever the costs hold, and which workgroup draws a tile changes no arithmetic: every frame of a
    sequence — first frame (no costs yet), ordered frames, frames of a moving camera (another view's costs), a re-run
    frame, frames in flight (with and without the strip workgroups) — is bit-identical to the frame with the
    feedback switched off (debug flag 0x10000000). A viewport whose workgroup count is no multiple of 8 and the
    instantiations with their own XCD shares (surfels: four runs per XCD) are covered."""
    from bevy_gaussian_splatting_amd.multiview import framebuffer_as_tensor
    kw = {"dense": {}, "scene_like": {"global_scale": 0.05},
          "surfel": {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True, "global_scale": 0.3},
          "aabb3d_depth": {"aabb": True}}[what]
    # (114 x 63 tiles: 1796 workgroups = 8 * 224 + 4 — no multiple of 8 — and more tile waves than the chip holds at six per
    # SIMD, so that the frame is ordered at all; 1336 x 1000 until the multisampled kernels went from five to six in round 6)
    W, Hh = (1816, 1000) if what == "scene_like" else (1920, 1080)
    s = CloudSettings(**kw)
    v = View.headless(W, Hh)
    views = [View.headless(W, Hh, yaw=0.03 * (k + 1)) for k in range(3)]
    depth_dev = None
    if what == "aabb3d_depth":
        d = np.full((Hh, W, 4), 0.02, np.float32)
        d[:, W // 2:, :] = 0.2
        depth_dev = plugin.upload_depth(d)
        for vk in [v] + views:
            vk.depth_device_ptr = depth_dev
    h = plugin.upload(cloud_1m)
    plugin.reset_adaptive_state()
    try:
        base = plugin.tile_order_counters()
        plugin.set_debug_flags(0x10000000)
        for _ in range(4):
            plain = plugin.render(h, v, s)
        refs = [plugin.render(h, vk, s) for vk in views]
        assert plugin.tile_order_counters() == base
        plugin.set_debug_flags(0)
        plugin.reset_adaptive_state()
        for k in range(6):
            img = plugin.render(h, v, s)
            assert np.array_equal(img, plain), (what, k)
        cost, ordered, made = (a - b for a, b in zip(plugin.tile_order_counters(), base))
        assert cost >= 6 and 1 <= ordered <= cost - 1, (what, cost, ordered)   # every frame but the first (and re-runs without a completed predecessor)
        assert 1 <= made < ordered, (what, made, ordered)                      # made once, then kept (refreshed every 8th frame)
        for k in range(10):                                                    # ... across a refresh
            assert np.array_equal(plugin.render(h, v, s), plain), (what, "kept order", k)
        assert plugin.tile_order_counters()[2] - base[2] >= made + 1, (what, made)
        for k, vk in enumerate(views):                                         # another view's costs: balance, never pixels
            assert np.array_equal(plugin.render(h, vk, s), refs[k]), (what, "moving", k)
        plugin.render(h, v, s)
        plugin.set_debug_flags(0x8000000)                                      # forced re-run: writes its own buffer again, reads its predecessor's
        for k in range(2):
            a = plugin.render(h, v, s)
            assert plugin.stats()["regrow_count"] >= 1
            assert np.array_equal(a, plain), (what, "re-run", k)
        plugin.set_debug_flags(0x40000000)                                     # the order made anew with every frame
        made = plugin.tile_order_counters()[2]
        for k in range(3):
            assert np.array_equal(plugin.render(h, v, s), plain), (what, "refreshed every frame", k)
        assert plugin.tile_order_counters()[2] - made == 3, what
        before = plugin.tile_order_counters()[1]
        plugin.set_async(True)
        plugin.set_pipeline_depth(8)
        for flags, want in ((0x20000000, False), (0, True), (0x4000000, True)):
            plugin.set_debug_flags(flags)
            for _ in range(40):
                plugin.render(h, v, s, download=False)
            plugin.synchronize()
            now = plugin.tile_order_counters()[1]
            assert (now > before) == want, (what, flags, before, now)
            before = now
            assert np.array_equal(framebuffer_as_tensor(plugin, Hh, W).cpu().numpy(), plain), (what, "in flight", flags)
    finally:
        plugin.set_debug_flags(0)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        h.free()
        if depth_dev is not None:
            plugin.device_free(depth_dev)
        plugin.reset_adaptive_state()


def test_first_frames_of_a_context_learn_one_by_one(plugin):
    """A context that has learnt nothing (list capacity, supertile level) used to send pipeline-depth async frames out on
    first guesses and re-run every one of them ("reruns: 8" in every first use). Now its async frames are completed one by
    one until a frame has run with everything it needed: a couple of re-runs instead of one per lane, and the frames
    completed early are handed out by bgs_pipeline_pop in order like any other."""
    from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor
    c = random_gaussians_3d_seeded(200_000, 9)
    views = [View.headless(1280, 720, yaw=0.01 * k) for k in range(8)]
    s = CloudSettings()
    h = plugin.upload(c)
    refs = [plugin.render(h, v, s) for v in views]
    assert not np.array_equal(refs[0], refs[1])
    try:
        for depth in (8, 3):
            plugin.reset_adaptive_state()
            before = plugin.adaptive_counters()
            plugin.set_async(True)
            plugin.set_pipeline_depth(depth)
            for rnd in range(2):                                   # round 0 learns, round 1 runs on what it learnt
                for k in range(depth):
                    plugin.render(h, views[k], s, download=False)
                    assert plugin.frames_in_flight() == k + 1
                for k in range(depth):
                    f32, _ = plugin.pipeline_pop()
                    got = device_ptr_as_tensor(f32, (720, 1280, 4), "<f4", "cuda:0").cpu().numpy()
                    assert np.array_equal(got, refs[k]), (depth, rnd, k)
                    assert plugin.frames_in_flight() == depth - 1 - k
                after = plugin.adaptive_counters()
                reruns = sum(after[n] - before[n] for n in ("reruns_sort", "reruns_lists", "reruns_instances"))
                assert reruns <= 3, (depth, rnd, reruns, before, after)
            plugin.set_async(False)
        # ANOTHER kind of frame in the same context (here: splats twenty times smaller, then the first kind again, then a
        # larger viewport): what the previous kind left may not fit — found out by ONE frame, not by every lane in flight
        plugin.set_async(True)
        plugin.set_pipeline_depth(8)
        small = CloudSettings(global_scale=0.05)
        big_views = [View.headless(1920, 1080, yaw=0.01 * k) for k in range(8)]
        for kind, (vs, st_) in enumerate(((views, small), (views, s), (big_views, s), (views, small))):
            before = plugin.adaptive_counters()
            for rnd in range(2):
                for k in range(8):
                    plugin.render(h, vs[k], st_, download=False)
                plugin.synchronize()
            after = plugin.adaptive_counters()
            reruns = sum(after[n] - before[n] for n in ("reruns_sort", "reruns_lists", "reruns_instances"))
            assert reruns <= 3, (kind, reruns, before, after)
    finally:
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.reset_adaptive_state()
    h.free()


@pytest.mark.parametrize("n,mode", [(300_000, "rayon"), (1_000_000, "rayon"), (1_000_000, "radix_far"), (2_000_000, "radix_far"),
                                    (2_600_000, "radix_far"), (5_000_000, "rayon"), (700_000, "std")])
def test_bucket_sort_with_every_splat_drawable(plugin, oracle, n, mode):
    """D = N (round 4's verdict, weak item 9: the headline camera's list is 88 % culled sentinels, and more than 786 k
    drawable pairs went through four digit passes): SortMode::Rayon / Std never cull (src/sort/rayon.rs:86-104), and a
    SortMode::Radix camera that sees the whole cloud keys every splat. Since round 5 the bucket path takes 256 * sub
    buckets — the 256 * sub - 1 exact quantile keys of a completed frame's list: sub = 2 at 1 M, 3 at 2 M (the table still
    travels in keygen's arguments), 6 at 2.6 M and 11 at 5 M (copied to the lane's device table ahead of keygen) — and
    bgs_sort's ordered keygen places its pairs by atomics plus ONE chain for the culled tail. Bit-exact with the oracle on
    the first call (digit passes) and on the later ones, under a slowly moving camera as well."""
    c = random_gaussians_3d_seeded(n, 300 + n % 97)
    if mode == "radix_far":
        s = CloudSettings()
        views = [View.perspective(transform_from((0.02 * k, 0.0, 120.0), (0.0, 0.0, 0.0, 1.0)), 1920, 1080) for k in range(3)]
    else:
        s = CloudSettings(sort_mode=SortMode.Rayon if mode == "rayon" else SortMode.Std)
        views = [View.headless(1920, 1080, yaw=0.002 * k) for k in range(3)]
    h = plugin.upload(c)
    try:
        plugin.reset_adaptive_state()
        paths = []
        for k, v in enumerate(views):
            got = plugin.sort(h, v, s)
            paths.append(plugin.stats()["sort_path"])
            ref = oracle.sort(c, v, s)
            assert plugin.stats()["draw_count"] == n
            assert np.array_equal(got["key"], ref["key"]), (k, paths)
            if mode == "radix_far":
                assert np.array_equal(got["index"], ref["index"]), (k, paths)
            else:   # the reference's CPU sort is unstable: equal keys may come in any order; ours is (key, index)
                assert np.array_equal(np.sort(got["index"]), np.arange(n, dtype=np.uint32))
                assert np.array_equal(got["index"], ref["index"])
        assert paths[0] == "onesweep" and paths[-1] == "bucket", paths
    finally:
        plugin.reset_adaptive_state()
    h.free()


def test_fine_buckets_on_small_lists_give_the_same_frames(plugin, oracle):
    """Debug flags 0x200 / 0x400: at least 768 buckets (767 quantile keys in keygen's arguments) / at least 1280 (1279 keys in
    the lane's device table) whatever the list's length — headline-sized lists normally take 256 buckets: rendered frames
    (chainless keygen) and bgs_sort (ordered keygen, culled tail) give the bits of the 256-bucket path and of the digit
    passes. Flag 0x800 (round 6): the WIDE buckets of the long lists (16 384 pairs, 1024-thread workgroups, 128 KB of LDS) on
    this short one, with each of the three tables."""
    c = random_gaussians_3d_seeded(250_000, 71)
    v, s = View.headless(1280, 720), CloudSettings()
    h = plugin.upload(c)
    ref_entries = oracle.sort(c, v, s)
    try:
        out = {}
        for name, flags in (("passes", 0x80000), ("coarse", 0), ("fine", 0x200), ("device_table", 0x400),
                            ("wide", 0x800), ("wide_fine", 0xA00), ("wide_device_table", 0xC00)):
            plugin.reset_adaptive_state()
            plugin.set_debug_flags(flags)
            for _ in range(3):
                img = plugin.render(h, v, s)
            st = plugin.stats()
            assert st["sort_path"] == ("onesweep" if name == "passes" else "bucket"), (name, st)
            for _ in range(2):
                e = plugin.sort(h, v, s)
            assert plugin.stats()["sort_path"] == ("onesweep" if name == "passes" else "bucket")
            out[name] = (img, e)
        for name in ("coarse", "fine", "device_table", "wide", "wide_fine", "wide_device_table"):
            assert np.array_equal(out[name][0], out["passes"][0]), name
            assert np.array_equal(out[name][1]["key"], ref_entries["key"]) and np.array_equal(out[name][1]["index"], ref_entries["index"]), name
    finally:
        plugin.set_debug_flags(0)
        plugin.reset_adaptive_state()
    h.free()


def test_rendered_frames_with_a_long_draw_list_take_the_wide_buckets(plugin, oracle):
    """Round 6: a camera that sees 1.8 M splats (past the 1.57 M pairs the narrow geometry holds in 768 buckets) sorts its RENDERED
    frames — the chainless keygen — into wide buckets (16 384 pairs, 1024-thread workgroups). Same frame, bit for bit, as
    with the digit passes (debug flag 0x80000) and with narrow buckets only (0x100); the draw list is the oracle's."""
    from bevy_gaussian_splatting_amd import transform_from
    c = random_gaussians_3d_seeded(1_800_000, 61)
    far = View.perspective(transform_from((0.0, 0.0, 120.0)), 480, 270)   # the whole cloud inside the frustum
    s = CloudSettings(global_scale=0.05)
    h = plugin.upload(c)
    try:
        out = {}
        for name, flags in (("passes", 0x80000), ("wide", 0), ("narrow", 0x100)):
            plugin.reset_adaptive_state()
            plugin.set_debug_flags(flags)
            for _ in range(3):
                img = plugin.render(h, far, s)
            st = plugin.stats()
            assert st["draw_count"] == 1_800_000
            assert st["sort_path"] == ("onesweep" if name == "passes" else "bucket"), (name, st)
            out[name] = img
        assert np.array_equal(out["wide"], out["passes"]) and np.array_equal(out["narrow"], out["passes"])
        plugin.set_debug_flags(0)
        for _ in range(2):
            e = plugin.sort(h, far, s)
        assert plugin.stats()["sort_path"] == "bucket"
        ref = oracle.sort(c, far, s)
        assert np.array_equal(e["key"], ref["key"]) and np.array_equal(e["index"], ref["index"])
    finally:
        plugin.set_debug_flags(0)
        plugin.reset_adaptive_state()
        h.free()


def test_forty_kinds_of_frame_stay_pipelined_after_their_first_visit(plugin):
    """Round 4 remembered the last 16 kinds of frame (FIFO) and hashed the cloud's address and the raw global_scale bits into
    the kind: a host cycling through more than 16 (cloud, viewport, mode) combinations, uploading a cloud per frame or
    animating the scale completed EVERY async frame inside its render call — correct images at the blocking rate, silently
    (ADVICE round 4, medium). Now: the set never forgets, the cloud counts by size and format, the scale by half octaves,
    and bgs_learning_counters shows the phase. 40 kinds, three cycles: only the first visit of a kind completes frames
    early; every frame is bit-identical to the blocking frame of the same inputs."""
    from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor
    clouds = [random_gaussians_3d_seeded(n, 70 + i) for i, n in enumerate((30_000, 45_000))]
    handles = [plugin.upload(c) for c in clouds]
    sizes = [(320, 192), (400, 240), (480, 272), (352, 208), (448, 256)]
    kinds = []
    for ci in range(2):
        for (w, hh) in sizes:
            for aabb in (False, True):
                for gs in (1.0, 0.2):
                    kinds.append((ci, View.headless(w, hh, yaw=0.02 * len(kinds)), CloudSettings(aabb=aabb, global_scale=gs), (hh, w)))
    assert len(kinds) == 40
    refs = [plugin.render(handles[ci], v, s) for ci, v, s, _ in kinds]
    try:
        plugin.reset_adaptive_state()
        assert plugin.learning_counters()["kinds_settled"] == 0
        plugin.set_async(True)
        plugin.set_pipeline_depth(4)
        early = []
        for cycle in range(3):
            for k, (ci, v, s, shape) in enumerate(kinds):
                for rep in range(3):
                    plugin.render(handles[ci], v, s, download=False)
                # pop the three frames of this kind and compare the last one
                while plugin.frames_in_flight() > 1:
                    plugin.pipeline_pop()
                f32, _ = plugin.pipeline_pop()
                got = device_ptr_as_tensor(f32, (*shape, 4), "<f4", "cuda:0").cpu().numpy()
                assert np.array_equal(got, refs[k]), (cycle, k)
            early.append(plugin.learning_counters()["early_frames"])
        lc = plugin.learning_counters()
        assert lc["kinds_settled"] == 40
        assert 40 <= early[0] <= 40 * 3, early            # the first visit of a kind: one to three frames completed early
        assert early[1] == early[0] and early[2] == early[0], early   # ... and never again: pipelined from then on

        # a NEW cloud handle of a known size for every frame, and an animated global_scale: still pipelined
        before = plugin.learning_counters()["early_frames"]
        ci, v, s, shape = kinds[0]
        for i in range(12):
            hx = plugin.upload(clouds[0])
            plugin.render(hx, v, s, download=False)
            plugin.synchronize()
            hx.free()
        for i in range(64):
            plugin.render(handles[0], v, CloudSettings(global_scale=1.0 + 0.004 * i), download=False)
            if plugin.frames_in_flight() >= 4:
                plugin.pipeline_pop()
        plugin.synchronize()
        after = plugin.learning_counters()["early_frames"]
        assert after - before <= 3, (before, after)     # (1.0 -> 1.25 crosses at most one half-octave step)
    finally:
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.reset_adaptive_state()
    for h in handles:
        h.free()


def test_a_kind_that_never_runs_clean_is_settled_on_anyway(plugin):
    """Debug flag 0x8000000 re-runs every frame: no frame of the kind ever completes "with everything it needed". The learning
    phase is bounded all the same (LEARN_MAX frames in a row of one kind), after which the kind's frames are pipelined."""
    c = random_gaussians_3d_seeded(40_000, 75)
    h = plugin.upload(c)
    v, s = View.headless(416, 240), CloudSettings()
    ref = plugin.render(h, v, s)
    from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor
    try:
        plugin.reset_adaptive_state()
        plugin.set_debug_flags(0x8000000)
        plugin.set_async(True)
        plugin.set_pipeline_depth(4)
        e0 = plugin.learning_counters()["early_frames"]
        for i in range(16):
            plugin.render(h, v, s, download=False)
            if plugin.frames_in_flight() >= 4:
                f32, _ = plugin.pipeline_pop()
                assert np.array_equal(device_ptr_as_tensor(f32, (240, 416, 4), "<f4", "cuda:0").cpu().numpy(), ref)
        plugin.synchronize()
        assert plugin.learning_counters()["early_frames"] - e0 == 3
    finally:
        plugin.set_debug_flags(0)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.reset_adaptive_state()
    h.free()


def test_two_alternating_kinds_that_never_run_clean_are_settled_on_too(plugin):
    """Round 5's advisor: the bound counted early frames of one kind "in a row", so a host that alternates two kinds which
    never run clean (two cameras with different viewports, every frame re-run) reset the streak with every frame and was
    completed inside bgs_render for ever. Counted per kind: LEARN_MAX early frames of EACH kind, none after."""
    c = random_gaussians_3d_seeded(40_000, 76)
    h = plugin.upload(c)
    va, vb, s = View.headless(416, 240), View.headless(384, 224), CloudSettings()
    try:
        plugin.reset_adaptive_state()
        plugin.set_debug_flags(0x8000000)   # every frame is re-run: no frame of either kind ever runs clean
        plugin.set_async(True)
        plugin.set_pipeline_depth(4)
        e0 = plugin.learning_counters()["early_frames"]
        for i in range(24):
            plugin.render(h, va if i % 2 == 0 else vb, s, download=False)
            if plugin.frames_in_flight() >= 4:
                plugin.pipeline_pop()
        plugin.synchronize()
        assert plugin.learning_counters()["early_frames"] - e0 == 6
    finally:
        plugin.set_debug_flags(0)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.reset_adaptive_state()
    h.free()


def test_a_kind_whose_saturating_tiles_hold_the_work_runs_the_midround_exit_rasteriser(plugin):
    """Round 6: tiles leave bit 15 in their cost word when every pixel went opaque before their list ended; tile_order_kernel
    sums the work of those tiles and of all tiles while it makes the order, the next frame's clean-up block hands the share
    to the host, and kinds at >= 30 % (trained-like 1 M: 36 %) run the rasteriser instantiation that looks for saturation
    inside a staging round — with frames in flight only (alone on the chip the launch ends with its longest lists, which do
    not saturate). Leaving early changes no bit: the frame equals the one with that instantiation switched off (debug flag
    0x1000000). The scene-like frame (no tile saturates) stays on the plain instantiation."""
    from bevy_gaussian_splatting_amd import trained_like_gaussians_3d_seeded
    v = View.headless(1920, 1080)
    for what in ("trained", "scene"):
        c = trained_like_gaussians_3d_seeded(1_000_000, 7) if what == "trained" else random_gaussians_3d_seeded(1_000_000, 2)
        s = CloudSettings() if what == "trained" else CloudSettings(global_scale=0.05)
        h = plugin.upload(c)
        try:
            plugin.reset_adaptive_state()
            plugin.set_debug_flags(0x1000000)
            want = plugin.render(h, v, s)
            plugin.set_debug_flags(0)
            plugin.reset_adaptive_state()
            plugin.set_async(True)
            plugin.set_pipeline_depth(4)
            for i in range(40):
                plugin.render(h, v, s, download=False)
                if plugin.frames_in_flight() >= 4:
                    plugin.pipeline_pop()
            got = plugin.render(h, v, s)
            ts = plugin.stats()["tile_saturation"]
            assert ts["known"], what
            if what == "trained":
                assert ts["work_share"] >= 0.30 and ts["midround_exit"], ts
            else:
                assert ts["work_share"] <= 0.15 and not ts["midround_exit"], ts
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), what
            plugin.synchronize()
            plugin.set_pipeline_depth(1)           # one frame at a time: the plain instantiation, whatever the share
            for i in range(3):
                got = plugin.render(h, v, s)
            assert not plugin.stats()["tile_saturation"]["midround_exit"], what
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), what
        finally:
            plugin.set_debug_flags(0)
            plugin.set_async(False)
            plugin.set_pipeline_depth(1)
            plugin.reset_adaptive_state()
            h.free()


def test_alternating_kinds_keep_their_own_rasteriser_choice(plugin):
    """Two kinds of frame of ONE cloud alternate on the same lanes (a host with two cameras): the trained-like cloud at its own
    splat sizes (saturating tiles hold a third of the work: mid-round exit) and at global_scale 0.05 (nothing saturates:
    the plain instantiation). A lane's tile order is made of the costs of its previous frame — of the OTHER kind here —, so
    the saturation share a frame reports belongs to that frame's kind, not to its own: each kind must settle on its own
    choice, and every frame must equal the blocking frame of its inputs bit for bit."""
    from bevy_gaussian_splatting_amd import trained_like_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor
    c = trained_like_gaussians_3d_seeded(1_000_000, 7)
    h = plugin.upload(c)
    v = View.headless(1920, 1080)
    kinds = [CloudSettings(), CloudSettings(global_scale=0.05)]
    try:
        plugin.reset_adaptive_state()
        refs = [plugin.render(h, v, s) for s in kinds]
        plugin.reset_adaptive_state()
        plugin.set_async(True)
        plugin.set_pipeline_depth(3)       # (odd: every lane sees both kinds in turn)
        for i in range(72):
            k = i % 2
            plugin.render(h, v, kinds[k], download=False)
            while plugin.frames_in_flight() > 2:
                plugin.pipeline_pop()
        # drain, comparing the last frame of each kind
        last = {}
        order = [(72 - plugin.frames_in_flight() + j) % 2 for j in range(plugin.frames_in_flight())]
        for k in order:
            f32, _ = plugin.pipeline_pop()
            last[k] = device_ptr_as_tensor(f32, (1080, 1920, 4), "<f4", "cuda:0").cpu().numpy().copy()
        for k in (0, 1):
            assert k in last and np.array_equal(last[k].view(np.uint32), refs[k].view(np.uint32)), k
        # the steady state, three more pairs of frames completed one by one: each kind on its own rasteriser
        for rep in range(3):
            for k in (0, 1):
                got = plugin.render(h, v, kinds[k])
                assert np.array_equal(got.view(np.uint32), refs[k].view(np.uint32)), (rep, k)
                ts = plugin.stats()["tile_saturation"]
                if ts["known"]:
                    assert ts["midround_exit"] == (k == 0), (rep, k, ts)
    finally:
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.reset_adaptive_state()
        h.free()


def test_rerun_keeps_the_output_state_the_frame_was_enqueued_with(plugin):
    """A frame whose supertile lists overflow is re-run when its lane completes (here: forced). If the caller changed the packed
    output format in between (bgs_set_output_rgba16f / _srgb8 / _packed_only complete nothing), the re-run must still
    produce what the frame was ENQUEUED for: an Rgba8UnormSrgb image of w*h*4 bytes in the caller's target — not
    8 B per pixel of Rgba16Float written past its end."""
    import torch
    c = random_gaussians_3d_seeded(120_000, 19)
    v = View.headless(640, 360)
    s = CloudSettings()
    h = plugin.upload(c)
    plugin.reset_adaptive_state()
    plugin.set_output_srgb8(True)
    plugin.render(h, v, s)
    from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor
    p8, _ = plugin.framebuffer_srgb8_device_ptr()
    want8 = device_ptr_as_tensor(p8, (360, 640, 4), "|u1", "cuda:0").cpu().numpy().copy()
    assert want8.any()
    plugin.reset_adaptive_state()
    for _ in range(3):                            # (a context that has learnt nothing completes its async frames at once)
        plugin.render(h, v, s)
    target = torch.zeros((2, 360, 640, 4), dtype=torch.uint8, device="cuda:0")   # second half = guard zone
    try:
        plugin.set_debug_flags(0x8000000)         # this frame WILL be re-run when its lane is completed
        plugin.set_async(True)
        plugin.set_pipeline_depth(2)
        plugin.set_srgb8_target(target[0].data_ptr())
        plugin.render(h, v, s, download=False)
        # while it is in flight: another format, packed-only, and the flag that made it overflow cleared
        plugin.set_output_srgb8(False)
        plugin.set_output_rgba16f(True)
        plugin.set_packed_only(True)
        plugin.set_debug_flags(0)
        plugin.synchronize()
        assert plugin.stats()["regrow_count"] >= 1
        got = target.cpu().numpy()
        assert np.array_equal(got[0], want8)
        assert not got[1].any()                   # nothing was written past the 4 B/px image
    finally:
        plugin.set_debug_flags(0)
        plugin.set_packed_only(False)
        plugin.set_output_rgba16f(False)
        plugin.set_output_srgb8(False)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
        plugin.reset_adaptive_state()
    h.free()

# ---------------------------------------------------------------------------------------------
# randomized sweep over camera / transform / settings combinations
# ---------------------------------------------------------------------------------------------
_SEED_BASE = int(os.environ.get("BGS_RANDOM_SEED_BASE", "0"))


@pytest.mark.parametrize("seed", range(_SEED_BASE, _SEED_BASE + int(os.environ.get("BGS_RANDOM_SEEDS", "24"))))
def test_randomized_configurations(plugin, oracle, seed):
    """BGS_RANDOM_SEEDS=N widens the sweep (a 2000-seed run is part of the round's evidence, profiles/README.md);
    BGS_RANDOM_SEED_BASE=B moves it to seeds B .. B+N-1 (exploratory sweeps over configurations not seen before)."""
    c, v, s = H.random_case(seed)
    cloud = c.to_f16() if seed % 4 == 3 else c
    cd = oracle.decode_f16(cloud) if cloud is not c else c
    plugin.set_binning("sort" if seed % 6 == 5 else "scan")
    h = plugin.upload(cloud)
    with _scene_depth(plugin, v):
        got = plugin.render(h, v, s)
    gs = plugin.sort(h, v, s)
    plugin.set_binning("scan")
    e = oracle.sort(cd, v, s)
    assert np.array_equal(gs["key"], e["key"]) and np.array_equal(gs["index"], e["index"])
    ref, amb = oracle.render(cd, e, v, s, with_ambiguity=True, depth=v.depth_host)
    _assert_image(ref, got, amb, frac_slack=0.01, what=f"seed {seed}: x{v.msaa_samples} depth={v.depth_host is not None} {s}",
                  overlay=s.visualize_bounding_box)
    h.free()


def _medium_seeds():
    """12 seeds by default plus seed 321 — round 2's one parity failure (an ill-conditioned 2DGS degeneracy decision that
    a 1-ulp difference in ln(opacity) flipped; since round 3 the log is correctly rounded on every side), kept in the
    default suite for good. BGS_RANDOM_MEDIUM_SEEDS=N widens the range (evidence runs: 350)."""
    n = int(os.environ.get("BGS_RANDOM_MEDIUM_SEEDS", "12"))
    seeds = list(range(_SEED_BASE, _SEED_BASE + n))
    return seeds + ([321] if 321 not in seeds else [])


@pytest.mark.parametrize("seed", _medium_seeds())
def test_randomized_configurations_medium(plugin, oracle, seed):
    """The same sweep at 40-250 k splats and up to 1280x720: many tiles and supertiles, ticket loops, both
    supertile rules (BGS_RANDOM_MEDIUM_SEEDS=N widens it; a 60-seed run is in profiles/)."""
    c, v, s = H.random_case(1000 + seed, medium=True)
    if os.environ.get("BGS_RANDOM_FORCE_SURFEL"):   # every seed on the 2DGS surfel (aabb) fragment path
        s.aabb, s.gaussian_mode = True, GaussianMode.Gaussian2d
    if s.global_scale > 1.0:
        s.global_scale = 0.3     # keeps the f64 oracle raster of a 1 Mpx frame in seconds
    cloud = c.to_f16() if seed % 4 == 3 else c
    cd = oracle.decode_f16(cloud) if cloud is not c else c
    plugin.set_binning("sort" if seed % 6 == 5 else "scan")
    h = plugin.upload(cloud)
    with _scene_depth(plugin, v):
        got = plugin.render(h, v, s)
        # round 6: the rasteriser's three instantiations (plain; mid-round exit for dense frames; mid-round exit with the sparse
        # frames' strip masks and staged sample offsets) differ in when a tile stops and in how a record reaches a strip, never
        # in a bit: forced on at whatever supertile level the frame has (0x20000) and forced off (0x1000000)
        if seed % 6 != 5:
            try:
                for flags in (0x20000, 0x1000000):
                    plugin.set_debug_flags(flags)
                    assert np.array_equal(plugin.render(h, v, s).view(np.uint32), got.view(np.uint32)), (seed, hex(flags))
            finally:
                plugin.set_debug_flags(0)
    gs = plugin.sort(h, v, s)
    plugin.set_binning("scan")
    e = oracle.sort(cd, v, s)
    assert np.array_equal(gs["key"], e["key"]) and np.array_equal(gs["index"], e["index"])
    ref, amb = oracle.render(cd, e, v, s, with_ambiguity=True, depth=v.depth_host)
    _assert_image(ref, got, amb, frac_slack=0.01, what=f"medium seed {seed}: x{v.msaa_samples} depth={v.depth_host is not None} {s}",
                  overlay=s.visualize_bounding_box)
    h.free()


def test_maximum_viewport_4096_square(plugin, oracle):
    """Largest target the ABI accepts: 256 x 256 tiles, 16 x 16 = 256 supertiles (every thread of the
    binning block owns one), 16.7 M pixels. Oracle-checked crops at three places incl. the far corner."""
    c = random_gaussians_3d_seeded(150_000, 51)
    v = View.headless(4096, 4096)
    for kw in ({"global_scale": 0.2}, {"global_scale": 0.2, "aabb": True}):
        s = CloudSettings(**kw)
        h = plugin.upload(c)
        got = plugin.render(h, v, s)
        assert got.shape == (4096, 4096, 4) and np.isfinite(got).all()
        e = oracle.sort(c, v, s)
        for (x0, y0) in ((0, 0), (2040, 2040), (4096 - 40, 4096 - 40)):
            win = (x0, y0, x0 + 40, y0 + 40)
            ref, amb = oracle.render(c, e, v, s, window=win, with_ambiguity=True)
            _assert_image(ref, got[y0:y0 + 40, x0:x0 + 40], amb, frac_slack=0.01, what=f"4096^2 {kw} {win}")
        h.free()
    with pytest.raises(Exception):
        plugin.render(plugin.upload(random_gaussians_3d_seeded(10, 1)), View.headless(4097, 64), CloudSettings())


def test_draw_modes(plugin, oracle, binning):
    """DrawMode::Selected / HighlightSelected (src/render/gaussian.wgsl:203-205, 423-427)."""
    from bevy_gaussian_splatting_amd import DrawMode
    c = random_gaussians_3d_seeded(8000, 29)
    c.position_visibility[:, 3] = (np.arange(len(c)) % 3 == 0).astype(np.float32)
    v = View.headless(192, 108)
    for dm in (DrawMode.Selected, DrawMode.HighlightSelected):
        for kw in ({}, {"aabb": True}, {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True}):
            s = CloudSettings(draw_mode=dm, **kw)
            for cloud in (c, c.to_f16()) if not kw else (c,):
                h = plugin.upload(cloud)
                got = plugin.render(h, v, s)
                cd = oracle.decode_f16(cloud) if cloud is not c else c
                e = oracle.sort(cd, v, s)
                ref, amb = oracle.render(cd, e, v, s, with_ambiguity=True)
                _assert_image(ref, got, amb, what=f"{dm.name} {kw}")
                vis, _ = oracle.instance_stats(cd, e, v, s)
                assert plugin.stats()["visible_count"] == vis
                h.free()


def test_zz_report_ambiguity_slack_use(oracle):
    """LAST test of the run (tests/conftest.py moves it behind every other collected test; run with -s to see it). Over
    every oracle comparison of this run: how many values were accepted only through the oracle's per-pixel ambiguity bound
    (quad-edge coverage flips, ill-conditioned surfel intersections / AABB conics) rather than the plain
    1e-3 + 1e-4 |ref| tolerance, and by how much the worst of them exceeds that tolerance. Teeth (from the runs under
    profiles/r5_v1/, profiles/r5/ and profiles/r5_v2/, at both edge bands; the default band is 5e-4 px since round 6): at
    most 2e-5 of the values; on the FIXED configurations (whole frames, variants, depth buffers, f16, ...) none more than
    0.025 beyond (largest seen: 0.0199) — one sample's share (a quarter) of a fragment
    of alpha exp(-4.5) * 0.8 and a colour of magnitude 15, the brightest the synthetic clouds hold, is 0.033; on the
    RANDOMIZED configurations (Msaa::Off on a third of the seeds: a flip is a whole fragment; global_opacity up to 2;
    every raster mode) none more than 0.24 (largest seen in 41 000 configurations: 0.19); under the bounding-box overlay,
    where a flip is one sample's share of a whole opaque fragment and of what it hides, none more than
    max(1, the frame's largest value) (largest seen: 1.45 on a frame that reaches 4.76)."""
    t = H.TOLERANCE
    v, n = t["values"], t["checked"]
    print(f"[tolerance accounting] edge band {oracle.lib().oracle_edge_band_px():g} px: {v} of {n} compared values "
          f"({100.0 * v / max(n, 1):.5f} %) beyond 1e-3 + 1e-4 |ref|, largest excess {t['max_excess']:.3e} on the fixed "
          f"configurations, {t['max_excess_randomized']:.3e} on the randomized ones "
          f"(frames with the bounding-box overlay, where a flip is a whole opaque fragment: {t['max_excess_overlay']:.3e}, "
          f"{t['max_excess_overlay_rel']:.3f} of max(1, the frame's largest value))")
    for rec in sorted(t["comparisons"], key=lambda r: -r["max_excess"])[:12]:
        print(f"    {rec['what'][:60]}: {rec['beyond_strict']} of {rec['values']}, excess {rec['max_excess']:.2e}, max |err| {rec['max_err']:.2e}, "
              f"max |ref| {rec['ref_absmax']:.2f}")
    # (round 6: the ceilings are what five rounds of runs showed plus a quarter — 1.99e-2 on the fixed configurations,
    # 0.19 on the randomized ones; 0.05 / 0.25 before)
    assert v <= 2e-5 * max(n, 1) + 50
    assert t["max_excess"] <= 0.025
    assert t["max_excess_randomized"] <= 0.24
    # under the overlay a flip swaps a splat's fragment for the frame's opaque (0.3, 1, 0.1, 1) — which also hides, or reveals,
    # everything behind it: one sample's share (0.25 at four samples per pixel, 0.5 at two, 1 at Msaa::Off) of
    # max(1, the frame's largest value). Seen: 0.25 / 0.45 / 0.50 / 0.62 in 300 forced-surfel configurations never run
    # before, identical with and without round 6's kernels; 1.45 on a frame whose colours reach 4.76 (medium seed 501815,
    # 4 samples: 0.30 of that bound) and 0.9-1.0 on Msaa::Off frames with colours <= 1 among 2000 + 800 further ones
    # (profiles/r6_v3/explore_500000). The bound is that product, with the share left at 1
    assert t["max_excess_overlay_rel"] <= 1.0


# ---------------------------------------------------------------------------------------------
# multisampling (CloudPipelineKey.sample_count) and the depth test against the view's depth buffer
# (src/render/mod.rs:357-424, 959-979)
# ---------------------------------------------------------------------------------------------
_MS_VARIANTS = {"obb3d": {}, "aabb3d": {"aabb": True}, "obb2d": {"gaussian_mode": GaussianMode.Gaussian2d},
                "surfel": {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True}}


@pytest.mark.parametrize("variant", sorted(_MS_VARIANTS))
@pytest.mark.parametrize("global_scale", [1.0, 0.1])
def test_multisampled_target_matches_the_oracle(plugin, oracle, binning, variant, global_scale):
    """Msaa::Sample4 (Bevy's default, what the reference's cameras render with) and Msaa::Off, every rasteriser variant,
    both binning pipelines, large and small splats: coverage per sample at the standard 4x positions, one shading per
    pixel at its centre, box resolve — against the oracle's multisampled target on every pixel. The two sample counts
    are different images (quad edges), and the single-sampled one is what rounds 1-3 produced."""
    c = random_gaussians_3d_seeded(60_000, 33)
    s = CloudSettings(global_scale=global_scale, **_MS_VARIANTS[variant])
    h = plugin.upload(c)
    imgs = {}
    for samples in (4, 1, 2, 8):     # Msaa::Sample4 (Bevy's default), Off, Sample2, Sample8
        v = View.headless(640, 360, msaa_samples=samples)
        got = plugin.render(h, v, s)
        e = oracle.sort(c, v, s)
        ref, amb = oracle.render(c, e, v, s, with_ambiguity=True)
        _assert_image(ref, got, amb, frac_slack=0.005, what=f"{variant} gs={global_scale} x{samples} {binning}")
        imgs[samples] = got
    d = np.abs(imgs[4] - imgs[1])
    # (a dense frame saturates within a few large splats: its quad edges are worth ~1e-3; small splats show them)
    assert d.max() > (2e-3 if global_scale < 1.0 else 2e-4), "4x and 1x must differ at quad edges"
    for a, b in ((2, 1), (2, 4), (8, 4)):
        assert np.abs(imgs[a] - imgs[b]).max() > (1e-3 if global_scale < 1.0 else 1e-4), f"{a}x and {b}x must differ at quad edges"
    print(f"[msaa {variant} gs={global_scale} {binning}] |4x - 1x|: mean {d.mean():.2e} max {d.max():.2e}, "
          f"values apart by more than 1e-3: {(d > 1e-3).mean():.2%}")
    h.free()


def test_sample_count_and_depth_pointer_are_validated(plugin):
    c = random_gaussians_3d_seeded(100, 1)
    h = plugin.upload(c)
    for bad in (3, 5, 16, 64):
        with pytest.raises(Exception) as ei:
            plugin.render(h, View.headless(64, 64, msaa_samples=bad), CloudSettings())
        assert "sample_count" in str(ei.value)
    # 0 = "not set" (a zero-initialised bgs_view): Msaa::default() = Sample4, bit for bit
    assert np.array_equal(plugin.render(h, View.headless(64, 64, msaa_samples=0), CloudSettings()),
                          plugin.render(h, View.headless(64, 64, msaa_samples=4), CloudSettings()))
    # the per-tile trace has no instantiation with a depth buffer: refused, not silently untraced
    vz = View.headless(64, 64, msaa_samples=4)
    pz = plugin.device_alloc(64 * 64 * 16)
    tr = plugin.device_alloc(4 * 4 * 32)
    vz.depth_device_ptr = pz
    plugin.set_tile_trace(tr)
    try:
        with pytest.raises(Exception) as ei:
            plugin.render(h, vz, CloudSettings())
        assert "trace" in str(ei.value)
    finally:
        plugin.set_tile_trace(None)
    plugin.device_free(tr)
    plugin.device_free(pz)
    v = View.headless(64, 64, msaa_samples=4)
    p = plugin.device_alloc(64 * 64 * 16 + 64)
    v.depth_device_ptr = p + 4                      # not aligned to one pixel's four samples
    with pytest.raises(Exception) as ei:
        plugin.render(h, v, CloudSettings())
    assert "depth_device_ptr" in str(ei.value)
    plugin.device_free(p)
    plugin.sort(h, View.headless(64, 64, msaa_samples=3), CloudSettings())   # the sort does not look at the samples
    h.free()


@pytest.mark.parametrize("variant", sorted(_MS_VARIANTS))
@pytest.mark.parametrize("samples", [1, 4, 8])
@pytest.mark.parametrize("depth", [False, True])
def test_bounding_box_overlay_matches_the_oracle(plugin, oracle, binning, variant, samples, depth):
    """CloudSettings::visualize_bounding_box (src/gaussian/settings.rs:95,117; pipeline key bit src/render/mod.rs:418,824;
    src/render/gaussian.wgsl:486-495): a fragment in the outer 8 % of its quad's uv square is (0.3, 1, 0.1, 1) — the
    quads' frames, opaque, over the splats. Every rasteriser variant, both binnings, 1 / 4 / 8 samples per pixel, with
    and without a scene depth buffer, against the oracle on every pixel; the overlay changes the image."""
    c = random_gaussians_3d_seeded(30_000, 41)
    v = View.headless(480, 270, msaa_samples=samples)
    v.clear_color = (0.05, 0.1, 0.2, 1.0)
    h = plugin.upload(c)
    # (global_scale 0.02: sub-pixel quads. A sample can be covered while the pixel centre the fragment is shaded at lies
    # several quad widths outside: fs_main's OBB discard, dot(uv, uv) > 9, fires BEFORE the overlay's test — found by the
    # round-5 exploratory sweep, seeds 70137 / 71390: one sample's share of an opaque frame fragment the reference drops)
    for gs in (0.25, 0.02):
        kw = dict(global_scale=gs, **_MS_VARIANTS[variant])
        s = CloudSettings(visualize_bounding_box=True, **kw)
        dhost, dptr = None, None
        if depth:
            dhost = H.random_depth_buffer(c, v, s, np.random.default_rng(5))
            dptr = plugin.upload_depth(dhost)
            v.depth_device_ptr = dptr
        try:
            got = plugin.render(h, v, s)
            plain = plugin.render(h, v, CloudSettings(**kw))
        finally:
            if dptr is not None:
                plugin.device_free(dptr)
                v.depth_device_ptr = 0
        e = oracle.sort(c, v, s)
        ref, amb = oracle.render(c, e, v, s, with_ambiguity=True, depth=dhost)
        _assert_image(ref, got, amb, frac_slack=0.02, what=f"bbox {variant} gs={gs} x{samples} depth={depth} {binning}", overlay=True)
        assert np.abs(got - plain).max() > 0.2
        # the frame colour really is there: pixels that are (nearly) pure (0.3, 1, 0.1) with alpha 1
        frame_px = (np.abs(got[..., :3] - np.array([0.3, 1.0, 0.1], np.float32)).max(axis=2) < 1e-3) & (np.abs(got[..., 3] - 1.0) < 1e-3)
        assert frame_px.sum() > (50 if gs > 0.1 else 5)
    h.free()


@pytest.mark.parametrize("variant", sorted(_MS_VARIANTS))
@pytest.mark.parametrize("samples", [1, 2, 4, 8])
def test_depth_buffer_occludes_splats(plugin, oracle, binning, variant, samples):
    """The view's depth attachment (Depth32Float, reverse-Z, GreaterEqual, no write: src/render/mod.rs:959-974) as a device
    buffer: a tilted, per-sample jittered plane through the middle of the cloud — tiles wholly in front of it, wholly
    behind it, and tiles whose pixels and samples disagree — against the oracle; a buffer of zeros changes no bit; a
    buffer of ones (the near plane) hides everything."""
    c = random_gaussians_3d_seeded(40_000, 35)
    v = View.headless(480, 270, msaa_samples=samples)
    v.clear_color = (0.1, 0.2, 0.3, 0.5)
    s = CloudSettings(global_scale=0.3, **_MS_VARIANTS[variant])
    h = plugin.upload(c)
    plain = plugin.render(h, v, s)
    e = oracle.sort(c, v, s)
    v.depth_host = H.random_depth_buffer(c, v, s, np.random.default_rng(5))
    with _scene_depth(plugin, v):
        got = plugin.render(h, v, s)
    ref, amb = oracle.render(c, e, v, s, with_ambiguity=True, depth=v.depth_host)
    _assert_image(ref, got, amb, frac_slack=0.005, what=f"depth {variant} x{samples} {binning}")
    assert np.abs(got - plain).max() > 0.05                      # the plane really cuts splats away
    assert np.array_equal(got[:64, :100], plain[:64, :100])      # ... and the corner without an occluder is untouched
    for value, expect_plain in ((0.0, True), (1.0, False)):
        v.depth_host = np.full((270, 480, samples), value, np.float32)
        with _scene_depth(plugin, v):
            g2 = plugin.render(h, v, s)
        if expect_plain:
            assert np.array_equal(g2, plain)
        else:
            assert np.allclose(g2, np.asarray(v.clear_color, np.float32))
    v.depth_host = None
    h.free()


def test_depth_buffer_with_frames_in_flight_and_a_rerun(plugin, oracle):
    """The depth pointer travels with the frame (FrameParams): pipelined frames with different depth buffers, and a
    frame that is re-run on its lane (forced: debug flag 0x8000000), each see their own."""
    from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor
    c = random_gaussians_3d_seeded(30_000, 36)
    s = CloudSettings(global_scale=0.3)
    h = plugin.upload(c)
    v0 = View.headless(320, 180)
    blocking = []
    bufs = []
    for k in range(4):
        d = H.random_depth_buffer(c, v0, s, np.random.default_rng(50 + k)) if k else None
        bufs.append(plugin.upload_depth(d) if d is not None else 0)
    views = []
    for k in range(4):
        vk = View.headless(320, 180)
        vk.depth_device_ptr = bufs[k]
        views.append(vk)
        blocking.append(plugin.render(h, vk, s))
    assert not np.array_equal(blocking[0], blocking[1]) and not np.array_equal(blocking[1], blocking[2])
    try:
        plugin.set_async(True)
        plugin.set_pipeline_depth(4)
        for flags in (0, 0x8000000):
            plugin.set_debug_flags(flags)
            for vk in views:
                plugin.render(h, vk, s, download=False)
            for k in range(4):
                f32, _ = plugin.pipeline_pop()
                got = device_ptr_as_tensor(f32, (180, 320, 4), "<f4", "cuda:0").cpu().numpy()
                assert np.array_equal(got, blocking[k]), (flags, k)
    finally:
        plugin.set_debug_flags(0)
        plugin.set_async(False)
        plugin.set_pipeline_depth(1)
    for p in bufs:
        if p:
            plugin.device_free(p)
    h.free()


# ---------------------------------------------------------------------------------------------
# configs[4] at size on one GPU: the eight cameras of the multi-GPU configuration (camera g = the headless camera
# yawed g * 45 degrees, one per GPU there), each at 1920x1080 on the 1 M-splat cloud
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("g", range(8))
def test_config4_camera_at_full_size(plugin, oracle, cloud_1m, g):
    """Sort bit-exact for every camera; the whole 1080p frame against the oracle for g in {1, 3, 6}, crops (centre,
    corner, the heavy bottom-right) for the others."""
    from bevy_gaussian_splatting_amd.multiview import headless_view
    v = headless_view(g, 1920, 1080)
    s = CloudSettings()
    h = plugin.upload(cloud_1m)
    gs = plugin.sort(h, v, s)
    e = oracle.sort(cloud_1m, v, s)
    assert np.array_equal(gs["key"], e["key"]) and np.array_equal(gs["index"], e["index"])
    if g in (1, 3, 6):
        _whole_frame_parity(plugin, oracle, cloud_1m, h, v, s, f"cfg4 camera {g} (yaw {45 * g} deg)")
    else:
        got = plugin.render(h, v, s)
        for (x0, y0) in ((936, 516), (0, 0), (1860, 1020)):
            win = (x0, y0, x0 + 48, y0 + 48)
            ref, amb = oracle.render(cloud_1m, e, v, s, window=win, with_ambiguity=True)
            _assert_image(ref, got[y0:y0 + 48, x0:x0 + 48], amb, frac_slack=0.01, what=f"cfg4 camera {g} {win}")
    h.free()


def _nccl_gather_worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["BGS_QUEUE_HOLDERS"] = "0"
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.multiview import BatchedFrameGather, headless_view
    W, Hh, frames, batch = 640, 360, 11, 4
    cloud = random_gaussians_3d_seeded(100_000, 2)
    got = []
    with GaussianSplattingPlugin(rank) as p:
        h = p.upload(cloud)
        v, s = headless_view(rank, W, Hh), CloudSettings()
        p.set_output_srgb8(True)
        p.render(h, v, s)                                              # this rank's own frame, blocking: the reference
        from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor
        ptr, _ = p.framebuffer_srgb8_device_ptr()
        own = device_ptr_as_tensor(ptr, (Hh, W, 4), "|u1", f"cuda:{rank}").cpu().numpy().copy()
        np.save(os.path.join(outdir, f"own_{rank}.npy"), own)
        bg = BatchedFrameGather((Hh, W, 4), torch.uint8, f"cuda:{rank}", batch=batch,
                                on_batch=lambda per_rank: got.append([t.cpu().numpy().copy() for t in per_rank]))
        p.set_async(True)
        p.set_pipeline_depth(4)
        p.set_packed_only(True)
        for i in range(frames):
            p.set_srgb8_target(bg.next_target().data_ptr())
            p.render(h, v, s, download=False)
            if p.frames_in_flight() >= 4:
                p.pipeline_pop()
                bg.frame_completed()
        while p.frames_in_flight():
            p.pipeline_pop()
            bg.frame_completed()
        bg.flush()
        if rank == 0:
            np.save(os.path.join(outdir, "received.npy"), np.array([bg.frames_received]))
            for r in range(world):
                np.save(os.path.join(outdir, f"gathered_{r}.npy"), np.concatenate([b[r] for b in got]))
        dist.barrier()
        h.free()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_nccl_frame_gather_worker_with_one_rank(tmp_path):
    """The worker of the two-device test below on the ONE device every box has (a process group of one rank on RCCL,
    the same zero-copy staged batches, pops and flush): everything but the second rank, so that the two-rank test
    cannot fail on a multi-GPU box for a reason a one-GPU box could have found."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_gather_worker, args=(0, 1, port, str(tmp_path)))
    p.start()
    p.join(500)
    assert p.exitcode == 0
    assert int(np.load(tmp_path / "received.npy")[0]) == 11
    own, seq = np.load(tmp_path / "own_0.npy"), np.load(tmp_path / "gathered_0.npy")
    assert seq.shape == (11, 360, 640, 4) and own.any()
    for f in seq:
        assert np.array_equal(f, own)


@pytest.mark.timeout(600)
def test_two_rank_nccl_frame_gather_end_to_end(tmp_path):
    """The N > 1 path of bench.py on real devices (needs two): two processes, one GPU and one camera each, RCCL gather of
    the Rgba8UnormSrgb frames through BatchedFrameGather (zero-copy: every frame rendered into its slot of the staging
    batch, packed-only); rank 0 must hold, for every rank, exactly the bytes that rank rendered on its own."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (the driver's multi-GPU box)")
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_nccl_gather_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
        assert p.exitcode == 0
    assert int(np.load(tmp_path / "received.npy")[0]) == 2 * 11
    for r in range(2):
        own = np.load(tmp_path / f"own_{r}.npy")
        seq = np.load(tmp_path / f"gathered_{r}.npy")
        assert seq.shape == (11, 360, 640, 4)
        for f in seq:
            assert np.array_equal(f, own), f"rank {r}: a gathered frame differs from the frame the rank rendered"
    assert not np.array_equal(np.load(tmp_path / "own_0.npy"), np.load(tmp_path / "own_1.npy"))
