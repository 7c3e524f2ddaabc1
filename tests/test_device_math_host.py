"""CPU pre-flight of the PRODUCT's per-splat arithmetic: bevy_gaussian_splatting_amd/csrc/
splat_math.h is compiled with g++ (tests/host_shim) and compared with the oracle. This catches
transcription slips in the device code without a GPU; it is not a product path (libbgs has no
host execution of these functions)."""
import numpy as np
import pytest

import helpers as H
from bevy_gaussian_splatting_amd import (
    CloudSettings, GaussianColorSpace, GaussianMode, RadixSortDepthBits, SortMode, View,
    random_gaussians_3d_seeded, transform_from, rotation_y)


@pytest.mark.parametrize("mode", [SortMode.Radix, SortMode.Rayon, SortMode.Std, SortMode.NONE])
@pytest.mark.parametrize("bits", [16, 24, 32])
def test_device_keys_and_order_are_bit_exact(oracle, mode, bits):
    c = random_gaussians_3d_seeded(60000, 7)
    c.position_visibility[:4, :3] = [[0, 1.5, 5], [np.nan, 0, 0], [np.inf, 0, 0], [0, 1.5, 4.9]]
    v = View.headless(640, 360, yaw=0.3)
    s = CloudSettings(sort_mode=mode, radix_sort_depth_bits=RadixSortDepthBits(bits),
                      transform=transform_from((0.5, -0.25, 1.0), rotation_y(0.2)))
    ref = oracle.sort(c, v, s)
    got = H.device_sorted_entries(c, v, s)
    assert np.array_equal(ref["key"], got["key"])
    assert np.array_equal(ref["index"], got["index"])


SCENES = {
    "obb3d": {},
    "aabb3d": {"aabb": True},
    "obb3d_fixed_radius_linear": {"opacity_adaptive_radius": False,
                                  "color_space": GaussianColorSpace.LinRec709Display, "global_scale": 0.3},
    "obb3d_sh1_opacity2": {"sh_degree": 1, "global_opacity": 2.0},
    "obb2d": {"gaussian_mode": GaussianMode.Gaussian2d},
    "aabb2d": {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True},
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_device_records_render_like_the_oracle(oracle, name):
    c = random_gaussians_3d_seeded(2500, 11)
    v = View.headless(160, 96)
    s = CloudSettings(**SCENES[name])
    e = oracle.sort(c, v, s)
    ref, amb = oracle.render(c, e, v, s, with_ambiguity=True)
    got = H.emulate_render(c, v, s)
    ok, err = H.tolerance_mask(ref, got, amb)
    assert ok.all(), f"max err {err.max():.3e}"
    strict, _ = H.tolerance_mask(ref, got, None)
    assert (~strict).sum() <= 0.002 * strict.size  # ambiguity slack is used by (almost) no pixel


def test_reference_tool_scenes(oracle):
    cases = [
        (H.visibility_test_cloud(), View.perspective(transform_from((0, 0, 5)), 128, 128),
         CloudSettings(sort_mode=SortMode.NONE, global_opacity=2.0, opacity_adaptive_radius=False)),
        (H.surfel_plane_cloud(), View.perspective(transform_from((0, 1.5, 20)), 192, 108),
         CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, transform=transform_from((5.0, 5.0, 0.0)))),
        (H.aabb_obb_pair_cloud(), View.headless(192, 108), CloudSettings(aabb=True)),
        (H.aabb_obb_pair_cloud(), View.headless(192, 108), CloudSettings()),
    ]
    for c, v, s in cases:
        e = oracle.sort(c, v, s)
        ref, amb = oracle.render(c, e, v, s, with_ambiguity=True)
        got = H.emulate_render(c, v, s)
        ok, err = H.tolerance_mask(ref, got, amb)
        assert ok.all(), f"max err {err.max():.3e}"


def test_tile_rect_covers_every_pixel_the_oracle_draws(oracle):
    """The conservative tile rectangle of project_splat must contain every pixel the oracle's
    exact coverage test accepts (otherwise binning would drop contributions)."""
    import ctypes
    c = random_gaussians_3d_seeded(4000, 13)
    v = View.headless(320, 192)
    for kw in ({}, {"aabb": True}, {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True}):
        s = CloudSettings(**kw)
        fpc = H.frame_params(len(c), v, s)
        e = oracle.sort(c, v, s)
        out = H.ShimOut()
        checked = 0
        for ent in e[: int((e["key"] != 0xFFFFFFFF).sum())][::7]:
            si = int(ent["index"])
            H.shim().shim_project(ctypes.byref(fpc), int(ent["key"]), H._fp(c.position_visibility[si]),
                                  H._fp(c.rotation[si]), H._fp(c.scale_opacity[si]),
                                  H._fp(c.spherical_harmonic[si]), H._fp(np.zeros(2, np.float32)),
                                  ctypes.byref(out))
            one = np.array([ent], dtype=e.dtype)
            vclear = View(v.world_from_view, v.view_from_world, v.clip_from_view, v.clip_from_world,
                          v.viewport, clear_color=(0, 0, 0, 0))
            img = oracle.render(c, one, vclear, s)
            ys, xs = np.nonzero(img[..., 3] != 0)
            if len(xs) == 0:
                continue
            assert out.draw
            assert out.tx0 * 16 <= xs.min() and xs.max() < (out.tx1 + 1) * 16
            assert out.ty0 * 16 <= ys.min() and ys.max() < (out.ty1 + 1) * 16
            checked += 1
        assert checked > 20


# ---------------------------------------------------------------------------------------------
# RasterizeMode colour variants (src/render/gaussian.wgsl:312-405), SURVEY 8(f) item 4
# ---------------------------------------------------------------------------------------------
from bevy_gaussian_splatting_amd import RasterizeMode, compute_aabb, PlanarGaussian3d


def _mode_settings(mode, c, **kw):
    mn, mx = compute_aabb(c)
    return CloudSettings(rasterize_mode=mode, position_min=mn, position_max=mx, num_classes=5, **kw)


def _classified(c):
    c = PlanarGaussian3d(c.position_visibility.copy(), c.spherical_harmonic, c.rotation, c.scale_opacity)
    c.position_visibility[:, 3] = (np.arange(len(c)) % 8).astype(np.float32)  # 0,1: plain colour; 2..7: classes 0..5
    return c


@pytest.mark.parametrize("mode", [RasterizeMode.Classification, RasterizeMode.Depth, RasterizeMode.Normal,
                                  RasterizeMode.Position, RasterizeMode.OpticalFlow])
@pytest.mark.parametrize("kw", [{}, {"aabb": True}, {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True}])
def test_rasterize_modes_device_math_matches_oracle(oracle, mode, kw):
    c = _classified(random_gaussians_3d_seeded(3000, 21))
    v = View.headless(128, 72)
    v.previous_clip_from_world = View.headless(128, 72, yaw=0.003).clip_from_world   # the camera turned a little
    v.delta_time = 1.0 / 120.0
    tr = transform_from((0.5, -0.25, 0.0), rotation_y(0.3))
    tr[:3, :3] *= np.float32(1.25)   # Transform::with_scale
    s = _mode_settings(mode, c, transform=tr, **kw)
    e = oracle.sort(c, v, s)
    ref, amb = oracle.render(c, e, v, s, with_ambiguity=True)
    got = H.emulate_render(c, v, s)
    ok, err = H.tolerance_mask(ref, got, amb)
    assert ok.all(), f"{mode.name} {kw}: max err {err.max():.3e}"
    assert np.abs(ref[..., :3]).max() > 0.05
    # the variants only change the colour: coverage/alpha is the Color image's
    ref_color = oracle.render(c, e, v, CloudSettings(transform=s.transform, **kw))
    assert np.array_equal(ref[..., 3], ref_color[..., 3])
    if mode == RasterizeMode.Classification:
        plain = PlanarGaussian3d(c.position_visibility.copy(), c.spherical_harmonic, c.rotation, c.scale_opacity)
        plain.position_visibility[:, 3] = 1.0   # visibility < 2: class_to_rgb returns the SH colour
        assert np.array_equal(oracle.render(plain, e, v, s), ref_color)


def test_rasterize_mode_closed_forms(oracle):
    """Per-splat colours against float64 closed forms."""
    import ctypes
    n = 5
    pv = np.array([[0, 0, -6, 1], [1, 0.5, -4, 3], [-1, 0.2, -9, 5.5], [0.3, -0.4, -5, 7], [0, 1, -7, 2]], np.float32)
    rot = np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1))
    rot[1] = [0.7071068, 0.7071068, 0, 0]      # 90 deg about x
    so = np.tile(np.array([[0.3, 0.2, 0.1, 0.7]], np.float32), (n, 1))
    sh = np.zeros((n, 48), np.float32)
    c = PlanarGaussian3d(pv, sh, rot, so)
    v = View.perspective(transform_from((0, 0, 0)), 64, 64)
    mn, mx = compute_aabb(c)
    assert np.allclose(mn, pv[:, :3].min(0) - 0.1, atol=1e-6) and np.allclose(mx, pv[:, :3].max(0) + 0.1, atol=1e-6)
    base = dict(position_min=mn, position_max=mx, num_classes=4, sort_mode=SortMode.NONE)
    e = oracle.sort(c, v, CloudSettings(**base))          # identity order
    cam = np.zeros(3)

    # Depth: range from entries[n-1] (min) and entries[1] (max); r/g/b ramps of depth.wgsl
    s = CloudSettings(rasterize_mode=RasterizeMode.Depth, **base)
    dmin, dmax = np.linalg.norm(pv[n - 1, :3] - cam), np.linalg.norm(pv[1, :3] - cam)
    assert np.allclose(oracle.depth_range(c, e, v, s), (dmin, dmax), rtol=1e-6)
    sm = lambda a, b, x: (lambda t: t * t * (3 - 2 * t))(np.clip((x - a) / (b - a), 0, 1))
    for i in range(n):
        nd = np.clip((np.linalg.norm(pv[i, :3] - cam) - dmin) / (dmax - dmin), 0, 1)
        want = (sm(0.5, 1.0, nd), 1 - abs(nd - 0.5) * 2, 1 - sm(0, 0.5, nd))
        assert np.allclose(list(oracle.vs_sorted(c, e, i, v, s).color)[:3], want, atol=2e-5)
    with pytest.raises(RuntimeError):
        oracle.vs(c, e[0], v, s)

    # Normal: third column of T*S*R in view space (camera at origin looking -z => view = world)
    s = CloudSettings(rasterize_mode=RasterizeMode.Normal, **base)
    assert np.allclose(list(oracle.vs(c, e[0], v, s).color)[:3], (0.5, 0.5, 1.0), atol=1e-6)   # +z
    # helpers.wgsl:137-157 feeds the row-major quaternion matrix to WGSL's column-major constructor, so its
    # third COLUMN is (2(xz - wy), 2(yz + wx), 1 - 2(xx + yy)) = (0, 1, 0) for w = x = 0.7071
    assert np.allclose(list(oracle.vs(c, e[1], v, s).color)[:3], (0.5, 1.0, 0.5), atol=1e-6)   # +y

    # Position: (p - min) / (max - min)
    s = CloudSettings(rasterize_mode=RasterizeMode.Position, **base)
    for i in range(n):
        want = (pv[i, :3].astype(np.float64) - mn) / (np.array(mx) - mn)
        assert np.allclose(list(oracle.vs(c, e[i], v, s).color)[:3], want, atol=1e-6)

    # Classification: visibility < 2 -> SH colour (0.5 -> sRGB->linear 0.2140); class k -> 50/50 mix with hue 2*pi*k/num_classes
    s = CloudSettings(rasterize_mode=RasterizeMode.Classification, **base)
    lin = ((0.5 + 0.055) / 1.055) ** 2.4
    assert np.allclose(list(oracle.vs(c, e[0], v, s).color)[:3], (lin,) * 3, atol=1e-6)
    import colorsys
    for i, cls in ((1, 1.0), (2, 3.5), (3, 5.0), (4, 0.0)):
        hue = (cls / 4.0) % 1.0
        want = 0.5 * lin + 0.5 * np.array(colorsys.hsv_to_rgb(hue, 1.0, 1.0))
        assert np.allclose(list(oracle.vs(c, e[i], v, s).color)[:3], want, atol=2e-5), (i, cls)

    # OpticalFlow: previous position == position for a 3D cloud, so a camera that did not move gives zero flow =
    # hsv(angle 0, saturation 0, value 1) = white; a moved camera gives hue = direction, saturation = min(|flow|, 1)
    s = CloudSettings(rasterize_mode=RasterizeMode.OpticalFlow, **base)
    assert list(oracle.vs(c, e[1], v, s).color)[:3] == pytest.approx([1.0, 1.0, 1.0])
    import copy, colorsys, math
    v2 = copy.copy(v)
    v2.previous_clip_from_world = View.perspective(transform_from((0.02, -0.01, 0)), 64, 64).clip_from_world
    v2.delta_time = 0.05
    for i in (1, 2, 3):
        p4 = np.append(pv[i, :3].astype(np.float64), 1.0)
        a, b = np.asarray(v2.clip_from_world, np.float64) @ p4, np.asarray(v2.previous_clip_from_world, np.float64) @ p4
        flow = (a[:2] / a[3] - b[:2] / b[3]) * np.array([0.5, -0.5]) / v2.delta_time
        ang = math.atan2(flow[1], flow[0]) % (2 * math.pi)
        want = colorsys.hsv_to_rgb(ang / (2 * math.pi), min(np.linalg.norm(flow), 1.0), 1.0)
        assert list(oracle.vs(c, e[i], v2, s).color)[:3] == pytest.approx(want, abs=3e-4)

    # the product's arithmetic gives the same per-splat colours
    for mode in (RasterizeMode.Depth, RasterizeMode.Normal, RasterizeMode.Position, RasterizeMode.Classification,
                 RasterizeMode.OpticalFlow):
        s = CloudSettings(rasterize_mode=mode, **base)
        fpc = H.frame_params(n, v, s)
        rng = np.array(oracle.depth_range(c, e, v, s) if mode == RasterizeMode.Depth else (0, 0), np.float32)
        out = H.ShimOut()
        drawn = 0
        for i in range(n):
            H.shim().shim_project(ctypes.byref(fpc), int(e[i]["key"]), H._fp(pv[i]), H._fp(rot[i]), H._fp(so[i]),
                                  H._fp(sh[i]), H._fp(rng), ctypes.byref(out))
            ref = oracle.vs_sorted(c, e, i, v, s)
            if out.draw:   # (the on-axis splat has a NaN OBB in the reference's own maths: nothing to draw)
                assert not ref.discard
                assert np.allclose(list(out.color), list(ref.color), atol=2e-6), (mode, i)
                drawn += 1
        assert drawn >= 3


@pytest.mark.parametrize("seed", range(10))
def test_randomized_configurations_device_math(oracle, seed):
    """CPU pre-flight of the randomized GPU sweep (tests/test_gpu_parity.py): the product's per-splat
    arithmetic, composited by the numpy emulation (per-sample coverage and depth test included), against the oracle."""
    c, v, s = H.random_case(seed)
    e = oracle.sort(c, v, s)
    assert np.array_equal(H.device_sorted_entries(c, v, s)["index"], e["index"])
    ref, amb = oracle.render(c, e, v, s, with_ambiguity=True, depth=v.depth_host)
    got = H.emulate_render(c, v, s)
    ok, err = H.tolerance_mask(ref, got, amb)
    assert ok.all(), f"seed {seed}: max err {err.max():.3e} settings {s}"


def test_draw_modes_device_math_matches_oracle(oracle):
    """DrawMode::Selected / HighlightSelected (src/render/gaussian.wgsl:203-205, 423-427)."""
    from bevy_gaussian_splatting_amd import DrawMode
    c = _classified(random_gaussians_3d_seeded(3000, 27))
    c.position_visibility[:, 3] = (np.arange(len(c)) % 3 == 0).astype(np.float32)   # every third splat selected
    v = View.headless(128, 72)
    base = oracle.render(c, oracle.sort(c, v, CloudSettings()), v, CloudSettings())
    for dm in (DrawMode.Selected, DrawMode.HighlightSelected):
        for kw in ({}, {"aabb": True}):
            s = CloudSettings(draw_mode=dm, **kw)
            e = oracle.sort(c, v, s)
            ref, amb = oracle.render(c, e, v, s, with_ambiguity=True)
            got = H.emulate_render(c, v, s)
            ok, err = H.tolerance_mask(ref, got, amb)
            assert ok.all(), f"{dm.name} {kw}: max err {err.max():.3e}"
        assert not np.allclose(ref, base)
    # closed forms: Selected draws exactly the selected subset; Highlight recolours it (0.3, 1, 0.1) at opacity 1
    s = CloudSettings(draw_mode=DrawMode.Selected)
    e = oracle.sort(c, v, s)
    sel = c.position_visibility[e["index"], 3] > 0.5
    only = oracle.render(c, e[sel], v, CloudSettings())
    assert np.array_equal(oracle.render(c, e, v, s), only)
    hs = CloudSettings(draw_mode=DrawMode.HighlightSelected)
    drawn = [i for i in range(0, 3000, 3) if not oracle.vs(c, (0, i), v, hs).discard][:5]
    for i in drawn:
        assert list(oracle.vs(c, (0, i), v, hs).color) == pytest.approx([0.3, 1.0, 0.1, 1.0])


def test_supertile_index_by_reciprocal_multiply_is_exact():
    """bgs_device.h supertile_div: tile / edge as (tile * (65536 / edge + 1)) >> 16 for every tile index a
    packed rectangle can hold (< 256) and every supertile edge the library can choose (<= 32)."""
    l = H.shim()
    for edge in range(1, 33):
        for tile in range(256):
            assert l.shim_supertile_div(tile, edge) == tile // edge, (tile, edge)


def test_xcd_runs_work_order_is_a_bijection_made_of_runs():
    """splat_math.h xcd_runs_item (the surfel rasteriser's work order: four runs per XCD): every work item exactly once
    for every grid size — a tile nobody draws or a tile drawn twice would be a wrong image — and XCD b % 8 really gets
    S contiguous runs."""
    import ctypes
    l = H.shim()
    for n in list(range(1, 70)) + [255, 256, 257, 2040, 2041, 2047, 4093, 16384, 65535]:
        for S in (1, 2, 3, 4, 8):
            out = np.empty(n, np.uint32)
            l.shim_xcd_runs_items(n, S, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
            assert np.array_equal(np.sort(out), np.arange(n, dtype=np.uint32)), (n, S)
            for xcd in range(min(8, n)):
                mine = out[xcd::8].astype(np.int64)          # this XCD's items in its dispatch order
                assert np.all(np.diff(mine) > 0), (n, S, xcd)  # ascending: runs are walked front to back
                assert int((np.diff(mine) != 1).sum()) <= S - 1, (n, S, xcd)
    # one run per XCD is the contiguous-band order (render_kernels.hip xcd_remap)
    n = 2040
    out = np.empty(n, np.uint32)
    l.shim_xcd_runs_items(n, 1, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    assert np.array_equal(out[0::8], np.arange(0, 255, dtype=np.uint32)) and out[1] == 255


@pytest.mark.parametrize("yaw", [0.0, 0.37])
def test_two_step_keys_are_the_reference_keys_at_every_frustum_threshold(oracle, yaw):
    """splat_math.h sort_key_fast + the divisions for the splats it is unsure about (what the chainless keygen tiles
    run) against sort_key (the reference's statement) and the oracle, on points within a few ulp of every threshold of
    in_frustum and of every guard band. Host build: the reciprocal is the correctly rounded one here and v_rcp_f32
    (1 ulp) on the device, which the 2^-20 bands cover with a factor of four to spare; the device itself is held to
    the same cloud by tests/test_gpu_parity.py."""
    v = View.headless(1920, 1080, yaw=yaw)
    c = H.frustum_boundary_cloud(v, 20_000, 5 + int(100 * yaw))
    s = CloudSettings()
    plain = H.device_keys(c, v, s)
    two_step, unsure = H.device_keys_two_step(c, v, s)
    assert np.array_equal(plain, two_step)
    e = oracle.sort(c, v, s)
    by_index = np.empty(len(c), np.uint32)
    by_index[e["index"]] = e["key"]
    assert np.array_equal(by_index, two_step)
    drawn = int((two_step != 0xFFFFFFFF).sum())
    assert 0.2 * len(c) < drawn < 0.8 * len(c)
    assert 0.05 * len(c) < unsure < 0.6 * len(c), unsure      # a fair share inside the bands, most of the cloud outside
    for mode in (SortMode.Rayon, SortMode.NONE):              # no frustum test in these keys: never unsure
        k2, u2 = H.device_keys_two_step(c, v, CloudSettings(sort_mode=mode))
        assert u2 == 0 and np.array_equal(k2, H.device_keys(c, v, CloudSettings(sort_mode=mode)))


LN_HARD_CASES = (0x65d890d3, 0x4c5d65a5, 0x4d604ebe, 0x41178feb, 0x3c413d3a, 0x6f31a8ec)


def test_ln_f32_cr_is_the_correctly_rounded_log():
    """csrc/exact_log.h (host build; the device runs the same binary64 operations) against (a) the oracle's x87
    logl rounded once, on a few million inputs — every special value, both ends of every binade, the six inputs the
    exhaustive run found nearest to a rounding boundary (2^-57.8 .. 2^-53.2 relative; on the last three of them a
    binary64 `log` rounded to binary32 is WRONG), a range of consecutive patterns below 1 (where opacities live) and
    random patterns — and (b) mpmath at 300 bits on the hard cases and a random sample. The exhaustive comparison over
    all 2 139 095 039 positive inputs is scripts/exact_log/check_exhaustive.cpp (0 mismatches; 28 s on 8 cores) and,
    for the device build, tests/test_gpu_parity.py::test_exact_log_on_the_device_every_positive_input."""
    import mpmath as mp
    from oracle import oracle as orc
    l = H.shim()
    rng = np.random.default_rng(5)
    binade_ends = np.array([[e << 23, (e << 23) | 0x7FFFFF, (e << 23) | 1] for e in range(0, 255)], np.uint32).ravel()
    bits = np.concatenate([
        np.array([0, 1, 2, 0x007FFFFF, 0x00800000, 0x3F800000, 0x3F7FFFFF, 0x3F800001, 0x7F7FFFFF, 0x7F800000, 0x7FC00000,
                  0x80000000, 0xBF800000, 0xFF800000], np.uint32),
        np.array(LN_HARD_CASES, np.uint32), binade_ends,
        np.arange(0x3F800000 - (1 << 21), 0x3F800000 + (1 << 16), dtype=np.uint32),   # [0.75, 1.0078]
        rng.integers(1, 0x7F800000, 2_000_000, dtype=np.uint32)])
    x = bits.view(np.float32)
    own = np.empty_like(x)
    l.shim_ln_f32(H._fp(x), x.size, H._fp(own))
    with np.errstate(all="ignore"):
        ref = orc.ln_f32(x)
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(own), nan)
    assert np.array_equal(own.view(np.uint32)[~nan], ref.view(np.uint32)[~nan])
    assert own[0] == -np.inf and own[11] == -np.inf and own[9] == np.inf and np.isnan(own[12]) and own[5] == 0.0
    # the same sum the GPU test compares, host build vs oracle
    assert int(l.shim_ln_f32_checksum(0x3F000000, 1 << 20)) == orc.ln_f32_checksum(0x3F000000, 1 << 20)

    mp.mp.prec = 300

    def cr(b):   # correctly rounded binary32 ln of the pattern b, from 300-bit arithmetic
        v = mp.log(mp.mpf(float(np.array([b], np.uint32).view(np.float32)[0])))
        f = np.float32(float(v))                                  # a double rounding; repaired against the exact value below
        cands = [f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))]
        return min(cands, key=lambda c: abs(mp.mpf(float(c)) - v))
    sample = list(LN_HARD_CASES) + [int(b) for b in rng.integers(1, 0x7F800000, 3000, dtype=np.uint32)]
    xs = np.array(sample, np.uint32).view(np.float32)
    got = np.empty_like(xs)
    l.shim_ln_f32(H._fp(xs), xs.size, H._fp(got))
    for b, g in zip(sample, got):
        assert np.float32(g) == cr(b), hex(b)
    # and the reason the oracle does not simply round a binary64 log: it misrounds three of the hard cases
    with np.errstate(all="ignore"):
        via_double = np.log(np.array(LN_HARD_CASES, np.uint32).view(np.float32).astype(np.float64)).astype(np.float32)
    assert (via_double != got[:len(LN_HARD_CASES)]).sum() >= 1


def test_2dgs_degeneracy_decision_agrees_on_the_known_ill_conditioned_case():
    """Round 2's one parity failure (seed 321 of the medium sweep, splat 18410, opacity 0.0232): `extent = mean2d^2 - ...`
    (gaussian_2d.wgsl:80-132) cancels to multiples of ulp(mean2d^2), so `extent < 1e-4` asks whether two f32 numbers are
    EQUAL, and with `opacity_adaptive_radius` both depend on ln(opacity). One ulp of that log flips the decision (the
    device took it from v_log_f32, the oracle from libm). Since round 3 every side uses the correctly rounded log
    (csrc/exact_log.h, oracle `ln_correctly_rounded`): the decision is the same number-for-number computation on host
    build, device and oracle — here for the known splat and for the opacities around it, where the decision does flip."""
    import ctypes
    from oracle import oracle as orc
    c, v, s = H.random_case(1000 + 321, medium=True)
    si = 18410
    assert s.gaussian_mode == GaussianMode.Gaussian2d and s.opacity_adaptive_radius
    e = orc.sort(c, v, s)
    key = int(e[e["index"] == si]["key"][0])
    fpc = H.frame_params(len(c), v, s)
    out = H.ShimOut()
    dr = np.zeros(2, np.float32)
    op = c.scale_opacity[si, 3]
    ops = [op]
    for _ in range(24):
        ops.append(np.nextafter(ops[-1], np.float32(0)))
    up = op
    for _ in range(24):
        up = np.nextafter(up, np.float32(1))
        ops.append(up)
    flips = set()
    for o in ops:
        so = c.scale_opacity[si].copy()
        so[3] = o
        H.shim().shim_project(ctypes.byref(fpc), key, H._fp(c.position_visibility[si]), H._fp(c.rotation[si]), H._fp(so),
                              H._fp(c.spherical_harmonic[si]), H._fp(dr), ctypes.byref(out))
        c2 = c.slice(si, si + 1)
        c2.scale_opacity = c2.scale_opacity.copy()
        c2.scale_opacity[0, 3] = o
        vs = orc.vs(c2, np.array([(key, 0)], orc.SORT_ENTRY_DTYPE)[0], v, s)
        oracle_draws = (not vs.discard) and vs.radius[0] > 0.0
        assert bool(out.draw) == bool(oracle_draws), float(o)
        if out.draw:
            assert abs(out.radius - vs.radius[0]) <= 1e-6 * abs(vs.radius[0]), float(o)
        flips.add(bool(out.draw))
    assert flips == {True, False}   # the neighbourhood really is on the edge: both outcomes occur, always on both sides
