"""The oracle against every known-answer the reference's own tests hold for the hot path
(tests/radix.rs), against float64 closed forms, and against the committed goldens."""
import json
import os

import numpy as np
import pytest

import helpers as H
from bevy_gaussian_splatting_amd import (
    CloudSettings, GaussianColorSpace, GaussianMode, PlanarGaussian3d, RadixSortDepthBits,
    ShaderDefines, SortMode, View, random_gaussians_3d_seeded, transform_from)
from bevy_gaussian_splatting_amd.gaussian import Gaussian3d, SphericalHarmonicCoefficients

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _radix_golden():
    with open(os.path.join(GOLDEN, "radix_keys.json")) as f:
        return json.load(f)


def test_radix_defines_table_matches_reference_test(oracle):
    # tests/radix.rs:42-62
    for bits, exp in _radix_golden()["table"].items():
        places, shift, parity = oracle.radix_defines(int(bits))
        assert (places, shift, parity) == (exp["digit_places"], exp["key_shift"], exp["initial_parity"])
        d = ShaderDefines.for_radix_depth_bits(int(bits))
        assert (d.radix_digit_places, d.radix_key_shift, d.radix_initial_parity()) == (places, shift, parity)
        # tests/radix.rs:65-79: the final pass lands in sorted_entries (bind-group parity 1)
        assert (parity + places - 1) % 2 == 1
    with pytest.raises(ValueError):
        oracle.radix_defines(20)


def test_radix_depth_keys_match_golden(oracle):
    # tests/radix.rs:96-106 helpers, values pinned in tests/golden/radix_keys.json
    for case in _radix_golden()["cases"]:
        _, shift, _ = oracle.radix_defines(case["depth_bits"])
        for e in case["entries"]:
            d2 = oracle.distance_squared(e["position"], case["camera"])
            assert int(np.array(d2, np.float32).view(np.uint32)) == e["dist2_bits"]
            assert oracle.radix_depth_key(d2, shift) == e["key"]


def test_radix_order_is_back_to_front_while_camera_moves():
    # tests/radix.rs:11-39: sort by key ascending => dist2 non-increasing, for 24 and 32 bits
    for case in _radix_golden()["cases"]:
        if case["depth_bits"] == 16:
            continue
        es = sorted(case["entries"], key=lambda e: e["key"])
        d = [np.array(e["dist2_bits"], np.uint32).view(np.float32) for e in es]
        assert all(d[i] >= d[i + 1] for i in range(len(d) - 1))


def test_radix_16_bit_key_collapses_close_depths():
    # tests/radix.rs:82-94
    case = [c for c in _radix_golden()["cases"] if c["depth_bits"] == 16 and c["camera"][0] < 0][0]
    assert case["entries"][0]["key"] == case["entries"][1]["key"]


def test_keygen_matches_known_answers_through_the_full_path(oracle):
    """radix_sort_a on the tests/radix.rs positions: camera at `cam` looking down -Z would
    cull them (they sit at +z), so look down +Z (rotate pi about Y) to keep them in frustum."""
    from bevy_gaussian_splatting_amd import rotation_y
    for case in _radix_golden()["cases"]:
        pv = np.array([[*e["position"], 1.0] for e in case["entries"]], np.float32)
        cloud = PlanarGaussian3d(pv, np.zeros((2, 48), np.float32), np.tile([1, 0, 0, 0], (2, 1)),
                                 np.tile([0.01, 0.01, 0.01, 0.5], (2, 1)))
        view = View.perspective(transform_from(case["camera"], rotation_y(np.pi)), 64, 64)
        s = CloudSettings(radix_sort_depth_bits=RadixSortDepthBits(case["depth_bits"]))
        keys = oracle.keygen(cloud, view, s)
        cam = view.world_position  # f32 camera actually used
        for i, e in enumerate(case["entries"]):
            _, shift, _ = oracle.radix_defines(case["depth_bits"])
            d2 = oracle.distance_squared(e["position"], cam)
            assert int(keys["key"][i]) == oracle.radix_depth_key(d2, shift)
            assert int(keys["key"][i]) != (0xFFFFFFFF >> shift)  # in frustum
        if np.allclose(cam, case["camera"], atol=0):
            assert [int(k) for k in keys["key"]] == [e["key"] for e in case["entries"]]


@pytest.mark.parametrize("n", [0, 1, 5, 1023, 1024, 1025, 5000])
@pytest.mark.parametrize("places", [2, 3, 4])
def test_oracle_radix_sort_is_a_stable_sort(oracle, n, places):
    rng = np.random.default_rng(n * 7 + places)
    keys = rng.integers(0, 1 << (8 * places), size=n, dtype=np.uint64).astype(np.uint32)
    if n > 10:
        keys[rng.integers(0, n, size=n // 3)] = keys[0]  # force ties
    e = np.empty(n, oracle.SORT_ENTRY_DTYPE)
    e["key"], e["index"] = keys, np.arange(n, dtype=np.uint32)
    out = oracle.radix_sort(e, places)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(out["index"], order.astype(np.uint32))
    assert np.array_equal(out["key"], keys[order])


def test_oracle_sort_contract_culled_last_ties_by_index(oracle):
    c = random_gaussians_3d_seeded(20000, 3)
    v = View.headless(320, 180)
    for bits in (16, 24, 32):
        s = CloudSettings(radix_sort_depth_bits=RadixSortDepthBits(bits))
        e = oracle.sort(c, v, s)
        k = e["key"].astype(np.int64)
        assert np.all(np.diff(k) >= 0)
        same = np.diff(k) == 0
        assert np.all(np.diff(e["index"].astype(np.int64))[same] > 0)
        sentinel = 0xFFFFFFFF >> (32 - bits)
        culled = e["key"] == sentinel
        assert culled.any() and not culled[0]
        assert np.all(culled[np.argmax(culled):])  # all culled entries form the tail
        assert sorted(e["index"].tolist()) == list(range(len(c)))


def test_oracle_rayon_mode_is_descending_distance(oracle):
    c = random_gaussians_3d_seeded(5000, 4)
    v = View.headless(64, 64)
    e = oracle.sort(c, v, CloudSettings(sort_mode=SortMode.Rayon))
    d = e["key"].view(np.float32)
    assert np.all(np.diff(d) <= 0)
    cam = v.world_position
    p = c.position_visibility[e["index"], :3]
    dd = ((p - cam) ** 2).sum(1)
    assert np.allclose(dd, d, rtol=1e-5)


def _single(pos, scale, opacity, sh0=(1.0, 0.5, 0.25), rot=(1, 0, 0, 0)):
    sh = SphericalHarmonicCoefficients()
    for c, v in enumerate(sh0):
        sh.set(c, v)
    return PlanarGaussian3d.from_interleaved([
        Gaussian3d(np.array([*pos, 1.0], np.float32), sh.coefficients, np.array(rot, np.float32),
                   np.array([*scale, opacity], np.float32))])


def test_single_isotropic_splat_closed_form_aabb(oracle):
    """SURVEY 8c fixture 6: on-axis isotropic splat of scale s at view depth z (AABB mode; OBB is
    NaN exactly on the axis). cov2d = ((s*f*H/z)^2 + 0.3) I in half-pixel^2 with f = 1/tan(pi/8);
    quad half-size = cutoff*sqrt(cov00) half-px; centre alpha = min(opacity*exp(power~0), .999)."""
    s_, z, H_, W_ = 0.2, 4.0, 128, 128
    cloud = _single((0.0, 0.0, -z), (s_, s_, s_), 0.6)
    view = View.perspective(transform_from((0, 0, 0)), W_, H_)
    st = CloudSettings(aabb=True, color_space=GaussianColorSpace.LinRec709Display, sh_degree=0)
    e = oracle.sort(cloud, view, st)
    vs = oracle.vs(cloud, e[0], view, st)
    f = 1.0 / np.tan(np.pi / 8)
    cov = (s_ * f * H_ / z) ** 2 + 0.3
    assert vs.discard == 0
    assert np.allclose([vs.cov2d[0], vs.cov2d[2]], cov, rtol=2e-5) and abs(vs.cov2d[1]) < 1e-3
    cutoff = np.sqrt(9 + 2 * np.log(0.6))
    assert np.isclose(vs.cutoff, cutoff, rtol=1e-6)
    assert np.isclose(vs.bb[3][2], cutoff * np.sqrt(cov), rtol=2e-5)  # radius_px (half-px)
    rgb = 0.5 + 0.28209479177387814 * np.array([1.0, 0.5, 0.25])
    assert np.allclose(list(vs.color)[:3], rgb, rtol=1e-6) and np.isclose(vs.color[3], 0.6)
    img = oracle.render(cloud, e, view, st)
    # the 4 centre pixels are half a pixel (= 1 half-px unit per axis ... d = 1 half-px) off-centre
    d = 1.0  # half-pixel units: pixel centre is 0.5 px = 1 half-px from the splat centre on each axis
    power = -0.5 * (d * d + d * d) / cov
    alpha = min(0.6 * np.exp(power), 0.999)
    centre = img[H_ // 2 - 1: H_ // 2 + 1, W_ // 2 - 1: W_ // 2 + 1]
    assert np.allclose(centre[..., :3], rgb * alpha, rtol=1e-4)
    assert np.allclose(centre[..., 3], 1.0)  # opaque clear colour
    # footprint: AABB square of half-size radius/2 pixels
    half = cutoff * np.sqrt(cov) / 2
    covered = (np.abs(img[..., :3]).sum(-1) > 0)
    ys, xs = np.nonzero(covered)
    assert abs((xs.max() - xs.min() + 1) - 2 * half) <= 1.5 and abs((ys.max() - ys.min() + 1) - 2 * half) <= 1.5


def test_obb_on_axis_splat_vanishes_like_the_reference(oracle):
    """e1 = normalize(0,0) = NaN for an exactly axis-aligned footprint; the quad is NaN
    and the splat contributes nothing (no epsilon is added)."""
    cloud = _single((0.0, 0.0, -4.0), (0.2, 0.2, 0.2), 0.6)
    view = View.perspective(transform_from((0, 0, 0)), 64, 64)
    img = oracle.render(cloud, oracle.sort(cloud, view, CloudSettings()), view, CloudSettings())
    assert np.all(img[..., :3] == 0)
    off = _single((0.3, 0.2, -4.0), (0.2, 0.2, 0.2), 0.6)
    img2 = oracle.render(off, oracle.sort(off, view, CloudSettings()), view, CloudSettings())
    assert np.abs(img2[..., :3]).sum() > 0


def test_visibility_render_thresholds_of_the_reference(oracle):
    """tests/visibility_render.rs:199-274: >= 64 pixels with a channel > 8/255 and max channel
    > 32/255 after the Rgba8UnormSrgb encode of the target."""
    cloud = H.visibility_test_cloud()
    view = View.perspective(transform_from((0, 0, 5)), 128, 128)
    s = CloudSettings(sort_mode=SortMode.NONE, global_opacity=2.0, opacity_adaptive_radius=False)
    img = oracle.render(cloud, oracle.sort(cloud, view, s), view, s)
    lin = np.clip(img[..., :3], 0, 1)
    srgb = np.where(lin <= 0.0031308, lin * 12.92, 1.055 * lin ** (1 / 2.4) - 0.055)
    u8 = np.round(srgb * 255)
    assert (u8.max(-1) > 8).sum() >= 64
    assert u8.max() > 32
    # hidden cloud: nothing drawn -> 0 such pixels (<= 8 allowed)
    empty = oracle.render(cloud, oracle.sort(cloud, view, s)[:0], view, s)
    assert (empty[..., :3] > 8 / 255).sum() == 0


def test_srgb_and_linear_colour_spaces(oracle):
    cloud = _single((0.2, 0.1, -3.0), (0.1, 0.1, 0.1), 0.5, sh0=(0.3, -3.0, 2.0))
    view = View.perspective(transform_from((0, 0, 0)), 32, 32)
    e = oracle.sort(cloud, view, CloudSettings())
    lin = oracle.vs(cloud, e[0], view, CloudSettings(color_space=GaussianColorSpace.LinRec709Display, sh_degree=0))
    srg = oracle.vs(cloud, e[0], view, CloudSettings(sh_degree=0))
    c = np.array(list(lin.color)[:3], np.float64)
    exp = np.where(c <= 0.04045, c / 12.92, ((np.maximum(c, 0) + 0.055) / 1.055) ** 2.4)
    assert np.allclose(list(srg.color)[:3], exp, rtol=1e-5)
    assert c[1] < 0 and c[2] > 1  # unclamped both ways


def test_f16_pack_matches_numpy_half_and_roundtrips(oracle):
    c = random_gaussians_3d_seeded(2000, 9)
    c.scale_opacity[:5, :3] = [1e-9, 6e-8, 7e4]  # underflow / subnormal / overflow to inf
    a = oracle.encode_f16(c)
    b = c.to_f16()
    assert np.array_equal(a.spherical_harmonic, b.spherical_harmonic)
    assert np.array_equal(a.rotation_scale_opacity, b.rotation_scale_opacity)
    d1, d2 = oracle.decode_f16(a), b.to_f32()
    for x, y in ((d1.spherical_harmonic, d2.spherical_harmonic), (d1.rotation, d2.rotation),
                 (d1.scale_opacity, d2.scale_opacity)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.allclose(d1.rotation, c.rotation, atol=1e-3)


@pytest.mark.parametrize("name,kw", [
    ("render_random2k_obb.npz", {}),
    ("render_random2k_aabb.npz", {"aabb": True}),
    ("render_random2k_2d_aabb.npz", {"gaussian_mode": GaussianMode.Gaussian2d, "aabb": True}),
])
def test_oracle_reproduces_committed_goldens(oracle, name, kw):
    g = np.load(os.path.join(GOLDEN, name))
    c = random_gaussians_3d_seeded(2000, 1)
    v = View.headless(96, 64)
    s = CloudSettings(**kw)
    e = oracle.sort(c, v, s)
    assert np.array_equal(e["key"], g["keys"]) and np.array_equal(e["index"], g["index"])
    img = oracle.render(c, e, v, s)
    assert np.allclose(img, g["rgba"], rtol=1e-5, atol=1e-6)


def test_render_window_equals_crop_of_full_frame(oracle):
    c = random_gaussians_3d_seeded(1500, 5)
    v = View.headless(96, 64)
    s = CloudSettings()
    e = oracle.sort(c, v, s)
    full = oracle.render(c, e, v, s)
    win = oracle.render(c, e, v, s, window=(17, 9, 70, 41))
    assert np.array_equal(win, full[9:41, 17:70])
