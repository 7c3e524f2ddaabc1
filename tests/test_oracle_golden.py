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
    for samples, tag in ((4, ""), (1, "_msaa1")):
        v.msaa_samples = samples
        img, amb = oracle.render(c, e, v, s, with_ambiguity=True)
        assert np.allclose(img, g["rgba" + tag], rtol=1e-5, atol=1e-6)
        assert np.allclose(amb, g["amb" + tag], rtol=1e-4, atol=1e-6)
    # the two sample counts are different images (quad edges: partial coverage), not a re-labelling of one
    assert np.abs(g["rgba"] - g["rgba_msaa1"]).max() > 1e-3


def test_render_window_equals_crop_of_full_frame(oracle):
    c = random_gaussians_3d_seeded(1500, 5)
    v = View.headless(96, 64)
    s = CloudSettings()
    e = oracle.sort(c, v, s)
    full = oracle.render(c, e, v, s)
    win = oracle.render(c, e, v, s, window=(17, 9, 70, 41))
    assert np.array_equal(win, full[9:41, 17:70])


# ---------------------------------------------------------------------------------------------------
# Independent float64 pins of the render half of the oracle (projection, quad, falloff, SH, surfel, blend).
# Nothing below re-types the WGSL: every expectation is derived from the geometry (camera model, world
# covariance, ray-plane intersection, the textbook real spherical harmonics via scipy) in float64.
# Where the reference deliberately or accidentally departs from the geometric truth, the departure is
# spelled out in the test that meets it (and in DESIGN.md section 2).
# ---------------------------------------------------------------------------------------------------
import math

from scipy.spatial.transform import Rotation
import scipy.special


def _cam(view):
    wfv = np.asarray(view.world_from_view, np.float64)
    cfv = np.asarray(view.clip_from_view, np.float64)
    return wfv[:3, :3], wfv[:3, 3], cfv[0, 0], cfv[1, 1]


def _pixel_of(view, pw):
    """World point -> framebuffer coordinates (x right, y down, origin at the top-left CORNER, pixel
    centres at +0.5): pinhole camera looking down -Z of the view frame, then the viewport transform."""
    Rc, tc, fx, fy = _cam(view)
    t = Rc.T @ (np.asarray(pw, np.float64) - tc)
    ndc = np.array([fx * t[0] / -t[2], fy * t[1] / -t[2]])
    return np.array([(ndc[0] + 1) / 2 * view.width, (1 - ndc[1]) / 2 * view.height])


def _screen_gaussian(view, pos, sigma_world):
    """Centre and 2x2 covariance (px^2) of the projected Gaussian: numerical Jacobian of _pixel_of."""
    pos = np.asarray(pos, np.float64)
    J = np.zeros((2, 3))
    for k in range(3):
        d = np.zeros(3)
        d[k] = 1e-6
        J[:, k] = (_pixel_of(view, pos + d) - _pixel_of(view, pos - d)) / 2e-6
    return _pixel_of(view, pos), J @ sigma_world @ J.T


def _sigma_world(scale, rot_wxyz, unit=True, gs=1.0, A=np.eye(3)):
    w, x, y, z = (float(v) for v in rot_wxyz)
    if unit:
        R = Rotation.from_quat([x, y, z, w]).as_matrix()
    else:
        # Rodrigues' form R = I + 2w[v]x + 2[v]x^2 evaluated WITHOUT normalising q: what the reference's
        # polynomial does for the slightly non-unit quaternions of its own tools (helpers.wgsl:137-157)
        vx = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
        R = np.eye(3) + 2 * w * vx + 2 * vx @ vx
    S2 = np.diag(np.square(np.asarray(scale, np.float64) * gs))
    return A @ R @ S2 @ R.T @ A.T


LOWPASS_PX2 = 0.3 / 4.0  # the reference adds 0.3 to the diagonal in HALF-pixel units (helpers.wgsl:44-46)


def _oracle_corners(vs, view):
    pr = np.array(vs.projected, np.float64)
    out = []
    for k in range(4):
        nx, ny = (pr[0] + vs.bb[k][0]) / pr[3], (pr[1] + vs.bb[k][1]) / pr[3]
        out.append([(nx + 1) / 2 * view.width, (1 - ny) / 2 * view.height])
    return np.array(out)


def _same_point_set(a, b, atol):
    a, b = np.asarray(a), np.asarray(b)
    used = set()
    for p in a:
        d = np.abs(b - p).max(1)
        j = int(np.argmin(d))
        if d[j] > atol or j in used:
            return False
        used.add(j)
    return True


# Sample positions inside a pixel (origin top-left, y down) of the multisample patterns, from the graphics APIs'
# specifications (Vulkan "standard sample locations", D3D11+ standard patterns, Metal's default positions) — typed
# here, not taken from the oracle: what MultisampleState { count: 4 } (src/render/mod.rs:975-979) means on every
# backend wgpu has.
_SAMPLE_POS_1_4 = {1: [(0.5, 0.5)], 4: [(0.375, 0.125), (0.875, 0.375), (0.125, 0.625), (0.625, 0.875)]}
# Msaa::Sample2 / Sample8 (typed from the Vulkan "standard sample locations" table / D3D11's standard patterns, in
# sixteenths of a pixel from the top-left corner — not taken from the oracle)
SAMPLE_POS_ALL = dict(_SAMPLE_POS_1_4)
SAMPLE_POS_ALL[2] = [(12 / 16, 12 / 16), (4 / 16, 4 / 16)]
SAMPLE_POS_ALL[8] = [(9 / 16, 5 / 16), (7 / 16, 11 / 16), (13 / 16, 9 / 16), (5 / 16, 3 / 16),
                     (3 / 16, 13 / 16), (1 / 16, 7 / 16), (11 / 16, 15 / 16), (15 / 16, 1 / 16)]


SAMPLE_POS = SAMPLE_POS_ALL   # every pixel-level pin below runs at Msaa::Off / Sample2 / Sample4 / Sample8


def test_oracle_sample_positions_are_the_standard_ones(oracle):
    for n, pos in SAMPLE_POS_ALL.items():
        assert np.allclose(oracle.sample_positions(n), pos)
        # every standard pattern is centred on the pixel centre
        assert np.allclose(np.mean(pos, axis=0), (0.5, 0.5))
    for bad in (3, 16):
        with pytest.raises(ValueError):
            oracle.sample_positions(bad)


ANISO_CASES = [
    # name, position, scale, rotation [w,x,y,z], unit quaternion?
    ("tools/compare_aabb_obb.rs:19-58", (0.0, 0.0, 0.0), (10.0, 1.0, 1.0), (0.89, 0.0, -0.432, 0.144), False),
    ("rotated off-axis", (0.5, -0.3, -2.0), (0.8, 0.15, 0.3),
     tuple(np.array([0.89, 0.0, -0.432, 0.144]) / np.linalg.norm([0.89, 0.0, -0.432, 0.144])), True),
    ("near the screen edge", (-2.2, 1.1, -1.0), (0.3, 0.05, 0.6), (0.5, 0.5, -0.5, 0.5), True),
]


@pytest.mark.parametrize("case", ANISO_CASES, ids=[c[0] for c in ANISO_CASES])
@pytest.mark.parametrize("aabb", [False, True])
def test_anisotropic_splat_covariance_and_quad_from_geometry(oracle, case, aabb):
    """Sigma' = J W (R S S^T R^T) W^T J^T from the camera model (numerical Jacobian), eigen-decomposed with
    numpy.linalg.eigh: the oracle's cov2d is 4 Sigma' + 0.3 I (half-pixel units, screen axes x right / y
    down), its quad the rectangle centre +- cutoff sqrt(l1) e1 +- cutoff sqrt(l2) e2 (OBB) or the square of
    half-size cutoff sqrt(l1) (AABB)."""
    _, pos, scale, rot, unit = case
    W_, H_ = 640, 360
    view = View.perspective(transform_from((0.3, 0.2, 4.0), (0.0, math.sin(0.1), 0.0, math.cos(0.1))), W_, H_)
    cloud = _single(pos, scale, 0.5, rot=rot)
    st = CloudSettings(aabb=aabb, opacity_adaptive_radius=False, sh_degree=0)
    e = oracle.sort(cloud, view, st)
    vs = oracle.vs(cloud, e[0], view, st)
    assert vs.discard == 0
    c, S = _screen_gaussian(view, pos, _sigma_world(scale, rot, unit))
    Sp = S + LOWPASS_PX2 * np.eye(2)
    assert np.allclose([vs.cov2d[0], vs.cov2d[1], vs.cov2d[2]], [4 * Sp[0, 0], 4 * Sp[0, 1], 4 * Sp[1, 1]], rtol=2e-4)
    lam, ev = np.linalg.eigh(Sp)  # ascending
    corners = _oracle_corners(vs, view)
    assert np.allclose(corners.mean(0), c, atol=2e-3 * max(1.0, math.sqrt(lam[1])))
    k = 3.0
    if aabb:
        r = k * math.sqrt(lam[1])
        expected = [c + [sx * r, sy * r] for sx in (-1, 1) for sy in (-1, 1)]
    else:
        expected = [c + sx * k * math.sqrt(lam[1]) * ev[:, 1] + sy * k * math.sqrt(lam[0]) * ev[:, 0]
                    for sx in (-1, 1) for sy in (-1, 1)]
    assert _same_point_set(corners, expected, atol=1e-3 * k * math.sqrt(lam[1]) + 0.02), (corners, expected)


@pytest.mark.parametrize("samples", [1, 2, 4, 8])
@pytest.mark.parametrize("aabb", [False, True])
@pytest.mark.parametrize("adaptive", [False, True])
def test_anisotropic_splat_image_is_the_projected_gaussian(oracle, aabb, adaptive, samples):
    """Every covered pixel of a single rotated anisotropic splat against the analytic projected Gaussian
    alpha = opacity exp(-1/2 d^T Sigma'^-1 d), colour premultiplied, over an opaque black target; the
    covered set against the analytic footprint. OBB with opacity_adaptive_radius: the reference ties the
    falloff to the QUAD (exp(-4.5 |uv|^2), gaussian.wgsl:474-480), not to Sigma', so a shrunk quad
    (cutoff < 3) also narrows the Gaussian by (3 / cutoff)^2 — reproduced, and stated here.
    samples = 4 (Msaa::Sample4, the pipeline's sample_count: src/render/mod.rs:357-424,975-979): the resolved pixel is
    the value shaded at the pixel CENTRE times the fraction of the pixel's sample positions the footprint covers."""
    W_, H_ = 200, 140
    view = View.perspective(transform_from((0.1, 0.0, 3.0)), W_, H_, msaa_samples=samples)
    pos, scale, rot, opacity = (0.15, -0.1, 0.0), (0.5, 0.12, 0.25), (0.8, 0.3, -0.4, 0.33166247903554), 0.6
    rot = tuple(np.array(rot) / np.linalg.norm(rot))
    cloud = _single(pos, scale, opacity, rot=rot)
    st = CloudSettings(aabb=aabb, opacity_adaptive_radius=adaptive, sh_degree=0,
                       color_space=GaussianColorSpace.LinRec709Display)
    e = oracle.sort(cloud, view, st)
    img = oracle.render(cloud, e, view, st).astype(np.float64)
    c, S = _screen_gaussian(view, pos, _sigma_world(scale, rot))
    Sp = S + LOWPASS_PX2 * np.eye(2)
    lam, ev = np.linalg.eigh(Sp)
    k = math.sqrt(9 + 2 * math.log(opacity)) if adaptive else 3.0
    ys, xs = np.mgrid[0:H_, 0:W_]

    def footprint(ox, oy):
        """(inside, distance to the footprint's edge in px, power) at the point (ox, oy) of every pixel"""
        d = np.stack([xs + ox - c[0], ys + oy - c[1]], -1)
        a1, a2 = d @ ev[:, 1], d @ ev[:, 0]
        if aabb:
            r = k * math.sqrt(lam[1])
            return ((np.abs(d[..., 0]) <= r) & (np.abs(d[..., 1]) <= r),
                    np.minimum(np.abs(np.abs(d[..., 0]) - r), np.abs(np.abs(d[..., 1]) - r)),
                    -0.5 * np.einsum("...i,ij,...j->...", d, np.linalg.inv(Sp), d))
        u, v = a1 / (k * math.sqrt(lam[1])), a2 / (k * math.sqrt(lam[0]))
        return ((np.abs(u) <= 1) & (np.abs(v) <= 1),
                np.minimum(np.abs(np.abs(a1) - k * math.sqrt(lam[1])), np.abs(np.abs(a2) - k * math.sqrt(lam[0]))),
                -0.5 * (a1 * a1 / lam[1] + a2 * a2 / lam[0]) * (3.0 / k) ** 2)

    _, _, power = footprint(0.5, 0.5)                 # shaded once per pixel, at its centre
    per_sample = [footprint(ox, oy) for ox, oy in SAMPLE_POS[samples]]
    coverage = np.mean([f[0] for f in per_sample], axis=0)   # fraction of the pixel's samples inside
    sure = np.min([f[1] for f in per_sample], axis=0) > 0.02  # a sample within 0.02 px of the edge may fall either side
    alpha = np.minimum(opacity * np.exp(power), 0.999)
    rgb = 0.5 + 0.28209479177387814 * np.array([1.0, 0.5, 0.25])
    covered = np.abs(img[..., :3]).sum(-1) > 0
    assert np.array_equal(covered[sure], (coverage > 0)[sure])
    assert (coverage > 0).sum() > 500
    m = (coverage > 0) & sure
    assert np.allclose(img[m][:, :3], (alpha * coverage)[m][:, None] * rgb, rtol=5e-4, atol=1e-6)
    if samples == 4:  # the footprint's rim really is partially covered, and the extrapolated centre value is what it holds
        partial = m & (coverage < 1)
        assert partial.sum() > 50 and (coverage[partial] * 4 == np.round(coverage[partial] * 4)).all()
    assert np.allclose(img[..., 3], 1.0)


def _complex_sph_harm(l, m, polar, azimuth):
    if hasattr(scipy.special, "sph_harm_y"):  # SciPy >= 1.15
        return scipy.special.sph_harm_y(l, m, polar, azimuth)
    return scipy.special.sph_harm(m, l, azimuth, polar)


def _real_sh_basis(d):
    """The 16 real spherical harmonics up to l = 3 at unit direction d, Condon-Shortley phase kept, in
    the order k = l^2 + l + m of the INRIA coefficient layout — from scipy's complex Y_l^m."""
    x, y, z = d
    theta, phi = math.acos(max(-1.0, min(1.0, z))), math.atan2(y, x)
    out = []
    for l in range(4):
        for m in range(-l, l + 1):
            Y = _complex_sph_harm(l, abs(m), theta, phi)
            out.append(Y.real if m == 0 else math.sqrt(2) * (Y.real if m > 0 else Y.imag))
    return np.array(out)


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
def test_sh_colour_against_scipy_real_spherical_harmonics(oracle, degree):
    """colour = 0.5 + sum_k Y_k(dir) sh[k] (material/spherical_harmonics.wgsl:34-68) with dir the unit
    vector from the camera to the splat, at 1000 random directions, against an independent basis."""
    n = 1000
    rng = np.random.default_rng(40 + degree)
    c = random_gaussians_3d_seeded(n, 11)
    view = View.headless(640, 360)
    c.position_visibility[:, :3] = (rng.uniform(-1, 1, (n, 3)) * [3.5, 2.0, 6.0] + [0, 1.5, -4.0]).astype(np.float32)
    st = CloudSettings(sh_degree=degree, color_space=GaussianColorSpace.LinRec709Display)
    cam = view.world_position.astype(np.float64)
    checked = 0
    for ent in oracle.sort(c, view, st):
        if ent["key"] == 0xFFFFFFFF:
            continue
        i = int(ent["index"])
        vs = oracle.vs(c, ent, view, st)
        if vs.discard:
            continue
        d = c.position_visibility[i, :3].astype(np.float64) - cam
        B = _real_sh_basis(d / np.linalg.norm(d))[: (degree + 1) ** 2]
        sh = c.spherical_harmonic[i].astype(np.float64).reshape(16, 3)[: (degree + 1) ** 2]
        assert np.allclose(list(vs.color)[:3], 0.5 + B @ sh, atol=3e-6)
        checked += 1
    assert checked > 300


def test_sh_direction_follows_the_model_transform(oracle):
    """The view direction is taken to the cloud's local frame (gaussian.wgsl:166-183,408-412): rotating
    cloud AND camera together must not change the colour; rotating the cloud alone must."""
    c = _single((0.4, 0.2, -3.0), (0.1, 0.1, 0.1), 0.5)
    rng = np.random.default_rng(3)
    c.spherical_harmonic[:] = rng.uniform(-1, 1, c.spherical_harmonic.shape).astype(np.float32)
    q = Rotation.from_euler("xyz", [0.3, -0.8, 0.5])
    Rm = q.as_matrix()
    cam0 = transform_from((0.0, 0.0, 1.0))
    v0 = View.perspective(cam0, 64, 64)
    lin = dict(color_space=GaussianColorSpace.LinRec709Display)
    col0 = list(oracle.vs(c, oracle.sort(c, v0, CloudSettings(**lin))[0], v0, CloudSettings(**lin)).color)[:3]
    tr = np.eye(4, dtype=np.float32)
    tr[:3, :3] = Rm
    cam1 = np.eye(4, dtype=np.float32)
    cam1[:3, :3] = Rm
    cam1[:3, 3] = Rm @ np.array([0.0, 0.0, 1.0])
    v1 = View.perspective(cam1, 64, 64)
    s1 = CloudSettings(transform=tr, **lin)
    col1 = list(oracle.vs(c, oracle.sort(c, v1, s1)[0], v1, s1).color)[:3]
    assert np.allclose(col0, col1, atol=2e-5)
    col2 = list(oracle.vs(c, oracle.sort(c, v0, s1)[0], v0, s1).color)[:3]
    assert not np.allclose(col0, col2, atol=1e-3)


def _surfel_case(samples=1):
    W_, H_ = 320, 200
    view = View.perspective(transform_from((0.2, 0.1, 3.0), (0.0, math.sin(0.075), 0.0, math.cos(0.075))), W_, H_,
                            msaa_samples=samples)
    q = Rotation.from_euler("xyz", [0.4, 0.7, -0.3])
    x, y, z, w = q.as_quat()
    pos, scale = (0.3, -0.2, -1.0), (0.5, 0.2, 0.01)
    return view, q.as_matrix(), (w, x, y, z), pos, scale


def _internal_frame(view, pw):
    """The 2DGS branch's own 'pixel' frame (helpers.wgsl:122-135 intrinsic_matrix applied to CLIP space):
    x = ndc.x * (P00 W / 2) + (W - 1) / 2, y likewise, y UP. Not framebuffer pixels: the projection's
    focal factor P00 = f / aspect is applied a second time. That is the reference's behaviour; the frame
    is only ever used consistently inside the surfel code, so it is modelled, not corrected."""
    Rc, tc, fx, fy = _cam(view)
    t = Rc.T @ (np.asarray(pw, np.float64) - tc)
    ndc = np.array([fx * t[0] / -t[2], fy * t[1] / -t[2]])
    return np.array([ndc[0] * fx * view.width / 2 + (view.width - 1) / 2,
                     ndc[1] * fy * view.height / 2 + (view.height - 1) / 2])


def test_surfel_ray_plane_intersection_and_bounds_from_geometry(oracle):
    """2DGS: (a) local_to_pixel must intersect a pixel's camera ray with the surfel plane — solved here as
    a 3x3 linear system in float64 (camera centre + t dir = p + u t_u + v t_v); (b) mean_2d / extent must be
    the centre / squared half-size of the bounding box of the projected cutoff ellipse — found here by
    sampling the ellipse (gaussian_2d.wgsl:80-156)."""
    view, R, rot, pos, scale = _surfel_case()
    W_, H_ = view.width, view.height
    cloud = _single(pos, scale, 0.7, rot=rot)
    st = CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, sh_degree=0, opacity_adaptive_radius=False)
    vs = oracle.vs(cloud, oracle.sort(cloud, view, st)[0], view, st)
    assert vs.discard == 0
    T = np.array(vs.local_to_pixel, np.float64).reshape(3, 3)
    tu, tv = R[:, 0] * scale[0], R[:, 1] * scale[1]
    Rc, tc, fx, fy = _cam(view)
    rng = np.random.default_rng(1)
    for pc in rng.uniform([60, 20], [300, 160], (200, 2)):
        ndc = np.array([(pc[0] - (W_ - 1) / 2) / (fx * W_ / 2), (pc[1] - (H_ - 1) / 2) / (fy * H_ / 2)])
        dw = Rc @ np.array([ndc[0] / fx, ndc[1] / fy, -1.0])
        u, v, _ = np.linalg.solve(np.stack([tu, tv, -dw], 1), tc - np.asarray(pos, np.float64))
        hu, hv = pc[0] * T[2] - T[0], pc[1] * T[2] - T[1]
        p = np.cross(hu, hv)
        assert np.allclose([p[0] / p[2], p[1] / p[2]], [u, v], rtol=2e-5, atol=2e-5)
    ang = np.linspace(0, 2 * np.pi, 20001)
    pts = np.array([_internal_frame(view, np.asarray(pos) + 3 * math.cos(a) * tu + 3 * math.sin(a) * tv) for a in ang])
    mn, mx = pts.min(0), pts.max(0)
    assert np.allclose(list(vs.mean_2d), (mn + mx) / 2, atol=2e-3)
    assert np.allclose(np.sqrt(list(vs.extent)), (mx - mn) / 2, rtol=2e-5)
    assert np.isclose(vs.radius[0], max((mx - mn) / 2)) or vs.radius[0] >= 3 * 0.707106


@pytest.mark.parametrize("samples", [1, 2, 4, 8])
def test_surfel_image_follows_the_intersection(oracle, samples):
    """Every pixel the surfel quad covers: alpha = opacity exp(-1/2 min(u^2 + v^2, 2 |mean_2d - pc|^2)) with
    (u, v) from the float64 ray-plane intersection at the coordinate pc the fragment stage derives for that
    pixel (gaussian.wgsl:440-455: pc = uv_quad * radius * (1, W/H) + mean_2d, the quad being the square of
    half-size radius / 2 framebuffer pixels about the projected centre). The quad is in framebuffer pixels,
    pc in the surfel code's own frame: the two scales do not agree in the reference, and are not made to.
    With 4 samples per pixel the value shaded at the pixel centre is weighted by the fraction of covered samples."""
    view, R, rot, pos, scale = _surfel_case(samples)
    W_, H_ = view.width, view.height
    opacity = 0.7
    cloud = _single(pos, scale, opacity, rot=rot)
    st = CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, sh_degree=0, opacity_adaptive_radius=False,
                       color_space=GaussianColorSpace.LinRec709Display)
    e = oracle.sort(cloud, view, st)
    vs = oracle.vs(cloud, e[0], view, st)
    img = oracle.render(cloud, e, view, st).astype(np.float64)
    tu, tv = R[:, 0] * scale[0], R[:, 1] * scale[1]
    Rc, tc, fx, fy = _cam(view)
    c = _pixel_of(view, pos)
    mean = np.array(list(vs.mean_2d), np.float64)
    radius = float(vs.radius[0])
    rgb = 0.5 + 0.28209479177387814 * np.array([1.0, 0.5, 0.25])
    checked = 0
    for j in range(H_):
        for i in range(W_):
            ux, uy = (i + 0.5 - c[0]) / (radius / 2), -(j + 0.5 - c[1]) / (radius / 2)
            su = [((i + ox - c[0]) / (radius / 2), (j + oy - c[1]) / (radius / 2)) for ox, oy in SAMPLE_POS[samples]]
            g = [max(abs(a), abs(b)) for a, b in su]
            if any(abs(x - 1) <= 1e-3 for x in g):
                continue                                   # a sample on the quad's edge: either side
            cover = sum(x < 1 for x in g) / samples
            if cover == 0:
                assert np.all(img[j, i, :3] == 0)
                continue
            pc = np.array([ux * radius, uy * radius * W_ / H_]) + mean
            ndc = np.array([(pc[0] - (W_ - 1) / 2) / (fx * W_ / 2), (pc[1] - (H_ - 1) / 2) / (fy * H_ / 2)])
            dw = Rc @ np.array([ndc[0] / fx, ndc[1] / fy, -1.0])
            u, v, _ = np.linalg.solve(np.stack([tu, tv, -dw], 1), tc - np.asarray(pos, np.float64))
            power = -0.5 * min(u * u + v * v, 2 * float(((mean - pc) ** 2).sum()))
            alpha = min(opacity * math.exp(power), 0.999)
            assert np.allclose(img[j, i, :3], cover * alpha * rgb, rtol=2e-3, atol=2e-6), (i, j)
            checked += 1
    assert checked > 2000


@pytest.mark.parametrize("samples", [1, 2, 4, 8])
def test_three_splat_stack_blends_back_to_front(oracle, samples):
    """Three overlapping splats at different depths (AABB, analytic alphas as above): the target must hold
    sum_i c_i a_i prod_{j nearer} (1 - a_j) over an opaque clear colour — premultiplied 'over' in
    back-to-front draw order (render/mod.rs:944-948) — and the farthest splat must be drawn first. With 4 samples per
    pixel every SAMPLE is such a stack (of the splats that cover it, each with the alpha shaded at the pixel centre) and
    the pixel is the mean of its samples."""
    W_, H_ = 96, 80
    view = View.perspective(transform_from((0.0, 0.0, 3.0)), W_, H_, clear_color=(0.2, 0.1, 0.05, 1.0), msaa_samples=samples)
    specs = [((0.05, 0.0, 0.0), (0.30, 0.20, 0.2), 0.7, (1.0, 0.0, 0.0)),
             ((-0.05, 0.05, -1.0), (0.45, 0.30, 0.2), 0.5, (0.0, 1.0, 0.0)),
             ((0.0, -0.05, -2.5), (0.90, 0.60, 0.2), 0.9, (0.0, 0.0, 1.0))]
    rot = tuple(np.array([0.9, 0.1, 0.2, -0.3]) / np.linalg.norm([0.9, 0.1, 0.2, -0.3]))
    gs = []
    for pos, scale, op, colour in specs:
        sh = SphericalHarmonicCoefficients()
        for ch, val in enumerate(colour):
            sh.set(ch, (val - 0.5) / 0.28209479177387814)
        gs.append(Gaussian3d(np.array([*pos, 1.0], np.float32), sh.coefficients, np.array(rot, np.float32),
                             np.array([*scale, op], np.float32)))
    cloud = PlanarGaussian3d.from_interleaved(gs)
    st = CloudSettings(aabb=True, opacity_adaptive_radius=False, sh_degree=0,
                       color_space=GaussianColorSpace.LinRec709Display)
    e = oracle.sort(cloud, view, st)
    assert [int(i) for i in e["index"]] == [2, 1, 0]  # farthest first
    img = oracle.render(cloud, e, view, st).astype(np.float64)
    ys, xs = np.mgrid[0:H_, 0:W_]
    C = np.tile(np.array([0.2, 0.1, 0.05]), (samples, H_, W_, 1))   # one target per sample
    unsure = np.zeros((H_, W_), bool)
    for pos, scale, op, colour in reversed(specs):  # back to front
        c, S = _screen_gaussian(view, pos, _sigma_world(scale, rot))
        Sp = S + LOWPASS_PX2 * np.eye(2)
        lam = np.linalg.eigvalsh(Sp)
        r = 3.0 * math.sqrt(lam[1])
        d = np.stack([xs + 0.5 - c[0], ys + 0.5 - c[1]], -1)      # shaded at the pixel centre
        shade = np.minimum(op * np.exp(-0.5 * np.einsum("...i,ij,...j->...", d, np.linalg.inv(Sp), d)), 0.999)
        for k, (ox, oy) in enumerate(SAMPLE_POS[samples]):
            ds = np.stack([xs + ox - c[0], ys + oy - c[1]], -1)
            inside = (np.abs(ds[..., 0]) <= r) & (np.abs(ds[..., 1]) <= r)
            unsure |= np.minimum(np.abs(np.abs(ds[..., 0]) - r), np.abs(np.abs(ds[..., 1]) - r)) < 0.02
            a = np.where(inside, shade, 0.0)
            C[k] = a[..., None] * np.array(colour) + C[k] * (1 - a[..., None])
    C = C.mean(0)                                                    # the resolve
    ok = ~unsure
    assert np.allclose(img[ok][:, :3], C[ok], rtol=5e-4, atol=2e-6)
    assert np.allclose(img[..., 3], 1.0, atol=1e-6)
    # the order matters: the same entries drawn front to back give a different image
    assert np.abs(oracle.render(cloud, e[::-1].copy(), view, st) - img).max() > 0.05


# ---------------------------------------------------------------------------------------------------
# A whole random SCENE against an independent float64 renderer: everything at once — cull, order, projected
# covariance, low-pass, eigen-decomposition, adaptive radius, quad coverage, falloff, SH colour with the view
# direction, sRGB -> linear, alpha clamp, back-to-front blending — derived from the camera model and the
# textbook formulas above, never from the WGSL. The single-splat pins above say WHICH piece is wrong when this
# one fails; this one says the pieces compose.
# ---------------------------------------------------------------------------------------------------
def _srgb_to_linear64(c):
    c = np.asarray(c, np.float64)
    return np.where(c <= 0.04045, c / 12.92, np.power(np.maximum((c + 0.055) / 1.055, 0.0), 2.4))


def _independent_scene(view, cloud, settings, depth=None):
    """[H, W, 4] float64 image of the scene + a mask of pixels within 0.02 px of some quad edge (their coverage
    decision cannot be pinned tighter than the rasteriser's rounding). depth: None or [H, W, samples] scene depth
    (reverse-Z): a quad — one depth for all its fragments, the centre's clip z / w, the vertex stage offsets xy only
    (gaussian.wgsl:395-417) — is drawn at a sample where its depth >= the stored one (CompareFunction::GreaterEqual,
    no depth write; src/render/mod.rs:959-974)."""
    W_, H_ = view.width, view.height
    Rc, tc, fx, fy = _cam(view)
    P = np.asarray(view.clip_from_view, np.float64)
    V = np.linalg.inv(np.asarray(view.world_from_view, np.float64))
    ys, xs = np.mgrid[0:H_, 0:W_]
    samples = SAMPLE_POS[view.msaa_samples]
    img = np.zeros((len(samples), H_, W_, 4))      # one target per sample; resolved (mean) at the end
    img[..., :] = np.asarray(view.clear_color, np.float64)
    edge_mask = np.zeros((H_, W_), bool)
    n = len(cloud)
    pos = cloud.position_visibility[:, :3].astype(np.float64)
    d2 = ((pos - tc) ** 2).sum(1)
    order = np.argsort(-d2, kind="stable")           # back to front: farthest first
    drawn = 0
    for i in order:
        clip = P @ V @ np.append(pos[i], 1.0)
        ndc = clip / (clip[3] + 1e-9)
        if not (abs(ndc[0]) < 1.1 and abs(ndc[1]) < 1.1 and abs(ndc[2] - 0.5) < 0.5):
            continue                                    # the reference culls on the CENTRE (transform.wgsl:5-14)
        scale = cloud.scale_opacity[i, :3].astype(np.float64)
        opacity = float(cloud.scale_opacity[i, 3])
        rot = cloud.rotation[i].astype(np.float64)
        c, S = _screen_gaussian(view, pos[i], _sigma_world(scale, rot, unit=False, gs=settings.global_scale))
        Sp = S + LOWPASS_PX2 * np.eye(2)
        lam, ev = np.linalg.eigh(Sp)
        k = math.sqrt(max(9 + 2 * math.log(opacity), 1e-6)) if settings.opacity_adaptive_radius else 3.0
        r1, r2 = k * math.sqrt(lam[1]), k * math.sqrt(max(lam[0], 0.0))

        def at(ox, oy):
            """(inside the footprint, within 0.02 px of its edge, power) at the point (ox, oy) of every pixel"""
            d = np.stack([xs + ox - c[0], ys + oy - c[1]], -1)
            a1, a2 = d @ ev[:, 1], d @ ev[:, 0]
            if settings.aabb:
                inside = (np.abs(d[..., 0]) <= r1) & (np.abs(d[..., 1]) <= r1)
                edge = np.minimum(np.abs(np.abs(d[..., 0]) - r1), np.abs(np.abs(d[..., 1]) - r1))
                power = -0.5 * np.einsum("...i,ij,...j->...", d, np.linalg.inv(Sp), d)
                near = (np.abs(d[..., 0]) <= r1 + 0.05) & (np.abs(d[..., 1]) <= r1 + 0.05)
            else:
                inside = (np.abs(a1) <= r1) & (np.abs(a2) <= r2)
                edge = np.minimum(np.abs(np.abs(a1) - r1), np.abs(np.abs(a2) - r2))
                power = -4.5 * ((a1 / r1) ** 2 + (a2 / r2) ** 2)   # tied to the quad (gaussian.wgsl:474-480)
                near = (np.abs(a1) <= r1 + 0.05) & (np.abs(a2) <= r2 + 0.05)
            return inside, near & (edge < 0.02), power

        _, _, power = at(0.5, 0.5)                      # the fragment is shaded once, at the pixel centre
        alpha = np.minimum(opacity * settings.global_opacity * np.exp(power), 0.999)
        dirv = pos[i] - tc
        B = _real_sh_basis(dirv / np.linalg.norm(dirv))
        rgb = 0.5 + B @ cloud.spherical_harmonic[i].astype(np.float64).reshape(16, 3)
        if settings.color_space != GaussianColorSpace.LinRec709Display:
            rgb = _srgb_to_linear64(rgb)
        for si, (ox, oy) in enumerate(samples):         # coverage per sample, the same source colour for all of them
            inside, on_edge, _ = at(ox, oy)
            edge_mask |= on_edge
            if depth is not None:
                inside = inside & (clip[2] / clip[3] >= depth[..., si].astype(np.float64))
            a = np.where(inside, alpha, 0.0)[..., None]
            src = np.concatenate([rgb[None, None, :] * a, a], -1)
            img[si] = src + img[si] * (1.0 - a)
        drawn += 1
    return img.mean(0), edge_mask, drawn


@pytest.mark.parametrize("samples", [1, 2, 4, 8])
@pytest.mark.parametrize("aabb", [False, True])
def test_random_scene_against_an_independent_float64_renderer(oracle, aabb, samples):
    """120 random anisotropic splats (unnormalised rotations, SH degree 3, sRGB colour space, adaptive radius, overlapping,
    some culled) at 96x64: the oracle's whole image against the independent renderer above, every pixel but the
    few with a sample position within 0.02 px of a quad edge — single-sampled (Msaa::Off) and with the pipeline's
    default 4 samples per pixel (coverage per sample, shading per pixel, box resolve)."""
    rng = np.random.default_rng(2024)
    n = 120
    c = random_gaussians_3d_seeded(n, 77)
    c.position_visibility[:, :3] = (rng.uniform(-1, 1, (n, 3)) * [2.6, 1.6, 2.5] + [0.0, 1.5, 0.5]).astype(np.float32)
    c.scale_opacity[:, :3] = rng.uniform(0.05, 0.45, (n, 3)).astype(np.float32)
    c.scale_opacity[:, 3] = rng.uniform(0.05, 0.9, n).astype(np.float32)
    c.spherical_harmonic[:] = rng.uniform(-0.6, 0.6, c.spherical_harmonic.shape).astype(np.float32)
    view = View.headless(96, 64, msaa_samples=samples)
    st = CloudSettings(aabb=aabb)
    cam = np.asarray(view.world_from_view, np.float64)[:3, 3]
    d2 = np.sort(((c.position_visibility[:, :3].astype(np.float64) - cam) ** 2).sum(1))
    assert np.diff(d2).min() > 1e-5 * d2.max()     # no two splats within f32 rounding of the same depth: one order
    e = oracle.sort(c, view, st)
    img = oracle.render(c, e, view, st).astype(np.float64)
    ref, edge, drawn = _independent_scene(view, c, st)
    assert 40 < drawn < n                            # a real mix of drawn and culled splats
    ok = ~edge
    assert ok.mean() > {1: 0.9, 2: 0.8, 4: 0.7, 8: 0.55}[samples]   # (more sample positions, more pixels with one near an edge)
    err = np.abs(img - ref)
    assert err[ok].max() < 2e-5, (err[ok].max(), np.abs(ref).max())   # measured: 1e-6 (f32 oracle vs float64 geometry)
    # and the scene is not trivial: most pixels see several splats
    assert (np.abs(ref[..., :3] - np.asarray(view.clear_color)[:3]).sum(-1) > 1e-3).mean() > 0.5


@pytest.mark.parametrize("samples", [1, 2, 4, 8])
@pytest.mark.parametrize("aabb", [False, True])
def test_random_scene_with_a_depth_buffer_against_an_independent_float64_renderer(oracle, aabb, samples):
    """The same 120-splat scene drawn against a scene depth buffer that differs from SAMPLE to sample: every stored depth
    is 0 (nothing in front), 1 (everything in front) or the midpoint between two neighbouring quad depths, drawn at
    random per pixel and sample — so every quad is cut at some samples of some pixels and none of the comparisons is
    within rounding of a tie. The oracle's depth test (per sample, GreaterEqual on the quad's one depth, no write)
    against the independent renderer's."""
    rng = np.random.default_rng(2024)
    n = 120
    c = random_gaussians_3d_seeded(n, 77)
    c.position_visibility[:, :3] = (rng.uniform(-1, 1, (n, 3)) * [2.6, 1.6, 2.5] + [0.0, 1.5, 0.5]).astype(np.float32)
    c.scale_opacity[:, :3] = rng.uniform(0.05, 0.45, (n, 3)).astype(np.float32)
    c.scale_opacity[:, 3] = rng.uniform(0.05, 0.9, n).astype(np.float32)
    c.spherical_harmonic[:] = rng.uniform(-0.6, 0.6, c.spherical_harmonic.shape).astype(np.float32)
    view = View.headless(96, 64, msaa_samples=samples)
    st = CloudSettings(aabb=aabb)
    P = np.asarray(view.clip_from_view, np.float64)
    V = np.linalg.inv(np.asarray(view.world_from_view, np.float64))
    clip = (P @ V @ np.concatenate([c.position_visibility[:, :3].astype(np.float64), np.ones((n, 1))], 1).T).T
    z = np.sort(clip[clip[:, 3] > 0.05, 2] / clip[clip[:, 3] > 0.05, 3])
    gaps = np.diff(z)
    mids = (z[:-1] + 0.5 * gaps)[gaps > 1e-4 * z[1:]]           # clear of every quad's depth by far more than f32 rounding
    assert len(mids) > 40
    levels = np.concatenate([[0.0, 1.0], mids]).astype(np.float32)
    depth = levels[rng.integers(0, len(levels), (view.height, view.width, samples))]
    e = oracle.sort(c, view, st)
    img = oracle.render(c, e, view, st, depth=depth).astype(np.float64)
    ref, edge, drawn = _independent_scene(view, c, st, depth=depth)
    open_ref, _, _ = _independent_scene(view, c, st)
    ok = ~edge
    err = np.abs(img - ref)
    assert err[ok].max() < 2e-5, (err[ok].max(), np.abs(ref).max())
    # the depth buffer matters: most pixels differ from the undepthed scene, and (4x) by amounts no whole-pixel test gives
    changed = np.abs(ref - open_ref).max(-1) > 1e-3
    assert changed.mean() > 0.5
    # a buffer that is the same at every sample of a pixel gives the per-pixel test: the two sample counts' oracles agree
    # with their own independent renderers there too (plane through the middle of the cloud)
    plane = np.full((view.height, view.width, samples), np.float32(np.median(mids)))
    plane[:, : view.width // 2] = 0.0
    img2 = oracle.render(c, e, view, st, depth=plane).astype(np.float64)
    ref2, edge2, _ = _independent_scene(view, c, st, depth=plane)
    assert np.abs(img2 - ref2)[~edge2].max() < 2e-5
    assert np.array_equal(img2[:, : view.width // 2 - 8], oracle.render(c, e, view, st)[:, : view.width // 2 - 8].astype(np.float64))


# ---------------------------------------------------------------------------------------------------
# The independent renderer, widened (round 4): the f16 storage, the precomputed-covariance plane and a whole scene of
# 2DGS surfels go through an independent float64 derivation as well.
# ---------------------------------------------------------------------------------------------------
def _scene_cloud(n, seed):
    rng = np.random.default_rng(seed)
    c = random_gaussians_3d_seeded(n, 77)
    c.position_visibility[:, :3] = (rng.uniform(-1, 1, (n, 3)) * [2.6, 1.6, 2.5] + [0.0, 1.5, 0.5]).astype(np.float32)
    c.scale_opacity[:, :3] = rng.uniform(0.05, 0.45, (n, 3)).astype(np.float32)
    c.scale_opacity[:, 3] = rng.uniform(0.05, 0.9, n).astype(np.float32)
    c.spherical_harmonic[:] = rng.uniform(-0.6, 0.6, c.spherical_harmonic.shape).astype(np.float32)
    return c


@pytest.mark.parametrize("samples", [1, 2, 4, 8])
def test_f16_cloud_scene_against_the_independent_renderer(oracle, samples):
    """The f16 planar storage (src/gaussian/f16.rs:29-55, src/render/planar.wgsl:117-176): the oracle draws the PACKED
    cloud (its own unpack: which half holds which value, coefficient pairing); the independent renderer draws the cloud
    whose attributes were rounded to binary16 by numpy — another implementation of round-to-nearest-even, and no
    packing at all. Same image => the pack / unpack conventions and the rounding are what the format says."""
    c = _scene_cloud(120, 2025)
    packed = c.to_f16()
    h = lambda a: a.astype(np.float16).astype(np.float32)
    rounded = PlanarGaussian3d(c.position_visibility, h(c.spherical_harmonic), h(c.rotation), h(c.scale_opacity))
    view = View.headless(96, 64, msaa_samples=samples)
    st = CloudSettings()
    e = oracle.sort(packed, view, st)
    img = oracle.render(packed, e, view, st).astype(np.float64)
    ref, edge, drawn = _independent_scene(view, rounded, st)
    assert 40 < drawn < len(c)
    err = np.abs(img - ref)
    assert err[~edge].max() < 2e-5, err[~edge].max()
    # ... and it is not the f32 cloud's image: the rounding is visible
    img32 = oracle.render(c, oracle.sort(c, view, st), view, st)
    assert np.abs(img32 - img).max() > 1e-3


def test_covariance_3d_plane_against_float64():
    """`precompute_covariance_3d` (src/gaussian/covariance.rs:4-41, f32.rs:218-251): the six entries (xx, xy, xz, yy, yz,
    zz) of M^T M with M = S R, R the reference's rotation polynomial of the (unnormalised) quaternion — against
    R_std S^2 R_std^T in float64, R_std = the textbook rotation of that quaternion evaluated without normalising."""
    from bevy_gaussian_splatting_amd.gaussian import covariance_3d_opacity
    c = random_gaussians_3d_seeded(500, 9)
    got = covariance_3d_opacity(c).astype(np.float64)
    for i in range(len(c)):
        S = _sigma_world(c.scale_opacity[i, :3], c.rotation[i], unit=False)
        want = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]
        assert np.allclose(got[i, :6], want, rtol=2e-5, atol=1e-6), i
        assert got[i, 6] == c.scale_opacity[i, 3]


def _internal_frame_many(view, pw):
    """_internal_frame for an [n, 3] array of world points."""
    Rc, tc, fx, fy = _cam(view)
    t = (np.asarray(pw, np.float64) - tc) @ Rc
    ndc = np.stack([fx * t[:, 0] / -t[:, 2], fy * t[:, 1] / -t[:, 2]], 1)
    return np.stack([ndc[:, 0] * fx * view.width / 2 + (view.width - 1) / 2,
                     ndc[:, 1] * fy * view.height / 2 + (view.height - 1) / 2], 1)


def _independent_surfel_scene(view, cloud, settings, depth=None):
    """[H, W, 4] float64 image of a scene of 2DGS surfels (GaussianMode::Gaussian2d, aabb) + the mask of pixels with a
    sample position within 0.02 px of a quad edge. Per surfel, from the geometry only: tangent vectors t_u, t_v = the
    first two columns of the rotation times the scales; the quad = the square of half-size radius / 2 framebuffer pixels
    about the projected centre, radius = the larger half-side of the bounding box of the projected cutoff ellipse in the
    surfel code's own frame (sampled; at least cutoff * 0.707106); alpha at a pixel = opacity exp(-1/2 min(u^2 + v^2,
    2 |mean_2d - pc|^2)) with (u, v) from the float64 ray-plane intersection at the coordinate pc the fragment stage
    derives for the pixel (see test_surfel_image_follows_the_intersection for the reference's two frames)."""
    W_, H_ = view.width, view.height
    Rc, tc, fx, fy = _cam(view)
    P = np.asarray(view.clip_from_view, np.float64)
    V = np.linalg.inv(np.asarray(view.world_from_view, np.float64))
    ys, xs = np.mgrid[0:H_, 0:W_]
    samples = SAMPLE_POS[view.msaa_samples]
    img = np.zeros((len(samples), H_, W_, 4))
    img[..., :] = np.asarray(view.clear_color, np.float64)
    edge_mask = np.zeros((H_, W_), bool)
    pos = cloud.position_visibility[:, :3].astype(np.float64)
    order = np.argsort(-((pos - tc) ** 2).sum(1), kind="stable")
    ang = np.linspace(0, 2 * np.pi, 4001)
    drawn = 0
    for i in order:
        clip = P @ V @ np.append(pos[i], 1.0)
        ndc = clip / (clip[3] + 1e-9)
        if not (abs(ndc[0]) < 1.1 and abs(ndc[1]) < 1.1 and abs(ndc[2] - 0.5) < 0.5):
            continue
        w, x, y, z = (float(v) for v in cloud.rotation[i])
        R = Rotation.from_quat([x, y, z, w]).as_matrix()
        sc = cloud.scale_opacity[i, :3].astype(np.float64) * settings.global_scale
        opacity = float(cloud.scale_opacity[i, 3])
        k = math.sqrt(max(9 + 2 * math.log(opacity), 1e-6)) if settings.opacity_adaptive_radius else 3.0
        tu, tv = R[:, 0] * sc[0], R[:, 1] * sc[1]
        pts = _internal_frame_many(view, pos[i] + k * np.cos(ang)[:, None] * tu + k * np.sin(ang)[:, None] * tv)
        mn, mx = pts.min(0), pts.max(0)
        mean, half = (mn + mx) / 2, (mx - mn) / 2
        radius = max(half[0], half[1], k * 0.707106)
        c = _pixel_of(view, pos[i])

        def quad_uv(ox, oy):
            return (xs + ox - c[0]) / (radius / 2), -(ys + oy - c[1]) / (radius / 2)

        uq, vq = quad_uv(0.5, 0.5)                          # shaded once per pixel, at its centre
        pcx, pcy = uq * radius + mean[0], vq * radius * W_ / H_ + mean[1]
        nx, ny = (pcx - (W_ - 1) / 2) / (fx * W_ / 2), (pcy - (H_ - 1) / 2) / (fy * H_ / 2)
        dw = np.einsum("ij,...j->...i", Rc, np.stack([nx / fx, ny / fy, -np.ones_like(nx)], -1))
        # camera centre + t dw = p + u t_u + v t_v  ->  [t_u t_v -dw] (u, v, t)^T = camera centre - p  (Cramer)
        b = tc - pos[i]
        det = np.einsum("i,...i->...", np.cross(tu, tv), -dw)
        u = np.einsum("i,...i->...", np.cross(b, tv), -dw) / det
        v = np.einsum("i,...i->...", np.cross(tu, b), -dw) / det
        power = -0.5 * np.minimum(u * u + v * v, 2 * ((mean[0] - pcx) ** 2 + (mean[1] - pcy) ** 2))
        alpha = np.minimum(opacity * settings.global_opacity * np.exp(power), 0.999)
        dirv = pos[i] - tc
        B = _real_sh_basis(dirv / np.linalg.norm(dirv))
        rgb = 0.5 + B @ cloud.spherical_harmonic[i].astype(np.float64).reshape(16, 3)
        if settings.color_space != GaussianColorSpace.LinRec709Display:
            rgb = _srgb_to_linear64(rgb)
        for si, (ox, oy) in enumerate(samples):
            us, vs = quad_uv(ox, oy)
            g = np.maximum(np.abs(us), np.abs(vs))
            edge_mask |= np.abs(g - 1) * (radius / 2) < 0.02
            inside = g <= 1
            if depth is not None:                           # the quad's one depth (the centre's), per sample, GreaterEqual
                inside = inside & (clip[2] / clip[3] >= depth[..., si].astype(np.float64))
            a = np.where(inside, alpha, 0.0)[..., None]
            img[si] = np.concatenate([rgb[None, None, :] * a, a], -1) + img[si] * (1.0 - a)
        drawn += 1
    return img.mean(0), edge_mask, drawn


@pytest.mark.parametrize("samples", [1, 2, 4, 8])
def test_surfel_scene_against_an_independent_float64_renderer(oracle, samples):
    """100 overlapping 2DGS surfels (unit quaternions within ~50 degrees of facing the camera, SH degree 3, sRGB colour
    space, adaptive radius) at 128x80, GaussianMode::Gaussian2d + aabb (the true surfel fragment path): the oracle's
    whole image against the independent surfel renderer above. Covers what the single-surfel pins cannot: order, blending
    and clipping of many surfels, each with its own two frames."""
    rng = np.random.default_rng(11)
    n = 100
    c = random_gaussians_3d_seeded(n, 5)
    c.position_visibility[:, :3] = (rng.uniform(-1, 1, (n, 3)) * [2.4, 1.4, 1.5] + [0.0, 1.5, 0.0]).astype(np.float32)
    q = Rotation.from_euler("xyz", rng.uniform(-0.8, 0.8, (n, 3))).as_quat()      # x, y, z, w
    c.rotation[:] = np.stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]], 1).astype(np.float32)
    c.scale_opacity[:, :3] = rng.uniform(0.08, 0.35, (n, 3)).astype(np.float32)
    c.scale_opacity[:, 3] = rng.uniform(0.2, 0.9, n).astype(np.float32)
    c.spherical_harmonic[:] = rng.uniform(-0.6, 0.6, c.spherical_harmonic.shape).astype(np.float32)
    view = View.headless(128, 80, msaa_samples=samples)
    st = CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True)
    e = oracle.sort(c, view, st)
    # none of the reference's degeneracy branches is taken in this scene (they are pinned by their own tests)
    live = [oracle.vs(c, ent, view, st) for ent in e if ent["key"] != 0xFFFFFFFF]
    assert all(v.discard == 0 and v.radius[0] > 0 for v in live)
    img = oracle.render(c, e, view, st).astype(np.float64)
    ref, edge, drawn = _independent_surfel_scene(view, c, st)
    assert drawn == len(live) and 60 < drawn <= n
    ok = ~edge
    assert ok.mean() > 0.6
    err = np.abs(img - ref)
    # (f32 ray-plane intersection against float64 geometry; the bound of the single-surfel pin, relative to the colour range)
    assert err[ok].max() < 4e-3 * max(1.0, np.abs(ref).max()), (err[ok].max(), np.abs(ref).max())
    assert np.quantile(err[ok], 0.99) < 2e-4
    assert (np.abs(ref[..., :3] - np.asarray(view.clear_color)[:3]).sum(-1) > 1e-3).mean() > 0.4
    # and against a depth buffer that differs from sample to sample (levels clear of every surfel's depth, as in
    # test_random_scene_with_a_depth_buffer_against_an_independent_float64_renderer)
    P = np.asarray(view.clip_from_view, np.float64)
    V = np.linalg.inv(np.asarray(view.world_from_view, np.float64))
    clip = (P @ V @ np.concatenate([c.position_visibility[:, :3].astype(np.float64), np.ones((n, 1))], 1).T).T
    z = np.sort(clip[clip[:, 3] > 0.05, 2] / clip[clip[:, 3] > 0.05, 3])
    gaps = np.diff(z)
    mids = (z[:-1] + 0.5 * gaps)[gaps > 1e-4 * z[1:]]
    levels = np.concatenate([[0.0, 1.0], mids]).astype(np.float32)
    depth = levels[rng.integers(0, len(levels), (view.height, view.width, samples))]
    img_d = oracle.render(c, e, view, st, depth=depth).astype(np.float64)
    ref_d, edge_d, _ = _independent_surfel_scene(view, c, st, depth=depth)
    err_d = np.abs(img_d - ref_d)[~edge_d]
    assert err_d.max() < 4e-3 * max(1.0, np.abs(ref_d).max()) and np.quantile(err_d, 0.99) < 2e-4
    assert (np.abs(ref_d - ref).max(-1) > 1e-3).mean() > 0.3     # the buffer cuts the scene
