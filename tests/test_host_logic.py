"""Host-side mirror of the reference's data model / settings (no GPU, no oracle)."""
import numpy as np
import pytest

from bevy_gaussian_splatting_amd import (
    CloudSettings, GaussianMode, PlanarGaussian3d, PlanarGaussian3dF16, RadixSortDepthBits,
    ShaderDefines, SortedEntries, SortMode, View, random_gaussians_3d_seeded)
from bevy_gaussian_splatting_amd import gaussian as G
from bevy_gaussian_splatting_amd.multiview import assign_views, headless_view


def test_shader_defines_geometry():
    # src/render/mod.rs:715-758
    d = ShaderDefines.for_radix_depth_bits(RadixSortDepthBits.Bits32)
    assert (d.radix_base, d.radix_bits_per_digit, d.workgroup_entries_c) == (256, 8, 1024)
    assert d.workgroup_entries_a == 256 * 4 * 4
    assert d.sorting_buffer_size == 256 * 4 * 4 + (5 + 256) * 4
    assert d.max_tile_count(1_000_000) == 977
    with pytest.raises(ValueError):
        ShaderDefines.for_radix_depth_bits(12)


def test_cloud_settings_defaults_match_reference():
    s = CloudSettings()  # src/gaussian/settings.rs:110-131
    assert (s.aabb, s.global_opacity, s.global_scale, s.opacity_adaptive_radius) == (False, 1.0, 1.0, True)
    assert s.sort_mode == SortMode.Radix and s.radix_sort_depth_bits == RadixSortDepthBits.Bits32
    assert s.gaussian_mode == GaussianMode.Gaussian3d
    n = s.to_native()
    assert list(n.transform) == [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    t = np.eye(4, dtype=np.float32); t[:3, 3] = [1, 2, 3]
    n2 = CloudSettings(transform=t).to_native()
    assert list(n2.transform)[12:15] == [1, 2, 3]  # column-major: translation in column 3


def test_sorted_entries_initialisation():
    se = SortedEntries.new(3, 5)  # src/sort/mod.rs:347-354
    assert se.sorted.shape == (15,) and np.all(se.sorted["key"] == 1)
    assert se.chunk(2)["index"].tolist() == [0, 1, 2, 3, 4]
    # auto_insert_sorted_entries sizes the asset with the square-padded length (src/sort/mod.rs:259-262) but the
    # sort and the draw stride by the CLOUD length (src/sort/rayon.rs:82-84, src/render/mod.rs:1548-1554)
    se = SortedEntries.for_cloud(2, 10)
    assert se.entry_count == 16 and se.sorted.shape == (32,)
    se.chunk(1, 10)["index"][:] = 7
    assert np.all(se.sorted["index"][10:20] == 7) and se.sorted["index"][9] == 9 and se.sorted["index"][20] == 4
    with pytest.raises(IndexError):
        se.chunk(4, 10)
    assert se.resized(2) is se and se.resized(3).sorted.shape == (48,) and np.all(se.resized(3).sorted["key"] == 1)


def test_random_cloud_distributions_and_determinism():
    a = random_gaussians_3d_seeded(20000, 5)
    b = random_gaussians_3d_seeded(20000, 5)
    assert np.array_equal(a.position_visibility, b.position_visibility)
    assert len(a) == 20000 and a.nbytes() == 20000 * 240
    p = a.position_visibility
    assert p[:, :3].min() >= -20 and p[:, :3].max() < 20 and np.all(p[:, 3] == 1)
    assert a.scale_opacity[:, :3].min() >= 0 and a.scale_opacity[:, :3].max() < 1
    assert a.scale_opacity[:, 3].max() < 0.8
    assert a.rotation.min() >= -1 and a.rotation.max() < 1
    assert abs(np.linalg.norm(a.rotation, axis=1).mean() - 1.0) > 0.05  # NOT normalised
    assert abs(a.spherical_harmonic.mean()) < 0.01


def test_f16_layout():
    c = random_gaussians_3d_seeded(16, 2)
    h = c.to_f16()
    assert isinstance(h, PlanarGaussian3dF16) and h.nbytes() == 16 * 128
    # low half = even coefficient (src/render/planar.wgsl:117-130)
    lo = (h.spherical_harmonic[:, 0] & 0xFFFF).astype(np.uint16).view(np.float16)
    assert np.array_equal(lo, c.spherical_harmonic[:, 0].astype(np.float16))
    # first value in the high half (src/gaussian/f16.rs:244-252)
    hi = (h.rotation_scale_opacity[:, 0] >> 16).astype(np.uint16).view(np.float16)
    assert np.array_equal(hi, c.rotation[:, 0].astype(np.float16))
    op = (h.rotation_scale_opacity[:, 3] & 0xFFFF).astype(np.uint16).view(np.float16)
    assert np.array_equal(op, c.scale_opacity[:, 3].astype(np.float16))
    back = h.to_f32()
    assert np.allclose(back.scale_opacity, c.scale_opacity, atol=1e-3)


def test_test_model_geometry():
    m = G.test_model(0)  # src/gaussian/formats/planar_3d.rs:193-251
    assert len(m) == 9 and np.array_equal(m.position_visibility[8], m.position_visibility[0])
    assert set(np.abs(m.position_visibility[:, :3]).ravel().tolist()) == {0.5}


def test_headless_camera_matches_reference_example():
    v = View.headless()  # examples/headless.rs:177-184
    assert np.allclose(v.world_position, [0, 1.5, 5])
    f = 1 / np.tan(np.pi / 8)
    assert np.isclose(v.clip_from_view[1, 1], f) and np.isclose(v.clip_from_view[0, 0], f / (1920 / 1080))
    assert v.clip_from_view[3, 2] == -1 and np.isclose(v.clip_from_view[2, 3], 0.1)
    p = v.clip_from_world @ np.array([0, 1.5, 5 - 0.1, 1], np.float32)  # point on the near plane
    assert np.isclose(p[2] / p[3], 1.0, atol=1e-5)  # reverse-Z: near -> 1
    assert np.allclose(v.view_from_world @ v.world_from_view, np.eye(4), atol=1e-6)


def test_view_assignment():
    assert assign_views(8, 8) == [[g] for g in range(8)]
    assert assign_views(8, 2) == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert assign_views(3, 4) == [[0], [1], [2], []]
    v = headless_view(2, 64, 32)
    assert v.camera.order == 2 and np.allclose(v.world_position, [0, 1.5, 5])
    fwd = v.world_from_view[:3, :3] @ np.array([0, 0, -1], np.float32)
    assert np.allclose(fwd, [-1, 0, 0], atol=1e-6)  # yawed 90 degrees about +Y


def test_compute_covariance_3d_cpu_twin():
    """src/gaussian/covariance.rs:4-41 and Covariance3dOpacity::from (src/gaussian/f32.rs:238-251)."""
    from bevy_gaussian_splatting_amd import compute_covariance_3d, covariance_3d_opacity
    assert np.allclose(compute_covariance_3d([1, 0, 0, 0], [2, 3, 4]), [4, 0, 0, 9, 0, 16])
    # a rotation leaves the trace (sum of squared scales) alone; Sigma = R^T S^2 R with the reference's R
    q = np.array([0.5, 0.5, 0.5, 0.5], np.float32)
    cov = compute_covariance_3d(q, [1, 2, 3])
    assert np.isclose(cov[0] + cov[3] + cov[5], 14.0)
    c = random_gaussians_3d_seeded(500, 3)
    planes = compute_covariance_3d(c.rotation, c.scale_opacity[:, :3])
    for i in (0, 7, 499):
        r, x, y, z = c.rotation[i].astype(np.float64)
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y)],
                      [2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x)],
                      [2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)]])   # rows = from_cols columns^T
        M = np.diag(c.scale_opacity[i, :3].astype(np.float64)) @ R
        S = M.T @ M
        assert np.allclose(planes[i], [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]], rtol=1e-5, atol=1e-6)
    co = covariance_3d_opacity(c)
    assert co.shape == (500, 8) and np.array_equal(co[:, 6], c.scale_opacity[:, 3]) and np.all(co[:, 7] == 0)


def test_supertile_level_rule_is_stable_and_direct():
    """frame_params.h next_supertile_level (what finish_lane applies to every completed frame): inside
    [1.6, 4] entries per visible splat the level stays; outside it jumps straight to the level the frame's
    geometry asks for; a step never lands on a level that would step back (no flip-flop)."""
    import ctypes
    import helpers as H
    sh = H.shim()
    edges = (ctypes.c_uint32 * 4)(6, 8, 16, 32)

    def nxt(ratio, lv):
        longer = ctypes.c_double(0.0)
        return sh.shim_next_supertile_level(ratio, lv, edges, ctypes.byref(longer)), longer.value

    def ratio_at(extent_tiles, lv):   # a splat of that extent overlaps (extent / edge + 1)^2 supertiles on average
        return (extent_tiles / edges[lv] + 1.0) ** 2

    for lv in range(4):
        for r in (1.6, 2.0, 3.99, 4.0):
            assert nxt(r, lv)[0] == lv
    assert nxt(14.8, 1) == (3, pytest.approx(((22.77 / 32 + 1) ** 2 / 14.8) * 16, rel=1e-2))   # the dense headline frame
    assert nxt(1.3, 1)[0] == 0 and nxt(1.3, 3)[0] == 0 and nxt(1.2, 0)[0] == 0                  # scene-like
    assert nxt(5.0, 3)[0] == 3 and nxt(100.0, 0)[0] == 3
    # from any level and any splat extent the rule settles within two completed frames and then stays
    for extent in np.geomspace(0.05, 400.0, 60):
        for start in range(4):
            lv = start
            seen = [lv]
            for _ in range(4):
                lv = nxt(ratio_at(extent, lv), lv)[0]
                seen.append(lv)
            assert seen[2] == seen[3] == seen[4], (extent, seen)
    assert sh.shim_pow2_ceil(0) == 1 and sh.shim_pow2_ceil(1) == 1 and sh.shim_pow2_ceil(5) == 8
    assert sh.shim_pow2_ceil(1 << 20) == 1 << 20 and sh.shim_pow2_ceil((1 << 40) + 1) == 1 << 31


def test_splitter_tables_are_only_accepted_when_ascending():
    """bucket(key) = number of splitters <= key orders the buckets only for an ascending table."""
    import ctypes
    import helpers as H
    sh = H.shim()
    rng = np.random.default_rng(2)
    good = np.sort(rng.integers(0, 1 << 32, 255, dtype=np.uint64).astype(np.uint32))
    ptr = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    assert sh.shim_splitters_ascending(ptr(good), 255) == 1
    assert sh.shim_splitters_ascending(ptr(np.full(255, 7, np.uint32)), 255) == 1   # all equal keys: still monotone
    bad = good.copy()
    bad[100], bad[101] = bad[101], bad[100]
    assert good[100] == good[101] or sh.shim_splitters_ascending(ptr(bad), 255) == 0
    # the search keygen runs (branchless count of entries <= key over 255 sorted values) is monotone in the key
    keys = np.sort(rng.integers(0, 1 << 32, 5000, dtype=np.uint64).astype(np.uint32))
    buckets = np.searchsorted(good, keys, side="right")
    assert np.all(np.diff(buckets) >= 0) and buckets.min() >= 0 and buckets.max() <= 255
