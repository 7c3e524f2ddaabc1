"""The C++ host layer (include/bgs.hpp) and examples/headless.cpp (the reference's examples/headless.rs).
CPU: it builds with plain g++ against the C ABI and fails loudly without a HIP device (no CPU fallback).
GPU: the frame it renders is bit-identical to the same call through the Python mirror, and the
Rgba8UnormSrgb PNG it writes decodes to the library's sRGB8 image."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLE = os.path.join(ROOT, "examples", "headless")


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "bevy_gaussian_splatting_amd", "csrc"), "-j4"], check=True,
                   capture_output=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "headless"], check=True, capture_output=True)


def _decode_png_rgba8(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + body) & 0xFFFFFFFF == crc, typ
        if typ == b"IHDR":
            w, h, depth, colour = struct.unpack(">IIBB", body[:10])
            assert (depth, colour) == (8, 6)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 4 * w)
    assert (raw[:, 0] == 0).all()  # filter type none
    return raw[:, 1:].reshape(h, w, 4)


def test_cpp_example_builds_and_fails_loudly_without_a_device(tmp_path):
    import torch

    _build()
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present: covered by the gpu test")
    r = subprocess.run([EXAMPLE, "--gaussian-count", "100", "--width", "64", "--height", "64", "--frames", "1",
                        "--output-dir", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "no usable HIP device" in r.stderr and "no CPU fallback" in r.stderr
    assert not os.path.exists(tmp_path / "0.png")


@pytest.mark.gpu
def test_cpp_example_matches_the_python_mirror(tmp_path):
    from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded

    _build()
    c = random_gaussians_3d_seeded(50_000, 9)
    planes = tmp_path / "cloud.bin"
    with open(planes, "wb") as f:
        f.write(struct.pack("<I", len(c)))
        for a in (c.position_visibility, c.spherical_harmonic, c.rotation, c.scale_opacity):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
    r = subprocess.run([EXAMPLE, "--cloud", str(planes), "--width", "640", "--height", "360", "--frames", "12",
                        "--output-dir", str(tmp_path), "--dump-f32", str(tmp_path / "frame.f32")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert '"frames_per_s"' in r.stdout
    got = np.fromfile(tmp_path / "frame.f32", np.float32).reshape(360, 640, 4)

    p = GaussianSplattingPlugin(0)
    h = p.upload(c)
    v = View.headless(640, 360)
    ref = p.render(h, v, CloudSettings())
    assert np.array_equal(got, ref)                       # same library, same bits
    p.set_output_srgb8(True)
    p.render(h, v, CloudSettings(), download=False)
    p.synchronize()
    from bevy_gaussian_splatting_amd.multiview import device_ptr_as_tensor
    ptr, nbytes = p.framebuffer_srgb8_device_ptr()
    srgb = device_ptr_as_tensor(ptr, (360, 640, 4), "|u1", "cuda:0").cpu().numpy()
    assert np.array_equal(_decode_png_rgba8(tmp_path / "0.png"), srgb)
    h.free()

    # the f16 planar format through the C++ packers and bgs_cloud_upload_f16
    r = subprocess.run([EXAMPLE, "--cloud", str(planes), "--f16", "--width", "640", "--height", "360", "--frames", "4",
                        "--output-dir", str(tmp_path), "--dump-f32", str(tmp_path / "frame16.f32")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    h16 = p.upload(c.to_f16())
    assert np.array_equal(np.fromfile(tmp_path / "frame16.f32", np.float32).reshape(360, 640, 4),
                          p.render(h16, v, CloudSettings()))
    h16.free()


# ---------------------------------------------------------------------------------------------
# include/bgs_host.hpp (f16 packing, sort policy, .ply loader) against the Python mirror — CPU only
# ---------------------------------------------------------------------------------------------
TOOL = os.path.join(ROOT, "tests", "cpp", "host_tool")


@pytest.fixture(scope="module")
def host_tool():
    subprocess.run(["make", "-C", os.path.join(ROOT, "bevy_gaussian_splatting_amd", "csrc"), "-j4"], check=True,
                   capture_output=True)
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-o", TOOL, os.path.join(ROOT, "tests", "cpp", "host_tool.cpp"),
                    "-L" + os.path.join(ROOT, "bevy_gaussian_splatting_amd", "csrc"), "-lbgs",
                    "-Wl,-rpath," + os.path.join(ROOT, "bevy_gaussian_splatting_amd", "csrc")], check=True)
    return TOOL


def _write_planes(path, c):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(c)))
        for a in (c.position_visibility, c.spherical_harmonic, c.rotation, c.scale_opacity):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())


def _read_planes(path):
    raw = open(path, "rb").read()
    (n,) = struct.unpack("<I", raw[:4])
    a = np.frombuffer(raw, np.float32, offset=4)
    pv, sh, rot, so = np.split(a, np.cumsum([4 * n, 48 * n, 4 * n]))
    return pv.reshape(n, 4), sh.reshape(n, 48), rot.reshape(n, 4), so.reshape(n, 4)


def test_cpp_settings_defaults_are_the_c_abi_defaults(host_tool):
    out = subprocess.run([host_tool, "settings"], capture_output=True, text=True, check=True).stdout.split()
    from bevy_gaussian_splatting_amd.settings import BgsSettings
    import ctypes
    assert int(out[0]) == ctypes.sizeof(BgsSettings) and out[1] == "1"


def test_cpp_f32_to_f16_is_ieee_round_to_nearest_even(host_tool, tmp_path):
    rng = np.random.default_rng(5)
    bits = np.concatenate([
        rng.integers(0, 1 << 32, 400_000, dtype=np.uint64).astype(np.uint32),               # anything, NaNs included
        (np.arange(0, 1 << 16, dtype=np.uint32) << np.uint32(13)) + np.uint32(0x38000000),   # around every normal half
        (np.arange(0, 1 << 16, dtype=np.uint32) << np.uint32(13)) + np.uint32(0x38000FFF),
        (np.arange(0, 1 << 16, dtype=np.uint32) << np.uint32(13)) + np.uint32(0x38001000),   # exact ties
        (np.arange(0, 1 << 16, dtype=np.uint32) << np.uint32(13)) + np.uint32(0x38001001),
        np.arange(0x33000000 - 64, 0x38800000, 4099, dtype=np.uint32),                       # the subnormal-half range
        np.array([0, 0x80000000, 0x7F800000, 0xFF800000, 0x477FE000, 0x477FEFFF, 0x477FF000, 0x33000000, 0x33000001],
                 np.uint32)])
    vals = bits.view(np.float32)
    vals.tofile(tmp_path / "in.f32")
    subprocess.run([host_tool, "half", str(tmp_path / "in.f32"), str(tmp_path / "out.u16")], check=True)
    got = np.fromfile(tmp_path / "out.u16", np.uint16)
    with np.errstate(over="ignore", invalid="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    nan = np.isnan(vals)
    assert np.array_equal(got[~nan], want[~nan])
    assert ((got[nan] & 0x7C00) == 0x7C00).all() and ((got[nan] & 0x03FF) != 0).all()   # NaN stays NaN


def test_cpp_f16_cloud_packing_matches_the_python_mirror(host_tool, tmp_path):
    from bevy_gaussian_splatting_amd import random_gaussians_3d_seeded
    c = random_gaussians_3d_seeded(3000, 12)
    _write_planes(tmp_path / "c.bin", c)
    subprocess.run([host_tool, "f16", str(tmp_path / "c.bin"), str(tmp_path / "h.bin")], check=True)
    raw = open(tmp_path / "h.bin", "rb").read()
    n = len(c)
    pv = np.frombuffer(raw, np.float32, 4 * n).reshape(n, 4)
    sh = np.frombuffer(raw, np.uint32, 24 * n, offset=16 * n).reshape(n, 24)
    rso = np.frombuffer(raw, np.uint32, 4 * n, offset=16 * n + 96 * n).reshape(n, 4)
    h = c.to_f16()
    assert np.array_equal(pv, h.position_visibility) and np.array_equal(sh, h.spherical_harmonic)
    assert np.array_equal(rso, h.rotation_scale_opacity)


def test_cpp_sort_trigger_policy_matches_the_python_mirror(host_tool):
    from bevy_gaussian_splatting_amd.sort_policy import SortConfig, SortTrigger, update_sort_trigger
    rng = np.random.default_rng(3)
    t, lines, want = 0.0, [], []
    trig, cfg = SortTrigger(), SortConfig(period_ms=250)
    pos = np.zeros(3, np.float32)
    for k in range(200):
        t += float(rng.choice([0.01, 0.1, 0.3]))
        if rng.random() < 0.4:
            pos = rng.integers(-3, 4, 3).astype(np.float32)
        lines.append(f"{t!r} {pos[0]} {pos[1]} {pos[2]} {k % 3}")
        trig.needs_sort = False
        update_sort_trigger(trig, pos, k % 3, cfg, now=lambda: t)
        want.append(f"{trig.camera_index} {int(trig.needs_sort)}")
    out = subprocess.run([host_tool, "trigger", "250"], input="\n".join(lines) + "\n", capture_output=True, text=True,
                         check=True).stdout.split("\n")
    assert out[:len(want)] == want


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_cpp_ply_loader_matches_the_python_mirror(host_tool, tmp_path, fmt):
    from bevy_gaussian_splatting_amd.io_ply import parse_ply_3d
    rng = np.random.default_rng(17)
    n = 100 if fmt == "ascii" else 1000
    names = ["x", "y", "z", "nx", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "label"]
    types = {nme: "float" for nme in names}
    types["label"] = "uchar"          # a non-float property: ignored by the loader
    cols = {nme: rng.normal(0, 2, n).astype(np.float32) for nme in names}
    cols["scale_2"] = cols["scale_2"] + np.float32(9.0)   # the +-4 clamp around the mean bites
    cols["label"] = rng.integers(0, 255, n).astype(np.uint8)
    header = "ply\nformat %s 1.0\ncomment made by the test\nelement vertex %d\n" % (fmt, n)
    header += "".join(f"property {types[nme]} {nme}\n" for nme in names) + "end_header\n"
    path = tmp_path / "c.ply"
    with open(path, "wb") as f:
        f.write(header.encode())
        if fmt == "ascii":
            for r in range(n):
                f.write((" ".join(repr(float(cols[nme][r])) if types[nme] == "float" else str(int(cols[nme][r]))
                                  for nme in names) + "\n").encode())
        else:
            order = "<" if fmt == "binary_little_endian" else ">"
            rec = np.zeros(n, np.dtype([(nme, order + ("f4" if types[nme] == "float" else "u1")) for nme in names]))
            for nme in names:
                rec[nme] = cols[nme]
            f.write(rec.tobytes())
    subprocess.run([host_tool, "ply", str(path), str(tmp_path / "out.bin")], check=True)
    pv, sh, rot, so = _read_planes(tmp_path / "out.bin")
    ref = parse_ply_3d(str(path))
    assert len(pv) == len(ref) == n + (32 - n % 32)
    assert np.array_equal(pv, ref.position_visibility) and np.array_equal(sh, ref.spherical_harmonic)
    # exp / sigmoid / sqrt come from libm here and from numpy there: equal to a few ulp
    assert np.allclose(rot, ref.rotation, rtol=3e-7, atol=0, equal_nan=True)
    assert np.allclose(so, ref.scale_opacity, rtol=1e-6, atol=0)


def test_cpp_ply_loader_rejects_what_the_reference_rejects(host_tool, tmp_path):
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nend_header\n0 0\n")
    r = subprocess.run([host_tool, "ply", str(bad), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "missing required properties" in r.stderr
    notply = tmp_path / "not.ply"
    notply.write_bytes(b"hello\n")
    r = subprocess.run([host_tool, "ply", str(notply), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "not a PLY file" in r.stderr


def test_cpp_sorted_entries_layout_matches_the_python_mirror(host_tool, tmp_path):
    from bevy_gaussian_splatting_amd.plugin import SortedEntries
    from bevy_gaussian_splatting_amd import SORT_ENTRY_DTYPE
    cams, n = 3, 1000
    out = subprocess.run([host_tool, "entries", str(cams), str(n), str(tmp_path / "e.bin")], capture_output=True, text=True,
                         check=True).stdout.split()
    raw = open(tmp_path / "e.bin", "rb").read()
    camera_count, entry_count = struct.unpack("<QQ", raw[:16])
    got = np.frombuffer(raw, SORT_ENTRY_DTYPE, offset=16)
    ref = SortedEntries.for_cloud(cams, n)
    ref.chunk(cams - 1, n)["key"][0] = 77
    assert (camera_count, entry_count) == (ref.camera_count, ref.entry_count) == (3, 32 * 32)
    assert np.array_equal(got, ref.sorted)
    # out-of-range chunk throws; a camera-count change re-creates the asset (key back to 1)
    assert out == ["1", str((cams + 1) * entry_count), "1"]


def test_cpp_compute_aabb_matches_the_python_mirror(host_tool, tmp_path):
    from bevy_gaussian_splatting_amd import compute_aabb, random_gaussians_3d_seeded
    c = random_gaussians_3d_seeded(5000, 21)
    _write_planes(tmp_path / "c.bin", c)
    out = subprocess.run([host_tool, "aabb", str(tmp_path / "c.bin")], capture_output=True, text=True, check=True).stdout.split()
    mn, mx = compute_aabb(c)
    assert out[0] == "1"
    assert np.array_equal(np.array(out[1:4], np.float32), np.array(mn, np.float32))
    assert np.array_equal(np.array(out[4:7], np.float32), np.array(mx, np.float32))


def test_cpp_gcloud_reader_matches_the_python_codec(host_tool, tmp_path):
    """include/bgs_host.hpp decode_gcloud / load_cloud against bevy_gaussian_splatting_amd.io_gcloud: a file
    of the regular shape (what the writer and serde produce), and irregular ones that serde also accepts —
    structs written as sequences, wider slots, missing fields (Default), an extra key."""
    from bevy_gaussian_splatting_amd import random_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.io_gcloud import _Builder, decode_gcloud, encode_gcloud
    c = random_gaussians_3d_seeded(777, 5)
    c.position_visibility[:, 3] = np.arange(777) % 5
    regular = tmp_path / "regular.gcloud"
    regular.write_bytes(encode_gcloud(c))
    subprocess.run([host_tool, "cloud", str(regular), str(tmp_path / "o.bin")], check=True)
    pv, sh, rot, so = _read_planes(tmp_path / "o.bin")
    assert np.array_equal(pv, c.position_visibility) and np.array_equal(sh, c.spherical_harmonic)
    assert np.array_equal(rot, c.rotation) and np.array_equal(so, c.scale_opacity)

    b = _Builder()
    n = 9
    pvv = [b.vector([b.floats(c.position_visibility[i, :3]), b.f32(float(c.position_visibility[i, 3]))]) for i in range(n)]  # as sequences
    shv = [b.map({"coefficients": b.floats(c.spherical_harmonic[i])}) for i in range(n)]
    rov = [b.map({"rotation": b.floats(c.rotation[i]), "zz_extra": b.uint(70000)}) for i in range(n)]
    sov = [b.map({"scale": b.floats(c.scale_opacity[i, :3])}) for i in range(n)]                     # opacity missing -> 0
    irregular = tmp_path / "irregular.gcloud"
    data = b.finish(b.map({"position_visibility": b.vector(pvv), "spherical_harmonic": b.vector(shv),
                           "rotation": b.vector(rov), "scale_opacity": b.vector(sov)}))
    irregular.write_bytes(data)
    ref = decode_gcloud(data)
    subprocess.run([host_tool, "cloud", str(irregular), str(tmp_path / "o2.bin")], check=True)
    pv, sh, rot, so = _read_planes(tmp_path / "o2.bin")
    assert np.array_equal(pv, ref.position_visibility) and np.array_equal(sh, ref.spherical_harmonic)
    assert np.array_equal(rot, ref.rotation) and np.array_equal(so, ref.scale_opacity)
    assert np.all(so[:, 3] == 0) and np.array_equal(pv[:, :3], c.position_visibility[:n, :3])

    # truncated / foreign bytes are an error, not a crash; other extensions are refused like the reference's loader
    for k, cut in enumerate((3, len(data) // 2, len(data) - 1)):
        bad = tmp_path / f"bad{k}.gcloud"
        bad.write_bytes(data[:cut])
        r = subprocess.run([host_tool, "cloud", str(bad), str(tmp_path / "o3.bin")], capture_output=True, text=True)
        assert r.returncode == 1 and "gcloud" in r.stderr, (cut, r.stderr)
    other = tmp_path / "cloud.splat"
    other.write_bytes(b"x")
    r = subprocess.run([host_tool, "cloud", str(other), str(tmp_path / "o4.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "only .ply and .gcloud supported" in r.stderr


def test_cpp_compute_covariance_3d_matches_the_python_mirror(host_tool, tmp_path):
    from bevy_gaussian_splatting_amd import random_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.gaussian import covariance_3d_opacity
    c = random_gaussians_3d_seeded(500, 8)
    _write_planes(tmp_path / "p.bin", c)
    subprocess.run([host_tool, "cov3d", str(tmp_path / "p.bin"), str(tmp_path / "cov.bin")], check=True)
    got = np.fromfile(tmp_path / "cov.bin", np.float32).reshape(-1, 8)
    assert np.array_equal(got, covariance_3d_opacity(c))


def _trained_like_statistics(pv, sh, rot, so):
    """What makes a cloud 'trained-like' (gaussian.py trained_like_gaussians_3d_seeded), as numbers two generators can be held to."""
    from bevy_gaussian_splatting_amd import compute_covariance_3d
    tang = np.sqrt(so[:, 0] * so[:, 1])
    rgb0 = 0.5 + 0.2820947917738781 * sh[:, :3]
    cov = compute_covariance_3d(rot[:2000], so[:2000, :3])
    S = np.zeros((len(cov), 3, 3))
    for k, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
        S[:, i, j] = S[:, j, i] = cov[:, k]
    w = np.linalg.eigvalsh(S)
    return {"quat_norm_err": float(np.abs(np.linalg.norm(rot, axis=1) - 1.0).max()),
            "eig_vs_scale": float(np.abs(np.sqrt(np.maximum(w, 0.0)) - np.sort(so[:2000, :3], axis=1)).max()),
            "tang_median": float(np.median(tang)), "tang_ln_sigma": float(np.std(np.log(tang))),
            "flat_ratio_median": float(np.median(so[:, 2] / tang)),
            "opaque_share": float((so[:, 3] > 0.6).mean()), "faint_share": float((so[:, 3] < 0.3).mean()),
            "opacity_range": (float(so[:, 3].min()), float(so[:, 3].max())),
            "dc_colour_range": (float(rgb0.min()), float(rgb0.max())), "higher_band_absmax": float(np.abs(sh[:, 3:]).max()),
            "extent": float(np.abs(pv[:, :3]).max()), "visibility": float(pv[:, 3].min())}


def test_trained_like_cloud_has_the_statistics_it_promises_in_both_generators(host_tool, tmp_path):
    """`trained_like_gaussians_3d_seeded` (Python) and `bgs::PlanarGaussian3d::trained_like` (C++): surfaces, log-normal flat
    splats aligned with them, bimodal opacity, DC-dominated SH with colours in [0, 1] — the same statistics from both."""
    from bevy_gaussian_splatting_amd import trained_like_gaussians_3d_seeded
    n = 60_000
    c = trained_like_gaussians_3d_seeded(n, 5)
    assert np.array_equal(c.position_visibility, trained_like_gaussians_3d_seeded(n, 5).position_visibility)   # seeded
    py = _trained_like_statistics(c.position_visibility, c.spherical_harmonic, c.rotation, c.scale_opacity)
    subprocess.run([host_tool, "trained", str(n), "5", str(tmp_path / "t.bin")], check=True)
    cpp = _trained_like_statistics(*_read_planes(tmp_path / "t.bin"))
    for st in (py, cpp):
        assert st["quat_norm_err"] < 1e-6 and st["eig_vs_scale"] < 1e-5          # unit quaternions; Sigma's axes are the scales
        assert 0.35 < st["tang_ln_sigma"] < 0.5 and 0.1 < st["flat_ratio_median"] < 0.2   # (geometric mean of two axes of sigma_ln 0.6)
        assert 0.5 < st["opaque_share"] < 0.62 and 0.3 < st["faint_share"] < 0.42   # bimodal
        assert 0.0 <= st["opacity_range"][0] and st["opacity_range"][1] <= 1.0
        assert 0.049 <= st["dc_colour_range"][0] and st["dc_colour_range"][1] <= 0.951 and st["higher_band_absmax"] < 0.1
        assert st["extent"] < 30.0 and st["visibility"] == 1.0
    assert abs(py["tang_median"] / cpp["tang_median"] - 1.0) < 0.35   # (different patch draws: the area per splat differs a little)
    # the view-dependent part is small next to the DC colour: per channel |sum_k c_k Y_k| <= 1.1 sum_k |c_k| (every basis
    # function of degree 1-3 is below 1.1 in magnitude), a few hundredths for nearly every splat
    dev = 1.1 * np.abs(c.spherical_harmonic[:, 3:].reshape(n, 15, 3)).sum(axis=1)
    assert float(np.quantile(dev, 0.999)) < 0.2 and float(np.median(dev)) < 0.12


def test_cpp_example_takes_the_reference_flag_names():
    """examples/headless.cpp parses the flag names of the reference's `GaussianSplattingViewer` (src/utils.rs:25-74:
    --width / --height / --msaa-samples / --input-cloud / --gaussian-count / --gaussian-seed / --gaussian-mode /
    --rasterization-mode / --radix-sort-depth-bits) with clap's value names (kebab case of the enum variants); a wrong
    value is refused before anything touches a GPU."""
    _build()
    for flag, bad, names in (("--gaussian-mode", "gaussian5d", ("gaussian2d", "gaussian3d")),
                             ("--rasterization-mode", "colour", ("classification", "color", "depth", "normal", "optical-flow", "position", "velocity")),
                             ("--radix-sort-depth-bits", "32", ("bits16", "bits24", "bits32"))):
        r = subprocess.run([EXAMPLE, flag, bad], capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and all(nm in r.stderr for nm in names), (flag, r.stderr)
    src = open(os.path.join(ROOT, "examples", "headless.cpp")).read()
    for flag in ("--width", "--height", "--msaa-samples", "--input-cloud", "--gaussian-count", "--gaussian-seed", "--gaussian-mode",
                 "--rasterization-mode", "--radix-sort-depth-bits"):
        assert f'"{flag}"' in src, flag


@pytest.mark.gpu
def test_cpp_example_with_the_reference_flags_renders_what_the_settings_say(tmp_path):
    """--gaussian-mode gaussian2d --radix-sort-depth-bits bits24 --msaa-samples 1 on a cloud handed over as planes: the
    frame is bit for bit the Python plugin's with the same CloudSettings and Msaa::Off."""
    from bevy_gaussian_splatting_amd import (CloudSettings, GaussianMode, GaussianSplattingPlugin, RadixSortDepthBits, View,
                                             trained_like_gaussians_3d_seeded)
    _build()
    c = trained_like_gaussians_3d_seeded(30_000, 3)
    _write_planes(tmp_path / "c.bin", c)
    r = subprocess.run([EXAMPLE, "--cloud", str(tmp_path / "c.bin"), "--width", "480", "--height", "270", "--frames", "4",
                        "--gaussian-mode", "gaussian2d", "--radix-sort-depth-bits", "bits24", "--msaa-samples", "1",
                        "--output-dir", str(tmp_path), "--dump-f32", str(tmp_path / "frame.f32")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "frame.f32", np.float32).reshape(270, 480, 4)
    p = GaussianSplattingPlugin(0)
    h = p.upload(c)
    want = p.render(h, View.headless(480, 270, msaa_samples=1),
                    CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, radix_sort_depth_bits=RadixSortDepthBits.Bits24))
    assert np.array_equal(got, want)
    h.free()
    p.close()


@pytest.mark.gpu
def test_cpp_example_loads_a_gcloud_file(tmp_path):
    """examples/headless --input-cloud x.gcloud (the reference viewer's flag; loader dispatch src/io/loader.rs:22-61):
    the C++ reader's cloud renders to the same bits as the Python reader's."""
    from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.io_gcloud import read_gcloud, write_gcloud

    _build()
    c = random_gaussians_3d_seeded(20_000, 14)
    path = tmp_path / "cloud.gcloud"
    write_gcloud(c, str(path))
    r = subprocess.run([EXAMPLE, "--input-cloud", str(path), "--width", "480", "--height", "270", "--frames", "6",
                        "--output-dir", str(tmp_path), "--dump-f32", str(tmp_path / "frame.f32")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "frame.f32", np.float32).reshape(270, 480, 4)
    p = GaussianSplattingPlugin(0)
    h = p.upload(read_gcloud(str(path)))
    assert np.array_equal(got, p.render(h, View.headless(480, 270), CloudSettings()))
    h.free()
    p.close()


@pytest.mark.parametrize("fmt", ["binary_little_endian", "binary_big_endian", "ascii"])
def test_ply_with_a_trailing_face_element_loads_in_both_hosts(host_tool, tmp_path, fmt):
    """A mesh-style export: the vertex element followed by `element face` with a list property. ply-rs parses
    it and the reference ignores it (src/io/ply.rs:76-91 reads only "vertex"); so do the Python and the C++
    loader, binary list rows included."""
    from bevy_gaussian_splatting_amd.io_ply import parse_ply_3d
    rng = np.random.default_rng(4)
    n, faces = 70, 13
    names = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    cols = {k: rng.normal(0, 1, n).astype(np.float32) for k in names}
    header = f"ply\nformat {fmt} 1.0\nelement vertex {n}\n" + "".join(f"property float {k}\n" for k in names)
    header += f"element face {faces}\nproperty list uchar int vertex_indices\nproperty uchar flags\nend_header\n"
    path = tmp_path / "mesh.ply"
    order = ">" if fmt == "binary_big_endian" else "<"
    with open(path, "wb") as f:
        f.write(header.encode())
        if fmt == "ascii":
            for r in range(n):
                f.write((" ".join(repr(float(cols[k][r])) for k in names) + "\n").encode())
            for r in range(faces):
                k = 3 + r % 3
                f.write((f"{k} " + " ".join(str(int(v)) for v in rng.integers(0, n, k)) + " 7\n").encode())
        else:
            rec = np.zeros(n, np.dtype([(k, order + "f4") for k in names]))
            for k in names:
                rec[k] = cols[k]
            f.write(rec.tobytes())
            for r in range(faces):
                k = 3 + r % 3
                f.write(bytes([k]) + rng.integers(0, n, k).astype(order + "i4").tobytes() + bytes([7]))
    ref = parse_ply_3d(str(path))
    assert len(ref) == 96 and np.array_equal(ref.position_visibility[:n, 0], cols["x"])
    subprocess.run([host_tool, "ply", str(path), str(tmp_path / "o.bin")], check=True)
    pv, sh, rot, so = _read_planes(tmp_path / "o.bin")
    assert np.array_equal(pv, ref.position_visibility) and np.array_equal(sh, ref.spherical_harmonic)
    assert np.allclose(rot, ref.rotation, rtol=3e-7, atol=0, equal_nan=True) and np.allclose(so, ref.scale_opacity, rtol=1e-6, atol=0)
    if fmt != "ascii":   # truncated inside the face element: an error, not a crash
        data = path.read_bytes()
        (tmp_path / "cut.ply").write_bytes(data[:-5])
        with pytest.raises(ValueError):
            parse_ply_3d(str(tmp_path / "cut.ply"))
        r = subprocess.run([host_tool, "ply", str(tmp_path / "cut.ply"), str(tmp_path / "o2.bin")], capture_output=True, text=True)
        assert r.returncode == 1 and "truncated" in r.stderr


def test_gcloud_general_reader_reports_malformed_input_as_value_error():
    from bevy_gaussian_splatting_amd import random_gaussians_3d_seeded
    from bevy_gaussian_splatting_amd.io_gcloud import decode_gcloud, encode_gcloud, flexbuffers_loads
    data = encode_gcloud(random_gaussians_3d_seeded(40, 3))
    rng = np.random.default_rng(0)
    for trial in range(300):
        bad = bytearray(data[: int(rng.integers(3, len(data)))]) if trial % 2 else bytearray(data)
        for _ in range(4):
            bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
        for fn in (flexbuffers_loads, lambda b: decode_gcloud(b, fast=False), decode_gcloud):
            try:
                fn(bytes(bad))
            except ValueError:
                pass   # the only exception type callers have to expect
