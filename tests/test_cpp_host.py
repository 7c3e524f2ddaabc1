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
