"""The C-ABI library: loads, exports every symbol include/bgs.h declares, fails loudly
without a device, and its two pure-host helpers agree with the Python mirror. No compute."""
import ctypes
import os
import re

import numpy as np
import pytest

from bevy_gaussian_splatting_amd import CloudSettings, View, _native
from bevy_gaussian_splatting_amd.camera import BgsView
from bevy_gaussian_splatting_amd.settings import BgsSettings

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared(headers=("bgs.h", "bgs_diag.h")):
    """Entry points declared by the seam (include/bgs.h) and the diagnostics header (include/bgs_diag.h)."""
    names = set()
    for hname in headers:
        text = open(os.path.join(ROOT, "include", hname)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(bgs_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_the_seam_header_carries_no_diagnostics():
    """include/bgs.h is the drop-in boundary: context, upload, sort, render, targets, frame pipeline. Test hooks,
    counters and experiment switches live in include/bgs_diag.h."""
    seam, diag = set(_declared(("bgs.h",))), set(_declared(("bgs_diag.h",)))
    assert not seam & diag
    assert {"bgs_create", "bgs_cloud_upload_f32", "bgs_sort", "bgs_render", "bgs_pipeline_pop"} <= seam
    assert {"bgs_set_debug_flags", "bgs_set_tile_trace", "bgs_selftest_ln_f32", "bgs_hbm_probe", "bgs_radix_sort_pairs",
            "bgs_set_queue_holders", "bgs_adaptive_counters", "bgs_graph_counters", "bgs_learning_counters",
            "bgs_tile_order_counters", "bgs_selftest_tile_order"} == diag
    # the seam proper, plus the multi-GPU frame gather ("ncclGather ... behind the boundary")
    assert {n for n in seam if n.startswith("bgs_comm_")} == {"bgs_comm_unique_id", "bgs_comm_create", "bgs_comm_gather", "bgs_comm_gather_after",
                                                               "bgs_comm_wait", "bgs_comm_stream", "bgs_comm_destroy"}
    assert len(seam) <= 48


def test_library_is_built_and_exports_every_declared_symbol():
    lib = _native.load()  # raises ImportError if the HIP extension is not built
    names = _declared()
    assert len(names) >= 28
    assert set(names) == set(_native.EXPORTED_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), f"libbgs.so does not export {n}"
    assert lib.bgs_version() == (0 << 16) | 4 == _native.ABI_VERSION


def test_the_library_exports_the_c_abi_and_nothing_else():
    """The host side is several translation units (round 6) whose shared C++ functions must stay inside the library: its
    dynamic symbol table defines exactly the entry points the two headers declare (csrc/libbgs.map)."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    lib = os.path.join(ROOT, "bevy_gaussian_splatting_amd", "csrc", "libbgs.so")
    _native.load()   # (builds it if it is stale)
    out = subprocess.run([nm, "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    defined = {line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip()}
    assert defined == set(_declared()), sorted(defined ^ set(_declared()))


def test_integration_doc_binds_every_exported_symbol():
    """INTEGRATION.md shows the reference-side binding (the `extern "C"` block a maintainer would add): it
    has to name every entry point the header declares, and nothing the library does not have."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    bound = set(re.findall(r"pub fn (bgs_[a-z0-9_]+)\(", block))
    assert bound == set(_declared())


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(BgsView) == (16 * 4 + 8 + 16 + 4) * 4 + 16 and BgsView.depth_device_ptr.offset % 8 == 0
    assert BgsView.sample_count.offset == (16 * 4 + 8 + 16 + 1) * 4
    assert ctypes.sizeof(BgsSettings) == (16 + 2 + 8 + 1 + 1 + 2 + 8) * 4
    assert ctypes.sizeof(_native.BgsSortEntry) == 8
    assert ctypes.sizeof(_native.BgsStats) == 6 * 4 + 4 + 4 + 4 + 4 + 8 + 8 + 8 + 4 * 4 + 8 + 8 + 8 + 8 + 8
    # the ctypes images include natural padding exactly like the C structs
    assert _native.BgsStats.instance_count.offset % 8 == 0


def test_abi_check_refuses_a_stale_binding():
    """bgs_abi_check is the binding's handshake (bgs_view grew by 16 bytes between 0.2 and 0.3 and
    nothing noticed a caller with the short struct): the right version and sizes pass, anything else is BGS_EINVAL
    with a message, and _native.load() performs it."""
    lib = _native.load()
    sizes = (ctypes.sizeof(BgsView), ctypes.sizeof(BgsSettings), ctypes.sizeof(_native.BgsStats))
    assert lib.bgs_abi_check(_native.ABI_VERSION, *sizes) == _native.BGS_OK
    assert lib.bgs_abi_check((0 << 16) | 2, *sizes) == _native.BGS_EINVAL
    assert b"built against 0.2" in lib.bgs_last_error(None)
    assert lib.bgs_abi_check(_native.ABI_VERSION, sizes[0] - 16, sizes[1], sizes[2]) == _native.BGS_EINVAL   # 0.2's bgs_view
    assert b"stale binding" in lib.bgs_last_error(None)
    assert lib.bgs_abi_check(_native.ABI_VERSION, sizes[0], sizes[1], sizes[2] + 8) == _native.BGS_EINVAL


def test_settings_default_equals_cloud_settings_default():
    lib = _native.load()
    s = BgsSettings()
    lib.bgs_settings_default(ctypes.byref(s))
    py = CloudSettings().to_native()
    for name, _ in BgsSettings._fields_:
        a, b = getattr(s, name), getattr(py, name)
        if name in ("transform", "position_min", "position_max", "reserved"):
            assert list(a) == list(b)
        else:
            assert a == b, name


def test_view_perspective_matches_python_mirror():
    lib = _native.load()
    v = View.headless(1920, 1080, yaw=0.4)
    out = BgsView()
    wfv = v.to_native().world_from_view
    lib.bgs_view_perspective(wfv, ctypes.c_float(np.pi / 4), ctypes.c_float(0.1), 1920, 1080, ctypes.byref(out))
    ref = v.to_native()
    assert abs(out.delta_time - 1 / 60) < 1e-7
    # Camera3d::default() carries Bevy's default Msaa (Sample4) and no scene depth
    assert out.sample_count == ref.sample_count == 4 and out.depth_device_ptr == 0
    for name in ("world_from_view", "view_from_world", "clip_from_view", "clip_from_world", "viewport", "clear_color",
                 "previous_clip_from_world"):
        assert np.allclose(list(getattr(out, name)), list(getattr(ref, name)), rtol=1e-5, atol=1e-6), name


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is covered on the CPU box")
    lib = _native.load()
    ctx = ctypes.c_void_p()
    rc = lib.bgs_create(0, ctypes.byref(ctx))
    assert rc == _native.BGS_EHIP and not ctx.value
    assert b"no usable HIP device" in lib.bgs_last_error(None)
    from bevy_gaussian_splatting_amd import GaussianSplattingPlugin
    with pytest.raises(_native.BgsError):
        GaussianSplattingPlugin(0)


def test_product_package_never_touches_the_oracle():
    """The product path must not import/link/execute anything under oracle/."""
    pkg = os.path.join(ROOT, "bevy_gaussian_splatting_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "bgs_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_header_is_plain_c_and_the_cpp_layer_is_standard_cpp17(tmp_path):
    """include/bgs.h must compile as strict C99 (it is the FFI surface a Rust / cgo / ctypes binding reads)
    and include/bgs.hpp + bgs_host.hpp as warning-free C++17 with no HIP headers on the include path."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = tmp_path / "abi.c"
    c.write_text('#include "bgs.h"\n#include "bgs_diag.h"\nint main(void) { bgs_settings s; bgs_settings_default(&s); '
                 "return (int)sizeof(bgs_view) == 0; }\n")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                    "-c", str(c), "-o", str(tmp_path / "abi.o")], check=True)
    cpp = tmp_path / "host.cpp"
    cpp.write_text('#include "bgs_host.hpp"\nint main() { bgs::CloudSettings s; return (int)s.to_native().sh_degree - 3; }\n')
    subprocess.run(["g++", "-std=c++17", "-pedantic", "-Wall", "-Wextra", "-Wshadow", "-Werror", "-I",
                    os.path.join(root, "include"), "-c", str(cpp), "-o", str(tmp_path / "host.o")], check=True)


def test_library_is_built_from_this_trees_kernel_sources(tmp_path):
    """`bgs_build_id()` = SHA-256 over csrc/*.hip + csrc/*.h (`_build_id.py`), compiled in by the Makefile; the id is
    also readable from the file's bytes without loading it, and a library carrying another id is refused."""
    from bevy_gaussian_splatting_amd import _build_id
    lib = _native.load()
    want = _build_id.kernel_source_sha256()
    assert lib.bgs_build_id().decode() == want and len(want) == 64
    assert _build_id.library_build_id(_native.LIB_PATH) == want
    stale = tmp_path / "libbgs_stale.so"
    data = open(_native.LIB_PATH, "rb").read()
    stale.write_bytes(data.replace(want.encode(), b"0" * 64))
    assert _build_id.library_build_id(str(stale)) == "0" * 64
    assert _build_id.library_build_id(str(tmp_path / "missing.so")) is None
    # the loader's verdict on such a file (auto-build off): refused, with both ids in the message
    import importlib
    old_path, old_env = _native.LIB_PATH, os.environ.get("BGS_NO_AUTOBUILD")
    os.environ["BGS_NO_AUTOBUILD"] = "1"
    _native.LIB_PATH = str(stale)
    try:
        with pytest.raises(ImportError) as ei:
            _native.ensure_current()
        assert "000000000000" in str(ei.value) and want[:12] in str(ei.value)
    finally:
        _native.LIB_PATH = old_path
        if old_env is None:
            del os.environ["BGS_NO_AUTOBUILD"]
        else:
            os.environ["BGS_NO_AUTOBUILD"] = old_env
    assert _native.ensure_current() == want
