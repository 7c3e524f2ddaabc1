"""Identity of the kernel sources: SHA-256 over every HIP source and header of libbgs (csrc/*.hip, csrc/*.h) and the
Makefile that holds the compiler flags they are built with.

The Makefile compiles it into libbgs.so (`bgs_build_id()`, and as the byte string `BGS_BUILD_ID=<hex>` so that it can
be read without loading the library); `_native.load()` refuses a library built from other sources and rebuilds it
instead, so a stale prebuilt binary can never be what the tests or `bench.py` ran. Counter files under `profiles/`
are stamped with the same hash. Not the commit id: that also changes with every documentation commit.

`python bevy_gaussian_splatting_amd/_build_id.py` prints the hash (used by csrc/Makefile)."""
from __future__ import annotations

import hashlib
import os
from typing import Optional

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
MARKER = b"BGS_BUILD_ID="


def kernel_source_sha256() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")) or name == "Makefile":   # (build_id.inc is generated FROM this hash: not a source)
            h.update(name.encode())
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def library_build_id(path: str) -> Optional[str]:
    """The id compiled into a libbgs.so, read from the file's bytes (no dlopen); None if there is none."""
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return None
    at = data.find(MARKER)
    while at >= 0:
        hexid = data[at + len(MARKER): at + len(MARKER) + 64]
        if len(hexid) == 64 and all(c in b"0123456789abcdef" for c in hexid):
            return hexid.decode()
        at = data.find(MARKER, at + 1)
    return None


if __name__ == "__main__":
    print(kernel_source_sha256())
