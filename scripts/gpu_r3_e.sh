#!/bin/bash
# Round 3, fifth GPU call: keygen as 512 threads x 8 (debug flag 0x8000000) vs 256 x 16, checked bit-exact first.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3_e}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import numpy as np
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded, SortMode
p = GaussianSplattingPlugin(0)
v = View.headless(1920, 1080)
for n in (600_000, 1_000_000, 2_500_001):
    c = random_gaussians_3d_seeded(n, 5); h = p.upload(c)
    for s in (CloudSettings(), CloudSettings(sort_mode=SortMode.Rayon)):
        p.set_debug_flags(0); p.reset_adaptive_state()
        a = p.sort(h, v, s); ia = p.render(h, v, s); ia2 = p.render(h, v, s)
        p.set_debug_flags(0x8000000); p.reset_adaptive_state()
        b = p.sort(h, v, s); ib = p.render(h, v, s); ib2 = p.render(h, v, s)
        assert np.array_equal(a["key"], b["key"]) and np.array_equal(a["index"], b["index"]), (n, s.sort_mode)
        assert np.array_equal(ia, ib) and np.array_equal(ia2, ib2), (n, s.sort_mode)
    h.free()
p.set_debug_flags(0)
print("512-thread keygen: sort entries and images bit-identical to the 256-thread one")
PY
echo "== A/B"; timeout 900 python scripts/ab_flags.py "dense scene 5m_dense 5m_scene" "0,0x8000000" 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_keygen_512.log
