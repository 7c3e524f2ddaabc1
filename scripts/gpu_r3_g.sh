#!/bin/bash
# Round 3: A/B of library variants (late SH fetch in project_bin; the same at a forced 128-VGPR budget), checked bit-identical first
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r3_g}; rm -rf $OUT; mkdir -p $OUT
cd $R
for v in gpurun_variants/*.so; do
BGS_LIB_OVERRIDE=$R/$v python - <<'PY' 2>&1 | grep -v -E "amdgpu.ids"
import numpy as np, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded, GaussianMode
from oracle import oracle
p = GaussianSplattingPlugin(0)
c = random_gaussians_3d_seeded(60_000, 9)
for cloud in (c, c.to_f16()):
    h = p.upload(cloud)
    dec = oracle.decode_f16(cloud) if cloud is not c else c
    for s in (CloudSettings(global_scale=0.5), CloudSettings(global_scale=0.5, sh_degree=1, aabb=True), CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True, global_scale=0.5)):
        v = View.headless(480, 270)
        img = p.render(h, v, s)
        e = oracle.sort(dec, v, s)
        ref, amb = oracle.render(dec, e, v, s, with_ambiguity=True)
        err = np.abs(img - ref); lim = 1e-3 + 1e-4 * np.abs(ref) + amb[..., None]
        assert (err <= lim).all(), (err.max())
    h.free()
print("variant ok vs oracle:", os.environ["BGS_LIB_OVERRIDE"])
PY
done
bash scripts/ab_variants.sh "dense scene 5m_dense 5m_scene surfel" 2 2>&1 | tee $OUT/ab_variants_sh_late.log
