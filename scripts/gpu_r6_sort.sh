#!/bin/bash
# round 6: the digit passes (onesweep_kernel) — unit tests, then blocking bgs_sort rates on the passes (flag 0x80000) and the
# instance-sort pipeline's tile sort, tree vs every library in gpurun_variants/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
shopt -s nullglob   # (no variants: the tree alone)
python -m pytest $R/tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "onesweep or sort_bit_exact or full_size_sort or bucket_sort" 2>&1 | tail -2
run() {
  for n in 1000000 5000000; do python $R/scripts/sort_rates.py $n rayon 0x80000 2>&1 | grep -v amdgpu.ids; done
  python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, View, random_gaussians_3d_seeded
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(1_000_000, 2))
p.set_binning("sort"); p.set_profiling_stride(1)
v, s = View.headless(1920, 1080), CloudSettings()
for _ in range(3): p.render(h, v, s, download=False)
acc = {}
for _ in range(6):
    p.render(h, v, s, download=False)
    for k, x in p.stats()["stage_ms"].items(): acc[k] = acc.get(k, 0.0) + x / 6
st = p.stats()
print("instance-sort pipeline, dense 1 M:", {k: round(x, 4) for k, x in acc.items()}, "instances", st["instance_count"],
      "tile sort GB/s", round(st["instance_count"] * 32 / (acc["tile_sort"] * 1e-3) / 1e9, 1))
PY
}
echo "== tree"; run
for v in $R/gpurun_variants/*.so; do echo "== $(basename $v)"; export BGS_LIB_OVERRIDE=$v; run; unset BGS_LIB_OVERRIDE; done
