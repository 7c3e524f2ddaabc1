#!/bin/bash
# Same-box A/B of library VARIANTS built from modified sources (what scripts/ab.sh did in round 2): every
# gpurun_variants/*.so is loaded through BGS_LIB_OVERRIDE (experiments only: bench.py and the tests refuse it),
# next to the tree's own library.   bash scripts/ab_variants.sh "<configs>" [rounds]
R=$GRAFT_REPO_ROOT
shopt -s nullglob   # (no variants: the tree alone)
for rep in $(seq 1 ${2:-2}); do
  echo "== tree"; python $R/scripts/ab_flags.py "${1:-dense scene}" 0 1 2>&1 | grep -v -E "amdgpu.ids|BGS_LIB_OVERRIDE"
  for v in $R/gpurun_variants/*.so; do
    echo "== $(basename $v)"; BGS_LIB_OVERRIDE=$v python $R/scripts/ab_flags.py "${1:-dense scene}" 0 1 2>&1 | grep -v -E "amdgpu.ids|BGS_LIB_OVERRIDE"
  done
done
