#!/bin/bash
# Same-box A/B of library variants on the blocking bgs_sort call: bash scripts/sort_variants.sh "<n> <mode> <flags>" [rounds]
R=$GRAFT_REPO_ROOT
shopt -s nullglob   # (no variants: the tree alone)
for rep in $(seq 1 ${2:-2}); do
  echo "== tree"; python $R/scripts/sort_rates.py ${1:-5000000 rayon 0} 2>&1 | grep -v -E "amdgpu.ids|BGS_LIB_OVERRIDE"
  for v in $R/gpurun_variants/*.so; do
    echo "== $(basename $v)"; BGS_LIB_OVERRIDE=$v python $R/scripts/sort_rates.py ${1:-5000000 rayon 0} 2>&1 | grep -v -E "amdgpu.ids|BGS_LIB_OVERRIDE"
  done
done
