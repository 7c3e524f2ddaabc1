#!/bin/bash
# same-box A/B of prebuilt library variants: ab/*.so are swapped in for libbgs.so one after another
# bash scripts/ab.sh [frames] [global_scale] [reps]
R=$GRAFT_REPO_ROOT
cp $R/bevy_gaussian_splatting_amd/csrc/libbgs.so /tmp/libbgs_orig.so
for rep in $(seq 1 ${3:-2}); do
for v in $R/ab/*.so; do
  cp $v $R/bevy_gaussian_splatting_amd/csrc/libbgs.so
  echo "== $(basename $v) rep $rep"
  python $R/scripts/loop_pipelined.py 1 ${1:-400} ${2:-1.0}
  python $R/scripts/loop_pipelined.py 6 ${1:-400} ${2:-1.0} 0 0 3
done
done
cp /tmp/libbgs_orig.so $R/bevy_gaussian_splatting_amd/csrc/libbgs.so
