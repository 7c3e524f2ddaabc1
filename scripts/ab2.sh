#!/bin/bash
# like ab.sh, on both the dense (1.0) and the scene-like (0.05) headline frame: bash scripts/ab2.sh [frames] [reps]
R=$GRAFT_REPO_ROOT
cp $R/bevy_gaussian_splatting_amd/csrc/libbgs.so /tmp/libbgs_orig.so
for rep in $(seq 1 ${2:-2}); do
for v in $R/ab/*.so; do
  cp $v $R/bevy_gaussian_splatting_amd/csrc/libbgs.so
  echo "== $(basename $v) rep $rep"
  python $R/scripts/loop_pipelined.py 1 ${1:-400} 1.0
  python $R/scripts/loop_pipelined.py 6 ${1:-400} 1.0 0 0 3
  python $R/scripts/loop_pipelined.py 1 ${1:-400} 0.05
  python $R/scripts/loop_pipelined.py 6 ${1:-400} 0.05 0 0 3
done
done
cp /tmp/libbgs_orig.so $R/bevy_gaussian_splatting_amd/csrc/libbgs.so
