#!/bin/bash
# ab2.sh + the 5 M f16 frame (6 lanes / 3 streams): bash scripts/ab3.sh [frames] [reps]
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
  N=5000000 F16=1 python $R/scripts/loop_pipelined.py 6 300 1.0 0 0 3
  N=5000000 F16=1 python $R/scripts/loop_pipelined.py 6 300 0.05 0 0 3
done
done
cp /tmp/libbgs_orig.so $R/bevy_gaussian_splatting_amd/csrc/libbgs.so
