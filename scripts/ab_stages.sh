#!/bin/bash
# same-box A/B of ab/*.so: per-stage kernel times (1 M f32 + 5 M f16 frames) and pipelined frames/s
R=$GRAFT_REPO_ROOT
LIB=$R/bevy_gaussian_splatting_amd/csrc/libbgs.so
cp $LIB /tmp/libbgs_orig.so
for v in $R/ab/*.so; do
  cp $v $LIB
  echo "== $(basename $v)"
  python $R/scripts/stage_times.py ${1:-5000000}
  python $R/scripts/loop_pipelined.py 6 600 1.0 0 0 3
done
cp /tmp/libbgs_orig.so $LIB
