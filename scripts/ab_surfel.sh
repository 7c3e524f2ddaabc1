#!/bin/bash
# same-box A/B of ab/*.so on the 2DGS surfel frames (stage times, dense + scene-like)
R=$GRAFT_REPO_ROOT
LIB=$R/bevy_gaussian_splatting_amd/csrc/libbgs.so
cp $LIB /tmp/libbgs_orig.so
for rep in 1 2; do
for v in $R/ab/*.so; do
  cp $v $LIB
  echo "== $(basename $v) rep $rep"
  python $R/scripts/bench_surfel.py 0
done
done
cp /tmp/libbgs_orig.so $LIB
