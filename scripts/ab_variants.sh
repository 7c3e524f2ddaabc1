#!/bin/bash
# same-box A/B of ab/*.so on the headline frame (single stream + 6 lanes / 3 streams) and on the surfel frames,
# then the GPU suite on the variant named by $1 (if any): bash scripts/ab_variants.sh [variant-for-tests]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_variants; mkdir -p $O
LIB=$R/bevy_gaussian_splatting_amd/csrc/libbgs.so
cp $LIB /tmp/libbgs_orig.so
for rep in 1 2; do
for v in $R/ab/*.so; do
  cp $v $LIB
  echo "== $(basename $v) rep $rep"
  python $R/scripts/loop_pipelined.py 1 400 1.0
  python $R/scripts/loop_pipelined.py 6 600 1.0 0 0 3
  python $R/scripts/loop_pipelined.py 6 600 0.05 0 0 3
done
done 2>&1 | tee $O/headline.log
for v in $R/ab/*.so; do
  cp $v $LIB
  echo "== $(basename $v)"
  python $R/scripts/bench_surfel.py 0 0x40
done 2>&1 | tee $O/surfel.log
if [ -n "$1" ]; then
  cp $R/ab/$1.so $LIB
  cd $R && timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 | tee $O/pytest_$1.log
fi
cp /tmp/libbgs_orig.so $LIB
