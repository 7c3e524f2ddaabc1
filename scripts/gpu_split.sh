#!/bin/bash
R=$GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
echo "baseline"; for d in 3 4; do python $R/scripts/loop_pipelined.py $d 300; done
for f in 32 64 96 128; do for ra in 0 1; do
  echo "== BGS_FRONT_CUS=$f RASTER_ALL=$ra"
  for d in 2 3 4 6; do BGS_FRONT_CUS=$f BGS_RASTER_ALL=$ra timeout 120 python $R/scripts/loop_pipelined.py $d 300; done
done; done
echo "== correctness under split"; BGS_FRONT_CUS=64 timeout 600 python -m pytest $R/tests/test_gpu_parity.py -m gpu -q -x -k "pipelined or async or parity_small" 2>&1 | tail -3
