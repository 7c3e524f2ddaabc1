#!/bin/bash
# round 2, depth-sort iteration: new parity tests, A/B of the bucket sort against the onesweep passes
# (debug flag 0x80000), rocprofv3 kernel stats of both.
mkdir -p gpurun_out/r2_sort
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_sort
echo "== pytest gpu (subset: ${1:-default})"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -x -s \
  -k "${1:-bucket or supertile_list or sort_bit_exact or full_size_sort or collisions or default_scale or async_frames or frame_graphs}" 2>&1 | tail -40 | tee $O/pytest_tail.txt
for gs in 1.0 0.05; do
for fl in 0 0x80000; do
  python $R/scripts/loop_pipelined.py 1 400 $gs 0 $fl
  python $R/scripts/loop_pipelined.py 6 600 $gs 0 $fl 3
done
done 2>&1 | tee $O/loops.txt
cd /tmp
for fl in 0 0x80000; do
  rm -rf /tmp/st$fl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st$fl -o t -- python $R/scripts/loop_pipelined.py 1 300 1.0 0 $fl > /tmp/st$fl.out 2>/tmp/st$fl.err
  cat /tmp/st$fl.out
  f=$(find /tmp/st$fl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_flags_$fl.csv && head -12 $f | cut -c1-200
done
