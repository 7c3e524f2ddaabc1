#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (all)"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -60
echo "== rocprof kernel trace"
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err
cd $GRAFT_REPO_ROOT
find /tmp/prof1 -name "*stats*" | head; 
for f in $(find /tmp/prof1 -name "*kernel_stats*.csv"); do cp $f gpurun_out/r1_kernel_stats.csv; head -30 $f; done
