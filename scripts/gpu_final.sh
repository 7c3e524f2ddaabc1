#!/bin/bash
# Round-end evidence run: smoke, all GPU tests, bench, other configs, rocprofv3 stats + PMC, timelines.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2_final}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -s 2>&1 | grep -E "passed|failed|error|ambiguity slack used" | tee $OUT/pytest_gpu_summary.log | tail -8
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -2 $OUT/bench.err
echo "== other configs"; timeout 900 python scripts/bench_configs.py > $OUT/bench_other_configs.json 2> $OUT/bench_other.err; tail -c 1500 $OUT/bench_other_configs.json; tail -2 $OUT/bench_other.err
echo "== rocprof"; bash scripts/gpu_profile.sh > $OUT/profile.log 2>&1; cp $R/gpurun_out/prof/* $OUT/ 2>/dev/null; tail -30 $OUT/profile.log
python $R/scripts/make_pmc_traffic.py "$R/gpurun_out/pmc_dense/counters.txt" "$OUT/pmc_traffic.json" "$TAG"
echo "== timelines"; bash scripts/gpu_timeline.sh 2>&1 | tail -6; cp $R/gpurun_out/timeline_depth*.json $OUT/
ls -la $OUT
