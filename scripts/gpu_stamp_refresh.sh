#!/bin/bash
# A SHORT evidence run for a tree whose kernels are those of the last full set (scripts/gpu_final.sh) but whose source hash
# moved (host-side changes): all GPU tests, smoke, the bench lines, rocprofv3 kernel stats + the PMC passes of the dense
# headline frame, the counter file with the new stamp, the bench line carrying it.
#   bash scripts/gpu_stamp_refresh.sh <tag>     then copy gpurun_out/<tag> to profiles/<tag> and its pmc_traffic.json to profiles/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r6_stamp}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q --timeout 1200 -s 2>&1 | grep -E "passed|failed|error|Error|libbgs build id|tolerance accounting" | tee $OUT/pytest_gpu_summary.log | tail -4
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json; tail -2 $OUT/bench.err
for i in 1 2; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_flags_$i.json 2>/dev/null; cut -c1-160 $OUT/bench_driver_flags_$i.json; done
echo "== rocprof"; bash scripts/gpu_profile.sh > $OUT/profile.log 2>&1; cp $R/gpurun_out/prof/* $OUT/ 2>/dev/null; tail -4 $OUT/profile.log
python $R/scripts/make_pmc_traffic.py "$R/gpurun_out/pmc_dense/counters.txt" "$OUT/pmc_traffic.json" "$TAG" | cut -c1-300
echo "== bench with the PMC stamp"; cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_stamped.json 2>/dev/null; python -c "
import json
d=json.load(open('$OUT/bench_stamped.json')); print(d['value'], d['roofline']['frac'], d['roofline'].get('traffic'))"
echo "== sort rates"; timeout 600 python scripts/sort_rates.py 2>&1 | grep -v amdgpu.ids | grep "flags      0x0" | tee $OUT/sort_rates_default.txt | cut -c1-170
ls $OUT
