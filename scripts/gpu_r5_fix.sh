#!/bin/bash
# Round 5, after gpu_final.sh r5_v1: the bench line with the (re-parsed) PMC stamp, and the medium / surfel sweeps again
# with the accounting test's classes (the parity tests of those sweeps passed in r5_v1; its cap on the excess was the
# fixed configurations' for the randomized ones too).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5_v1
cd $R
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_stamped.json 2>/dev/null; python -c "
import json
d=json.load(open('$OUT/bench_stamped.json')); print(d['value'], json.dumps(d['roofline'].get('valu'))[:900], d['roofline'].get('traffic'))"
SMALL=2000 MEDIUM=350 SURFEL=120 bash scripts/gpu_sweeps.sh r5_v1 2>&1 | tail -n 8
cp $R/gpurun_out/tolerance_accounting_band_2e-3.json $OUT/tolerance_accounting_surfel_sweep_band_2e-3.json
BGS_ORACLE_EDGE_BAND_PX=5e-4 BGS_RANDOM_SEED_BASE=80000 SMALL=2500 MEDIUM=400 SURFEL=150 bash scripts/gpu_sweeps.sh r5_v1/explore_80000 2>&1 | tail -n 8
cp $R/gpurun_out/tolerance_accounting_band_5e-4.json $OUT/explore_80000/
