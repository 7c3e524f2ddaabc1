#!/bin/bash
# The three randomized parity sweeps (defaults 600 small, 200 medium, 120 medium forced onto the surfel path; evidence size since round 3: SMALL=2000 MEDIUM=350 SURFEL=120):
# bash scripts/gpu_sweeps.sh <tag>   -> gpurun_out/<tag>/randomized_*.log   (SMALL= MEDIUM= SURFEL= override the seed counts, BGS_RANDOM_SEED_BASE=B starts at seed B;
# MEDIUM=350 includes seed 321, round 2's one failure (fixed in round 3 by the correctly rounded log)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-sweeps}; mkdir -p $OUT
cd $R
BGS_RANDOM_SEEDS=${SMALL:-600} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_randomized_configurations and not medium or test_zz" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|tolerance accounting|^ +[a-z].*excess" | tail -n 30 > $OUT/randomized_small_seeds.log
BGS_RANDOM_MEDIUM_SEEDS=${MEDIUM:-200} timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_randomized_configurations_medium or test_zz" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|tolerance accounting|^ +[a-z].*excess" | tail -n 30 > $OUT/randomized_medium_seeds.log
BGS_RANDOM_FORCE_SURFEL=1 BGS_RANDOM_MEDIUM_SEEDS=${SURFEL:-120} timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_randomized_configurations_medium or test_zz" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|tolerance accounting|^ +[a-z].*excess" | tail -n 30 > $OUT/randomized_medium_surfel_seeds.log
tail -n 2 $OUT/randomized_*.log
