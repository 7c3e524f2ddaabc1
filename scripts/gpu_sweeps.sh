#!/bin/bash
# The three randomized parity sweeps at evidence size (600 small, 200 medium, 120 medium forced onto the surfel path):
# bash scripts/gpu_sweeps.sh <tag>   -> gpurun_out/<tag>/randomized_*.log   (SMALL= MEDIUM= SURFEL= override the seed counts;
# MEDIUM=350 includes seed 321, the known failure of DESIGN.md section 2 until scripts/patches/exact_cutoff_log.diff is in)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-sweeps}; mkdir -p $OUT
cd $R
BGS_RANDOM_SEEDS=${SMALL:-600} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_randomized_configurations and not medium or test_zz" 2>&1 | grep -v amdgpu.ids | tail -3 > $OUT/randomized_600_seeds.log
BGS_RANDOM_MEDIUM_SEEDS=${MEDIUM:-200} timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_randomized_configurations_medium or test_zz" 2>&1 | grep -v amdgpu.ids | tail -3 > $OUT/randomized_medium_200_seeds.log
BGS_RANDOM_FORCE_SURFEL=1 BGS_RANDOM_MEDIUM_SEEDS=${SURFEL:-120} timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "test_randomized_configurations_medium or test_zz" 2>&1 | grep -v amdgpu.ids | tail -3 > $OUT/randomized_medium_surfel_120_seeds.log
tail -n 2 $OUT/randomized_*.log
