#!/bin/bash
# round 6, first call: parity subset on the tree's library, then the same-box A/B of the variants in gpurun_variants/
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest gpu (subset)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -x -k "${1:-binning_modes or multisampled_target or whole_frame_parity_1m or cost_ordered or sort_bit_exact}" 2>&1 | tail -5
bash $R/scripts/ab_variants.sh "${2:-dense scene}" ${3:-2} 2>&1 | tee $R/gpurun_out/ab_variants.txt
