#!/bin/bash
# round 6: parity subset on the tree's library, then an A/B of debug-flag variants (scripts/ab_flags.py)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest gpu (subset)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -x -k "${1:-binning_modes or multisampled_target or whole_frame_parity_1m or cost_ordered or sort_bit_exact}" 2>&1 | tail -5
python $R/scripts/ab_flags.py "${2:-dense scene}" "${3:-0}" ${4:-2} 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/ab_flags.txt
