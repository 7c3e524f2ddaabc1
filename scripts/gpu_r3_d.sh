#!/bin/bash
# Round 3, fourth GPU call: packed cloud records — the full GPU suite, then the per-config throughput table.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3_d}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 1200 -s 2>&1 | grep -E "passed|failed|error|Error|whole frame" | tee $OUT/pytest_gpu_summary.log | tail -12
echo "== configs"; timeout 900 python scripts/ab_flags.py "dense scene surfel 2d_obb 5m_dense 5m_scene" "0" 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_packed_records.log
