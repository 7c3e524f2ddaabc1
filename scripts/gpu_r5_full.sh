#!/bin/bash
# Round 5: the whole GPU suite (edge band 2e-3 px, the default), then the six whole-frame tests again at 5e-4 px.
set -u
OUT=gpurun_out/${1:-r5_full}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 3000 python -m pytest tests -m gpu -q -s --timeout 1500 > $OUT/pytest_gpu_full.log 2>&1
echo "rc=$?" >> $OUT/pytest_gpu_full.log
grep -E "passed|failed|FAILED|ERROR|tolerance accounting" $OUT/pytest_gpu_full.log | tail -n 20
cp gpurun_out/tolerance_accounting_band_2e-3.json $OUT/ 2>/dev/null
BGS_ORACLE_EDGE_BAND_PX=5e-4 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "whole_frame_parity or config4_camera or zz_report" > $OUT/pytest_whole_frames_band_5e-4.log 2>&1
echo "rc=$?" >> $OUT/pytest_whole_frames_band_5e-4.log
grep -E "passed|failed|FAILED|tolerance accounting|whole frame" $OUT/pytest_whole_frames_band_5e-4.log | tail -n 20
cp gpurun_out/tolerance_accounting_band_5e-4.json $OUT/ 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "whole_frame_parity or config4_camera or zz_report" > $OUT/pytest_whole_frames_band_2e-3.log 2>&1
echo "rc=$?" >> $OUT/pytest_whole_frames_band_2e-3.log
grep -E "passed|failed|FAILED|tolerance accounting" $OUT/pytest_whole_frames_band_2e-3.log | tail -n 5
cp gpurun_out/tolerance_accounting_band_2e-3.json $OUT/tolerance_accounting_whole_frames_band_2e-3.json 2>/dev/null
