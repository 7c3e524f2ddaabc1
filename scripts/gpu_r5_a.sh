#!/bin/bash
# Round 5, first GPU call: the new host logic and instantiations (kinds of frame, ABI check, native RCCL gather,
# Msaa::Sample2 / Sample8, bounding-box overlay), then a bench line with the driver's flags.
set -u
OUT=gpurun_out/r5_a
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python -m pytest tests/test_native_gather.py tests/test_abi.py -m gpu -x -q -s > $OUT/pytest_gather.log 2>&1
echo "gather rc=$?" >> $OUT/pytest_gather.log
timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -s \
  -k "forty_kinds or never_runs_clean or sample_count_and_depth or bounding_box or multisampled_target or depth_buffer_occludes or first_frames or test_randomized_configurations or binning_modes_give or rerun_keeps or zz_report" \
  > $OUT/pytest_new.log 2>&1
echo "new rc=$?" >> $OUT/pytest_new.log
true
echo "bench rc=$?" >> $OUT/bench.err
tail -n 3 $OUT/pytest_gather.log; tail -n 15 $OUT/pytest_new.log; tail -c 600 $OUT/bench.err
