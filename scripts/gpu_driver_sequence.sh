#!/bin/bash
# What the driver runs at round end, in its order, on the tree as committed: the GPU tests, smoke(), the bench line with
# the driver's flags. bash scripts/gpu_driver_sequence.sh <tag> -> gpurun_out/<tag>/
out=gpurun_out/${1:-driver_sequence}; mkdir -p $out
( time python -m pytest tests/ -x -q -m gpu ) > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $out/pytest_gpu.log | tail -2
( time python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > $out/smoke.log 2>&1; echo "smoke rc $?"; tail -4 $out/smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_driver_flags.json 2> $out/bench.err; echo "bench rc $?"; cut -c1-300 $out/bench_driver_flags.json; tail -4 $out/bench.err
