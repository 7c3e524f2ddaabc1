#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu (all)"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r1_scan.json 2> gpurun_out/bench_r1_scan.err
cat gpurun_out/bench_r1_scan.json; tail -3 gpurun_out/bench_r1_scan.err
