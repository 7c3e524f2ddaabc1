#!/bin/bash
# first GPU pass: smoke, parity tests, bench, kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -6
nproc
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20
echo "== pytest gpu"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 400 2>&1 | tail -40
echo "== bench"
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r1_first.json 2> gpurun_out/bench_r1_first.err
tail -c 3000 gpurun_out/bench_r1_first.json; tail -5 gpurun_out/bench_r1_first.err
