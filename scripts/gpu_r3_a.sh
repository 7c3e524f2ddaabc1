#!/bin/bash
# Round 3, first GPU call: correctness first (full GPU suite incl. whole-frame parity and the exhaustive device log,
# the wide randomized sweeps), then the rasteriser's per-tile trace and one bench line.   bash scripts/gpu_r3_a.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3_a}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 1200 -s 2>&1 | grep -E "passed|failed|error|Error|ambiguity slack used|whole frame|libbgs build id" | tee $OUT/pytest_gpu_summary.log | tail -25
echo "== sweeps"; SMALL=${SMALL:-2000} MEDIUM=${MEDIUM:-350} SURFEL=${SURFEL:-120} bash scripts/gpu_sweeps.sh $TAG 2>&1 | tail -8
echo "== tile traces"
for c in dense scene surfel 5m_scene; do timeout 300 python scripts/tile_trace.py $c $OUT/tile_trace_$c.json > $OUT/tile_trace_$c.log 2>&1; tail -c 600 $OUT/tile_trace_$c.log | head -c 300; echo; done
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json | cut -c1-3000; tail -3 $OUT/bench.err
ls -la $OUT
