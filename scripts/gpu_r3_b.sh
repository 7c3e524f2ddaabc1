#!/bin/bash
# Round 3, second GPU call: A/B of the rasteriser work-item permutation (0x1000000 = plain band order) and the surfel
# per-strip cull (0x2000000 = round 2's tile-level cull), traces after the change, the tests the first call failed.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3_b}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "graphs or exact_log or binning_modes or 2dgs or surfel or rerun or supertile or render_parity_small" 2>&1 | tail -4
echo "== A/B"; timeout 900 python scripts/ab_flags.py "dense scene surfel surfel_scene 5m_scene" "0,0x1000000,0x2000000" 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_spread_and_strips.log
echo "== traces"
for c in dense surfel; do timeout 300 python scripts/tile_trace.py $c $OUT/tile_trace_$c.json > $OUT/tile_trace_$c.log 2>&1; grep -E "raster_ms|launch_span" $OUT/tile_trace_$c.log; done
ls $OUT
