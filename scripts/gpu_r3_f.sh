#!/bin/bash
# Round 3, sixth GPU call: grouped (chainless) bucket placement — its tests, then A/B against the chained placement
# (debug flag 0x1000000), single stream and 8 lanes / 4 streams.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3_f}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "grouped or bucket or splitter or graphs or pipelined or overflow or sort_bit_exact or full_size_sort or randomized_configurations_medium or render_parity_small or cloud_past" 2>&1 | tail -6
echo "== A/B"; timeout 900 python scripts/ab_flags.py "dense scene 5m_dense 5m_scene" "0,0x1000000" 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_grouped_vs_chained.log
cd /tmp; rm -rf /tmp/p_dense
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dense -o trace -- python $R/scripts/loop_render.py 1.0 40 > /dev/null 2> /tmp/dense.err
for f in $(find /tmp/p_dense -name "*kernel_stats.csv"); do cp $f $OUT/dense_kernel_stats.csv; awk -F'",' '{print substr($1,1,60)}' $f | head -8; awk -F, '{print $(NF-6),$(NF-5),$(NF-4)}' $f | head -8; done
