#!/bin/bash
# Round 3, third GPU call: tests touched by the keygen change, priority experiment A/B, FETCH_SIZE calibration for
# gathers, counter evidence for the dense headline frame and the two slowest configs.   bash scripts/gpu_r3_c.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3_c}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "sort or modes or pipelined or graphs or draw_modes or edge_cases or multi_camera" 2>&1 | tail -3
echo "== A/B priority"; timeout 600 python scripts/ab_flags.py "dense scene surfel 5m_scene" "0,0x4000000" 2 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_progressive_priority.log
echo "== gather calibration"; bash scripts/gpu_gather_fetch.sh $OUT 2>&1 | tail -12
echo "== PMC dense"; PMC_GROUPS="0 1 2 3 4 5 6" bash scripts/gpu_pmc.sh 1.0 dense > /dev/null 2>&1; cp $R/gpurun_out/pmc_dense/counters.txt $OUT/dense_pmc_counters.txt
echo "== PMC surfel"; PMC_GROUPS="0 3 5 6" bash scripts/gpu_pmc.sh surfel surfel > /dev/null 2>&1; cp $R/gpurun_out/pmc_surfel/counters.txt $OUT/surfel_pmc_counters.txt
echo "== PMC 5m_scene"; PMC_GROUPS="0 3 5 6" bash scripts/gpu_pmc.sh 5m_scene 5m_scene > /dev/null 2>&1; cp $R/gpurun_out/pmc_5m_scene/counters.txt $OUT/5m_scene_pmc_counters.txt
cd /tmp
for w in surfel 5m_scene; do
  rm -rf /tmp/p_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$w -o trace -- python $R/scripts/loop_render.py $w 40 > /dev/null 2> /tmp/$w.err
  for f in $(find /tmp/p_$w -name "*kernel_stats.csv"); do cp $f $OUT/${w}_kernel_stats.csv; cut -c1-150 $f | head -8; done
done
grep -E "raster_scan|project_bin|keygen" $OUT/*_pmc_counters.txt | grep -E "SQ_INSTS_VALU |TRANS_F32|FETCH_SIZE|WRITE_SIZE|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_BUSY_CYCLES|FMA_F64" | cut -c1-170
ls $OUT
