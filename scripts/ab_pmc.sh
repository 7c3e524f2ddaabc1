#!/bin/bash
# PMC counters of the rasteriser for every library variant in ab/*.so: bash scripts/ab_pmc.sh <global_scale>
R=$GRAFT_REPO_ROOT
cp $R/bevy_gaussian_splatting_amd/csrc/libbgs.so /tmp/libbgs_orig.so
for v in $R/ab/*.so; do
  cp $v $R/bevy_gaussian_splatting_amd/csrc/libbgs.so
  t=$(basename $v .so)
  bash $R/scripts/gpu_pmc.sh ${1:-1.0} ab_$t > /dev/null 2>&1
  echo "== $t (global_scale ${1:-1.0})"
  grep raster_scan $R/gpurun_out/pmc_ab_$t/counters.txt | grep -E "SQ_INSTS_VALU |SQ_ACTIVE_INST_VALU|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_INSTS_SALU|SQ_ACTIVE_INST_SCA|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|GRBM_GUI_ACTIVE|SQ_INSTS_VALU_TRANS|SQ_THREAD_CYCLES_VALU|SQ_ACTIVE_INST_LDS|SQ_INSTS_LDS|SQ_ACTIVE_INST_MISC|SQ_WAIT_ANY" | awk '{print $2"<"$3, $4, $NF}'
done
cp /tmp/libbgs_orig.so $R/bevy_gaussian_splatting_amd/csrc/libbgs.so
