#!/bin/bash
# PMC counters per kernel for one workload: bash scripts/gpu_pmc.sh <what> <tag>   (<what>: see scripts/loop_render.py)
# Every group is its own rocprofv3 run with --pmc only (never combined with trace domains).
GS=$1; TAG=$2
# PMC_GROUPS="0 3 5 6" picks counter groups by position (default: all); 12 blocking frames per run
SEL=" ${PMC_GROUPS:-0 1 2 3 4 5 6} "
GI=-1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$TAG; rm -f $R/gpurun_out/pmc_$TAG/counters.txt
cd /tmp
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  GI=$((GI+1)); case "$SEL" in *" $GI "*) ;; *) continue;; esac
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-30)
  rm -rf /tmp/p_$tag
  rocprofv3 --pmc $pmc --output-format csv -d /tmp/p_$tag -o pmc -- python $R/scripts/loop_render.py $GS 12 > /dev/null 2> /tmp/pmc_$tag.err
  for f in $(find /tmp/p_$tag -name "*counter_collection.csv"); do python - "$f" <<'PY' >> $R/gpurun_out/pmc_$TAG/counters.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "rocclr" in k: continue
    for c, v in cs.items():
        print(f"{k:62s} {c:24s} calls {len(v):4d} mean {sum(v)/len(v):16.1f}")
PY
  done
  tail -2 /tmp/pmc_$tag.err | cut -c1-200
done
cat $R/gpurun_out/pmc_$TAG/counters.txt
