#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace + stats of the bench command, and PMC passes
# (separate runs; counters never combined with trace domains) for the dense headline workload.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rm -rf /tmp/p_trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -o trace -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /tmp/trace.err
for f in $(find /tmp/p_trace -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done
for f in $(find /tmp/p_trace -name "*kernel_trace.csv"); do python - "$f" "$OUT/bench_kernel_trace_summary.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size")))
with open(sys.argv[2], "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline\n")
    f.write("(bench.py runs, in order: headline dense scan-binning frames, bgs_sort calls, scene-like frames, instance-sort frames)\n")
    f.write(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} vgpr lds\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        d = [x[0] for x in v]
        f.write(f"{k:72s} {len(d):6d} {sum(d)/len(d)/1e3:9.2f} {min(d)/1e3:9.2f} {max(d)/1e3:9.2f} {v[-1][1]} {v[-1][2]}\n")
print(open(sys.argv[2]).read())
PY
done
# dense-only loop: per-kernel durations without mixing workloads
rm -rf /tmp/p_dense
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dense -o trace -- python $R/scripts/loop_render.py 1.0 40 > /dev/null 2> /tmp/dense.err
for f in $(find /tmp/p_dense -name "*kernel_stats.csv"); do cp $f $OUT/dense_kernel_stats.csv; cat $f | cut -c1-160; done
bash $R/scripts/gpu_pmc.sh 1.0 dense > /dev/null 2>&1
cp $R/gpurun_out/pmc_dense/counters.txt $OUT/dense_pmc_counters.txt
grep -E "FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|SQ_ACTIVE_INST_VALU |SQ_INSTS_VALU |SQ_WAVE_CYCLES" $OUT/dense_pmc_counters.txt | cut -c1-130
