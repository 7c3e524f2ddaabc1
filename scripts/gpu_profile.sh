#!/bin/bash
# rocprofv3: kernel trace + stats, then PMC passes (separate runs, no trace domains mixed in)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -o trace -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof/bench_under_trace.json 2> /tmp/trace.err
find /tmp/p_trace -name "*.csv" | head -20
for f in $(find /tmp/p_trace -name "*kernel_stats.csv"); do cp $f $R/gpurun_out/prof/kernel_stats.csv; done
for f in $(find /tmp/p_trace -name "*kernel_trace.csv"); do python - "$f" "$R/gpurun_out/prof/kernel_trace_summary.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size")))
with open(sys.argv[2], "w") as f:
    f.write(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} vgpr lds grid wg\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        d = [x[0] for x in v]
        f.write(f"{k:72s} {len(d):6d} {sum(d)/len(d)/1e3:9.2f} {min(d)/1e3:9.2f} {max(d)/1e3:9.2f} {v[-1][1]} {v[-1][2]} {v[-1][3]} {v[-1][4]}\n")
print(open(sys.argv[2]).read())
PY
done
# PMC passes
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pmc --output-format csv -d /tmp/p_$tag -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/pmc_$tag.err
  for f in $(find /tmp/p_$tag -name "*counter_collection.csv"); do python - "$f" "$R/gpurun_out/prof/pmc_$tag.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    for k, cs in agg.items():
        for c, v in cs.items():
            f.write(f"{k:72s} {c:24s} calls {len(v):5d} mean {sum(v)/len(v):16.1f}\n")
print(open(sys.argv[2]).read()[:6000])
PY
  done
done
ls -la $R/gpurun_out/prof
