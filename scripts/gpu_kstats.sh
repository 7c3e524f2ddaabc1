#!/bin/bash
# rocprofv3 kernel stats of the single-stream headline loop (and optionally other global scales)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/kstats
cd /tmp
for gs in ${GSS:-1.0 0.05}; do
  rm -rf /tmp/ks
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o t -- python $R/scripts/loop_pipelined.py 1 300 $gs 0 ${FLAGS:-0} > /tmp/ks.out 2>/tmp/ks.err
  cat /tmp/ks.out
  f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/kstats/kernel_stats_gs$gs.csv
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) < 5: continue
    print(f"  {r['Name'][:58]:58s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
PY
done
