#!/bin/bash
# rocprofv3 kernel timelines of the pipelined loop (1 lane; 3 and 6 lanes on 3 streams; 8 lanes on 4 streams, the
# default) -> gpurun_out/timeline_depth*.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for d in ${DEPTHS:-1 3 6 8}; do
  rm -rf /tmp/tl$d
  st=3; [ $d -ge 8 ] && st=4
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$d -o t -- python $R/scripts/loop_pipelined.py $d 60 ${1:-1.0} 0 0 $st > /tmp/tl$d.out 2>/tmp/tl$d.err
  cat /tmp/tl$d.out
  for f in $(find /tmp/tl$d -name "*kernel_trace.csv"); do python $R/scripts/timeline_analysis.py $f 150 > $R/gpurun_out/timeline_depth$d.json; done
done
python - <<'PY'
import json, os
for d in (1, 3, 6, 8):
    p = f"{os.environ['GRAFT_REPO_ROOT']}/gpurun_out/timeline_depth{d}.json"
    if os.path.exists(p):
        t = json.load(open(p))
        print(d, {k: t[k] for k in ("dispatches", "span_us", "busy_union_us", "idle_frac", "sum_kernel_us", "concurrency_hist_us")})
        print({k[:24]: (v["calls"], v["avg_us"], v["alone_us"]) for k, v in t["kernels"].items()})
PY
