#!/bin/bash
# quick perf iteration: parity subset + single-stream / pipelined loops + per-kernel timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest gpu (subset)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "${1:-sort or onesweep or parity_small or full_size or pipelined}" 2>&1 | tail -5
cd /tmp
for d in 1 3; do
  rm -rf /tmp/tl$d
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$d -o t -- python $R/scripts/loop_pipelined.py $d 60 > /tmp/tl$d.out 2>/tmp/tl$d.err
  cat /tmp/tl$d.out
  for f in $(find /tmp/tl$d -name "*kernel_trace.csv"); do python $R/scripts/timeline_analysis.py $f 200 > $R/gpurun_out/timeline_depth$d.json; done
done
for d in 1 2 3 4; do python $R/scripts/loop_pipelined.py $d 200; done
python $R/scripts/loop_pipelined.py 3 200 0.05
python - <<'PY'
import json, os
for d in (1, 3):
    p = f"{os.environ['GRAFT_REPO_ROOT']}/gpurun_out/timeline_depth{d}.json"
    if os.path.exists(p):
        t = json.load(open(p))
        print(d, {k: t[k] for k in ("span_us", "busy_union_us", "idle_frac", "sum_kernel_us", "concurrency_hist_us")})
        print({k[:22]: (v["avg_us"], v["alone_us"]) for k, v in t["kernels"].items()})
PY
