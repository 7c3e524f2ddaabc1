#!/bin/bash
# tests of the new features + bench + pipelined timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest gpu (all)"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x 2>&1 | tail -15
echo "== hbm probe"
timeout 120 python -c "
from bevy_gaussian_splatting_amd import GaussianSplattingPlugin
p = GaussianSplattingPlugin(0)
for nb in (1<<26, 1<<28, 1<<30): print(nb, p.hbm_probe(nb, 20))
"
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r1_v7.json 2> gpurun_out/bench_r1_v7.err
cat gpurun_out/bench_r1_v7.json; tail -3 gpurun_out/bench_r1_v7.err
echo "== pipelined timelines"
cd /tmp
for d in 1 2 3 4; do
  rm -rf /tmp/tl$d
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$d -o t -- python $R/scripts/loop_pipelined.py $d 60 > /tmp/tl$d.out 2>/tmp/tl$d.err
  cat /tmp/tl$d.out
  for f in $(find /tmp/tl$d -name "*kernel_trace.csv"); do python $R/scripts/timeline_analysis.py $f 200 > $R/gpurun_out/timeline_depth$d.json; cp $f $R/gpurun_out/kernel_trace_depth$d.csv; done
done
python - <<'PY'
import json, os
for d in (1, 2, 3, 4):
    p = f"{os.environ['GRAFT_REPO_ROOT']}/gpurun_out/timeline_depth{d}.json"
    if os.path.exists(p):
        t = json.load(open(p))
        print(d, {k: t[k] for k in ("dispatches", "span_us", "busy_union_us", "idle_frac", "sum_kernel_us", "concurrency_hist_us")})
        print({k: (v["avg_us"], v["alone_us"], v["with_raster_us"]) for k, v in t["kernels"].items()})
PY
