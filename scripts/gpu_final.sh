#!/bin/bash
# Round-end evidence run: smoke, all GPU tests, bench, other configs, rocprofv3 stats + PMC, timelines.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r1_final}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -5
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -2 $OUT/bench.err
echo "== other configs"; timeout 900 python scripts/bench_configs.py > $OUT/bench_other_configs.json 2> $OUT/bench_other.err; tail -c 1500 $OUT/bench_other_configs.json; tail -2 $OUT/bench_other.err
echo "== rocprof"; bash scripts/gpu_profile.sh > $OUT/profile.log 2>&1; cp $R/gpurun_out/prof/* $OUT/ 2>/dev/null; tail -30 $OUT/profile.log
python - "$R/gpurun_out/pmc_dense/counters.txt" "$OUT/pmc_traffic.json" <<'PY'
import sys, json, re, collections
vals = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"\s*(?:void )?(?:bgs::)?([a-z_]+kernel)(?:<[^>]*>)?\s+(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE)\s+calls\s+\d+\s+mean\s+([\d.]+)", line)
    if m:
        vals[m.group(1)][m.group(2)] = float(m.group(3))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python scripts/loop_render.py 1.0 12; headline dense workload (1M splats, 1080p). FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); WRITE_SIZE is taken as reported. Units: KB per launch in the counters, bytes here.",
       "kernels": {}}
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out["kernels"][k] = {"fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"],
                             "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
                             "valu_wave_instructions": v.get("SQ_INSTS_VALU"), "gui_active_cycles": v.get("GRBM_GUI_ACTIVE")}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"]))
PY
echo "== timelines"; bash scripts/gpu_timeline.sh 2>&1 | tail -6; cp $R/gpurun_out/timeline_depth*.json $OUT/
ls -la $OUT
