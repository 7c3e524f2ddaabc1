#!/bin/bash
# FETCH_SIZE calibration for gathers (scripts/micro/gather_fetch.hip): bash scripts/gpu_gather_fetch.sh <outdir>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/gather_fetch}; mkdir -p $OUT
cd /tmp; rm -rf /tmp/gf
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/gf -o gf -- $R/scripts/micro/gather_fetch > $OUT/gather_fetch_requests.txt 2> /tmp/gf.err
for f in $(find /tmp/gf -name "*counter_collection.csv"); do python - "$f" $OUT/gather_fetch_requests.txt <<'PY' | tee $OUT/gather_fetch_calibration.txt
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
req = {}
for line in open(sys.argv[2]):
    m = re.match(r"(\S+kernel(?:<\d+>)?).*requests\s+([\d.]+) KB", line)
    if m: req[m.group(1)] = float(m.group(2))
print("kernel                          FETCH_SIZE_KB   requested_KB   requested/FETCH   FETCH bytes per gather")
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "kernel" not in k: continue
    fetch = float(r["Counter_Value"])
    want = req.get(k, 0.0)
    n = 1 << 22
    per = "" if k.startswith("stream") else f"{(fetch * 1024 - n * 4 / 2) / n:8.1f} (indices taken off at x2)"
    print(f"{k:30s} {fetch:14.1f} {want:14.1f} {want / fetch if fetch else 0:17.3f}   {per}")
PY
done
cat $OUT/gather_fetch_requests.txt
