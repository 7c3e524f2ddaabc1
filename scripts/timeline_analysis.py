"""Overlap analysis of a rocprofv3 --kernel-trace CSV: how busy is the GPU, and what runs concurrently?
python scripts/timeline_analysis.py <kernel_trace.csv> [skip_first_n_dispatches]"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    name = name.replace("void bgs::", "").replace("bgs::", "")[:40]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
ev.sort()
ev = ev[skip:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
span = t1 - t0
# union busy time + concurrency-weighted time via sweep
pts = []
for s, e, n, q in ev:
    pts.append((s, 1, n))
    pts.append((e, -1, n))
pts.sort(key=lambda p: (p[0], p[1]))
active = collections.Counter()
busy = 0
conc_time = collections.Counter()      # level -> ns
alone = collections.Counter()          # kernel -> ns during which ONLY this kernel name is active
with_raster = collections.Counter()    # kernel -> ns overlapped with a raster kernel
last = pts[0][0]
for t, d, n in pts:
    dt = t - last
    if dt > 0:
        level = sum(active.values())
        conc_time[level] += dt
        if level > 0:
            busy += dt
            names = [k for k, c in active.items() if c > 0]
            if len(names) == 1:
                alone[names[0]] += dt
            if any("raster" in k for k in names):
                for k in names:
                    with_raster[k] += dt
    active[n] += d
    last = t
agg = collections.defaultdict(list)
for s, e, n, q in ev:
    agg[n].append(e - s)
queues = sorted({q for _, _, _, q in ev})
out = {
    "dispatches": len(ev), "queues": queues, "span_us": span / 1e3, "busy_union_us": busy / 1e3,
    "idle_frac": 1 - busy / span, "sum_kernel_us": sum(e - s for s, e, _, _ in ev) / 1e3,
    "concurrency_hist_us": {str(k): round(v / 1e3, 1) for k, v in sorted(conc_time.items())},
    "kernels": {k: {"calls": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 2), "total_us": round(sum(v) / 1e3, 1),
                    "alone_us": round(alone[k] / 1e3, 1), "with_raster_us": round(with_raster[k] / 1e3, 1)}
                for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))},
}
print(json.dumps(out, indent=1))
