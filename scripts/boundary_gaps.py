"""Idle time at the launch boundaries of a rocprofv3 kernel trace, by (kernel before -> kernel after):
python scripts/boundary_gaps.py <kernel_trace.csv> [skip first N dispatches]"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(sys.argv[2]) if len(sys.argv) > 2 else 0:]
short = lambda n: n.split("(")[0].replace("void bgs::", "").replace("bgs::", "")[:28]
gaps = collections.defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    v.sort()
    print("%-30s -> %-30s n %4d  gap us: median %6.2f  p10 %6.2f  p90 %6.2f" % (k[0], k[1], len(v), v[len(v) // 2], v[len(v) // 10], v[(9 * len(v)) // 10]))
