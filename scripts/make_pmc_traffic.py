"""counters.txt of scripts/gpu_pmc.sh -> pmc_traffic.json, stamped with the hash of the kernel sources it was
measured on (bench.py quotes roofline.traffic / roofline.valu only when that hash matches the tree).
usage: python scripts/make_pmc_traffic.py <counters.txt> <out.json> [tag]"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha256  # noqa: E402

vals = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    # (the counters file cuts kernel names at 60 characters: a template argument list may end without its '>')
    m = re.match(r"\s*(?:void )?(?:bgs::)?([a-z_]+kernel)(?:<.*?)?\s{2,}(FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU|"
                 r"SQ_ACTIVE_INST_VALU|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE|SQ_INSTS_VALU_[A-Z0-9_]+)\s+calls\s+(\d+)\s+mean\s+([\d.]+)", line)
    if m:
        vals[m.group(1)][m.group(2)] = float(m.group(4))
        vals[m.group(1)]["calls"] = int(m.group(3))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, never combined with trace domains) "
                 "-- python scripts/loop_render.py 1.0 12; headline dense workload (1M splats, 1080p). FETCH_SIZE is "
                 "doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); WRITE_SIZE is taken as "
                 "reported. Units: KB per launch in the counters, bytes here. Per-kernel means over the launches of "
                 "the run (template instances of one kernel are pooled).",
       "measured": sys.argv[3] if len(sys.argv) > 3 else "",
       "kernel_source_sha256": kernel_source_sha256(),
       "kernels": {}}
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out["kernels"][k] = {"fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"],
                             "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
                             "valu_wave_instructions": v.get("SQ_INSTS_VALU"),
                             # per-class counts (bench.py prices them: transcendental 8 clocks, fp64 4, the rest 2);
                             # None when the class pass did not run
                             "valu_trans_wave_instructions": v.get("SQ_INSTS_VALU_TRANS_F32"),
                             "valu_f64_wave_instructions": (None if "SQ_INSTS_VALU_FMA_F64" not in v else
                                                            v.get("SQ_INSTS_VALU_FMA_F64", 0) + v.get("SQ_INSTS_VALU_ADD_F64", 0) +
                                                            v.get("SQ_INSTS_VALU_MUL_F64", 0) + v.get("SQ_INSTS_VALU_TRANS_F64", 0)),
                             "valu_classes": {k[len("SQ_INSTS_VALU_"):].lower(): x for k, x in v.items() if k.startswith("SQ_INSTS_VALU_")},
                             "gui_active_cycles": v.get("GRBM_GUI_ACTIVE"), "launches_sampled": v.get("calls")}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"]))
