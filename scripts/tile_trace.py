"""Per-tile trace of the default rasteriser (bgs_set_tile_trace): where every tile's wave ran, when, and on how much work.

python scripts/tile_trace.py <config> [out.json]     config: dense | scene | surfel | 5m_scene | 5m_dense
Single-stream blocking frames (the kernel alone on the chip). Prints and stores: the distribution of tile-wave
lifetimes, when the launch's waves finish (the tail), the load per SIMD (tiles / records blended / busy span), and how
duration follows work. s_memtime ticks are converted with the clock rate measured against the launch's own HIP-event
duration."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bevy_gaussian_splatting_amd import (CloudSettings, GaussianMode, GaussianSplattingPlugin,  # noqa: E402
                                         random_gaussians_3d_seeded)
from bevy_gaussian_splatting_amd.multiview import headless_view  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "dense"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
W, H = 1920, 1080
if cfg in ("dense", "scene"):
    cloud = random_gaussians_3d_seeded(1_000_000, 2)
    s = CloudSettings(global_scale=1.0 if cfg == "dense" else 0.05)
elif cfg == "surfel":
    cloud = random_gaussians_3d_seeded(1_000_000, 2)
    s = CloudSettings(gaussian_mode=GaussianMode.Gaussian2d, aabb=True)
elif cfg in ("5m_scene", "5m_dense"):
    cloud = random_gaussians_3d_seeded(5_000_000, 3).to_f16()
    s = CloudSettings(global_scale=0.05 if cfg == "5m_scene" else 1.0)
else:
    raise SystemExit("config: dense | scene | surfel | 5m_scene | 5m_dense")
p = GaussianSplattingPlugin(0)
h = p.upload(cloud)
v = headless_view(0, W, H)
for _ in range(30):   # adaptive state (supertile level, list capacity, splitters) settles
    p.render(h, v, s, download=False)
tx, ty = (W + 15) // 16, (H + 15) // 16
trace = torch.zeros((tx * ty, 8), dtype=torch.int32, device="cuda:0")
p.set_tile_trace(trace.data_ptr())
raster_ms = []
for _ in range(8):
    p.render(h, v, s, download=False)
    raster_ms.append(p.stats()["stage_ms"]["raster"])
torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
p.set_tile_trace(None)
untraced = []
for _ in range(8):
    p.render(h, v, s, download=False)
    untraced.append(p.stats()["stage_ms"]["raster"])
st = p.stats()

t0 = t[:, 0] | (t[:, 1] << 32)
t1 = t[:, 2] | (t[:, 3] << 32)
hw, xcc = t[:, 4], t[:, 5] & 0xF
scanned, blended, staged = t[:, 6], t[:, 7] & 0xFFFF, t[:, 7] >> 16
wave_slot, simd, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
dur = (t1 - t0).astype(np.float64)
# s_memtime is a PER-CU time base (counters of different CUs are millions of ticks apart): a launch's timeline is taken
# relative to the first wave of the same CU; one tick is one shader clock
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
cu_ids, cu_inv = np.unique(cu_key, return_inverse=True)
start = np.full(len(cu_ids), np.iinfo(np.int64).max)
np.minimum.at(start, cu_inv, t0)
rel_end = (t1 - start[cu_inv]).astype(np.float64)
rel_start = (t0 - start[cu_inv]).astype(np.float64)
span_ticks = rel_end.max()
launch_us = float(np.median(raster_ms)) * 1e3
tick_us = launch_us / span_ticks   # calibrated on this launch: the clock the kernel actually ran at
simd_key = cu_key * 4 + simd
keys, inv = np.unique(simd_key, return_inverse=True)
tiles_per_simd = np.bincount(inv)
blended_per_simd = np.bincount(inv, weights=blended)
dur_per_simd = np.bincount(inv, weights=dur)
last_end_per_simd = np.zeros(len(keys))
np.maximum.at(last_end_per_simd, inv, rel_end)
q = lambda a, f: float(np.quantile(a, f))  # noqa: E731
res = {
    "config": cfg, "tiles": int(tx * ty), "visible_splats": st["visible_count"], "coarse_entries": st["instance_count"],
    "raster_ms_traced_median": float(np.median(raster_ms)), "raster_ms_untraced_median": float(np.median(untraced)),
    "tick_ns": tick_us * 1e3, "clock_GHz": 1e-3 / tick_us, "launch_span_ticks": float(span_ticks), "launch_span_us": launch_us,
    "wave_life_us": {k: q(dur, f) * tick_us for k, f in (("p10", .1), ("p50", .5), ("p90", .9), ("p99", .99), ("max", 1.0))}
    | {"mean": float(dur.mean()) * tick_us},
    "wave_start_us": {"p50": q(rel_start, .5) * tick_us, "p99": q(rel_start, .99) * tick_us, "max": float(rel_start.max()) * tick_us},
    "waves_finished_at_us": {k: q(rel_end, f) * tick_us for k, f in (("p25", .25), ("p50", .5), ("p75", .75), ("p90", .9), ("p99", .99), ("all", 1.0))},
    "per_tile": {"scanned": [q(scanned, .5), q(scanned, .99), float(scanned.max())],
                 "staged": [q(staged, .5), q(staged, .99), float(staged.max())],
                 "blended": [q(blended, .5), q(blended, .99), float(blended.max())], "columns": "p50, p99, max"},
    "corr_duration_vs": {"blended": float(np.corrcoef(dur, blended)[0, 1]), "staged": float(np.corrcoef(dur, staged)[0, 1]),
                         "scanned": float(np.corrcoef(dur, scanned)[0, 1])},
    "simds_used": int(len(keys)), "cus_used": int(len(cu_ids)), "xcds": sorted(int(x) for x in np.unique(xcc)),
    "per_simd": {"tiles": [int(tiles_per_simd.min()), float(np.median(tiles_per_simd)), int(tiles_per_simd.max())],
                 "blended_records": [float(blended_per_simd.min()), float(np.median(blended_per_simd)), float(blended_per_simd.max())],
                 "last_wave_ends_us": [q(last_end_per_simd, f) * tick_us for f in (.0, .1, .5, .9, 1.0)],
                 "columns": "min, median, max (last_wave_ends: p0 p10 p50 p90 p100)"},
    "corr_simd_last_end_vs_blended": float(np.corrcoef(last_end_per_simd, blended_per_simd)[0, 1]),
    "corr_simd_last_end_vs_tiles": float(np.corrcoef(last_end_per_simd, tiles_per_simd)[0, 1]),
}
# time-resolved occupancy: waves alive in 20 slices of the launch
edges = np.linspace(0, span_ticks, 21)
alive = [int(((rel_start < b) & (rel_end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
res["waves_alive_per_5pct_slice"] = alive
# duration of a tile per blended record, by how late the tile finishes
order = np.argsort(rel_end)
tail = order[-len(order) // 20:]
res["slowest_5pct_tiles"] = {"blended_mean": float(blended[tail].mean()), "staged_mean": float(staged[tail].mean()),
                             "scanned_mean": float(scanned[tail].mean()), "life_us_mean": float(dur[tail].mean()) * tick_us,
                             "all_tiles_blended_mean": float(blended.mean()), "all_tiles_life_us_mean": float(dur.mean()) * tick_us}
print(json.dumps(res, indent=1))
if out_path:
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    np.savez_compressed(out_path.replace(".json", ".npz"), t0=t0, t1=t1, hw=hw, xcc=xcc, scanned=scanned, blended=blended, staged=staged)
