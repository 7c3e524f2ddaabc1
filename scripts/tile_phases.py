"""Where a tile wave's life goes (experiment builds with -DBGS_PHASE_TRACE=1, loaded through BGS_LIB_OVERRIDE):
python scripts/tile_phases.py <config>      config: dense | scene
Per tile: wave start -> raster_tile entered -> first candidates tested -> first round staged -> last record blended -> end."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded  # noqa: E402
from bevy_gaussian_splatting_amd.multiview import headless_view  # noqa: E402
cfg = sys.argv[1] if len(sys.argv) > 1 else "dense"
W, H = 1920, 1080
if cfg == "trained":
    from bevy_gaussian_splatting_amd import trained_like_gaussians_3d_seeded
    cloud, s = trained_like_gaussians_3d_seeded(1_000_000, 7), CloudSettings()
else:
    cloud = random_gaussians_3d_seeded(1_000_000, 2)
    s = CloudSettings(global_scale=1.0 if cfg == "dense" else 0.05)
p = GaussianSplattingPlugin(0)
h = p.upload(cloud)
v = headless_view(0, W, H)
for _ in range(30):
    p.render(h, v, s, download=False)
nt = ((W + 15) // 16) * ((H + 15) // 16)
trace = torch.zeros((nt * 3, 4), dtype=torch.int32, device="cuda:0")
p.set_tile_trace(trace.data_ptr())
ms = []
for _ in range(6):
    p.render(h, v, s, download=False)
    ms.append(p.stats()["stage_ms"]["raster"])
torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
p.set_tile_trace(None)
a = t[:2 * nt].reshape(nt, 8)
ph = t[2 * nt:]
t0 = a[:, 0] | (a[:, 1] << 32)
t1 = a[:, 2] | (a[:, 3] << 32)
hw, xcc = a[:, 4], a[:, 5] & 0xF
blended = a[:, 7] & 0xFFFF
staged = a[:, 7] >> 16
scanned = a[:, 6]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ids, inv = np.unique(cu_key, return_inverse=True)
start = np.full(len(ids), np.iinfo(np.int64).max)
np.minimum.at(start, inv, t0)
span = float((t1 - start[inv]).max())
tick_us = float(np.median(ms)) * 1e3 / span
rs = (t0 - start[inv]) * tick_us
life = (t1 - t0) * tick_us
first = rs < 5
names = ["enter raster_tile", "first candidates tested", "first round staged", "last record blended", "wave end"]
print(cfg, "raster ms", np.median(ms), "tick ns", tick_us * 1e3, "stats", {k: p.stats()[k] for k in ("visible_count", "instance_count", "list_capacity")})
print("per tile: scanned p50 %d mean %.0f max %d | staged p50 %d mean %.0f max %d | blended p50 %d mean %.0f max %d" % (
    np.median(scanned), scanned.mean(), scanned.max(), np.median(staged), staged.mean(), staged.max(), np.median(blended), blended.mean(), blended.max()))
for nm, m in (("round 1", first), ("round 2", ~first)):
    print(nm, int(m.sum()), "tiles; wave start p50 %.1f us; life mean %.1f us; blended mean %.1f" % (np.median(rs[m]), life[m].mean(), blended[m].mean()))
    prev = np.zeros(m.sum())
    for i in range(5):
        cur = (ph[m, i] * tick_us) if i < 4 else life[m]
        d = cur - prev
        print("   %-26s +%.2f us (p10 %.2f p90 %.2f)  cumulative %.2f" % (names[i], d.mean(), np.quantile(d, .1), np.quantile(d, .9), cur.mean()))
        prev = cur
