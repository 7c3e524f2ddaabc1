"""Per-tile rasteriser counters (debug bit 128): cycles, groups scanned, records staged, blended.
python scripts/tile_stats.py [global_scale]"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
gs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(1_000_000, 2))
v = headless_view(0)
s = CloudSettings(global_scale=gs)
for _ in range(3):
    p.render(h, v, s)
p.set_debug_flags(128)
img = p.render(h, v, s)
p.set_debug_flags(0)
tiles = 120 * 68
st = img.reshape(-1, 4)[:tiles].view(np.uint32).astype(np.float64)
cyc, groups, staged, kept = st.T
print(f"gs={gs}: tiles {tiles}; cycles mean {cyc.mean():.0f} median {np.median(cyc):.0f} p90 {np.percentile(cyc,90):.0f} "
      f"p99 {np.percentile(cyc,99):.0f} max {cyc.max():.0f}")
print(f"groups mean {groups.mean():.1f} max {groups.max():.0f}; staged mean {staged.mean():.1f} max {staged.max():.0f}; "
      f"kept mean {kept.mean():.1f} max {kept.max():.0f}")
A = np.stack([np.ones_like(cyc), groups, staged, kept], 1)
coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
print("strips / (4*kept) = %.3f; kept/staged = %.3f" % (cyc.sum() / (4 * kept.sum()), kept.sum()/staged.sum()))
print("cycles ~ %.0f + %.1f*groups + %.1f*staged + %.1f*kept" % tuple(coef))
order = np.argsort(-cyc)[:10]
for t in order:
    print(f"  tile ({int(t)%120:3d},{int(t)//120:2d}) cycles {cyc[t]:.0f} groups {groups[t]:.0f} staged {staged[t]:.0f} kept {kept[t]:.0f}")
hist, edges = np.histogram(cyc, bins=12)
print("cycle histogram:", [(int(e), int(c)) for e, c in zip(edges[:-1], hist)])
# map of cycles by tile row (mean per row)
print("row means:", [int(x) for x in cyc.reshape(68, 120).mean(1)[::4]])
print("col means:", [int(x) for x in cyc.reshape(68, 120).mean(0)[::8]])
