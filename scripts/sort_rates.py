"""Device time of blocking bgs_sort calls (keygen + depth sort) with and without the bucket path, on lists with every splat
drawable and on the headline camera's: python scripts/sort_rates.py  (profiles/r5_notes.md section 1).
python scripts/sort_rates.py <n> <rayon|radix_far|radix_headline> <flags>: that one combination only, 100 timed calls — the
form scripts/gpu_r5_sort_prof.sh runs under rocprofv3 for the per-kernel averages."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import (CloudSettings, GaussianSplattingPlugin, SortMode, View,  # noqa: E402
                                         random_gaussians_3d_seeded, transform_from)

only = (int(sys.argv[1]), sys.argv[2], int(sys.argv[3], 0)) if len(sys.argv) > 3 else None
if not only:
    pass
reps = 100 if only else 20
p = GaussianSplattingPlugin(0)
for n, seed in ((1_000_000, 2), (2_000_000, 5), (5_000_000, 3)):
    if only and n != only[0]:
        continue
    c = random_gaussians_3d_seeded(n, seed)
    h = p.upload(c)
    far = View.perspective(transform_from((0.0, 0.0, 120.0), (0.0, 0.0, 0.0, 1.0)), 1920, 1080)
    for name, v, s in (("rayon", View.headless(1920, 1080), CloudSettings(sort_mode=SortMode.Rayon)), ("radix_far", far, CloudSettings()),
                       ("radix_headline", View.headless(1920, 1080), CloudSettings())):
        for flags in (0x80000, 0, 0x100, 0x800):   # 0x80000: never the bucket path; 0x100 / 0x800: the bucket path with narrow / wide buckets whatever the length
            if only and (name, flags) != only[1:]:
                continue
            p.set_debug_flags(flags); p.reset_adaptive_state(); p.set_profiling_stride(1)
            for _ in range(4):
                p.sort(h, v, s, download=False)
            ms = kg = ds = 0.0
            for _ in range(reps):
                p.sort(h, v, s, download=False)
                st = p.stats()
                ms += st["total_ms"]; kg += st["stage_ms"]["keygen"]; ds += st["stage_ms"]["depth_sort"]
            print(f"{n:8d} {name:15s} flags {flags:#8x} {st['sort_path']:9s} D={st['draw_count']:8d} total {ms / reps * 1e3:7.1f} us "
                  f"keygen {kg / reps * 1e3:6.1f} sort {ds / reps * 1e3:6.1f}  {n / (ms / reps * 1e-3) / 1e9:6.2f} Gsplats/s "
                  f"{88.0 * n / (ms / reps * 1e-3) / 1e9:7.1f} GB/s on 88 B per splat", flush=True)
    p.set_debug_flags(0)
    h.free()
