#!/bin/bash
# Round 5: bucket sort with 256 * sub buckets (D = N lists), ordered keygen without per-bucket chains.
set -u
OUT=gpurun_out/r5_c
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "sort or bucket or fine_buckets or onesweep or multi_camera or rerun or stale or pipelined or first_frames or splitter" > $OUT/pytest.log 2>&1
echo "rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED" $OUT/pytest.log | tail -n 12
python - > $OUT/sort_rates.txt 2>&1 <<'PY'
import time, json
from bevy_gaussian_splatting_amd import *
p = GaussianSplattingPlugin(0)
for n, seed in ((1_000_000, 2), (5_000_000, 3)):
    c = random_gaussians_3d_seeded(n, seed)
    h = p.upload(c)
    far = View.perspective(transform_from((0.0, 0.0, 120.0), (0.0, 0.0, 0.0, 1.0)), 1920, 1080)
    for name, v, s in (("rayon", View.headless(1920, 1080), CloudSettings(sort_mode=SortMode.Rayon)), ("radix_far", far, CloudSettings()),
                       ("radix_headline", View.headless(1920, 1080), CloudSettings())):
        for flags in (0x80000, 0):
            p.set_debug_flags(flags); p.reset_adaptive_state(); p.set_profiling_stride(1)
            for _ in range(4): p.sort(h, v, s, download=False)
            ms = kg = ds = 0.0
            for _ in range(20):
                p.sort(h, v, s, download=False); st = p.stats()
                ms += st["total_ms"]; kg += st["stage_ms"]["keygen"]; ds += st["stage_ms"]["depth_sort"]
            print(f"{n:8d} {name:15s} flags {flags:#8x} {st['sort_path']:9s} D={st['draw_count']:8d} total {ms/20*1e3:7.1f} us keygen {kg/20*1e3:6.1f} sort {ds/20*1e3:6.1f}  {n/(ms/20*1e-3)/1e9:6.2f} Gsplats/s", flush=True)
    p.set_debug_flags(0)
    h.free()
PY
cat $OUT/sort_rates.txt
