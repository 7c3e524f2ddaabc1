"""Soak with everything that adapts in motion: an orbiting camera (draw count drifts -> sticky grid hint),
splat size switching between scene-like and dense (supertile level jumps, list capacities regrow, bucket
splitters go stale and the frame is re-run), 8 lanes on 4 streams, frame
graphs on for every other phase, and (SOAK_MIXED, default on) Msaa 1/2/4/8 and the bounding-box overlay cycling with
the phases, so that the context keeps switching between kinds of frame it has and has not settled on. Every phase ends with a check against a blocking, directly launched frame.
python scripts/soak_dynamic.py [frames]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view, framebuffer_as_tensor
total = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
mixed = os.environ.get("SOAK_MIXED", "1") != "0"        # phases also cycle Msaa 1/2/4/8 and the bounding-box overlay
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(1_000_000, 2))
views = [headless_view(g) for g in range(8)]          # camera yawed in 45 degree steps
p.set_profiling(0)
p.set_pipeline_depth(8)
phase_len = 2000
t0 = time.perf_counter()
done = 0
phase = 0
while done < total:
    gs = (0.05, 1.0, 0.3)[phase % 3]
    ms = (4, 4, 4, 1, 8, 2)[phase % 6] if mixed else 4          # Msaa::Sample4 is the reference's default
    bbox = mixed and phase % 7 == 5
    s = CloudSettings(global_scale=gs, visualize_bounding_box=bbox)
    for v in views:
        v.msaa_samples = ms
    p.set_async(True)
    p.set_graphs(phase % 2 == 1)
    pvs = [p.prepare(v, s) for v in views]
    for f in range(phase_len):
        p.render(h, pvs[(f // 50) % 8], download=False)
    p.synchronize()
    last = (((phase_len - 1) // 50) % 8)
    got = framebuffer_as_tensor(p, 1080, 1920).cpu().numpy()
    p.set_async(False)
    ref = p.render(h, views[last], s)
    assert np.array_equal(got, ref), f"phase {phase} (global_scale {gs}, Msaa {ms}, bbox {bbox}): pipelined frame differs from the blocking one"
    done += phase_len
    phase += 1
    if os.environ.get("SOAK_VERBOSE"):
        print(f"phase {phase} gs {gs} ms {ms} bbox {bbox} graphs {phase % 2 == 0}: {time.perf_counter() - t0:.2f} s so far; {p.adaptive_counters()} graphs {p.graph_counters()}", flush=True)
dt = time.perf_counter() - t0
c, r = p.graph_counters()
st = p.stats()
print(f"{done} frames in {phase} phases, {dt:.1f} s ({done / dt:.0f} fps incl. checks); graph captures {c}, replays {r}; "
      f"frames re-run for capacity {st['regrow_count']}, last sort path {st['sort_path']}, list capacity {st['list_capacity']}; "
      f"{p.adaptive_counters()}; learning {p.learning_counters()}; all phase checks passed")
