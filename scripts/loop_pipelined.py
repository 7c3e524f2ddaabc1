"""Pipelined frames of the headline workload (for rocprofv3 timelines):
python scripts/loop_pipelined.py <depth> [frames] [global_scale] [profiling 0/1] [debug flags] [streams]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
gs = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
prof = int(sys.argv[4]) if len(sys.argv) > 4 else 0
flags = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0
streams = int(sys.argv[6]) if len(sys.argv) > 6 else 0
p = GaussianSplattingPlugin(0)
n = int(os.environ.get("N", "1000000"))   # N=5000000 F16=1: the 5 M f16 cloud of BASELINE configs[2]
c = random_gaussians_3d_seeded(n, 2 if n == 1_000_000 else 3)
h = p.upload(c.to_f16() if os.environ.get("F16") else c)
v = headless_view(0)
s = CloudSettings(global_scale=gs)
p.set_async(True)
p.set_pipeline_depth(depth)
p.set_pipeline_streams(streams)
p.set_profiling(prof)
p.set_debug_flags(flags)
if os.environ.get("BGS_LOOP_GRAPHS"):
    p.set_graphs(True)   # frames replayed from a captured hipGraph
pv = p.prepare(v, s)
for _ in range(10):
    p.render(h, pv, download=False)
p.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    p.render(h, pv, download=False)
p.synchronize()
dt = time.perf_counter() - t0
print(f"flags {flags:#x} streams {streams} depth {depth} frames {frames}: {frames / dt:.1f} fps, {1e6 * dt / frames:.1f} us/frame")
