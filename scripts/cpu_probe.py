import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "-", e)
from bevy_gaussian_splatting_amd import *
from oracle import oracle
oracle.build()
c = random_gaussians_3d_seeded(1_000_000, 2); v = View.headless()
for th in (128, 64, 32, 16, 8, 4, 1):
    oracle.set_threads(th)
    for mode in (SortMode.Rayon, SortMode.Radix):
        s = CloudSettings(sort_mode=mode)
        oracle.sort(c, v, s)
        t = time.perf_counter(); oracle.sort(c, v, s); dt = time.perf_counter() - t
        print(th, mode, f"{dt*1e3:.1f} ms")
