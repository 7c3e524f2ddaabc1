"""Which pixels of a randomized medium case are out of tolerance: python scripts/debug_seed.py <seed>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H
from bevy_gaussian_splatting_amd import GaussianSplattingPlugin
from oracle import oracle
seed = int(sys.argv[1])
c, v, s = H.random_case(1000 + seed, medium=True)
if s.global_scale > 1.0:
    s.global_scale = 0.3     # as the test does
cloud = c.to_f16() if seed % 4 == 3 else c
p = GaussianSplattingPlugin(0)
p.set_binning("sort" if seed % 6 == 5 else "scan")
h = p.upload(cloud)
if getattr(v, "depth_host", None) is not None:
    v.depth_device_ptr = p.upload_depth(v.depth_host)
got = p.render(h, v, s)
for m in (1, 4):   # the same pixel at the other sample count / with the other binning
    v.msaa_samples = m
    for b in ("scan", "sort"):
        p.set_binning(b)
        print("samples", m, b, "max err", np.abs(p.render(h, v, s) - oracle.render(oracle.decode_f16(cloud) if seed % 4 == 3 else c, oracle.sort(oracle.decode_f16(cloud) if seed % 4 == 3 else c, v, s), v, s, depth=getattr(v, "depth_host", None))).max())
v.msaa_samples = H.random_case(1000 + seed, medium=True)[1].msaa_samples
p.set_binning("sort" if seed % 6 == 5 else "scan")
cc = oracle.decode_f16(cloud) if seed % 4 == 3 else c
e = oracle.sort(cc, v, s)
ref, amb = oracle.render(cc, e, v, s, with_ambiguity=True, depth=getattr(v, "depth_host", None))
ok, err = H.tolerance_mask(ref, got, amb)
ys, xs = np.where(~ok.all(axis=2))
for y, x in zip(ys, xs):
    print("pixel", x, y, "got", got[y, x], "ref", ref[y, x], "amb", amb[y, x] if amb is not None else None, flush=True)
np.save("gpurun_out/debug_seed_got.npy", got[max(0, ys.min() - 8):ys.max() + 9, max(0, xs.min() - 8):xs.max() + 9])
