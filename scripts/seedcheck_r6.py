import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import helpers as H
from oracle import oracle
from bevy_gaussian_splatting_amd import GaussianSplattingPlugin, GaussianMode
p = GaussianSplattingPlugin(0)
for seed in (300196, 300047, 300097, 300034):
    c, v, s = H.random_case(1000 + seed, medium=True)
    s.aabb, s.gaussian_mode = True, GaussianMode.Gaussian2d
    if s.global_scale > 1.0: s.global_scale = 0.3
    cloud = c.to_f16() if seed % 4 == 3 else c
    cd = oracle.decode_f16(cloud) if cloud is not c else c
    p.set_binning("sort" if seed % 6 == 5 else "scan")
    h = p.upload(cloud)
    got = p.render(h, v, s)
    e = oracle.sort(cd, v, s)
    ref, amb = oracle.render(cd, e, v, s, with_ambiguity=True)
    strict, err = H.tolerance_mask(ref, got, None)
    lim = 1e-3 + 1e-4 * np.abs(ref)
    over = ~strict
    okamb = (err <= lim + amb[..., None] if amb.ndim == 2 else err <= lim + amb)
    trained = np.random.default_rng(11_000_000 + 1000 + seed).random() < 1/6
    print(seed, "trained" if trained else "random", "msaa", v.msaa_samples, "gs", s.global_scale, "gopacity", round(s.global_opacity,2), "beyond strict", int(over.sum()),
          "max excess %.3f" % float((err - lim)[over].max() if over.any() else 0), "all within ambiguity:", bool(okamb.all()), "binning", "sort" if seed % 6 == 5 else "scan")
    h.free()
