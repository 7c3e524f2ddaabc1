"""Seed 321 (medium sweep), pixel (551, 102): which splats does the GPU composite there that the oracle does not?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import helpers as H
from bevy_gaussian_splatting_amd import GaussianSplattingPlugin, PlanarGaussian3d
from oracle import oracle
c, v, s = H.random_case(1000 + 321, medium=True)
p = GaussianSplattingPlugin(0)
X, Y = 551, 102
def sub(idx):
    return PlanarGaussian3d(c.position_visibility[idx].copy(), c.spherical_harmonic[idx].copy(), c.rotation[idx].copy(), c.scale_opacity[idx].copy())
def probe(name, idx):
    cc = sub(idx)
    h = p.upload(cc); got = p.render(h, v, s); h.free()
    e = oracle.sort(cc, v, s); ref = oracle.render(cc, e, v, s, window=(X, Y, X + 1, Y + 1))
    r = ref[0, 0] if ref.shape[0] == 1 else ref[Y, X]
    print(name, len(idx), "gpu", got[Y, X], "oracle", r, "diff", np.abs(got[Y, X] - r).max(), flush=True)
    return got[Y, X], r
allidx = np.arange(len(c))
two = np.array([111095, 118762])
probe("all", allidx)
probe("two", two)
rest = np.setdiff1d(allidx, two)
g, r = probe("rest", rest)
# bisect the rest for the splat(s) that reach the pixel on the GPU
cand = rest
while len(cand) > 1:
    half = cand[:len(cand) // 2]
    cc = sub(half); h = p.upload(cc); got = p.render(h, v, s); h.free()
    hit = np.abs(got[Y, X] - r).max() > 1e-4
    cand = half if hit else cand[len(cand) // 2:]
print("culprit", cand, c.position_visibility[cand], c.rotation[cand], c.scale_opacity[cand], flush=True)
probe("culprit alone", cand)
