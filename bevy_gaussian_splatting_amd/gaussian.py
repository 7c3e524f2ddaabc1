"""Cloud data model: mirror of the reference's `Gaussian3d` / `PlanarGaussian3d`.

Reference: src/gaussian/formats/planar_3d.rs:28-54 (item + planar SoA),
src/gaussian/f32.rs (PositionVisibility, Rotation [w,x,y,z], ScaleOpacity),
src/material/spherical_harmonics.rs:114-120 (48 f32, index 3*k + c),
src/gaussian/f16.rs:29-55,244-263 (packed f16 planes),
src/gaussian/formats/planar_3d.rs:120-191 (random cloud distributions),
src/gaussian/formats/planar_3d.rs:193-251 (test_model).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SH_DEGREE = 3
SH_CHANNELS = 3
SH_COEFF_COUNT_PER_CHANNEL = (SH_DEGREE + 1) ** 2
SH_COEFF_COUNT = SH_COEFF_COUNT_PER_CHANNEL * SH_CHANNELS  # 48, already a multiple of 4
HALF_SH_COEFF_COUNT = SH_COEFF_COUNT // 2


@dataclass
class Gaussian3d:
    """One splat (AoS item, src/gaussian/formats/planar_3d.rs:45-54)."""

    position_visibility: np.ndarray  # [x, y, z, visibility]
    spherical_harmonic: np.ndarray  # [48]
    rotation: np.ndarray  # [w, x, y, z]
    scale_opacity: np.ndarray  # [sx, sy, sz, opacity]


class SphericalHarmonicCoefficients:
    """src/material/spherical_harmonics.rs:114-160: `set(channel_index, value)` writes raw
    coefficient `index` (so index 0/1/2 = DC of R/G/B)."""

    def __init__(self):
        self.coefficients = np.zeros(SH_COEFF_COUNT, dtype=np.float32)

    def set(self, index: int, value: float) -> None:
        self.coefficients[index] = value


def _f32(a, shape) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


class PlanarGaussian3d:
    """Planar (SoA) cloud: four planes of `n` rows
    (bevy_interleave `Planar` derive of `Gaussian3d`, src/gaussian/formats/planar_3d.rs:28-43)."""

    def __init__(self, position_visibility, spherical_harmonic, rotation, scale_opacity):
        n = len(position_visibility)
        self.position_visibility = _f32(position_visibility, (n, 4))
        self.spherical_harmonic = _f32(spherical_harmonic, (n, SH_COEFF_COUNT))
        self.rotation = _f32(rotation, (n, 4))
        self.scale_opacity = _f32(scale_opacity, (n, 4))

    def __len__(self) -> int:
        return self.position_visibility.shape[0]

    def len_sqrt_ceil(self) -> int:
        """src/gaussian/interface.rs: entry count is rounded up to a square by the caller."""
        return int(np.ceil(np.sqrt(len(self))))

    @staticmethod
    def from_interleaved(gaussians) -> "PlanarGaussian3d":
        gaussians = list(gaussians)
        n = len(gaussians)
        pv = np.zeros((n, 4), np.float32)
        sh = np.zeros((n, SH_COEFF_COUNT), np.float32)
        rot = np.zeros((n, 4), np.float32)
        so = np.zeros((n, 4), np.float32)
        for i, g in enumerate(gaussians):
            pv[i] = g.position_visibility
            sh[i] = g.spherical_harmonic
            rot[i] = g.rotation
            so[i] = g.scale_opacity
        return PlanarGaussian3d(pv, sh, rot, so)

    def iter(self):
        for i in range(len(self)):
            yield Gaussian3d(
                self.position_visibility[i].copy(),
                self.spherical_harmonic[i].copy(),
                self.rotation[i].copy(),
                self.scale_opacity[i].copy(),
            )

    def nbytes(self) -> int:
        return (
            self.position_visibility.nbytes
            + self.spherical_harmonic.nbytes
            + self.rotation.nbytes
            + self.scale_opacity.nbytes
        )

    def to_f16(self) -> "PlanarGaussian3dF16":
        return PlanarGaussian3dF16.from_f32(self)

    def slice(self, start: int, stop: int) -> "PlanarGaussian3d":
        return PlanarGaussian3d(
            self.position_visibility[start:stop],
            self.spherical_harmonic[start:stop],
            self.rotation[start:stop],
            self.scale_opacity[start:stop],
        )


def _pack_f32s_to_u32(upper: np.ndarray, lower: np.ndarray) -> np.ndarray:
    """src/gaussian/f16.rs:244-252: IEEE round-to-nearest-even f32->f16, first argument in
    the high half."""
    with np.errstate(over="ignore"):  # values beyond the half range become +-inf, as in the reference
        u = np.asarray(upper, dtype=np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
        l = np.asarray(lower, dtype=np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
    return (u << np.uint32(16)) | l


def _unpack_u32(v: np.ndarray):
    hi = (v >> np.uint32(16)).astype(np.uint16).view(np.float16).astype(np.float32)
    lo = (v & np.uint32(0xFFFF)).astype(np.uint16).view(np.float16).astype(np.float32)
    return hi, lo


class PlanarGaussian3dF16:
    """f16 planar cloud (the reference's dormant `f16` storage:
    src/render/bindings.wgsl:102-140, src/gaussian/f16.rs:29-55).

    position_visibility stays f32 [n,4]; spherical_harmonic is u32 [n,24] with the EVEN
    coefficient in the low half (src/render/planar.wgsl:117-130); rotation_scale_opacity is
    u32 [n,4] = [rot0|rot1], [rot2|rot3], [s0|s1], [s2|opacity], first value in the high half.
    """

    def __init__(self, position_visibility, spherical_harmonic_h2, rotation_scale_opacity):
        n = len(position_visibility)
        self.position_visibility = _f32(position_visibility, (n, 4))
        self.spherical_harmonic = np.ascontiguousarray(spherical_harmonic_h2, dtype=np.uint32)
        self.rotation_scale_opacity = np.ascontiguousarray(rotation_scale_opacity, dtype=np.uint32)
        if self.spherical_harmonic.shape != (n, HALF_SH_COEFF_COUNT):
            raise ValueError("spherical_harmonic must be [n, 24] u32")
        if self.rotation_scale_opacity.shape != (n, 4):
            raise ValueError("rotation_scale_opacity must be [n, 4] u32")

    def __len__(self) -> int:
        return self.position_visibility.shape[0]

    def nbytes(self) -> int:
        return (
            self.position_visibility.nbytes
            + self.spherical_harmonic.nbytes
            + self.rotation_scale_opacity.nbytes
        )

    @staticmethod
    def from_f32(cloud: PlanarGaussian3d) -> "PlanarGaussian3dF16":
        sh = cloud.spherical_harmonic
        sh_h2 = _pack_f32s_to_u32(sh[:, 1::2], sh[:, 0::2])
        rot, so = cloud.rotation, cloud.scale_opacity
        rso = np.stack(
            [
                _pack_f32s_to_u32(rot[:, 0], rot[:, 1]),
                _pack_f32s_to_u32(rot[:, 2], rot[:, 3]),
                _pack_f32s_to_u32(so[:, 0], so[:, 1]),
                _pack_f32s_to_u32(so[:, 2], so[:, 3]),
            ],
            axis=1,
        )
        return PlanarGaussian3dF16(cloud.position_visibility, sh_h2, rso)

    def to_f32(self) -> PlanarGaussian3d:
        """Decode exactly as the shader does (src/render/planar.wgsl:117-176)."""
        n = len(self)
        hi, lo = _unpack_u32(self.spherical_harmonic)
        sh = np.empty((n, SH_COEFF_COUNT), np.float32)
        sh[:, 0::2] = lo
        sh[:, 1::2] = hi
        r = self.rotation_scale_opacity
        r0h, r0l = _unpack_u32(r[:, 0])
        r1h, r1l = _unpack_u32(r[:, 1])
        s0h, s0l = _unpack_u32(r[:, 2])
        s1h, s1l = _unpack_u32(r[:, 3])
        rot = np.stack([r0h, r0l, r1h, r1l], axis=1)
        so = np.stack([s0h, s0l, s1h, s1l], axis=1)
        return PlanarGaussian3d(self.position_visibility, sh, rot, so)


def random_gaussians_3d_seeded(n: int, seed: int) -> PlanarGaussian3d:
    """Seeded synthetic cloud with the reference's distributions
    (src/gaussian/formats/planar_3d.rs:120-168,182-191): rotation ~ U(-1,1)^4 (NOT
    normalised), position ~ U(-20,20)^3 with visibility 1, scale ~ U(0,1)^3,
    opacity ~ U(0,0.8), SH ~ U(-1,1)^48.

    The reference draws from `rand::StdRng` (ChaCha12, third-party, not reproduced); this
    build uses numpy PCG64 with the same per-field order, so clouds have the same
    statistics but not the same bits."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rot = rng.uniform(-1.0, 1.0, size=(n, 4)).astype(np.float32)
    pv = np.empty((n, 4), np.float32)
    pv[:, :3] = rng.uniform(-20.0, 20.0, size=(n, 3)).astype(np.float32)
    pv[:, 3] = 1.0
    so = np.empty((n, 4), np.float32)
    so[:, :3] = rng.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    so[:, 3] = rng.uniform(0.0, 0.8, size=n).astype(np.float32)
    sh = rng.uniform(-1.0, 1.0, size=(n, SH_COEFF_COUNT)).astype(np.float32)
    return PlanarGaussian3d(pv, sh, rot, so)


def trained_like_gaussians_3d_seeded(n: int, seed: int, patches: int = 96) -> PlanarGaussian3d:
    """A synthetic cloud with the STATISTICS of a trained 3DGS asset, for workloads the reference's own generator does
    not give (round 6; the reference demos trained assets — README.md:88 `scenes/icecream.gcloud` — but ships none, and
    `random_gaussians_3d` above is unit-scale splats with SH ~ U(-1, 1)^48: colours up to 15, a camera inside the cloud):

    * positions on SURFACES: `patches` rectangles (2-12 units wide, random pose inside the (-20, 20)^3 box the random
      clouds fill, so the headless camera sees the scene the same way) with a splat density proportional to their area
      and N(0, 0.02) of noise along the normal — what structure-from-motion points and the splats grown from them look like;
    * scales LOG-NORMAL around the spacing of the splats on their patch (median ~ sqrt(area per splat), sigma_ln 0.6),
      FLAT: the axis along the patch normal is 5-25 % of the tangential ones;
    * rotations: unit quaternions that turn the local z axis onto the patch normal, a random angle about it (the PLY
      loader normalises quaternions, src/io/ply.rs:118-124);
    * opacity BIMODAL: 60 % Beta(8, 1.2) (opaque surface splats), 40 % Beta(1.2, 6) (the translucent haze training leaves);
    * SH: the DC term such that the colour 0.5 + 0.2821 sh0 lies in [0.05, 0.95] (a base colour per patch plus per-splat
      variation), the higher bands N(0, 0.015) halving by band — the view-dependent part moves a colour by a few
      hundredths, so colours stay in [0, 1] but for the odd splat at the ends of the range.

    Meant for `CloudSettings(global_scale=1.0)`. Same per-field order in `bgs::PlanarGaussian3d::trained_like`
    (include/bgs.hpp; std::mt19937_64 there, numpy PCG64 here: the same statistics, not the same bits)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = int(patches)
    centre = rng.uniform(-18.0, 18.0, size=(P, 3))
    # patch frames: normal from a random direction, two tangents
    nrm = rng.normal(size=(P, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    helper = np.where(np.abs(nrm[:, :1]) < 0.9, np.array([[1.0, 0.0, 0.0]]), np.array([[0.0, 1.0, 0.0]]))
    tu = np.cross(nrm, helper)
    tu /= np.linalg.norm(tu, axis=1, keepdims=True)
    tv = np.cross(nrm, tu)
    size = rng.uniform(2.0, 12.0, size=(P, 2))
    area = size[:, 0] * size[:, 1]
    base_rgb = rng.uniform(0.15, 0.85, size=(P, 3))
    pid = rng.choice(P, size=n, p=area / area.sum())
    uv = rng.uniform(-0.5, 0.5, size=(n, 2)) * size[pid]
    off = rng.normal(0.0, 0.02, size=n)
    pos = centre[pid] + uv[:, :1] * tu[pid] + uv[:, 1:] * tv[pid] + off[:, None] * nrm[pid]
    pv = np.empty((n, 4), np.float32)
    pv[:, :3] = pos.astype(np.float32)
    pv[:, 3] = 1.0
    # scales: log-normal around the splat spacing of the patch, flat along the normal
    spacing = np.sqrt(area.sum() / max(n, 1))
    tang = spacing * np.exp(rng.normal(0.0, 0.6, size=(n, 2)))
    flat = rng.uniform(0.05, 0.25, size=n) * np.sqrt(tang[:, 0] * tang[:, 1])
    so = np.empty((n, 4), np.float32)
    so[:, 0] = tang[:, 0]; so[:, 1] = tang[:, 1]; so[:, 2] = flat
    pick = rng.uniform(size=n) < 0.6
    so[:, 3] = np.where(pick, rng.beta(8.0, 1.2, size=n), rng.beta(1.2, 6.0, size=n)).astype(np.float32)
    # rotation [w, x, y, z]: local axes (tu', tv', normal) with tu' = tu turned by a random angle about the normal
    ang = rng.uniform(0.0, 2.0 * np.pi, size=n)
    ca, sa = np.cos(ang)[:, None], np.sin(ang)[:, None]
    ax = ca * tu[pid] + sa * tv[pid]
    ay = -sa * tu[pid] + ca * tv[pid]
    az = nrm[pid]
    # the shader's rotation matrix is built COLUMN-wise from the quaternion and used as M = S * R (rows of R scaled): row i
    # of R is the world direction of local axis i (helpers.wgsl:137-168) -> quaternion of the matrix whose ROWS are ax, ay, az
    R = np.stack([ax, ay, az], axis=1)                      # R[n, row, col]
    # the reference's constructor: R[row 0] = (1 - 2(y^2 + z^2), 2(xy - rz), 2(xz + ry)), ... (see compute_covariance_3d):
    # it is the transpose of the usual rotation matrix of (r, x, y, z), so take the quaternion of R^T
    Rt = np.transpose(R, (0, 2, 1))
    tr = Rt[:, 0, 0] + Rt[:, 1, 1] + Rt[:, 2, 2]
    qw = np.sqrt(np.maximum(1.0 + tr, 1e-12)) / 2.0
    qx = np.copysign(np.sqrt(np.maximum(1.0 + Rt[:, 0, 0] - Rt[:, 1, 1] - Rt[:, 2, 2], 0.0)) / 2.0, Rt[:, 2, 1] - Rt[:, 1, 2])
    qy = np.copysign(np.sqrt(np.maximum(1.0 - Rt[:, 0, 0] + Rt[:, 1, 1] - Rt[:, 2, 2], 0.0)) / 2.0, Rt[:, 0, 2] - Rt[:, 2, 0])
    qz = np.copysign(np.sqrt(np.maximum(1.0 - Rt[:, 0, 0] - Rt[:, 1, 1] + Rt[:, 2, 2], 0.0)) / 2.0, Rt[:, 1, 0] - Rt[:, 0, 1])
    rot = np.stack([qw, qx, qy, qz], axis=1)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    # SH: DC-dominated, colours in [0.05, 0.95]
    rgb = np.clip(base_rgb[pid] + rng.normal(0.0, 0.08, size=(n, 3)), 0.05, 0.95)
    sh = np.zeros((n, SH_COEFF_COUNT), np.float32)
    sh[:, 0:3] = ((rgb - 0.5) / 0.2820947917738781).astype(np.float32)
    band_sigma = np.concatenate([np.full(3, 0.015), np.full(5, 0.0075), np.full(7, 0.004)])   # coefficients 1..15
    sh[:, 3:] = (rng.normal(size=(n, 15, 3)) * band_sigma[None, :, None]).reshape(n, 45).astype(np.float32)
    return PlanarGaussian3d(pv, sh, rot.astype(np.float32), so)


def random_gaussians_3d(n: int) -> PlanarGaussian3d:
    """src/gaussian/formats/planar_3d.rs:171-180 (thread RNG -> OS entropy seed)."""
    return random_gaussians_3d_seeded(n, int(np.random.SeedSequence().entropy % (1 << 63)))


def test_model(seed: int = 0) -> PlanarGaussian3d:
    """`PlanarGaussian3d::test_model()` geometry (src/gaussian/formats/planar_3d.rs:193-251):
    8 splats at (+-0.5)^3 + a duplicate of the first; identity rotation, scale 0.125,
    opacity 0.125, random SH (seeded here)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base_sh = rng.uniform(-1.0, 1.0, size=SH_COEFF_COUNT).astype(np.float32)
    gs = []
    for x in (-0.5, 0.5):
        for y in (-0.5, 0.5):
            for z in (-0.5, 0.5):
                sh = base_sh.copy()
                rng.shuffle(sh)
                gs.append(
                    Gaussian3d(
                        np.array([x, y, z, 1.0], np.float32),
                        sh,
                        np.array([1.0, 0.0, 0.0, 0.0], np.float32),
                        np.array([0.125, 0.125, 0.125, 0.125], np.float32),
                    )
                )
    gs.append(gs[0])
    return PlanarGaussian3d.from_interleaved(gs)


def compute_covariance_3d(rotation, scale) -> np.ndarray:
    """CPU twin of the shader's covariance (src/gaussian/covariance.rs:4-41): `S = diag(scale)`, `R` from
    the [w, x, y, z] quaternion with glam's column constructor, `M = S * R`, `Sigma = M^T * M`; returns the
    six unique entries [xx, xy, xz, yy, yz, zz] in float32 (what `Covariance3dOpacity::from`,
    src/gaussian/f32.rs:238-251, stores when the `precompute_covariance_3d` feature is on).
    Accepts one splat ([4], [3]) or planes ([n, 4], [n, 3])."""
    q = np.atleast_2d(np.asarray(rotation, np.float32))
    sc = np.atleast_2d(np.asarray(scale, np.float32))[:, :3]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    one, two = np.float32(1.0), np.float32(2.0)
    # columns of R (Mat3::from_cols)
    c0 = np.stack([one - two * (y * y + z * z), two * (x * y - r * z), two * (x * z + r * y)], 1)
    c1 = np.stack([two * (x * y + r * z), one - two * (x * x + z * z), two * (y * z - r * x)], 1)
    c2 = np.stack([two * (x * z - r * y), two * (y * z + r * x), one - two * (x * x + y * y)], 1)
    R = np.stack([c0, c1, c2], 2).astype(np.float32)              # R[n, row, col]
    M = (sc[:, :, None] * R).astype(np.float32)                   # S * R: row i scaled by scale_i
    Sigma = np.einsum("nki,nkj->nij", M, M).astype(np.float32)    # M^T * M
    out = np.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2], Sigma[:, 2, 2]], 1)
    return out[0] if np.asarray(rotation).ndim == 1 else out


def covariance_3d_opacity(cloud: "PlanarGaussian3d") -> np.ndarray:
    """`Covariance3dOpacity` plane ([n, 8] float32: cov3d[6], opacity, pad) of a cloud
    (src/gaussian/f32.rs:218-251)."""
    out = np.zeros((len(cloud), 8), np.float32)
    out[:, :6] = compute_covariance_3d(cloud.rotation, cloud.scale_opacity[:, :3])
    out[:, 6] = cloud.scale_opacity[:, 3]
    return out
