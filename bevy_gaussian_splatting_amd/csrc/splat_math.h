// splat_math.h — per-splat arithmetic of the hot path (device functions).
//
// What the reference computes per quad vertex in vs_points (src/render/gaussian.wgsl:184-436)
// and per key in radix_sort_a (src/sort/radix.wgsl:86-101), done ONCE per splat here.
//
// Evaluation order is fixed (see DESIGN.md "arithmetic contract"): the translation units
// that include this header are compiled with -ffp-contract=off so that
//   dot(a,b)   = ((a0*b0 + a1*b1) + a2*b2) [+ a3*b3]
//   (M*v)[r]   = (((M[0][r]*v0 + M[1][r]*v1) + M[2][r]*v2) + M[3][r]*v3)
//   (A*B)[c][r]= ((A[0][r]*B[c][0] + A[1][r]*B[c][1]) + A[2][r]*B[c][2])
// which makes the sort keys and every cull decision bit-exact with the parity oracle.
//
// BGS_HD expands to __host__ __device__ under hipcc and to nothing under g++; the g++
// build exists only for tests/test_device_math_host.py (a CPU pre-flight of this header
// against the oracle, so transcription slips are caught without a GPU). It is not a
// fallback: libbgs never calls these functions on the host.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "bgs_device.h"
#include "exact_log.h"

#if defined(__HIPCC__)
#define BGS_HD __host__ __device__ __forceinline__
#else
#define BGS_HD static inline
#endif

// Work order of the rasteriser's grid with S contiguous runs per XCD (render_kernels.hip: xcd_remap_runs): workgroup b
// runs on XCD b % 8 (observed dispatch; a speed assumption only) and is the (b / 8)-th of that XCD's q + (xcd < r)
// work items, n = 8 q + r. The items [0, n) are cut into 8 S runs dealt round-robin — run (s, x) belongs to XCD x and
// holds cnt_x / S (+ 1 for the first cnt_x % S runs) items — and the XCD's items are its runs one after the other.
// A bijection of [0, n) onto itself for every n and S >= 1.
BGS_HD uint32_t xcd_runs_item(const uint32_t b, const uint32_t n, const uint32_t S) {
    const uint32_t q = n / 8u, r = n % 8u, xcd = b % 8u;
    uint32_t i = b / 8u, start = 0u;
    for (uint32_t s = 0u; s < S; ++s) {
        uint32_t mine = 0u, before = 0u, row = 0u;
        for (uint32_t x = 0u; x < 8u; ++x) {
            const uint32_t cnt = q + (x < r ? 1u : 0u);
            const uint32_t len = cnt / S + (s < cnt % S ? 1u : 0u);
            if (x < xcd) before += len;
            if (x == xcd) mine = len;
            row += len;
        }
        if (i < mine) return start + before + i;
        i -= mine;
        start += row;
    }
    return n;   // not reached for b < n
}

// The one transcendental that feeds only a COLOUR (never a cull decision, a sort key or a quad) — the 2.4 power of
// srgb_to_linear — uses the hardware exp2/log2 on the device (~1 ulp); the host build keeps libm. Everything that
// reaches a compare is +, -, *, /, sqrt (correctly rounded on gfx950) or ln_f32_cr (exact_log.h): ln(opacity) of
// the adaptive cutoff goes into the 2DGS degeneracy tests, so it is the correctly rounded value on every side.
#if defined(__HIP_DEVICE_COMPILE__)
#define BGS_FAST_POW(x, y) __builtin_amdgcn_exp2f((y) * __builtin_amdgcn_logf(x))
#else
#define BGS_FAST_POW(x, y) powf(x, y)
#endif

namespace bgs {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct M3 { float m[9]; };  // column-major m[3*c + r]

BGS_HD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}

BGS_HD float dot2(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
BGS_HD float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
BGS_HD V3 mul3(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
BGS_HD V3 sub3(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
BGS_HD V3 scale3(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
BGS_HD V3 normalize3(V3 a) {
    float len = sqrtf(dot3(a, a));
    return V3{a.x / len, a.y / len, a.z / len};
}

BGS_HD V4 m4_mul_point(const float* m, V3 p) {  // M * vec4(p, 1)
    V4 r;
    r.x = ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12] * 1.0f;
    r.y = ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13] * 1.0f;
    r.z = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] * 1.0f;
    r.w = ((m[3] * p.x + m[7] * p.y) + m[11] * p.z) + m[15] * 1.0f;
    return r;
}

BGS_HD M3 m3_cols(V3 a, V3 b, V3 c) { return M3{{a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z}}; }
BGS_HD M3 m3_transpose(const M3& a) {
    return M3{{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}};
}
BGS_HD M3 m3_mul(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
            r.m[3 * c + rr] = (a.m[rr] * b.m[3 * c] + a.m[3 + rr] * b.m[3 * c + 1]) +
                              a.m[6 + rr] * b.m[3 * c + 2];
    return r;
}
BGS_HD M3 m3_from_m4(const float* m) {
    return M3{{m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]}};
}

// src/render/transform.wgsl:5-8
BGS_HD V4 world_to_clip(const FrameParams& fp, V3 world_pos) {
    V4 h = m4_mul_point(fp.clip_from_world, world_pos);
    float d = h.w + 0.000000001f;
    return V4{h.x / d, h.y / d, h.z / d, h.w / d};
}
// src/render/transform.wgsl:10-14
BGS_HD bool in_frustum(V4 c) {
    return fabsf(c.x) < 1.1f && fabsf(c.y) < 1.1f && fabsf(c.z - 0.5f) < 0.5f;
}

// in_frustum(world_to_clip(fp, p)) — the SAME verdict, bit for bit — without the three IEEE divisions wherever the
// verdict is clear (sort_key_fast below; sort_key_kind is the reference's statement as it stands). keygen runs one wave per SIMD and the divisions of a thread's 16 splats queue behind each other
// (v_div_scale -> v_div_fmas goes through VCC): they were half of the kernel's arithmetic time.
//   q~ = h * rcp(d) differs from fl(h / d) by less than 2^-22 relative for a normal d (v_rcp_f32: 1 ulp, the product:
//   half an ulp, the correctly rounded quotient: half an ulp), so with a guard band of 2^-20 around each threshold
//     |q~x|, |q~y| < 1.1 (1 - 2^-20)  and  2^-20 < q~z < 1 - 2^-20      =>  every compare of in_frustum is true,
//     |q~x| or |q~y| > 1.1 (1 + 2^-20)  or  q~z < -2^-20  or  q~z > 1 + 2^-20  =>  one of them is false
//   (fabsf(c.z - 0.5f) < 0.5f is true on [2^-20, 1 - 2^-20] with 2^-25 of rounding to spare and false for every c.z < 0
//   and every c.z > 1). Anything else — a value inside a band, a NaN, a d outside [2^-100, 2^100] (where rcp or the
//   product could leave the normal range) — takes the divisions. tests/test_gpu_parity.py puts a million points
//   within a few ulp of every threshold.
#if defined(__HIP_DEVICE_COMPILE__)
#define BGS_RCP_APPROX(x) __builtin_amdgcn_rcpf(x)
#else
#define BGS_RCP_APPROX(x) (1.0f / (x))
#endif
// Returns 1 (inside), 0 (outside) or 2 (not clear: the caller takes the divisions, in_frustum(world_to_clip(..))).
// Straight-line on purpose — no branch, bitwise combinations of the compares (each false for a NaN) — so that the
// splats a thread owns are independent chains the compiler interleaves.
BGS_HD uint32_t frustum_verdict_fast(const FrameParams& fp, V3 world_pos) {
    const V4 h = m4_mul_point(fp.clip_from_world, world_pos);
    const float d = h.w + 0.000000001f;
    const float ad = fabsf(d);
    constexpr float E = 0x1p-20f, LO = 1.1f * (1.0f - 0x1p-20f), HI = 1.1f * (1.0f + 0x1p-20f);
    const float r = BGS_RCP_APPROX(d);
    const float qx = fabsf(h.x * r), qy = fabsf(h.y * r), qz = h.z * r;
    const bool normal = (ad >= 0x1p-100f) & (ad <= 0x1p100f);
    const bool out = normal & ((qx > HI) | (qy > HI) | (qz < -E) | (qz > 1.0f + E));
    const bool in = normal & (qx < LO) & (qy < LO) & (qz > E) & (qz < 1.0f - E);
    return in ? 1u : (out ? 0u : 2u);
}
// Sort key of one splat for every SortMode.
//   SORT_RADIX: src/sort/radix.wgsl:86-101 (cull + inverted distance bits, >> key_shift)
//   SORT_RAYON/STD: src/sort/rayon.rs:91-97 stores bits(dist2) and sorts DESCENDING; the
//     device sorts ascending on ~bits and un-inverts in the last pass, so this returns ~bits.
//   SORT_NONE: src/sort/mod.rs:347-354 (key = 1, draw order = index order)
// KIND is the branch of the three a frame takes (0: SORT_NONE, 1: SORT_RADIX, 2: the CPU sorts' key), so that a
// kernel can pick it once per launch instead of once per splat.
template <int KIND>
BGS_HD uint32_t sort_key_kind(const FrameParams& fp, V3 pos) {
    if constexpr (KIND == 0) {
        return 1u;
    } else {
        V4 t4 = m4_mul_point(fp.transform, pos);
        V3 tp{t4.x, t4.y, t4.z};
        V3 cam{fp.cam[0], fp.cam[1], fp.cam[2]};
        if constexpr (KIND == 2) {
            V3 d = sub3(cam, tp);
            float dist2 = (d.x * d.x + d.y * d.y) + d.z * d.z;
            // a NaN key's sign/payload is platform-dependent (x86 vs gfx950) and its order is
            // unspecified in the reference (partial_cmp -> Equal): store the canonical quiet NaN
            const uint32_t bits = dist2 != dist2 ? 0x7FC00000u : f2u(dist2);
            return 0xFFFFFFFFu - bits;
        } else {
            uint32_t key = KEY_CULLED;
            V4 clip = world_to_clip(fp, tp);
            V3 diff = sub3(tp, cam);
            float dist2 = dot3(diff, diff);
            uint32_t key_distance = 0xFFFFFFFFu - f2u(dist2);
            if (in_frustum(clip)) key = key_distance;
            return key >> fp.key_shift;
        }
    }
}
// The same key in two steps, for a thread that owns several splats (keygen): sort_key_fast is straight-line code and
// says `unsure` where the frustum verdict needs the divisions; the caller then asks sort_key_kind for those splats.
template <int KIND>
BGS_HD uint32_t sort_key_fast(const FrameParams& fp, V3 pos, bool& unsure) {
    unsure = false;
    if constexpr (KIND != 1) {
        return sort_key_kind<KIND>(fp, pos);
    } else {
        V4 t4 = m4_mul_point(fp.transform, pos);
        V3 tp{t4.x, t4.y, t4.z};
        V3 cam{fp.cam[0], fp.cam[1], fp.cam[2]};
        V3 diff = sub3(tp, cam);
        float dist2 = dot3(diff, diff);
        const uint32_t key_distance = 0xFFFFFFFFu - f2u(dist2);
        const uint32_t v = frustum_verdict_fast(fp, tp);
        unsure = v == 2u;
        return (v == 1u ? key_distance : KEY_CULLED) >> fp.key_shift;
    }
}
BGS_HD uint32_t sort_key(const FrameParams& fp, V3 pos) {
    if (fp.sort_mode == SORT_NONE) return sort_key_kind<0>(fp, pos);
    if (fp.sort_mode != SORT_RADIX) return sort_key_kind<2>(fp, pos);
    return sort_key_kind<1>(fp, pos);
}

// src/render/helpers.wgsl:137-157 (column-major constructor), rotation = [w, x, y, z]
BGS_HD M3 rotation_matrix(const float* rot) {
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    return m3_cols(
        V3{1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - r * z), 2.0f * (x * z + r * y)},
        V3{2.0f * (x * y + r * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - r * x)},
        V3{2.0f * (x * z - r * y), 2.0f * (y * z + r * x), 1.0f - 2.0f * (x * x + y * y)});
}
// src/render/helpers.wgsl:159-168
BGS_HD M3 scale_matrix(const float* scale, float gs) {
    return m3_cols(V3{scale[0] * gs, 0.0f, 0.0f}, V3{0.0f, scale[1] * gs, 0.0f},
                   V3{0.0f, 0.0f, scale[2] * gs});
}

// src/render/gaussian_3d.wgsl:49-72 followed by src/render/helpers.wgsl:8-47.
// cov3d_pre != nullptr: the PRECOMPUTE_COVARIANCE_3D variant (gaussian_3d.wgsl:77-88, planar.wgsl:132-152):
// the six covariance entries come from the cloud (Covariance3dOpacity, src/gaussian/f32.rs:218-251) and
// compute_cov3d is skipped — with it the model transform's linear part and global_scale, which the
// reference applies inside compute_cov3d only.
// The matrix products are written out with their STRUCTURE used: S is diagonal, J has a zero column and two zero
// entries, Sigma and T Sigma T^T are symmetric, and only the upper-left 2x2 block of the projected covariance is read.
// Every term that is kept is the reference's term in the reference's order ((a0 b0 + a1 b1) + a2 b2); what is dropped
// is a multiplication by a structural zero and the addition of its product, which in IEEE arithmetic change nothing but
// the sign of a zero result (x + (+-0) = x, and y * 0 = +-0 for every finite y) — and NaN propagation for non-finite
// scales, where the reference's 0 * inf poisons entries this form leaves finite. A symmetric entry [i][j] is the same
// three products added in the same order as [j][i] (IEEE multiplication commutes), so it is formed once.
BGS_HD void cov2d_3dgs(const FrameParams& fp, V3 position, const float* scale, const float* rot,
                       const float* cov3d_pre, float out[3]) {
    float c0, c1, c2, c3, c4, c5;
    if (cov3d_pre) {
        c0 = cov3d_pre[0]; c1 = cov3d_pre[1]; c2 = cov3d_pre[2]; c3 = cov3d_pre[3]; c4 = cov3d_pre[4]; c5 = cov3d_pre[5];
    } else {
        const M3 R = rotation_matrix(rot);
        const float s0 = scale[0] * fp.global_scale, s1 = scale[1] * fp.global_scale, s2 = scale[2] * fp.global_scale;
        // M = S R (helpers.wgsl:12): row r of R scaled by s_r. Column-major m[3 c + r].
        float M[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) { M[3 * c] = s0 * R.m[3 * c]; M[3 * c + 1] = s1 * R.m[3 * c + 1]; M[3 * c + 2] = s2 * R.m[3 * c + 2]; }
        // Sigma = M^T M: Sigma[c][r] = dot(column r of M, column c of M), summed left to right over the ROWS of M
        auto sig = [&](int r, int c) { return (M[3 * r] * M[3 * c] + M[3 * r + 1] * M[3 * c + 1]) + M[3 * r + 2] * M[3 * c + 2]; };
        const float g00 = sig(0, 0), g01 = sig(0, 1), g02 = sig(0, 2), g11 = sig(1, 1), g12 = sig(1, 2), g22 = sig(2, 2);
        const float G[9] = {g00, g01, g02, g01, g11, g12, g02, g12, g22};   // symmetric, column-major
        const float* tm = fp.transform;   // T = mat3(transform[0].xyz, [1].xyz, [2].xyz): T[c][r] = tm[4 c + r]
        // A = T Sigma: A[c][r] = (T[0][r] G[c][0] + T[1][r] G[c][1]) + T[2][r] G[c][2]
        float A[9];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
                A[3 * c + r] = (tm[r] * G[3 * c] + tm[4 + r] * G[3 * c + 1]) + tm[8 + r] * G[3 * c + 2];
        // TS = A T^T: TS[c][r] = (A[0][r] T^T[c][0] + A[1][r] T^T[c][1]) + A[2][r] T^T[c][2], T^T[c][k] = T[k][c] = tm[4 k + c]
        auto ts = [&](int r, int c) { return (A[r] * tm[c] + A[3 + r] * tm[4 + c]) + A[6 + r] * tm[8 + c]; };
        c0 = ts(0, 0); c1 = ts(1, 0); c2 = ts(2, 0); c3 = ts(1, 1); c4 = ts(2, 1); c5 = ts(2, 2);
    }
    // Vrk = (c0 c1 c2 / c1 c3 c4 / c2 c4 c5), symmetric: transpose(Vrk) = Vrk
    const float Vrk[9] = {c0, c1, c2, c1, c3, c4, c2, c4, c5};

    V4 t = m4_mul_point(fp.view_from_world, position);
    float sI = 1.0f / (t.z * t.z);
    // J = cols (fx / tz, 0, -(fx tx) sI), (0, -fy / tz, (fy ty) sI), (0, 0, 0)      (helpers.wgsl:20-33)
    const float j00 = fp.focal_x / t.z, j02 = -(fp.focal_x * t.x) * sI;
    const float j11 = -fp.focal_y / t.z, j12 = (fp.focal_y * t.y) * sI;
    // W = transpose(mat3(view_from_world)): W[k][r] = vfw[4 r + k]. Tm = W J; its third column is zero.
    // Tm[0][r] = (W[0][r] j00 + W[1][r] 0) + W[2][r] j02,  Tm[1][r] = (W[0][r] 0 + W[1][r] j11) + W[2][r] j12
    const float* vw = fp.view_from_world;
    float t0[3], t1[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        t0[r] = vw[4 * r] * j00 + vw[4 * r + 2] * j02;
        t1[r] = vw[4 * r + 1] * j11 + vw[4 * r + 2] * j12;
    }
    // B = Tm^T Vrk^T, rows 0 and 1: B[c][r] = (Tm[r][0] Vrk[c][0] + Tm[r][1] Vrk[c][1]) + Tm[r][2] Vrk[c][2]
    float b0[3], b1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        b0[c] = (t0[0] * Vrk[3 * c] + t0[1] * Vrk[3 * c + 1]) + t0[2] * Vrk[3 * c + 2];
        b1[c] = (t1[0] * Vrk[3 * c] + t1[1] * Vrk[3 * c + 1]) + t1[2] * Vrk[3 * c + 2];
    }
    // cov = B Tm: cov[c][r] = (B[0][r] Tm[c][0] + B[1][r] Tm[c][1]) + B[2][r] Tm[c][2]; entries [0][0], [0][1], [1][1]
    out[0] = ((b0[0] * t0[0] + b0[1] * t0[1]) + b0[2] * t0[2]) + 0.3f;
    out[1] = (b1[0] * t0[0] + b1[1] * t0[1]) + b1[2] * t0[2];
    out[2] = ((b1[0] * t1[0] + b1[1] * t1[1]) + b1[2] * t1[2]) + 0.3f;
}

// src/render/helpers.wgsl:49-120: bb.xy (NDC offset) and bb.zw for one quad corner.
BGS_HD void bounding_box_clip(const FrameParams& fp, const float c2d[3], V2 dir, float cutoff,
                              float out[4]) {
    float det = c2d[0] * c2d[2] - c2d[1] * c2d[1];
    float trace = c2d[0] + c2d[2];
    float mid = 0.5f * trace;
    float discriminant = fmaxf(0.0f, mid * mid - det);
    float term = sqrtf(discriminant);
    float lambda1 = mid + term;
    float lambda2 = fmaxf(mid - term, 0.0f);
    float x_axis_length = sqrtf(lambda1);
    float y_axis_length = sqrtf(lambda2);
    if (fp.aabb) {
        float radius_px = cutoff * fmaxf(x_axis_length, y_axis_length);
        out[0] = (radius_px / fp.viewport_w) * dir.x;
        out[1] = (radius_px / fp.viewport_h) * dir.y;
        out[2] = radius_px * dir.x;
        out[3] = radius_px * dir.y;
        return;
    }
    float a = (c2d[0] - c2d[2]) * (c2d[0] - c2d[2]);
    float b = sqrtf(a + 4.0f * c2d[1] * c2d[1]);
    float major_radius = sqrtf((c2d[0] + c2d[2] + b) * 0.5f);
    float minor_radius = sqrtf((c2d[0] + c2d[2] - b) * 0.5f);
    V2 bounds{cutoff * major_radius, cutoff * minor_radius};
    V2 ev{-c2d[1], lambda1 - c2d[0]};
    float evlen = sqrtf(dot2(ev, ev));
    V2 e1{ev.x / evlen, ev.y / evlen};
    V2 e2{e1.y, -e1.x};
    V2 sv{dir.x * bounds.x, dir.y * bounds.y};
    V2 rv{dot2(sv, V2{e1.x, e2.x}), dot2(sv, V2{e1.y, e2.y})};
    out[0] = rv.x * fp.inv_viewport_w;   // rv * (1.0 / viewport): the reciprocal is a constant of the frame
    out[1] = rv.y * fp.inv_viewport_h;
    out[2] = rv.x;
    out[3] = rv.y;
}

// src/render/gaussian_2d.wgsl:49-78
BGS_HD void bounding_box_cov2d(const FrameParams& fp, const float extent[2], V2 dir, float cutoff,
                               float out[4]) {
    const float filter_size = 0.707106f;
    if (extent[0] < 1.e-4f || extent[1] < 1.e-4f) {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        return;
    }
    float rx = sqrtf(extent[0]), ry = sqrtf(extent[1]);
    float mr = fmaxf(fmaxf(rx, ry), cutoff * filter_size);
    out[0] = (mr / fp.viewport_w) * dir.x;
    out[1] = (mr / fp.viewport_h) * dir.y;
    out[2] = mr;
    out[3] = mr;
}

struct Surfel {
    float T[9];
    float mean_x, mean_y;
    float extent[2];
};

// src/render/gaussian_2d.wgsl:80-132 with intrinsic_matrix (src/render/helpers.wgsl:122-135)
BGS_HD void cov2d_surfel(const FrameParams& fp, V3 gp, const float* rot, const float* scale,
                         float cutoff, Surfel& o) {
#pragma unroll
    for (int i = 0; i < 9; ++i) o.T[i] = 0.0f;
    o.mean_x = o.mean_y = 0.0f;
    o.extent[0] = o.extent[1] = 0.0f;

    M3 T_r = m3_from_m4(fp.transform);
    M3 S = scale_matrix(scale, fp.global_scale);
    M3 R = rotation_matrix(rot);
    M3 L = m3_mul(m3_mul(T_r, m3_transpose(R)), S);
    const float wfl[3][4] = {{L.m[0], L.m[1], L.m[2], 0.0f},
                             {L.m[3], L.m[4], L.m[5], 0.0f},
                             {gp.x, gp.y, gp.z, 1.0f}};
    const float* cfw = fp.clip_from_world;
    float AB[4][3];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            AB[c][r] = ((wfl[r][0] * cfw[0 * 4 + c] + wfl[r][1] * cfw[1 * 4 + c]) +
                        wfl[r][2] * cfw[2 * 4 + c]) +
                       wfl[r][3] * cfw[3 * 4 + c];
    // clip_from_view[0].x * viewport.z / 2.0 == focal_x / 2.0 (same operation order)
    const float fx = fp.focal_x / 2.0f;
    const float fy = fp.focal_y / 2.0f;
    const float K[3][4] = {{fx, 0.0f, 0.0f, (fp.viewport_w - 1.0f) / 2.0f},
                           {0.0f, fy, 0.0f, (fp.viewport_h - 1.0f) / 2.0f},
                           {0.0f, 0.0f, 0.0f, 1.0f}};
    float T[9];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            T[3 * c + r] = ((AB[0][r] * K[c][0] + AB[1][r] * K[c][1]) + AB[2][r] * K[c][2]) +
                           AB[3][r] * K[c][3];
    V3 test{cutoff * cutoff, cutoff * cutoff, -1.0f};
    V3 T0{T[0], T[1], T[2]}, T1{T[3], T[4], T[5]}, T2{T[6], T[7], T[8]};
    float d = dot3(mul3(test, T2), T2);
    if (fabsf(d) < 1.0e-4f) return;
    V3 f = scale3(1.0f / d, test);
    float mx = dot3(f, mul3(T0, T2));
    float my = dot3(f, mul3(T1, T2));
    float tx = dot3(mul3(f, T0), T0);
    float ty = dot3(mul3(f, T1), T1);
    o.extent[0] = mx * mx - tx;
    o.extent[1] = my * my - ty;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.T[i] = T[i];
    o.mean_x = mx;
    o.mean_y = my;
}

// src/material/spherical_harmonics.wgsl:22-32
// (colour only, no decision downstream: the two divisions by constants are multiplications by the rounded reciprocals,
// within an ulp of the quotient; the power is hardware exp2 / log2 as before)
BGS_HD float srgb_to_linear1(float c) {
    if (c <= 0.04045f) return c * (1.0f / 12.92f);
    return BGS_FAST_POW((c + 0.055f) * (1.0f / 1.055f), 2.4f);
}

// src/render/gaussian.wgsl:166-183
// normalize(v) as ONE reciprocal of the length and three multiplications (the view direction feeds the colour only)
BGS_HD V3 normalize3_rcp(V3 a) {
    const float inv = 1.0f / sqrtf(dot3(a, a));
    return V3{a.x * inv, a.y * inv, a.z * inv};
}
// The normalised basis vectors of the model transform are constants of the frame (FrameParams::basis, formed on the
// host by the reference's operations).
BGS_HD V3 world_to_local_direction(V3 dir, const float* basis) {
    V3 bx{basis[0], basis[1], basis[2]}, by{basis[3], basis[4], basis[5]}, bz{basis[6], basis[7], basis[8]};
    return normalize3_rcp(V3{dot3(bx, dir), dot3(by, dir), dot3(bz, dir)});
}

// SH basis constants, src/material/spherical_harmonics.wgsl:3-20
#define BGS_SHC0 0.28209479177387814f
#define BGS_SHC1 0.4886025119029199f
#define BGS_SHC4 1.0925484305920792f
#define BGS_SHC6 0.31539156525252005f
#define BGS_SHC8 0.5462742152960396f
#define BGS_SHC9 0.5900435899266435f
#define BGS_SHC10 2.890611442640554f
#define BGS_SHC11 0.4570457994644658f
#define BGS_SHC12 0.3731763325901154f
#define BGS_SHC14 1.445305721320277f

// Basis weights w[k] = shc[k] * basis_k(dir) (src/material/spherical_harmonics.wgsl:34-68);
// colour = 0.5 + sum_k w[k] * sh[3k..3k+2]. Weights past the requested degree are 0.
BGS_HD void sh_weights(V3 d, uint32_t degree, float w[16]) {
    V3 s = mul3(d, d);
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = 0.0f;
    w[0] = BGS_SHC0;
    if (degree > 0) {
        w[1] = -BGS_SHC1 * d.y;
        w[2] = BGS_SHC1 * d.z;
        w[3] = -BGS_SHC1 * d.x;
    }
    if (degree > 1) {
        w[4] = BGS_SHC4 * d.x * d.y;
        w[5] = -BGS_SHC4 * d.y * d.z;
        w[6] = BGS_SHC6 * (2.0f * s.z - s.x - s.y);
        w[7] = -BGS_SHC4 * d.x * d.z;
        w[8] = BGS_SHC8 * (s.x - s.y);
    }
    if (degree > 2) {
        w[9] = -BGS_SHC9 * d.y * (3.0f * s.x - s.y);
        w[10] = BGS_SHC10 * d.x * d.y * d.z;
        w[11] = -BGS_SHC11 * d.y * (4.0f * s.z - s.x - s.y);
        w[12] = BGS_SHC12 * d.z * (2.0f * s.z - 3.0f * s.x - 3.0f * s.y);
        w[13] = -BGS_SHC11 * d.x * (4.0f * s.z - s.x - s.y);
        w[14] = BGS_SHC14 * d.z * (s.x - s.y);
        w[15] = -BGS_SHC9 * d.x * (s.x - 3.0f * s.y);
    }
}

// View-dependent direction used for the colour (src/render/gaussian.wgsl:408-412).
BGS_HD V3 sh_direction(const FrameParams& fp, V3 transformed_position) {
    V3 cam{fp.cam[0], fp.cam[1], fp.cam[2]};
    V3 rdw = normalize3_rcp(sub3(transformed_position, cam));
    return world_to_local_direction(rdw, fp.basis);
}

// ---- colour variants other than RASTERIZE_COLOR (src/render/gaussian.wgsl:312-405) -------------
BGS_HD float clamp1(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// WGSL smoothstep(low, high, x)
BGS_HD float smoothstep1(float low, float high, float x) {
    float t = clamp1((x - low) / (high - low), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
// src/material/depth.wgsl:3-11
BGS_HD V3 depth_to_rgb(float depth, float min_depth, float max_depth) {
    float nd = clamp1((depth - min_depth) / (max_depth - min_depth), 0.0f, 1.0f);
    float r = smoothstep1(0.5f, 1.0f, nd);
    float g = 1.0f - fabsf(nd - 0.5f) * 2.0f;
    float b = 1.0f - smoothstep1(0.0f, 0.5f, nd);
    return V3{r, g, b};
}
// bevy_render 0.19 color_operations.wgsl hsv_to_rgb (third-party, restated; hue in radians):
// k = (n + h / (pi/3)) % 6, n = (5,3,1); rgb = v - v*s*max(0, min(k, min(4-k, 1)))
BGS_HD float hsv_channel(float n, float h, float s, float v) {
    float k = fmodf(n + h / 1.047197551f, 6.0f);
    return v - v * s * fmaxf(0.0f, fminf(k, fminf(4.0f - k, 1.0f)));
}
// src/material/classification.wgsl:9-27
BGS_HD V3 class_to_rgb(const FrameParams& fp, float visualization, V3 sh_color) {
    if (visualization < 2.0f) return sh_color;
    float class_idx = visualization - 2.0f;
    float hue = (class_idx / (float)fp.num_classes) * 6.283185307f;
    V3 c{hsv_channel(5.0f, hue, 1.0f, 1.0f), hsv_channel(3.0f, hue, 1.0f, 1.0f), hsv_channel(1.0f, hue, 1.0f, 1.0f)};
    // mix(a, b, 0.5) = a * (1 - 0.5) + b * 0.5
    return V3{sh_color.x * (1.0f - 0.5f) + c.x * 0.5f, sh_color.y * (1.0f - 0.5f) + c.y * 0.5f,
              sh_color.z * (1.0f - 0.5f) + c.z * 0.5f};
}
// RASTERIZE_OPTICAL_FLOW (gaussian.wgsl:369-375; src/material/optical_flow.wgsl:16-53). For a 3D cloud
// previous_transformed_position == transformed_position (gaussian.wgsl:201), so the flow is the
// camera's. hsv_to_rgb is bevy_render's (third party, restated: see hsv_channel).
BGS_HD V3 optical_flow_rgb(const FrameParams& fp, V3 tp) {
    const float* c = fp.clip_from_world;
    const float* q = fp.prev_clip_from_world;
    V4 a = m4_mul_point(c, tp), b = m4_mul_point(q, tp);
    const float mx = (a.x / a.w - b.x / b.w) * 0.5f, my = (a.y / a.w - b.y / b.w) * -0.5f;
    const float fx = mx / fp.delta_time, fy = my / fp.delta_time;
    const float radius = sqrtf(fx * fx + fy * fy);
    float angle = atan2f(fy, fx);
    if (angle < 0.0f) angle += 6.283185307f;
    const float m = clamp1(radius, 0.0f, 1.0f);
    return V3{hsv_channel(5.0f, angle, m, 1.0f), hsv_channel(3.0f, angle, m, 1.0f), hsv_channel(1.0f, angle, m, 1.0f)};
}

// Depth-mode range endpoints: length(transform * vec4(p, 1) - camera) (gaussian.wgsl:329-340)
BGS_HD float distance_to_camera(const FrameParams& fp, V3 pos) {
    V4 t = m4_mul_point(fp.transform, pos);
    V3 d = sub3(V3{t.x, t.y, t.z}, V3{fp.cam[0], fp.cam[1], fp.cam[2]});
    return sqrtf(dot3(d, d));
}
// RASTERIZE_NORMAL (gaussian.wgsl:349-368): third column of L = T*S*R, taken to view space
BGS_HD V3 normal_rgb(const FrameParams& fp, const float* rot, const float* scale) {
    M3 R = rotation_matrix(rot);
    M3 S = scale_matrix(scale, fp.global_scale);
    M3 T = m3_from_m4(fp.transform);
    M3 L = m3_mul(m3_mul(T, S), R);
    const float x = L.m[6], y = L.m[7], z = L.m[8];
    const float* m = fp.view_from_world;
    V4 wn;
    wn.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * 0.0f;
    wn.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * 0.0f;
    wn.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * 0.0f;
    wn.w = ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * 0.0f;
    float len = sqrtf(((wn.x * wn.x + wn.y * wn.y) + wn.z * wn.z) + wn.w * wn.w);
    return V3{0.5f * (wn.x / len + 1.0f), 0.5f * (wn.y / len + 1.0f), 0.5f * (wn.z / len + 1.0f)};
}

// src/render/gaussian.wgsl:229-235
BGS_HD float cutoff_radius(const FrameParams& fp, float opacity) {
    if (!fp.adaptive_radius) return 3.0f;
    // ln(opacity) reaches DECISIONS (cutoff^2 -> d, extent -> the `< 1e-4` tests of gaussian_2d.wgsl:49-78,104-132, an
    // ill-conditioned cancellation): the correctly rounded value, bit-identical on host, device and oracle
    return sqrtf(fmaxf(9.0f + 2.0f * ln_f32_cr(opacity), 0.000001f));
}

// The four clip-space quad corners -> pixel-space parallelogram (centre, u axis, v axis).
// Corner k position = (projected.xy + bb_k.xy) / projected.w, viewport transform y-down
// (src/render/gaussian.wgsl:219-227,429-433). Returns false for a degenerate / NaN quad.
struct QuadPx {
    float cx, cy;      // centre
    float m00, m01, m10, m11;  // uv = M * (pixel - centre)
    float minx, maxx, miny, maxy;
};

BGS_HD bool quad_to_pixels(const FrameParams& fp, V4 projected, const float bb[4][4], QuadPx& q) {
    float X[4], Y[4];
    // the perspective divide of the four corners shares its divisor: one correctly rounded reciprocal, eight products
    // (within 1.5 ulp of the quotients; positions feed no decision finer than the tile rectangle's guard band)
    const float inv_w = 1.0f / projected.w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float nx = (projected.x + bb[k][0]) * inv_w;
        float ny = (projected.y + bb[k][1]) * inv_w;
        X[k] = (nx + 1.0f) * 0.5f * fp.viewport_w;
        Y[k] = (1.0f - ny) * 0.5f * fp.viewport_h;
    }
    // vertex 0 = uv(-1,-1), 1 = (-1,+1), 2 = (+1,-1), 3 = (+1,+1)
    const float esx = X[2] - X[0], esy = Y[2] - Y[0];
    const float etx = X[1] - X[0], ety = Y[1] - Y[0];
    const float det = esx * ety - esy * etx;
    if (!(fabsf(det) > 0.0f) || !(fabsf(det) < INFINITY)) return false;
    const float inv = 2.0f / det;
    q.cx = X[0] + 0.5f * (esx + etx);
    q.cy = Y[0] + 0.5f * (esy + ety);
    q.m00 = ety * inv;
    q.m01 = -etx * inv;
    q.m10 = -esy * inv;
    q.m11 = esx * inv;
    q.minx = fminf(fminf(X[0], X[1]), fminf(X[2], X[3]));
    q.maxx = fmaxf(fmaxf(X[0], X[1]), fmaxf(X[2], X[3]));
    q.miny = fminf(fminf(Y[0], Y[1]), fminf(Y[2], Y[3]));
    q.maxy = fmaxf(fmaxf(Y[0], Y[1]), fmaxf(Y[2], Y[3]));
    if (!(q.minx <= q.maxx) || !(q.miny <= q.maxy)) return false;  // NaN
    return true;
}

// Conservative inclusive tile bounds of a pixel-space bounding box (one-pixel guard band;
// the rasteriser applies the exact coverage test). Returns false if no tile is touched.
BGS_HD bool tile_rect(const FrameParams& fp, const QuadPx& q, int& tx0, int& ty0, int& tx1, int& ty1) {
    const float W = (float)fp.width, H = (float)fp.height;
    float fx0 = floorf(q.minx - 1.5f), fx1 = ceilf(q.maxx + 0.5f);
    float fy0 = floorf(q.miny - 1.5f), fy1 = ceilf(q.maxy + 0.5f);
    fx0 = fmaxf(fx0, 0.0f);
    fy0 = fmaxf(fy0, 0.0f);
    fx1 = fminf(fx1, W - 1.0f);
    fy1 = fminf(fy1, H - 1.0f);
    if (!(fx0 <= fx1) || !(fy0 <= fy1)) return false;
    tx0 = (int)fx0 / TILE_PX;
    tx1 = (int)fx1 / TILE_PX;
    ty0 = (int)fy0 / TILE_PX;
    ty1 = (int)fy1 / TILE_PX;
    return true;
}

// ------------------------------------------------------------------------------------
// Whole vertex stage for one draw-list entry: cull, cutoff, geometry, colour.
// `sh.load_all(c)` fetches the splat's 48 SH coefficients (index 3*k + channel) into c[]. It is
// called BEFORE the geometry so the 12 x 16-byte loads are in flight while the covariance math
// runs (a per-coefficient fetch inside the accumulation loop serialises 16 memory round trips).
// ------------------------------------------------------------------------------------
struct Projected {
    bool visible;   // passed the vertex-stage cull (key != ~0 and in_frustum)
    bool draw;      // visible AND the quad is non-degenerate and touches the target
    float color[4];
    QuadPx quad;
    float p[5];     // Record.p payload
    Surfel surfel;  // 2D only
    float radius;   // 2D: bb.zw
    float ndc_z;    // the quad's depth: position.z / position.w (gaussian.wgsl:429-433), constant over the quad
    int tx0, ty0, tx1, ty1;
};

// ANY_MODE = false compiles the benchmarked RASTERIZE_COLOR path only (the mode tests fold away and
// the kernel keeps its register budget); true honours fp.rasterize_mode.
// cov3d_pre: the splat's precomputed covariance (six floats) or nullptr; with it rot / so[0..2] are unused.
template <bool ANY_MODE, class ShFn>
BGS_HD void project_splat(const FrameParams& fp, uint32_t key, V3 pos, const float rot[4],
                          const float so[4], ShFn sh, const ColorInputs& ci, Projected& o,
                          const float* cov3d_pre = nullptr) {
    const uint32_t mode = ANY_MODE ? fp.rasterize_mode : RASTERIZE_COLOR;
    o.visible = false;
    o.draw = false;
    bool discard_quad = key == 0xFFFFFFFFu;                        // gaussian.wgsl:196
    V4 t4 = m4_mul_point(fp.transform, pos);                       // :198-200
    V3 tp{t4.x, t4.y, t4.z};
    V4 projected = world_to_clip(fp, tp);                          // :210
    discard_quad = discard_quad || !in_frustum(projected);         // :211
    if (ANY_MODE && fp.draw_mode == 1u) discard_quad = discard_quad || ci.visibility < 0.5f;  // DRAW_SELECTED :203-205
    if (discard_quad) return;                                      // :214-218
    o.visible = true;
    const bool sh_color = mode == RASTERIZE_COLOR || mode == RASTERIZE_CLASSIFICATION;
    float shc[48];
    if (sh_color) sh.load_all(shc);

    const float opacity = so[3];
    const float cutoff = cutoff_radius(fp, opacity);               // :229-235
    float bb[4][4];
    const V2 corners[4] = {{-1.0f, -1.0f}, {-1.0f, 1.0f}, {1.0f, -1.0f}, {1.0f, 1.0f}};
    o.p[0] = o.p[1] = o.p[2] = o.p[3] = o.p[4] = 0.0f;
    o.radius = 0.0f;
    float conic[3] = {0.0f, 0.0f, 0.0f};
    if (fp.gaussian_mode == 0u) {                                  // GAUSSIAN_2D :237-255
        cov2d_surfel(fp, tp, rot, so, cutoff, o.surfel);
#pragma unroll
        for (int k = 0; k < 4; ++k) bounding_box_cov2d(fp, o.surfel.extent, corners[k], cutoff, bb[k]);
        o.radius = bb[0][2];
    } else {                                                       // GAUSSIAN_3D :257-306
        float c2d[3];
        cov2d_3dgs(fp, tp, so, rot, cov3d_pre, c2d);
#pragma unroll
        for (int k = 0; k < 4; ++k) bounding_box_clip(fp, c2d, corners[k], cutoff, bb[k]);
        if (fp.aabb) {
            float det = c2d[0] * c2d[2] - c2d[1] * c2d[1];
            float det_inv = 1.0f / det;
            conic[0] = c2d[2] * det_inv;
            conic[1] = -c2d[1] * det_inv;
            conic[2] = c2d[0] * det_inv;
            o.radius = bb[3][2];  // radius_px * (+1)
        }
    }
    if (!quad_to_pixels(fp, projected, bb, o.quad)) return;
    if (!tile_rect(fp, o.quad, o.tx0, o.ty0, o.tx1, o.ty1)) return;
    o.ndc_z = projected.z / projected.w;                           // :429-433 position = (projected.xy + bb.xy, projected.zw)
    if (!fp.aabb) {
        o.p[0] = o.quad.m00; o.p[1] = o.quad.m01; o.p[2] = o.quad.m10; o.p[3] = o.quad.m11;
    } else {
        // axis-aligned square: m01 = m10 = 0 exactly
        o.p[0] = o.quad.m00;
        o.p[1] = o.quad.m11;
        if (fp.gaussian_mode != 0u) {
            const float r2 = o.radius * o.radius;
            o.p[2] = conic[0] * r2;
            o.p[3] = conic[1] * r2;
            o.p[4] = conic[2] * r2;
        }
    }

    float r = 0.0f, g = 0.0f, b = 0.0f;                            // var rgb = vec3(0.0)  :312
    if (sh_color) {
        // RASTERIZE_COLOR :406-417 / RASTERIZE_CLASSIFICATION :315-328, get_color planar.wgsl:334-339
        V3 dir = sh_direction(fp, tp);
        float w[16];
        sh_weights(dir, fp.sh_degree, w);
        r = 0.5f; g = 0.5f; b = 0.5f;
        // coefficients past the requested degree are never touched (may be garbage). The degree is a constant of the
        // frame: a scalar branch per band (same sums in the same order as one loop with a per-coefficient test)
        auto band = [&](const int k0, const int k1) {
#pragma unroll
            for (int k = k0; k < k1; ++k) {
                r += w[k] * shc[3 * k];
                g += w[k] * shc[3 * k + 1];
                b += w[k] * shc[3 * k + 2];
            }
        };
        band(0, 1);
        if (fp.sh_degree > 0) band(1, 4);
        if (fp.sh_degree > 1) band(4, 9);
        if (fp.sh_degree > 2) band(9, 16);
        if (fp.color_space != 1u) {                                // planar.wgsl:91-106
            r = srgb_to_linear1(r);
            g = srgb_to_linear1(g);
            b = srgb_to_linear1(b);
        }
        if (mode == RASTERIZE_CLASSIFICATION) {
            V3 c = class_to_rgb(fp, ci.visibility, V3{r, g, b});
            r = c.x; g = c.y; b = c.z;
        }
    } else if (mode == RASTERIZE_DEPTH) {             // :329-347
        V3 d = sub3(tp, V3{fp.cam[0], fp.cam[1], fp.cam[2]});
        V3 c = depth_to_rgb(sqrtf(dot3(d, d)), ci.min_distance, ci.max_distance);
        r = c.x; g = c.y; b = c.z;
    } else if (mode == RASTERIZE_NORMAL) {            // :348-368
        V3 c = normal_rgb(fp, rot, so);
        r = c.x; g = c.y; b = c.z;
    } else if (mode == RASTERIZE_OPTICAL_FLOW) {                   // :369-375
        V3 c = optical_flow_rgb(fp, tp);
        r = c.x; g = c.y; b = c.z;
    } else if (mode == RASTERIZE_POSITION) {          // :376-377
        r = (tp.x - fp.pos_min[0]) / (fp.pos_max[0] - fp.pos_min[0]);
        g = (tp.y - fp.pos_min[1]) / (fp.pos_max[1] - fp.pos_min[1]);
        b = (tp.z - fp.pos_min[2]) / (fp.pos_max[2] - fp.pos_min[2]);
    }
    o.color[0] = r; o.color[1] = g; o.color[2] = b;
    o.color[3] = opacity * fp.global_opacity;
    if (ANY_MODE && fp.draw_mode == 2u && ci.visibility > 0.5f) {  // HIGHLIGHT_SELECTED :423-427
        o.color[0] = 0.3f; o.color[1] = 1.0f; o.color[2] = 0.1f; o.color[3] = 1.0f;
    }
    o.draw = true;
}

}  // namespace bgs
