/*
 * bgs_oracle.h — CPU oracle for the sort + rasterize hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under bevy_gaussian_splatting_amd/ (the product) may
 * include, link, import or execute this. Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py use it, and only as the checker / reported baseline.
 *
 * What it is: a plain-C, single-precision restatement of the reference's algorithm,
 * function by function, each citing the reference file:line it follows
 * (mosure/bevy_gaussian_splatting v8.0.1). The reference is Rust + WGSL and cannot be
 * compiled or executed in the build container (no cargo/rustc, no wgpu, no GPU), so
 * there is no oracle/_ref build.
 *
 * PINNING STATUS
 *   - sort keys / order / (places, shift, parity) table: PINNED by the reference's own
 *     known-answer tests tests/radix.rs:9-106 (fixtures in tests/golden/radix_keys.json).
 *   - per-splat projection, SH colour, per-pixel falloff and blending: no output of a RUNNING reference
 *     exists to compare with (none can run here), so these are pinned by INDEPENDENT float64 derivations
 *     instead (tests/test_oracle_golden.py; nothing there re-types the WGSL): cov2d and the four quad
 *     corners from the camera model with a numerical Jacobian + numpy.linalg.eigh (OBB and AABB, incl. the
 *     pair of tools/compare_aabb_obb.rs); every covered pixel of an anisotropic splat against the projected
 *     Gaussian; SH degree 0-3 against scipy's spherical harmonics at 1000 directions; the 2DGS
 *     local_to_pixel / mean_2d / extent against a ray-plane linear solve and the sampled cutoff ellipse, and
 *     the surfel image against the same solve; a three-splat stack against the closed-form "over" sum; a WHOLE
 *     random scene (120 overlapping splats, SH3, sRGB, adaptive radius, a third culled; OBB and AABB) against an
 *     independent numpy float64 renderer, every pixel off the quad edges to 1e-6; plus
 *     the coarse thresholds of tests/visibility_render.rs:245-274 applied to the oracle's image. Where the
 *     reference itself departs from the geometric truth (OBB falloff tied to the quad, the 2DGS frame's doubled
 *     focal factor and half-size quad, the on-axis OBB NaN) the oracle follows the reference and the tests
 *     say so.
 *   - ln(opacity) of the adaptive cutoff is the CORRECTLY ROUNDED binary32 logarithm (x87 logl rounded once;
 *     checked against the product's binary64 implementation on all 2 139 095 039 positive inputs): it reaches the
 *     2DGS degeneracy decisions, and WGSL leaves the precision of log open.
 *   - STILL PARITY UNPINNED: third-party arithmetic outside the reference tree — Bevy's projection matrix
 *     helper, wgpu's rasterisation rules / blend unit, glam, bevy_render 0.19.0's hsv_to_rgb (Classification
 *     and OpticalFlow colour variants) — restated and labelled where used.
 *
 * Arithmetic contract (so "bit-exact sort order" is well defined; WGSL leaves the
 * evaluation order of dot()/matrix products implementation-defined):
 *   - every operation is IEEE-754 binary32, round-to-nearest-even, NO fused multiply-add
 *     (compile with -ffp-contract=off);
 *   - dot(a,b)      = ((a0*b0 + a1*b1) + a2*b2) [+ a3*b3]
 *   - (M * v)[r]    = (((M[0][r]*v0 + M[1][r]*v1) + M[2][r]*v2) + M[3][r]*v3)
 *   - (A * B)[c][r] = ((A[0][r]*B[c][0] + A[1][r]*B[c][1]) + A[2][r]*B[c][2])
 *   - the rasteriser between the vertex and fragment stages (pixel coverage and linear
 *     attribute interpolation, fixed-function in the reference) is evaluated in float64
 *     ("ideal rasteriser"), its outputs rounded to binary32 before fs_main.
 *   - the colour target is unclamped f32 (an ideal Rgba16Float/hdr target,
 *     src/render/mod.rs:917-921), blended in draw order with
 *     dst = src + dst * (1 - src.a)  (PREMULTIPLIED_ALPHA_BLENDING, src/render/mod.rs:946).
 */
#ifndef BGS_ORACLE_H
#define BGS_ORACLE_H

#include <stdint.h>

#include "../include/bgs.h" /* boundary structs only: bgs_view, bgs_settings, bgs_sort_entry */

#ifdef __cplusplus
extern "C" {
#endif

/* Planar f32 cloud (src/gaussian/formats/planar_3d.rs:45-54). */
typedef struct oracle_cloud {
    uint32_t n;
    const float* position_visibility; /* n*4  */
    const float* spherical_harmonic;  /* n*48 */
    const float* rotation;            /* n*4 [w,x,y,z] */
    const float* scale_opacity;       /* n*4  */
} oracle_cloud;

/* ShaderDefines::for_radix_depth_bits (src/render/mod.rs:715-758): bits -> places, key
 * shift, initial ping-pong parity. Returns 0, or -1 for an unsupported bit count. */
int oracle_radix_defines(uint32_t depth_bits, uint32_t* digit_places, uint32_t* key_shift,
                         uint32_t* initial_parity);

/* tests/radix.rs:96-106 helper pair: dist2 (scalar L->R) and the key derived from it. */
float oracle_distance_squared(const float position[3], const float camera[3]);
/* ln(x) correctly rounded to binary32 (x87 logl rounded once): the log of the adaptive cutoff, gaussian.wgsl:229-235 */
float oracle_ln_f32(float x);
void oracle_ln_f32_array(const float* x, uint32_t n, float* out);
/* the wrap-around sum bgs_selftest_ln_f32 (include/bgs.h) forms on the device, from the oracle's log */
uint64_t oracle_ln_f32_checksum(uint32_t first_bits, uint32_t count);
uint32_t oracle_radix_depth_key(float dist2, uint32_t key_shift);

/* radix_sort_a keygen (src/sort/radix.wgsl:86-101) for SORT_RADIX;
 * rayon keygen (src/sort/rayon.rs:86-98) for SORT_RAYON/STD;
 * SortedEntries::new (src/sort/mod.rs:347-354) for SORT_NONE. */
int oracle_keygen(const float* position_visibility, uint32_t n, const bgs_view* view,
                  const bgs_settings* settings, bgs_sort_entry* out);

/* radix_sort_b + radix_sort_c_* (src/sort/radix.wgsl:109-279): stable LSD, 8-bit
 * digits, `places` passes over 1024-entry tiles. entries is sorted in place; tmp has n. */
void oracle_radix_sort(bgs_sort_entry* entries, uint32_t n, uint32_t places,
                       bgs_sort_entry* tmp);

/* par_sort_unstable_by descending f32 (src/sort/rayon.rs:100-104). Ties (unspecified in
 * the reference) are broken by ascending index so the result is deterministic. */
void oracle_sort_descending_f32(bgs_sort_entry* entries, uint32_t n);

/* keygen + the sort selected by settings->sort_mode. out has n entries. */
int oracle_sort(const float* position_visibility, uint32_t n, const bgs_view* view,
                const bgs_settings* settings, bgs_sort_entry* out);

/* Everything vs_points (src/render/gaussian.wgsl:184-436) outputs for one instance,
 * for all four quad vertices. */
typedef struct oracle_vs_out {
    int32_t discard;            /* position = 0,0,0,0 (degenerate quad)                 */
    float projected[4];         /* world_to_clip(transformed_position)                  */
    float bb[4][4];             /* per vertex: xy = NDC offset, zw = second pair        */
    float color[4];             /* rgb, opacity * global_opacity                        */
    float cov2d[3];             /* 3D: (a,b,c)                                          */
    float conic[3];             /* AABB 3D                                              */
    float cutoff;
    float local_to_pixel[9];    /* 2D: columns u,v,w                                    */
    float mean_2d[2];           /* 2D                                                   */
    float extent[2];            /* 2D                                                   */
    float radius[2];            /* 2D: bb.zw                                            */
} oracle_vs_out;

int oracle_vs(const oracle_cloud* cloud, bgs_sort_entry entry, const bgs_view* view,
              const bgs_settings* settings, oracle_vs_out* out);

/* Same for instance `instance` of a sorted list; required for RasterizeMode::Depth, whose colour
 * depends on get_entry(1) and get_entry(count - 1) (src/render/gaussian.wgsl:331-340). */
int oracle_vs_sorted(const oracle_cloud* cloud, const bgs_sort_entry* entries, uint32_t count,
                     uint32_t instance, const bgs_view* view, const bgs_settings* settings,
                     oracle_vs_out* out);
/* out = {min_distance, max_distance} of the Depth mode. */
int oracle_depth_range(const oracle_cloud* cloud, const bgs_sort_entry* entries, uint32_t count,
                       const bgs_view* view, const bgs_settings* settings, float out[2]);

/* Draw entries[0..count) in order into the pixel window [x0,x1) x [y0,y1) of the
 * viewport-sized target. rgba_out is (y1-y0)*(x1-x0)*4 floats, row 0 = y0 (top).
 * ambiguity_out (may be NULL, same pixel count, 1 float each) accumulates, per pixel, an
 * upper bound on how much the result could change if every coverage / discard decision
 * that lies within rounding distance of its threshold flipped. Uses OpenMP over rows. */
int oracle_render(const oracle_cloud* cloud, const bgs_sort_entry* entries, uint32_t count,
                  const bgs_view* view, const bgs_settings* settings, int32_t x0, int32_t y0,
                  int32_t x1, int32_t y1, float* rgba_out, float* ambiguity_out);
/* The same draw into a multisampled target with view->sample_count samples per pixel (1 or 4:
 * MultisampleState.count = Msaa::samples(), src/render/mod.rs:357-424,975-979; coverage and depth test per sample,
 * shading once per pixel at its centre, box resolve) and tested against a scene depth buffer (HOST memory here:
 * viewport.w * viewport.h * sample_count floats [y][x][sample], Depth32Float / GreaterEqual / no write,
 * src/render/mod.rs:959-974; NULL = none; bgs_view.depth_device_ptr is ignored by the oracle).
 * oracle_render is this with depth = NULL. Sample counts: 1, 2, 4, 8 (Msaa::Off / Sample2 / Sample4 / Sample8 at the
 * graphics APIs' standard positions; 0 = not set = 4); -5 for anything else. */
int oracle_render_depth(const oracle_cloud* cloud, const bgs_sort_entry* entries, uint32_t count,
                        const bgs_view* view, const bgs_settings* settings, int32_t x0, int32_t y0,
                        int32_t x1, int32_t y1, const float* depth, float* rgba_out, float* ambiguity_out);
/* The same draw into a colour attachment of the reference's STORAGE format (src/render/mod.rs:917-921,944-948):
 * target_format 0 = binary32 samples (= oracle_render_depth: the ideal target), 1 = Rgba8UnormSrgb, 2 = Rgba16Float.
 * With a packed format every covered sample is read, blended and stored ROUNDED at every blend (source colour clamped
 * to [0, 1] first for the fixed-point format), and the resolve target holds the mean of the stored samples in the same
 * format. rgba_out receives the resolved texels' VALUES as float (decoded to linear for sRGB8). The per-blend rounding
 * is fixed-function behaviour of wgpu / the graphics API: restated from the Vulkan / D3D format-conversion rules,
 * PARITY UNPINNED. -6 for an unknown format. */
int oracle_render_target(const oracle_cloud* cloud, const bgs_sort_entry* entries, uint32_t count,
                         const bgs_view* view, const bgs_settings* settings, int32_t x0, int32_t y0,
                         int32_t x1, int32_t y1, const float* depth, int target_format, float* rgba_out,
                         float* ambiguity_out);
/* The Rgba8UnormSrgb codes (R, G, B, A bytes) of n RGBA values: exact rounding of 255 * OETF by threshold search — for
 * values that ARE stored texels (oracle_render_target's output) this is the read-back of the attachment. */
void oracle_srgb8_codes(const float* rgba, uint32_t n, uint8_t* out);
/* The sample positions of a pixel (x, y pairs, origin = the pixel's top-left corner, y down). */
int oracle_sample_positions(uint32_t sample_count, float* xy_out);

/* f16 planes -> f32 planes (src/render/planar.wgsl:117-176). */
void oracle_decode_f16(uint32_t n, const uint32_t* sh_h2 /* n*24 */,
                       const uint32_t* rot_scale_opacity /* n*4 */, float* sh_out /* n*48 */,
                       float* rotation_out /* n*4 */, float* scale_opacity_out /* n*4 */);
/* f32 planes -> f16 planes (src/gaussian/f16.rs:38-55,244-252; SH pairing
 * src/render/planar.wgsl:283-292). Round-to-nearest-even like the `half` crate. */
void oracle_encode_f16(uint32_t n, const float* sh /* n*48 */, const float* rotation,
                       const float* scale_opacity, uint32_t* sh_h2_out,
                       uint32_t* rot_scale_opacity_out);

/* Per-frame tile-instance statistics of the reference-equivalent quad set: number of
 * non-discarded splats and the sum over splats of 16x16 tiles overlapped by the quad's
 * pixel bounding box (used to size benches and to cross-check the binning stage). */
int oracle_instance_stats(const oracle_cloud* cloud, const bgs_sort_entry* entries,
                          uint32_t count, const bgs_view* view, const bgs_settings* settings,
                          uint32_t* visible_out, uint64_t* tile_instances_out);

/* Store of the f32 target into the reference's colour attachment format Rgba8UnormSrgb
 * (src/render/mod.rs:917-921, examples/headless.rs:120-123): per channel clamp to [0,1], RGB
 * through the sRGB transfer function (third-party wgpu / Vulkan format conversion: parity
 * unpinned), round to nearest unorm8; alpha linear. out = n*4 bytes R,G,B,A. */
void oracle_encode_srgb8(const float* rgba, uint32_t n, uint8_t* out);

/* The quad-edge ambiguity band of oracle_render's ambiguity bound, in pixels (default 5e-4 since round 6, 2e-3 before; accepted range [0, 0.5]). */
void oracle_set_edge_band_px(double px);
double oracle_edge_band_px(void);

int oracle_max_threads(void);
/* OpenMP threads used by keygen / render from now on (bench.py: the pinned 1-core baseline). */
void oracle_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
