// bgs_device.h — structs shared by the HIP kernels and the host orchestration (libbgs).
// Plain C++ (no HIP types) so the host-side pre-flight test can compile it with g++.
#pragma once
#include <stdint.h>

namespace bgs {

constexpr int TILE_PX = 16;              // 16x16-pixel raster tiles
constexpr uint32_t RADIX_BITS = 8;       // src/render/mod.rs:716
constexpr uint32_t RADIX_BASE = 256;     // src/render/mod.rs:720
constexpr uint32_t KEY_CULLED = 0xFFFFFFFFu;

// sort_mode values (include/bgs.h)
constexpr uint32_t SORT_NONE = 0, SORT_RADIX = 1, SORT_RAYON = 2, SORT_STD = 3;

// Everything the per-splat stages read that is constant over a frame: the used parts of
// Bevy's View uniform (src/render/bindings.wgsl:3-9) and CloudUniform
// (src/render/mod.rs:995-1009). The frame's first kernel gets it by value and leaves a
// copy in device memory for the others (kernels.h, KeygenLaunch).
struct FrameParams {
    float transform[16];        // gaussian_uniforms.transform, column-major
    float view_from_world[16];
    float clip_from_world[16];  // = unjittered_clip_from_world
    float cam[3];               // view.world_position
    float focal_x, focal_y;     // clip_from_view[0].x * W, clip_from_view[1].y * H (helpers.wgsl:20-23)
    float viewport_w, viewport_h;
    float global_opacity, global_scale;
    uint32_t n;                 // gaussian_uniforms.count
    uint32_t key_shift;         // RADIX_KEY_SHIFT
    uint32_t gaussian_mode;     // 0 = 2D, 1 = 3D
    uint32_t aabb;
    uint32_t adaptive_radius;
    uint32_t color_space;       // 0 = sRGB-encoded SH, 1 = linear
    uint32_t sh_degree;
    uint32_t sort_mode;
    int32_t width, height;      // render target size in pixels
    int32_t tiles_x, tiles_y;
    uint32_t debug;             // experiment switches (bgs_set_debug_flags); 0 in production
    uint32_t rasterize_mode;    // RasterizeMode discriminant (include/bgs.h)
    uint32_t num_classes;
    float pos_min[3], pos_max[3];  // gaussian_uniforms.min / .max (Position mode)
    uint32_t draw_mode;            // 0 All, 1 Selected, 2 HighlightSelected
    float prev_clip_from_world[16];  // previous_view_uniforms.clip_from_world (OpticalFlow)
    float delta_time;                // globals.delta_time
    float clear[4];                  // the camera's clear colour (premultiplied RGBA)
    uint64_t srgb8_target;           // device address the frame's Rgba8UnormSrgb image goes to; 0 = the lane's own
    // Depth-sort path of the frame (sort_kernels.hip): 0 = onesweep digit passes, 1 = bucket sort (keygen
    // places the drawable pairs into key-range buckets, one kernel sorts each bucket in LDS)
    uint32_t sort_path;
    // MultisampleState.count of the pipeline = Msaa::samples() of the camera (src/render/mod.rs:357-424,975-979): 1, 2, 4 or 8
    uint32_t sample_count;
    // the view's depth attachment (Depth32Float, reverse-Z, [y][x][sample] floats) the quads are tested against with
    // GreaterEqual (src/render/mod.rs:959-974), as a device address; 0 = none
    uint64_t depth_ptr;
    // Per-frame constants the vertex stage used to re-derive per splat (each a few correctly rounded divisions / square
    // roots of uniform data: ~130 instructions per thread). Formed by fill_frame_params on the host with the same IEEE
    // operations in the same order, so every bit downstream is what it was:
    float basis[9];            // normalize(transform[0].xyz), [1].xyz, [2].xyz  (gaussian.wgsl:166-176)
    float inv_viewport_w, inv_viewport_h;   // 1.0 / viewport.zw                (helpers.wgsl:116-117)
    uint32_t visualize_bbox;   // CloudSettings::visualize_bounding_box (src/render/gaussian.wgsl:486-495): the quads' frames
};
static_assert(sizeof(FrameParams) % 8 == 0 && sizeof(FrameParams) / 4 <= 256, "keygen copies it with one block");

// rasterize_mode values (include/bgs.h)
constexpr uint32_t RASTERIZE_CLASSIFICATION = 0, RASTERIZE_COLOR = 1, RASTERIZE_DEPTH = 2,
                   RASTERIZE_NORMAL = 3, RASTERIZE_OPTICAL_FLOW = 4, RASTERIZE_POSITION = 5;

// Per-splat inputs of the non-Color colour variants (src/render/gaussian.wgsl:312-405).
struct ColorInputs {
    float visibility;    // position_visibility.w (Classification: class id + 2)
    float min_distance;  // Depth: |M * p(sorted[count-1]) - cam|
    float max_distance;  // Depth: |M * p(sorted[1]) - cam|
};

// Projected record, one per draw-list rank, stored in front-to-back order.
// 12 dwords = 48 B (R = 48). Meaning of p[] by pipeline variant:
//   OBB (3D or 2D): p = {m00, m01, m10, m11, -}:  uv = M * (pixel_centre - centre)
//   AABB 3D:        p = {m00, m11, A, B, C}: uv = (m00*dx, m11*dy) and
//                   power = -0.5*(A*u*u + C*v*v) + B*u*v with (A,B,C) = conic * radius_px^2
//                   (fs_main uses d = -major_minor = -radius_px * uv, gaussian.wgsl:456-458)
// z = the quad's depth: position.z / position.w of src/render/gaussian.wgsl:429-433, constant over the quad, in (0, 1)
// for everything that passes in_frustum (reverse-Z: 1 = near plane); what the depth test compares
// (src/render/mod.rs:959-974). The packed tile rectangle travels in the coarse list entries / the emit kernel's LDS,
// not in the record.
struct Record {
    float cx, cy;       // quad centre in pixels (viewport origin at 0,0, y down)
    float p[5];
    float r, g, b, a;   // colour (linear, unclamped) and opacity * global_opacity
    float z;
};
static_assert(sizeof(Record) == 48, "Record must be 48 bytes");

// 2DGS surfel record for the AABB path (src/render/gaussian_2d.wgsl:134-156): 24 dwords = 96 B.
struct RecordSurfel {
    float cx, cy;       // quad centre in pixels
    float m00, m11;     // uv = (m00*dx, m11*dy) (axis-aligned square quad)
    float radius;       // input.radius (both components equal), half-pixel units
    float mean_x, mean_y;
    float T[9];         // A = T1 x T2, B = T2 x T0, C = T0 x T1 of local_to_pixel's columns (render_kernels.hip: stage_surfel)
    float r, g, b, a;
    float z;            // the quad's depth (see Record)
    uint32_t pad[3];
};
static_assert(sizeof(RecordSurfel) == 96, "RecordSurfel must be 96 bytes");

// Bucket sort geometry (sort_kernels.hip). keygen places every drawable pair into one of BUCKET_COUNT
// key-range buckets (fixed slot regions of BUCKET_CAP pairs); bucket_sort_kernel sorts each bucket in the
// LDS of one workgroup. bucket(key) = number of splitters <= key, the splitters being the 255 keys at
// the 1/256-quantiles of a recently completed frame's sorted list: balanced by construction while the
// view changes slowly, and monotone in the key whatever the table holds (order never depends on it).
constexpr uint32_t BUCKET_COUNT = 256;       // key ranges the splitters define
constexpr uint32_t BUCKET_CAP = 4096;        // pairs per bucket: 32 KB of LDS
constexpr uint32_t BUCKET_FINE = 2048;       // fine key ranges inside a bucket (counting sort + rank among equals)
constexpr uint32_t BUCKET_FINE_MAX = 1024;   // more pairs than this in one fine range: give up (ties), onesweep re-run
// Draw lists longer than 256 buckets hold (round 5: a camera that sees the whole cloud, SortMode::Rayon / Std — D = N):
// 256 * sub buckets, sub <= BUCKET_SUB_MAX, with the 256 * sub - 1 keys at the 1 / (256 sub)-quantiles of a completed
// frame's list as splitters — exact quantiles, so the buckets stay balanced whatever the key distribution is (equal KEY
// intervals inside a 1/256-quantile range were tried first: the first and the last range of a distance-keyed list
// span many binades and nearly all of their pairs fall into one interval). Up to sub = BUCKET_SUB_KERNARG the table
// travels in keygen's kernel arguments (4 KB in all); a longer one (up to 4095 keys: 12.6 M drawable pairs) is copied to
// the lane's device table ahead of keygen (one 16 KB host-to-device copy on the frame's stream: frames of that size last
// hundreds of microseconds).
constexpr uint32_t BUCKET_SUB_MAX = 16;
constexpr uint32_t BUCKET_SUB_KERNARG = 3;
constexpr uint32_t BUCKET_MAX = BUCKET_COUNT * BUCKET_SUB_MAX;
constexpr uint32_t BUCKET_TARGET = 2048;     // pairs per bucket the host aims at when it picks `sub`
// Lists the narrow buckets would need more than BUCKET_SUB_KERNARG x 256 of (round 6): WIDE buckets, a quarter as many
constexpr uint32_t BUCKET_CAP_WIDE = 16384;  // 128 KB of a CU's 160 KB of LDS
#ifndef BGS_BUCKET_FINE_WIDE
#define BGS_BUCKET_FINE_WIDE 8192            // (half a word each: sort_kernels.hip; 4096: a word each, A/B)
#endif
constexpr uint32_t BUCKET_FINE_WIDE = BGS_BUCKET_FINE_WIDE;
constexpr uint32_t BUCKET_TARGET_WIDE = 10240;
struct SplitterTable {
    uint32_t key[BUCKET_COUNT * BUCKET_SUB_KERNARG];   // sub <= BUCKET_SUB_KERNARG: key[0 .. 256 * sub - 2] ascending quantile keys
    const uint32_t* device_keys;                       // sub > BUCKET_SUB_KERNARG: the same, in device memory
    uint32_t sub;                                      // 1 .. BUCKET_SUB_MAX: the table defines 256 * sub buckets
    // WIDE buckets (round 6; lists past BUCKET_COUNT * BUCKET_SUB_KERNARG * BUCKET_TARGET pairs): slot regions of BUCKET_CAP_WIDE
    // pairs, sorted by bucket_sort_kernel's 1024-thread instantiation in 128 KB of LDS
    uint32_t wide;
};
// what the host keeps per view slot (a completed frame's quantile keys, any sub)
struct SplitterKeys { uint32_t key[BUCKET_MAX]; uint32_t sub; };

// Device-resident control block, zeroed at the start of every frame by one memset.
struct Control {
    uint32_t draw_count;      // entries of the draw list that reach the vertex stage (V' or N)
    uint32_t instance_count;  // (tile, rank) instances emitted (clamped to capacity)
    uint32_t instance_total_lo, instance_total_hi;  // unclamped 64-bit total
    uint32_t overflow;        // 1 if instance_total > capacity
    uint32_t error;           // device watchdog (bounded spins)
    uint32_t visible_count;   // splats that pass the vertex-stage cull (stats)
    uint32_t splat_count;     // N, parked on the device so the sort kernels read every size the same way
    uint32_t sort_overflow;   // bucket sort gave up (1 bucket over capacity, 2 too many equal keys): re-run with onesweep
    uint32_t bucket_max;      // fullest bucket (stats; one-level placement only)
    uint32_t strip_tiles;     // tiles this frame's rasteriser drew with four strip waves (the consumed heavy-tile list; stats)
    // (host copy only) from the cost plane the lane's tile order was last made of (tile_order_kernel, TileCost bit 15): the share
    // of the frame's tile work that was in tiles which ended saturated, x 0x7FFF; 0xFFFFFFFF = no order
    uint32_t saturated_tiles_prev;
    uint32_t pad0[20];        // the read-mostly header owns its 128-byte line (see ticket)
    // dynamic tile ids, one word per kernel launch of the frame, each in its OWN 128-byte line: every
    // block of a launch does a returning atomic on its ticket and the L2 retires same-line atomics one
    // at a time (~8 ns), so a load of draw_count queued behind them on a shared line waited for all.
    uint32_t ticket[16][32];
    // bits of max |r|, |g|, |b| over the frame's drawn records (non-negative floats order like their bits): the
    // rasterisers scale their transmittance cut-off by it (render_kernels.hip, frame_t_eps). Its own 128-byte
    // line for the same reason as the tickets: every block of the project kernel polls / bumps it.
    uint32_t color_max_bits;
    uint32_t pad1[31];
    uint32_t hist_depth[4][RADIX_BASE];  // global digit histograms of the depth keys
    uint32_t hist_tile[2][RADIX_BASE];   // digit 0 = tile x, digit 1 = tile y
    uint32_t coarse_total[RADIX_BASE];   // scan binning: entries in each supertile's ordered list
    uint32_t splitters[BUCKET_MAX];      // the 256 * sub - 1 quantile keys of THIS frame's sorted list (keygen key space; sub =
                                         // FrameCleanup::split_sub): the host hands them to later frames' keygen (SplitterTable)
    uint32_t bucket_count[BUCKET_MAX];   // bucket sort: pairs in each bucket (keygen's returning atomics); the frame's 256 * sub first
};
constexpr uint32_t CONTROL_HEADER_WORDS = 12;  // draw_count .. saturated_tiles_prev: what a frame reports to the host


// packed tile rectangle x0 | x1 << 8 | y0 << 16 | y1 << 24 (inclusive); x0 > x1 = touches no tile
constexpr uint32_t RECT_EMPTY = 0x000000FFu;
constexpr uint32_t MAX_SUPERTILES = 256;  // coarse bins ride the 256-wide chained scan
constexpr uint32_t MAX_SUPERTILES_PER_AXIS = 32;  // column / row masks of project_bin
// Supertiles are squares of E x E tiles, E any integer (6 at 1080p: 20 x 12 = 240 bins): tile / E as
// (tile * M) >> 16 with M = 65536 / E + 1, exact for tile < 256 and E <= 32.
inline uint32_t supertile_mul(uint32_t edge) { return 65536u / edge + 1u; }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t supertile_div(uint32_t tile, uint32_t mul) { return (tile * mul) >> 16; }

// binning modes (bgs_set_binning)
constexpr uint32_t BINNING_SCAN = 0;  // ordered coarse lists + lazy per-tile scan (default)
constexpr uint32_t BINNING_SORT = 1;  // (tile, rank) instances + stable radix sort on the tile id

// look-back status word: flag in the top 2 bits, 30-bit value
constexpr uint32_t STATUS_FLAG_SHIFT = 30;
constexpr uint32_t STATUS_VALUE_MASK = (1u << STATUS_FLAG_SHIFT) - 1u;
constexpr uint32_t STATUS_AGGREGATE = 1u << STATUS_FLAG_SHIFT;
constexpr uint32_t STATUS_PREFIX = 2u << STATUS_FLAG_SHIFT;

}  // namespace bgs
