/*
 * bgs.h — C ABI of libbgs: the MI355X-native Gaussian-splat sort + rasterize path.
 *
 * This is the drop-in boundary for ONE hot path of mosure/bevy_gaussian_splatting
 * (citations are into the reference tree):
 *
 *   cloud upload   <- RenderAsset upload of the planar cloud
 *                     (src/lib.rs:65-68, src/render/mod.rs:279-313,
 *                      layout src/gaussian/formats/planar_3d.rs:28-54)
 *   bgs_sort       <- run_radix_sort   (src/sort/radix.rs:616-756, kernels src/sort/radix.wgsl)
 *                     rayon_sort       (src/sort/rayon.rs:27-130)   [SortMode::Rayon ordering]
 *   bgs_render     <- DrawGaussianInstanced::render (src/render/mod.rs:1513-1569),
 *                     vs_points / fs_main (src/render/gaussian.wgsl:184-505),
 *                     blend state (src/render/mod.rs:944-948)
 *
 * The reference has no FFI of its own (it is a Bevy plugin); these entry points are
 * what a `hip-sys`-style Rust binding would declare.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary, the library never aborts;
 *   - every function returns BGS_OK (0) or a negative bgs_status; the message for the
 *     last failure on a context is available from bgs_last_error();
 *   - all matrices are column-major float[16] (glam / WGSL convention): m[4*c + r];
 *   - host pointers are borrowed for the duration of the call only;
 *   - device memory is owned by the library behind opaque handles;
 *   - a bgs_ctx binds one HIP device (and a few HIP streams of its own) and is NOT re-entrant
 *     (matches Bevy: one render thread); distinct contexts are independent.
 *   - there is NO CPU fallback: without a usable HIP device bgs_create fails with
 *     BGS_EHIP.
 *
 * This header is the seam and nothing else: context, cloud upload, sort, render, the frame's targets and
 * the frame pipeline. Diagnostics, test hooks and experiment switches (per-tile trace, counters of the
 * adaptive machinery, the HBM probe, the radix kernel on caller keys, the ln self-test, debug flags, the
 * queue-holder switch) are declared in bgs_diag.h; libbgs.so exports both sets.
 */
#ifndef BGS_H
#define BGS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGS_VERSION_MAJOR 0
#define BGS_VERSION_MINOR 4 /* 0.4: bgs_abi_check, bgs_comm_* (the multi-GPU frame gather), sample_count 0 = default; struct layouts as 0.3
                               (0.3: bgs_view gained sample_count + depth_device_ptr, 16 bytes longer; bgs_stats 16 bytes longer than 0.2's) */

typedef enum bgs_status {
    BGS_OK = 0,
    BGS_EINVAL = -1,    /* bad argument / unsupported setting               */
    BGS_ENOMEM = -2,    /* host or device allocation failed                 */
    BGS_EHIP = -3,      /* HIP runtime error (no device, launch failure...) */
    BGS_ECAPACITY = -4, /* tile-instance buffer could not be grown enough   */
    BGS_EINTERNAL = -5  /* device-side watchdog tripped (bounded spin)      */
} bgs_status;

typedef struct bgs_ctx bgs_ctx;
typedef struct bgs_cloud bgs_cloud;

/* The fields of Bevy's `View` uniform that the reference shaders read
 * (src/render/helpers.wgsl:18-38, src/render/transform.wgsl:5-8,
 *  src/render/gaussian_2d.wgsl:104, src/sort/radix.wgsl:90) plus the clear colour
 * (examples/headless.rs:70). The caller supplies every matrix explicitly, exactly as
 * Bevy does on the CPU, so the library never re-derives one from another. */
typedef struct bgs_view {
    float world_from_view[16]; /* camera transform; world_position = column 3        */
    float view_from_world[16];
    float clip_from_view[16];
    float clip_from_world[16]; /* also used as unjittered_clip_from_world (no TAA)   */
    float viewport[4];         /* x, y, width, height in pixels (View.viewport)      */
    float clear_color[4];      /* linear RGBA the target is cleared to before draws  */
    /* RasterizeMode::OpticalFlow only (src/material/optical_flow.wgsl:16-53): the PREVIOUS frame's
     * clip_from_world (Bevy's previous_view_uniforms) and globals.delta_time in seconds */
    float previous_clip_from_world[16];
    float delta_time;
    /* The camera's `Msaa` component as the pipeline is specialised on it: CloudPipelineKey.sample_count =
     * msaa.samples() (src/render/mod.rs:357,412,422) -> MultisampleState.count (:975-979). 1 = Msaa::Off, 2 / 4 / 8 =
     * Msaa::Sample2 / Sample4 / Sample8; 4 is Bevy's default, i.e. what every camera of the reference's examples and
     * tests runs with (nothing in the reference sets Msaa; src/utils.rs:34 `msaa_samples` is never read). With N > 1
     * samples coverage (and the depth test) is decided per sample at the graphics APIs' standard sample positions, the
     * fragment stage runs once per pixel at the pixel centre, every covered sample blends the same source colour, and
     * the image handed back is the resolved one (mean of the samples). 0 (a zero-initialised bgs_view) means "not set"
     * and renders with Msaa::default() = 4; anything else is BGS_EINVAL. bgs_view_perspective sets 4. */
    uint32_t sample_count;
    uint32_t reserved[2];
    /* The view's depth attachment the draw is tested against (src/render/mod.rs:959-974: Depth32Float,
     * CompareFunction::GreaterEqual — reverse-Z —, depth_write_enabled false): a DEVICE pointer to
     * viewport.w * viewport.h * sample_count floats laid out [y][x][sample] (sample order: the standard pattern's) (what Bevy's opaque passes left), or
     * 0 = no scene depth (every fragment passes, as against a buffer cleared to 0.0). A quad's depth is its
     * splat's constant NDC z (src/render/gaussian.wgsl:429-433). The memory must stay valid and unchanged until
     * the frame has completed (bgs_device_alloc / bgs_upload below serve hosts without a HIP runtime). */
    uint64_t depth_device_ptr;
    uint64_t reserved_ptr;
} bgs_view;

/* gaussian_mode: src/gaussian/settings.rs:17-22 */
#define BGS_GAUSSIAN_2D 0u
#define BGS_GAUSSIAN_3D 1u
/* color_space: src/render/mod.rs:1066-1069 */
#define BGS_COLOR_SRGB 0u
#define BGS_COLOR_LINEAR 1u
/* sort_mode: src/sort/mod.rs:46-58 */
#define BGS_SORT_NONE 0u  /* draw in entry order (key=1,index=i  src/sort/mod.rs:347-354) */
#define BGS_SORT_RADIX 1u /* GPU radix semantics: cull + inverted key, stable            */
#define BGS_SORT_RAYON 2u /* CPU-sort semantics: key=bits(dist2), descending, no cull    */
#define BGS_SORT_STD 3u   /* same ordering contract as RAYON (src/sort/std_sort.rs)      */

/* rasterize_mode: discriminants of RasterizeMode (src/gaussian/settings.rs:38-47); selects the
 * colour the vertex stage gives a splat (src/render/gaussian.wgsl:312-417). OpticalFlow needs the
 * previous frame's clip_from_world + delta_time in bgs_view. Velocity needs a 4D cloud: BGS_EINVAL. */
#define BGS_RASTERIZE_CLASSIFICATION 0u
#define BGS_RASTERIZE_COLOR 1u /* default */
#define BGS_RASTERIZE_DEPTH 2u
#define BGS_RASTERIZE_NORMAL 3u
#define BGS_RASTERIZE_OPTICAL_FLOW 4u
#define BGS_RASTERIZE_POSITION 5u
#define BGS_RASTERIZE_VELOCITY 6u

/* draw_mode: src/gaussian/settings.rs:6-12 -> DRAW_SELECTED / HIGHLIGHT_SELECTED shader defs
 * (src/render/mod.rs:889-893; src/render/gaussian.wgsl:203-205, 423-427) */
#define BGS_DRAW_ALL 0u
#define BGS_DRAW_SELECTED 1u           /* splats with visibility < 0.5 are not drawn         */
#define BGS_DRAW_HIGHLIGHT_SELECTED 2u /* splats with visibility > 0.5 drawn (0.3, 1, 0.1, 1) */

/* CloudUniform (src/render/mod.rs:995-1009) + the CloudSettings fields that select the
 * pipeline specialisation (src/gaussian/settings.rs:87-132, src/render/mod.rs:770-896). */
typedef struct bgs_settings {
    float transform[16];              /* model matrix (GlobalTransform)                  */
    float global_opacity;             /* default 1                                       */
    float global_scale;               /* default 1                                       */
    uint32_t gaussian_mode;           /* BGS_GAUSSIAN_3D default                         */
    uint32_t aabb;                    /* 0 = OBB (default), 1 = AABB   settings.rs:113   */
    uint32_t opacity_adaptive_radius; /* default 1                     settings.rs:116   */
    uint32_t color_space;             /* BGS_COLOR_SRGB default                          */
    uint32_t radix_depth_bits;        /* 16 | 24 | 32 (default 32)     settings.rs:52-77 */
    uint32_t sh_degree;               /* 0..3; reference is compile-time sh3             */
    uint32_t sort_mode;               /* BGS_SORT_RADIX default                          */
    uint32_t rasterize_mode;          /* BGS_RASTERIZE_COLOR default   settings.rs:38-47 */
    uint32_t num_classes;             /* default 1 (Classification)    settings.rs:124   */
    uint32_t draw_mode;               /* BGS_DRAW_ALL default          settings.rs:6-12  */
    uint32_t visualize_bounding_box;  /* default 0: VISUALIZE_BOUNDING_BOX (settings.rs:95,117; key bit
                                         src/render/mod.rs:418,824; src/render/gaussian.wgsl:486-495): fragments in the
                                         outer 8 % of a quad's uv square are (0.3, 1, 0.1, 1) — the quads' frames  */
    uint32_t reserved[1];
    float position_min[4];            /* CloudUniform.min/max = the cloud entity's Aabb   */
    float position_max[4];            /* (src/render/mod.rs:1070-1071); Position mode     */
} bgs_settings;

/* src/sort/mod.rs:324-329 */
typedef struct bgs_sort_entry {
    uint32_t key;
    uint32_t index;
} bgs_sort_entry;

/* Per-stage wall time (HIP events on the context's stream) and counters of the most
 * recent bgs_sort/bgs_render call. Times in milliseconds. */
#define BGS_STAGE_KEYGEN 0   /* keygen + 4-place global digit histogram             */
#define BGS_STAGE_DEPTH_SORT 1 /* all onesweep passes over the depth keys            */
#define BGS_STAGE_PROJECT 2  /* per-splat projection + SH + tile-instance emission   */
#define BGS_STAGE_TILE_SORT 3 /* stable radix passes over tile ids                   */
#define BGS_STAGE_RANGES 4   /* per-tile [start,end)                                 */
#define BGS_STAGE_RASTER 5   /* per-16x16-tile front-to-back blend                   */
#define BGS_STAGE_COUNT 6

typedef struct bgs_stats {
    float stage_ms[BGS_STAGE_COUNT];
    float total_ms;              /* first to last event of the call                    */
    uint32_t splat_count;        /* N                                                  */
    uint32_t visible_count;      /* V: splats that pass the frustum test               */
    uint32_t draw_count;         /* D: entries that go through the radix passes / reach the
                                    vertex stage (key != culled sentinel)               */
    uint32_t sort_path;          /* depth sort of the call: 0 = onesweep digit passes, 1 = bucket sort
                                    (one launch; chosen per frame)        */
    uint64_t instance_count;     /* I: (tile, splat) instances emitted                 */
    uint64_t instance_capacity;  /* BGS_BINNING_SORT: tile instances the lane's buffers hold (0 until that mode ran) */
    uint32_t tiles_x, tiles_y;
    uint32_t depth_passes;       /* radix digit places used for the depth keys         */
    uint32_t tile_passes;        /* radix passes used for the tile ids                 */
    uint64_t algorithmic_bytes;  /* SURVEY 8(d) bytes_frame (or bytes_sort) of the call */
    uint32_t regrow_count;       /* times a frame was re-run because a data-dependent capacity was
                                    too small (tile instances, supertile lists, bucket sort) */
    uint32_t binning_mode;       /* BGS_BINNING_* used by the call                     */
    uint32_t frames_averaged;    /* stage_ms/total_ms = mean over this many frames (async: all
                                    frames since the last read-back, at most 64)        */
    uint32_t list_capacity;      /* BGS_BINNING_SCAN: entries each supertile list could hold  */
    uint64_t list_entries_allocated; /* BGS_BINNING_SCAN: 8-byte list entries allocated for the lane (all supertile
                                    lists together); instance_count is how many of them the frame filled */
    uint32_t strip_tiles;        /* tiles of the frame that were drawn by four strip waves instead of one wave (dense
                                    frames: the tiles a completed frame found heavy); 0 otherwise */
    uint32_t tile_saturation;    /* BGS_BINNING_SCAN frames drawn in cost order: bit 16 = known, bits 0-14 = share (x 0x7FFF) of
                                    the tile work of the completed frame the order was made of that was in tiles whose
                                    every pixel went opaque before their list ended — what selects the mid-round-exit
                                    rasteriser for a kind of frame; bit 31: this frame ran it. 0 otherwise */
} bgs_stats;

/* ---- lifecycle ------------------------------------------------------------------ */
int bgs_create(int hip_device, bgs_ctx** out);
void bgs_destroy(bgs_ctx* ctx);
const char* bgs_last_error(const bgs_ctx* ctx); /* ctx may be NULL: global message   */
uint32_t bgs_version(void);                     /* (major << 16) | minor             */
/* A binding's handshake, once at start-up: the version and the struct sizes IT was built with. BGS_EINVAL (message via
 * bgs_last_error(NULL)) unless they are this library's — the structs are passed by pointer, so a binding built
 * against another layout (0.2's bgs_view was 16 bytes shorter) would otherwise have the library read past its
 * struct. C callers: bgs_abi_check(BGS_ABI_VERSION, sizeof(bgs_view), sizeof(bgs_settings), sizeof(bgs_stats)). */
#define BGS_ABI_VERSION ((BGS_VERSION_MAJOR << 16) | BGS_VERSION_MINOR)
int bgs_abi_check(uint32_t version, uint32_t sizeof_view, uint32_t sizeof_settings, uint32_t sizeof_stats);
/* Identity of the kernel sources the library was compiled from: the SHA-256 (64 hex digits) over the HIP
 * sources and headers of libbgs (csrc/ .hip and .h files). The Python binding, the tests and bench.py refuse a library whose id is not the
 * tree's (and rebuild it); counter files under profiles/ carry the same stamp. */
const char* bgs_build_id(void);

/* Fill defaults equal to CloudSettings::default() (src/gaussian/settings.rs:110-131)
 * with an identity transform. */
void bgs_settings_default(bgs_settings* out);

/* Convenience: build a bgs_view the way Bevy builds its View uniform for
 * `Camera3d::default()` (infinite reverse-Z right-handed perspective) from a camera
 * world transform. Pure host arithmetic in f32. clear_color is set to opaque black, sample_count to 4
 * (Msaa::default() = Sample4), depth_device_ptr to 0. */
void bgs_view_perspective(const float world_from_view[16], float fov_y_radians,
                          float near_plane, uint32_t width, uint32_t height,
                          bgs_view* out);

/* ---- cloud upload (planar SoA, src/gaussian/formats/planar_3d.rs:45-54) ---------- */
/* f32 planes: position_visibility[n][4], spherical_harmonic[n][48] (index 3*k+c),
 * rotation[n][4] as [w,x,y,z], scale_opacity[n][4] (sx,sy,sz,opacity). */
int bgs_cloud_upload_f32(bgs_ctx* ctx, uint32_t n, const float* position_visibility,
                         const float* spherical_harmonic, const float* rotation,
                         const float* scale_opacity, bgs_cloud** out);
/* f16 planes (src/render/bindings.wgsl:102-140, src/gaussian/f16.rs:29-55):
 * position_visibility stays f32; spherical_harmonic[n][24] u32 (low half = even
 * coefficient, src/render/planar.wgsl:117-130); rotation_scale_opacity[n][4] u32 =
 * [rot0|rot1], [rot2|rot3], [s0|s1], [s2|opacity] with the FIRST value in the high half. */
int bgs_cloud_upload_f16(bgs_ctx* ctx, uint32_t n, const float* position_visibility,
                         const uint32_t* spherical_harmonic_h2,
                         const uint32_t* rotation_scale_opacity, bgs_cloud** out);
/* The `precompute_covariance_3d` storage variant (src/gaussian/f32.rs:218-251 Covariance3dOpacity,
 * src/render/planar.wgsl:132-152, src/render/gaussian_3d.wgsl:77-88): instead of rotation and scale the
 * cloud carries covariance_3d_opacity[n][8] = cov3d[6] (xx, xy, xz, yy, yz, zz of M^T M, M = S R:
 * src/gaussian/covariance.rs:4-41), opacity, pad. The vertex stage then skips compute_cov3d — and with
 * it the model transform's linear part and global_scale, which the reference applies only there.
 * 3D gaussian mode only (BGS_EINVAL for 2DGS and for RasterizeMode::Normal: both need rotation / scale). */
int bgs_cloud_upload_cov3d_f32(bgs_ctx* ctx, uint32_t n, const float* position_visibility,
                               const float* spherical_harmonic, const float* covariance_3d_opacity,
                               bgs_cloud** out);
void bgs_cloud_free(bgs_ctx* ctx, bgs_cloud* cloud);
uint32_t bgs_cloud_len(const bgs_cloud* cloud);

/* ---- the hot path ----------------------------------------------------------------- */
/* Depth sort for one view. Result is kept on the device (consumed by the next
 * bgs_render with the same cloud/view/settings) and, if host_out != NULL, copied to
 * host_out[n]. Order contract for BGS_SORT_RADIX: ascending key, ties by ascending
 * splat index, culled splats (key all-ones) last  (src/sort/radix.wgsl:109-279). */
int bgs_sort(bgs_ctx* ctx, const bgs_cloud* cloud, const bgs_view* view,
             const bgs_settings* settings, bgs_sort_entry* host_out);

/* Sort (always re-done: the benchmark sorts every frame) + project + bin +
 * rasterize one view. Output: viewport.w x viewport.h RGBA f32, premultiplied, linear,
 * unclamped, row 0 = top. rgba_host_out may be NULL (result stays on the device). */
int bgs_render(bgs_ctx* ctx, const bgs_cloud* cloud, const bgs_view* view,
               const bgs_settings* settings, float* rgba_host_out);

/* Device pointer + size in bytes of the framebuffer written by the last bgs_render
 * (for an RCCL gather or zero-copy interop). Valid until the next render/destroy. */
int bgs_framebuffer_device_ptr(bgs_ctx* ctx, void** dptr, uint64_t* bytes);
/* Also produce, with every bgs_render, the frame in the reference's colour-attachment format
 * TextureFormat::Rgba8UnormSrgb (src/render/mod.rs:917-921, examples/headless.rs:120-123): linear
 * RGB -> sRGB transfer -> unorm8, alpha linear, 4 bytes per pixel R,G,B,A. Default off. */
int bgs_set_output_srgb8(bgs_ctx* ctx, int enabled);
int bgs_framebuffer_srgb8_device_ptr(bgs_ctx* ctx, void** dptr, uint64_t* bytes);
/* The same for an hdr camera: the reference's colour attachment is then TextureFormat::Rgba16Float
 * (src/render/mod.rs:917-921). Every bgs_render also produces the frame as IEEE binary16 RGBA (8 bytes per
 * pixel, round to nearest even, overflow to inf), linear and unclamped. Exclusive with the sRGB8 output
 * (enabling one disables the other); bgs_set_srgb8_target / bgs_pipeline_pop's second pointer then carry
 * this image. Like the 8-bit image it is ONE conversion of the f32 result, not a per-blend quantisation. */
int bgs_set_output_rgba16f(bgs_ctx* ctx, int enabled);
int bgs_framebuffer_rgba16f_device_ptr(bgs_ctx* ctx, void** dptr, uint64_t* bytes);
/* Frames that write a packed image (sRGB8 or Rgba16Float) skip the f32 target (33 MB of writes per 1080p
 * frame) — for consumers that only ship the packed frame (the multi-GPU gather). BGS_BINNING_SCAN only.
 * bgs_framebuffer_device_ptr / a host copy of such a frame is BGS_EINVAL; bgs_pipeline_pop gives NULL. */
int bgs_set_packed_only(bgs_ctx* ctx, int enabled);
/* The NEXT bgs_render writes its Rgba8UnormSrgb image (width * height * 4 bytes) to this caller-owned
 * device memory instead of the lane's own buffer (one-shot; NULL cancels). Lets a consumer that ships
 * frames in batches (the multi-GPU gather) have each frame land in its slot of the batch with no copy
 * and no extra synchronisation. Implies srgb8 output for that frame; bgs_pipeline_pop /
 * bgs_framebuffer_srgb8_device_ptr then return this pointer for it. */
int bgs_set_srgb8_target(bgs_ctx* ctx, void* device_ptr);
/* Copy `bytes` from device memory this library handed out (a framebuffer, an sRGB8 image, the sorted
 * entries) to host memory, for hosts that do not link the HIP runtime themselves. Blocking. The
 * memory must belong to a COMPLETED frame: after a blocking call, bgs_pipeline_pop or bgs_synchronize. */
int bgs_download(bgs_ctx* ctx, const void* device_ptr, void* host_out, uint64_t bytes);
/* Device memory for what the CALLER hands to the library by device pointer (bgs_view.depth_device_ptr,
 * bgs_set_srgb8_target), for hosts that do not link the HIP runtime themselves: allocate / free on the context's
 * device, and a blocking host-to-device copy. bgs_device_free completes the frames in flight first. */
int bgs_device_alloc(bgs_ctx* ctx, uint64_t bytes, void** device_ptr_out);
int bgs_device_free(bgs_ctx* ctx, void* device_ptr);
int bgs_upload(bgs_ctx* ctx, void* device_ptr, const void* host_in, uint64_t bytes);

/* Frame pipelining. A single stream of this path's kernels is latency bound at 1M splats, so the
 * context can keep up to 8 frames in flight ("lanes", each with its own per-frame buffers, run on a
 * few shared HIP streams, see bgs_set_pipeline_streams): with bgs_set_async(1), successive bgs_render(..., NULL) calls go round-robin
 * over the lanes and overlap on the GPU. A lane's previous frame is completed (stream wait +
 * watchdog check) when the lane is reused. bgs_pipeline_pop completes the OLDEST frame in flight
 * and returns its f32 / sRGB8 framebuffers (valid until that lane is reused, i.e. for depth-1 more
 * enqueues). bgs_framebuffer_device_ptr & co refer to the most recently enqueued frame. */
int bgs_set_pipeline_depth(bgs_ctx* ctx, uint32_t lanes /* 1..8, default 1 */);
int bgs_pipeline_pop(bgs_ctx* ctx, void** rgba_f32_dptr_or_null, void** rgba8_dptr_or_null);
int bgs_frames_in_flight(bgs_ctx* ctx, uint32_t* count);

/* Device pointer of the sorted entries of the last call: after bgs_sort all n entries (culled
 * ones last); after bgs_render only the drawable prefix (*n = entries that reach the vertex
 * stage), because the culled tail is never needed for drawing. */
int bgs_sorted_entries_device_ptr(bgs_ctx* ctx, void** dptr, uint32_t* n);

/* Block until all work queued on the context's stream has finished; in async mode this is also
 * where the last frame's device watchdog word is checked (BGS_EINTERNAL) and its stats collected. */
int bgs_synchronize(bgs_ctx* ctx);
/* Async frames (default off). When on, bgs_render(..., rgba_host_out = NULL) under
 * BGS_BINNING_SCAN only ENQUEUES the frame and returns: that pipeline needs no host round trip
 * (every data-dependent size stays on the device; a frame whose supertile lists or bucket sort outgrow
 * their capacity is re-run on its lane when it is completed, before anyone sees it), so successive
 * frames run back-to-back on the GPU. Any call that needs results (a host copy, bgs_get_stats,
 * bgs_framebuffer_device_ptr, bgs_synchronize, bgs_sort) completes the pending frame first. */
int bgs_set_async(bgs_ctx* ctx, int enabled);
/* The hipStream_t (as void*) the context launches on, so a caller can order its own
 * HIP work (e.g. torch / RCCL) against it. */
int bgs_stream(bgs_ctx* ctx, void** hip_stream);

/* How splats are binned to 16x16 tiles (results are bit-identical between the two):
 *   BGS_BINNING_SCAN (default): ordered coarse (supertile) lists built during projection, each
 *     tile's rasteriser scans its list lazily and stops at saturation; instance_count in the
 *     stats is then the number of coarse list entries.
 *   BGS_BINNING_SORT: materialise every (tile, splat) instance and stable-radix-sort them by
 *     tile id ("tile-major|depth keys"), then per-tile ranges; instance_count = instances. */
#define BGS_BINNING_SCAN 0u
#define BGS_BINNING_SORT 1u
int bgs_set_binning(bgs_ctx* ctx, uint32_t mode);

/* Forget what completed frames taught the context: draw-count hint (grid sizes), drawable key range and
 * the bucket-sort back-off, supertile rule, list-capacity hint. Buffers stay allocated. The next frames
 * behave like the first frames of a fresh context (onesweep passes, default capacities, re-run on
 * overflow). Completes the frames in flight. */
int bgs_reset_adaptive_state(bgs_ctx* ctx);
/* HIP-event timing level: 0 = none, 1 = frame start/end only (total_ms), 2 = every stage
 * (default). Each recorded event costs a few microseconds of GPU timeline. */
int bgs_set_profiling(bgs_ctx* ctx, int enabled);
/* Record the events only on every Nth frame (default 1 = every frame); stage_ms is then the mean
 * over the timed frames since the last read-back. */
int bgs_set_profiling_stride(bgs_ctx* ctx, uint32_t every_nth_frame);
/* Stats of the most recent bgs_sort / bgs_render. Synchronises the stream. */
int bgs_get_stats(bgs_ctx* ctx, bgs_stats* out);

/* Streams the lanes run on: lane i uses stream i % min(streams, depth); 0 = one stream per lane.
 * Default 4: the HIP runtime multiplexes a process's streams onto 4 hardware queues (per priority), and only
 * queues run concurrently, so 4 streams — each on its own queue, the library parks idle streams to get them
 * dealt out that way — is one frame per queue; with more lanes than streams (depth 8 on 4 streams) a stream
 * already holds the next frame of a sibling lane while one executes and never waits for the host.
 * Measured on the headline workload (round 2): 8 lanes on 4 streams 19.0 k frames/s, 4 on 4 18.9 k, 8 on 8
 * 19.0 k, 6 on 3 16.9 k, 6 on 6 17.6 k, and anything that spreads unevenly over the 4 queues loses (8 on 6 16.7 k,
 * 5 on 5 15.6 k). BGS_QUEUE_HOLDERS=0 in the environment keeps the library from parking its idle streams: for
 * processes whose other streams (RCCL's) already hold the queues. */
int bgs_set_pipeline_streams(bgs_ctx* ctx, uint32_t streams);
/* Frame graphs (opt-in, default off). An asynchronous BINNING_SCAN frame (bgs_set_async) in the
 * steady state is replayed from a hipGraph captured once per (lane, parity): the frame's first kernel
 * receives the new view/settings as its arguments (one graph-node update) and leaves them in device
 * memory for the kernels behind it. A replay costs the host ~10 us instead of ~19 us for 7 direct
 * launches, but runs ~5 % slower on the GPU than the same launches issued directly (measured), so it
 * is meant for hosts short of CPU time. The graph is rebuilt when anything else a launch depends on
 * changes (cloud, buffers, viewport size, pipeline variant, grid sizes, debug flags); frames whose
 * stages are timed with events (bgs_set_profiling) are always launched directly. */
int bgs_set_graphs(bgs_ctx* ctx, int enabled);

/* ---- multi-GPU: the gather of finished frames ------------------------------------------------------------------
 * Views are independent units (the reference keys its sort state and entry offsets by camera: src/sort/mod.rs:143-150,
 * src/render/mod.rs:1548-1554): one process per GPU, camera g on rank g, a replica of the cloud on every GPU, no
 * exchange during sort / rasterise. The ONE collective is the gather of finished frames to a root rank: RCCL
 * ncclGather (grouped send / receive over xGMI, every non-root rank on its own link), enqueued on a stream the
 * communicator owns. librccl is opened when the first bgs_comm_* call needs it (hosts that never gather do not need
 * it installed).
 *   rank 0:      bgs_comm_unique_id(id)   -> ship the 128 bytes to every rank by the host's own means (a file, MPI,
 *                                            the launcher's rendezvous)
 *   every rank:  bgs_comm_create(ctx, id, world_size, rank, &comm)      (collective: returns once all ranks called it)
 *   per batch:   bgs_comm_gather(ctx, comm, root, send, bytes, recv, &ticket)   asynchronous; `send` (device memory
 *                holding COMPLETED frames: after bgs_pipeline_pop / bgs_synchronize — e.g. the batch buffer the frames
 *                were written to through bgs_set_srgb8_target) and `recv` (root only: world_size * bytes, rank r's
 *                block at r * bytes) must stay untouched until that gather has completed
 *                bgs_comm_wait(ctx, comm, ticket)  blocks until the gather with that ticket (and every earlier one) has
 *                completed on this rank; ticket 0 = every gather enqueued so far. Double buffering: gather batch k,
 *                render batch k + 1 into the other buffer, wait for ticket k - 1 before reusing its buffer.
 * A communicator belongs to the context's device and, like the context, is driven by one thread. */
#define BGS_COMM_ID_BYTES 128
typedef struct bgs_comm bgs_comm;
int bgs_comm_unique_id(uint8_t id_out[BGS_COMM_ID_BYTES]);
int bgs_comm_create(bgs_ctx* ctx, const uint8_t id[BGS_COMM_ID_BYTES], uint32_t world_size, uint32_t rank, bgs_comm** out);
int bgs_comm_gather(bgs_ctx* ctx, bgs_comm* comm, uint32_t root, const void* send_device_ptr, uint64_t bytes_per_rank,
                    void* recv_device_ptr_root_only, uint64_t* ticket_out_or_null);
/* The same gather ORDERED ON THE DEVICE behind work that is still running (round 6): the communicator's stream first
 * waits — a stream-to-stream dependency, no host round trip — for everything enqueued so far on `hip_stream`, or, with
 * hip_stream == NULL, for every frame this context has enqueued and not yet completed (all lanes). The host may then
 * enqueue the gather of a batch right behind the batch's last bgs_render instead of popping / synchronising its frames
 * first (bgs_comm_gather's contract, a host round trip per batch and rank). What it cannot know: a frame the library
 * re-runs later because a data-dependent capacity turned out too small (bgs_adaptive_counters: reruns_sort, reruns_lists)
 * has been sent in the state of its first attempt — a host that gathers this way checks those counters when it pops the
 * batch's frames (they do not move on a settled view) and gathers the batch again, or uses bgs_comm_gather. */
int bgs_comm_gather_after(bgs_ctx* ctx, bgs_comm* comm, uint32_t root, const void* send_device_ptr, uint64_t bytes_per_rank,
                          void* recv_device_ptr_root_only, void* hip_stream_or_null, uint64_t* ticket_out_or_null);
int bgs_comm_wait(bgs_ctx* ctx, bgs_comm* comm, uint64_t ticket);
int bgs_comm_stream(bgs_ctx* ctx, bgs_comm* comm, void** hip_stream); /* the stream the gathers run on */
void bgs_comm_destroy(bgs_ctx* ctx, bgs_comm* comm);
#ifdef __cplusplus
}
#endif
#endif /* BGS_H */
