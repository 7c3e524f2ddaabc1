// bgs_frame.hip — libbgs host side: the frame engine. Lanes and their buffers, the adaptive state completed frames leave,
// one frame's launches (directly or as a captured graph), completion with its capacity checks and re-runs.
// (The comment at the top of bgs_api.hip is the overview.)
#include "bgs_context.h"

thread_local std::string bgs_host::g_error;

namespace bgs_host {


// Idle "queue holder" streams (see assign_streams): PROCESS-global, three per device, created once before the first
// context of that device creates its own streams and kept for the life of the process — a process with several
// contexts (multi-camera, multi-cloud) parks three streams in all, not three per context, so its streams keep
// being dealt out evenly over the runtime's four hardware queues. -1: follow BGS_QUEUE_HOLDERS (default on).
constexpr int MAX_DEVICES = 64;
hipStream_t g_queue_holders[MAX_DEVICES][3] = {};
int g_queue_holders_mode = -1;
std::mutex g_queue_holders_mutex;   // contexts of one process may be created from different threads

int fail(bgs_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->error = msg;
    g_error = msg;
    return status;
}


size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Lane i runs on stream i % S (S = bgs_set_pipeline_streams, default: the pipeline depth, i.e. a
// stream per lane). With S < depth a stream holds the NEXT frame of a sibling lane while one executes,
// so the stream never waits for the host between frames.
int assign_streams(bgs_ctx* ctx) {
    const int S = ctx->num_streams > 0 ? std::min(ctx->num_streams, ctx->depth) : ctx->depth;
    for (int i = 0; i < MAX_LANES; ++i) {
        Lane& L = ctx->lanes[i];
        if (!L.done) continue;
        const int si = i < ctx->depth ? i % S : i;
        if (!ctx->streams[si]) {
            // The HIP runtime multiplexes a process's streams onto at most 4 hardware queues per priority
            // (GPU_MAX_HW_QUEUES), and only queues are concurrent: two streams on one queue run one after the other.
            // It creates a NEW queue for every new stream until the 4 exist and only then spreads further streams
            // by reference count — and the process's null stream already holds one. Left alone, our streams 0, 1, 2
            // get a queue each and stream 3 joins stream 2's (seen in the runtime's log, AMD_LOG_LEVEL=3): four
            // frames on three queues, two of them serialised — 14.0 k frames/s with 8 lanes on 4 streams where a
            // process that had initialised RCCL (whose idle streams happen to hold the queues) ran 19.2 k. So the
            // context parks three idle streams on the queues FIRST; ours are then dealt out evenly over all four,
            // the null stream's included (8 lanes / 4 streams 18.2 k, 8 / 8 19.0 k; scripts/queues_probe.py).
            // Priorities other than the default do not help: their queue pools are separate but slower (high:
            // 13.7 k at 8 / 4, 14.4 k at 6 / 6).
            // bgs_set_queue_holders(0) or BGS_QUEUE_HOLDERS=0 in the environment switches this off: a process whose
            // other streams already hold the queues (RCCL's, after a process group was initialised: bench.py's gather
            // path does) is better off without three more co-tenants on them (that path: 18.0 k frames/s without,
            // 13.6 k with). The holders are process-global (one set per device, however many contexts exist).
            {
                std::lock_guard<std::mutex> lock(g_queue_holders_mutex);
                const char* qh_env = std::getenv("BGS_QUEUE_HOLDERS");
                const bool park = g_queue_holders_mode >= 0 ? g_queue_holders_mode != 0 : !(qh_env && qh_env[0] == '0');
                if (park && ctx->device >= 0 && ctx->device < MAX_DEVICES && !g_queue_holders[ctx->device][0])
                    for (auto& qh : g_queue_holders[ctx->device]) HIP_TRY(ctx, hipStreamCreateWithFlags(&qh, hipStreamNonBlocking));
            }
            HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->streams[si], hipStreamNonBlocking));
        }
        L.stream = ctx->streams[si];
    }
    return BGS_OK;
}

int lane_create(bgs_ctx* ctx, Lane& L) {
    if (L.done) return L.stream ? BGS_OK : assign_streams(ctx);
    HIP_TRY(ctx, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    for (auto& slot : L.ev_ring)
        for (auto& ev : slot) HIP_TRY(ctx, hipEventCreate(&ev));
    void* h = nullptr;
    HIP_TRY(ctx, hipHostMalloc(&h, sizeof(Control), hipHostMallocDefault));
    L.h_ctl = (Control*)h;
    std::memset(L.h_ctl, 0, sizeof(Control));
    void* hd = nullptr;
    HIP_TRY(ctx, hipHostGetDevicePointer(&hd, h, 0));
    L.h_ctl_dev = (Control*)hd;
    L.d_fp = dev_alloc<FrameParams>(1);
    if (!L.d_fp) return fail(ctx, BGS_ENOMEM, "hipMalloc(frame params) failed");
    return assign_streams(ctx);
}

void graph_destroy(FrameGraph& g) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
    g = FrameGraph();
}

void lane_destroy(Lane& L) {
    if (L.stream) (void)hipStreamSynchronize(L.stream);
    for (auto& g : L.graph) graph_destroy(g);
    if (L.d_fp) (void)hipFree(L.d_fp);
    if (L.scratch) (void)hipFree(L.scratch);
    for (auto e : L.entries) if (e) (void)hipFree(e);
    if (L.culled) (void)hipFree(L.culled);
    for (auto e : L.inst) if (e) (void)hipFree(e);
    if (L.records) (void)hipFree(L.records);
    if (L.rects) (void)hipFree(L.rects);
    if (L.coarse) (void)hipFree(L.coarse);
    if (L.bucket_slots) (void)hipFree(L.bucket_slots);
    if (L.d_split_keys) (void)hipFree(L.d_split_keys);
    if (L.h_split_keys) (void)hipHostFree(L.h_split_keys);
    for (auto h : L.heavy) if (h) (void)hipFree(h);
    for (auto c : L.cost) if (c) (void)hipFree(c);
    if (L.order) (void)hipFree(L.order);
    if (L.fb) (void)hipFree(L.fb);
    if (L.fb8) (void)hipFree(L.fb8);
    if (L.h_ctl) (void)hipHostFree(L.h_ctl);
    for (auto& slot : L.ev_ring)
        for (auto ev : slot) if (ev) (void)hipEventDestroy(ev);
    if (L.done) (void)hipEventDestroy(L.done);
    L = Lane();
}

// (Re)build the zeroed scratch region for n splats and inst_cap instances.
int ensure_scratch(bgs_ctx* ctx, Lane& L, uint32_t n, uint64_t inst_cap) {
    if (L.scratch && n <= L.scratch_n && inst_cap <= L.scratch_inst_cap) return BGS_OK;
    n = std::max(n, L.scratch_n);
    inst_cap = std::max(inst_cap, L.scratch_inst_cap);
    // depth sort may use either tile size; size for the smaller one
    const size_t depth_tiles = ((size_t)n + sort_tile_size(false) - 1) / sort_tile_size(false) + 1;
    const size_t scan_tiles = ((size_t)n + 255) / 256 + 1;
    const size_t inst_tiles = (inst_cap + sort_tile_size(true) - 1) / sort_tile_size(true) + 1;
    size_t off = align_up(sizeof(Control), 256);
    const size_t off_depth = off;
    off += align_up(4 * depth_tiles * RADIX_BASE * sizeof(uint32_t), 256);
    const size_t off_scan = off;
    off += align_up(scan_tiles * sizeof(unsigned long long), 256);
    const size_t off_tile = off;
    off += align_up(2 * inst_tiles * RADIX_BASE * sizeof(uint32_t), 256);
    const size_t off_ranges = off;
    off += align_up((size_t)RADIX_BASE * RADIX_BASE * sizeof(uint2), 256);
    const size_t off_bin = off;
    off += align_up(scan_tiles * MAX_SUPERTILES * sizeof(uint32_t), 256);
    const size_t off_part = off;
    off += align_up((((size_t)n + KEYGEN_TILE - 1) / KEYGEN_TILE + 1) * sizeof(uint32_t), 256);
    const size_t off_ctl1 = off;  // the lane's second Control block (see FrameCleanup)
    off += align_up(sizeof(Control), 256);
    if (L.scratch) { (void)hipFree(L.scratch); L.scratch = nullptr; }
    void* p = nullptr;
    if (hipMalloc(&p, off) != hipSuccess) return fail(ctx, BGS_ENOMEM, "hipMalloc(scratch) failed");
    L.scratch = (uint8_t*)p;
    L.scratch_bytes = off;
    L.off_depth_status = off_depth;
    L.off_scan_status = off_scan;
    L.off_tile_status = off_tile;
    L.off_ranges = off_ranges;
    L.off_bin_status = off_bin;
    L.off_part_status = off_part;
    L.off_ctl1 = off_ctl1;
    L.scratch_clean = false;
    L.scratch_n = n;
    L.scratch_inst_cap = inst_cap;
    return BGS_OK;
}

int ensure_entries(bgs_ctx* ctx, Lane& L, uint32_t n) {
    if (n <= L.entries_cap && L.entries[0]) return BGS_OK;
    for (auto& e : L.entries) { if (e) (void)hipFree(e); e = nullptr; }
    if (L.culled) { (void)hipFree(L.culled); L.culled = nullptr; }
    if (L.rects) { (void)hipFree(L.rects); L.rects = nullptr; }
    for (auto& e : L.entries) {
        e = dev_alloc<uint2>(n);
        if (!e) return fail(ctx, BGS_ENOMEM, "hipMalloc(sort entries) failed");
    }
    L.culled = dev_alloc<uint2>(n);
    if (!L.culled) return fail(ctx, BGS_ENOMEM, "hipMalloc(culled entries) failed");
    L.entries_cap = n;
    return BGS_OK;
}

// BINNING_SCAN renders only: the packed tile rectangle per rank (project_kernel -> bin_kernel); freed with the entries
int ensure_rects(bgs_ctx* ctx, Lane& L) {
    if (L.rects) return BGS_OK;
    L.rects = dev_alloc<uint32_t>(L.entries_cap);
    if (!L.rects) return fail(ctx, BGS_ENOMEM, "hipMalloc(tile rectangles) failed");
    return BGS_OK;
}

int ensure_instances(bgs_ctx* ctx, Lane& L, uint64_t cap) {
    if (cap <= L.inst_cap && L.inst[0]) return BGS_OK;
    for (auto& e : L.inst) { if (e) (void)hipFree(e); e = nullptr; }
    L.inst_cap = 0;
    for (auto& e : L.inst) {
        e = dev_alloc<uint2>(cap);
        if (!e) return fail(ctx, BGS_ENOMEM, "hipMalloc(tile instances) failed");
    }
    L.inst_cap = cap;
    return BGS_OK;
}

int ensure_records(bgs_ctx* ctx, Lane& L, size_t bytes) {
    if (bytes <= L.records_bytes && L.records) return BGS_OK;
    if (L.records) (void)hipFree(L.records);
    L.records = nullptr;
    L.records_bytes = 0;
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess)
        return fail(ctx, BGS_ENOMEM, "hipMalloc(records) failed");
    L.records = p;
    L.records_bytes = bytes;
    return BGS_OK;
}

uint32_t pow2_ceil(uint64_t v) { return pow2_ceil_u32(v); }

// Supertile lists: `num_st` lists of `cap` (rank, tile rect) entries each. `cap` follows the longest list
// seen so far (ctx->coarse_cap_hint, never more than n: a list holds each rank at most once); a frame
// that overflows a list is detected when it completes (coarse_total > cap) and re-run with larger lists.
int ensure_coarse(bgs_ctx* ctx, Lane& L, uint32_t n, uint32_t num_st, uint32_t* cap_out) {
    const uint32_t n1 = std::max<uint32_t>(n, 1);
    if (ctx->coarse_cap_hint == 0)
        ctx->coarse_cap_hint = (ctx->debug_flags & 0x100000u) ? 64u : std::max<uint32_t>(pow2_ceil(n1 / 64u), 4096u);  // a first guess: a frame that outgrows it is re-run
    const uint32_t want = std::min<uint32_t>(n1, ctx->coarse_cap_hint);
    const size_t need = (size_t)num_st * want;
    // (grown when too small; a lane keeps what it has when the hint falls — a context that alternates between
    // views of different density would otherwise free and allocate every frame)
    if (need > L.coarse_entries || !L.coarse) {
        if (need * 8u > (64ull << 30))
            return fail(ctx, BGS_ECAPACITY, "coarse bin lists would exceed 64 GiB; use bgs_set_binning(ctx, 1)");
        if (L.coarse) (void)hipFree(L.coarse);
        L.coarse = nullptr;
        L.coarse_entries = 0;
        L.coarse = dev_alloc<uint32_t>(2 * need);
        if (!L.coarse)
            return fail(ctx, BGS_ENOMEM, "hipMalloc(coarse lists) failed: " + std::to_string((need * 8u) >> 20) +
                                             " MiB per lane (8 B x supertiles x longest list); fewer lanes (bgs_set_pipeline_depth) need less");
        L.coarse_entries = need;
    }
    // everything that is allocated is used (a lane that grew for an earlier frame keeps its longer lists;
    // not under debug flag 0x100000, which exists to exercise the overflow path)
    *cap_out = (ctx->debug_flags & 0x100000u) ? want : (uint32_t)std::min<size_t>(L.coarse_entries / num_st, n1);
    return BGS_OK;
}

int ensure_bucket_slots(bgs_ctx* ctx, Lane& L, uint32_t sub, bool wide) {
    const uint32_t units = sub * (wide ? BUCKET_CAP_WIDE / BUCKET_CAP : 1u);   // (in narrow subs: 256 * BUCKET_CAP pairs = 8 MB each)
    if (L.bucket_slots && units <= L.bucket_sub_cap) return BGS_OK;
    if (L.bucket_slots) (void)hipFree(L.bucket_slots);   // (waits for the device: no frame in flight still uses them)
    L.bucket_sub_cap = 0;
    L.bucket_slots = dev_alloc<uint2>((size_t)BUCKET_COUNT * units * BUCKET_CAP);
    if (!L.bucket_slots) return fail(ctx, BGS_ENOMEM, "hipMalloc(bucket sort slots) failed");
    L.bucket_sub_cap = units;
    return BGS_OK;
}

int ensure_heavy(bgs_ctx* ctx, Lane& L, uint32_t tiles) {
    if (L.heavy[0] && tiles <= L.heavy_tiles) return BGS_OK;
    // (hipFree waits for the device: no frame in flight still reads the old buffers)
    L.heavy_done = nullptr;
    for (auto& h : L.heavy) { if (h) (void)hipFree(h); h = nullptr; }
    L.heavy_tiles = 0;
    for (auto& h : L.heavy) {
        void* p = nullptr;
        if (hipMalloc(&p, heavy_feedback_bytes(tiles)) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, BGS_ENOMEM, "hipMalloc(heavy-tile feedback) failed"); }
        h = (uint8_t*)p;
    }
    L.heavy_tiles = tiles;
    return BGS_OK;
}

int ensure_cost(bgs_ctx* ctx, Lane& L, uint32_t tiles) {
    if (L.cost[0] && tiles <= L.cost_tiles) return BGS_OK;
    L.cost_done = nullptr;
    for (auto& c : L.cost) { if (c) (void)hipFree(c); c = nullptr; }
    if (L.order) { (void)hipFree(L.order); L.order = nullptr; }
    L.order_grid = 0xFFFFFFFFu;
    L.cost_tiles = 0;
    for (auto& c : L.cost) {
        void* p = nullptr;
        if (hipMalloc(&p, tile_cost_bytes(tiles)) != hipSuccess || hipMemset(p, 0, tile_cost_bytes(tiles)) != hipSuccess) {
            (void)hipGetLastError();
            if (p) (void)hipFree(p);
            return fail(ctx, BGS_ENOMEM, "hipMalloc(tile cost feedback) failed");
        }
        c = (uint16_t*)p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, tile_order_bytes(tiles)) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, BGS_ENOMEM, "hipMalloc(tile order) failed"); }
    L.order = (uint16_t*)p;
    L.cost_tiles = tiles;
    return BGS_OK;
}

int ensure_framebuffer(bgs_ctx* ctx, Lane& L, uint32_t w, uint32_t h, bool want8) {
    const size_t px = (size_t)w * h;
    if (px > L.fb_pixels || !L.fb) {
        if (L.fb) (void)hipFree(L.fb);
        L.fb = dev_alloc<float4>(px);
        if (!L.fb) return fail(ctx, BGS_ENOMEM, "hipMalloc(framebuffer) failed");
        L.fb_pixels = px;
    }
    if (want8 && (px > L.fb8_pixels || !L.fb8)) {
        if (L.fb8) (void)hipFree(L.fb8);
        L.fb8 = dev_alloc<uint32_t>(2 * px);  // room for either packed format (4 or 8 bytes per pixel)
        if (!L.fb8) return fail(ctx, BGS_ENOMEM, "hipMalloc(srgb8 framebuffer) failed");
        L.fb8_pixels = px;
    }
    L.fb_w = w;
    L.fb_h = h;
    return BGS_OK;
}

int validate(bgs_ctx* ctx, const bgs_cloud* cloud, const bgs_view* view, const bgs_settings* s, bool render) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (!cloud || !view || !s) return fail(ctx, BGS_EINVAL, "cloud, view and settings must be non-NULL");
    if (s->radix_depth_bits != 16 && s->radix_depth_bits != 24 && s->radix_depth_bits != 32)
        return fail(ctx, BGS_EINVAL, "radix_depth_bits must be 16, 24 or 32");
    if (s->gaussian_mode > BGS_GAUSSIAN_3D) return fail(ctx, BGS_EINVAL, "gaussian_mode must be 2D or 3D");
    if (s->sh_degree > 3) return fail(ctx, BGS_EINVAL, "sh_degree must be 0..3");
    if (s->sort_mode > BGS_SORT_STD) return fail(ctx, BGS_EINVAL, "unknown sort_mode");
    if (s->color_space > BGS_COLOR_LINEAR) return fail(ctx, BGS_EINVAL, "unknown color_space");
    if (s->rasterize_mode == BGS_RASTERIZE_VELOCITY)
        return fail(ctx, BGS_EINVAL, "rasterize_mode Velocity is outside the path (4D clouds only)");
    if (s->rasterize_mode == BGS_RASTERIZE_OPTICAL_FLOW && !(view->delta_time > 0.0f))
        return fail(ctx, BGS_EINVAL, "rasterize_mode OpticalFlow needs bgs_view.delta_time > 0");
    if (s->rasterize_mode > BGS_RASTERIZE_VELOCITY) return fail(ctx, BGS_EINVAL, "unknown rasterize_mode");
    if (s->draw_mode > BGS_DRAW_HIGHLIGHT_SELECTED) return fail(ctx, BGS_EINVAL, "unknown draw_mode");
    if (s->rasterize_mode == BGS_RASTERIZE_CLASSIFICATION && s->num_classes == 0)
        return fail(ctx, BGS_EINVAL, "num_classes must be >= 1");
    if (render && cloud->ptrs.format == CLOUD_COV3D &&
        (s->gaussian_mode != BGS_GAUSSIAN_3D || s->rasterize_mode == BGS_RASTERIZE_NORMAL))
        return fail(ctx, BGS_EINVAL, "a precomputed-covariance cloud has no rotation / scale: 3D gaussian mode only, no Normal raster mode");
    if (render) {
        // MultisampleState.count = Msaa::samples() (src/render/mod.rs:357-424,975-979): Off and Sample4 are built
        // (0 = "not set": a zero-initialised bgs_view gets Msaa::default() = Sample4, fill_frame_params)
        const uint32_t samples = view->sample_count ? view->sample_count : 4u;
        if (samples != 1u && samples != 2u && samples != 4u && samples != 8u)
            return fail(ctx, BGS_EINVAL, "bgs_view.sample_count must be 1 (Msaa::Off), 2, 4 (Msaa::Sample4, Bevy's default), 8 or 0 (= 4); got " +
                                             std::to_string(view->sample_count));
        if (ctx->tile_trace && (view->depth_device_ptr || s->visualize_bounding_box || samples == 2u || samples == 8u))
            return fail(ctx, BGS_EINVAL, "the per-tile trace (bgs_set_tile_trace) has no instantiation with a depth buffer, the bounding-box "
                                         "overlay or 2 / 8 samples per pixel: a frame would leave the trace buffer untouched");
        if (view->depth_device_ptr % (4u * samples) != 0u)
            return fail(ctx, BGS_EINVAL, "bgs_view.depth_device_ptr must be aligned to one pixel's samples (4 * sample_count bytes)");
        const float w = view->viewport[2], h = view->viewport[3];
        if (!(w >= 1.0f) || !(h >= 1.0f) || w > 4096.0f || h > 4096.0f || w != std::floor(w) || h != std::floor(h))
            return fail(ctx, BGS_EINVAL, "viewport width/height must be integers in [1, 4096]");
    }
    return BGS_OK;
}

int enqueue_frame(bgs_ctx* ctx, Lane& L, const bgs_cloud* cloud, const bgs_view* view, const bgs_settings* s,
                  bool render, bool allow_graph);

// camera pose of a view: world position and viewing direction (-Z of the view frame)
void view_pose(const bgs_view* v, float pos[3], float fwd[3]) {
    for (int k = 0; k < 3; ++k) { pos[k] = v->world_from_view[12 + k]; fwd[k] = -v->world_from_view[8 + k]; }
    const float len = std::sqrt(fwd[0] * fwd[0] + fwd[1] * fwd[1] + fwd[2] * fwd[2]);
    if (len > 0.0f) for (int k = 0; k < 3; ++k) fwd[k] /= len;
}

// The splitter slot that fits a frame (same cloud, sort mode and model transform, camera within 5 % of the
// slot's reach and 10 degrees of its direction), or -1.
int find_splitter_slot(bgs_ctx* ctx, const bgs_cloud* cloud, const bgs_view* view, const bgs_settings* s) {
    float pos[3], fwd[3];
    view_pose(view, pos, fwd);
    int best = -1;
    float best_d = 0.0f;
    for (int i = 0; i < bgs_ctx::SPLITTER_SLOTS; ++i) {
        const auto& sl = ctx->split_slots[i];
        if (!sl.epoch || sl.cloud != cloud || sl.n != cloud->ptrs.n || sl.sort_mode != s->sort_mode) continue;
        if (std::memcmp(sl.transform, s->transform, sizeof sl.transform) != 0) continue;
        if (s->sort_mode == BGS_SORT_RADIX &&
            (std::memcmp(sl.clip_from_view, view->clip_from_view, sizeof sl.clip_from_view) != 0 ||
             sl.viewport_wh[0] != view->viewport[2] || sl.viewport_wh[1] != view->viewport[3]))
            continue;
        const float dx = pos[0] - sl.pos[0], dy = pos[1] - sl.pos[1], dz = pos[2] - sl.pos[2];
        const float d = std::sqrt(dx * dx + dy * dy + dz * dz);
        const float c = fwd[0] * sl.fwd[0] + fwd[1] * sl.fwd[1] + fwd[2] * sl.fwd[2];
        if (!(d <= 0.05f * sl.reach) || !(c >= 0.9848f)) continue;
        const float score = d / std::max(sl.reach, 1e-30f) + (1.0f - c);
        if (best < 0 || score < best_d) { best = i; best_d = score; }
    }
    return best;
}

// What bgs_ctx::kinds is keyed by: a hash of the inputs the data-dependent capacities depend on (never 0). The cloud
// enters by its size and storage format, not by its address (a host that uploads a new cloud per frame — a stream of
// same-sized captures — stays pipelined; a different cloud of the same size is at worst a re-run, as for any stale
// hint), the global scale to half an octave (an animated scale crosses a step now and then, not with every frame).
uint64_t frame_kind(const bgs_cloud* cloud, const bgs_view* view, const bgs_settings* s) {
    int32_t scale_step = INT32_MIN;
    if (s->global_scale > 0.0f && std::isfinite(s->global_scale)) scale_step = (int32_t)std::lround(2.0 * std::log2((double)s->global_scale));
    const uint64_t words[6] = {((uint64_t)cloud->ptrs.n << 32) | cloud->ptrs.format, ((uint64_t)s->gaussian_mode << 32) | s->aabb,
                               ((uint64_t)(uint32_t)scale_step << 32) | s->opacity_adaptive_radius,
                               ((uint64_t)(uint32_t)view->viewport[2] << 32) | (uint32_t)view->viewport[3],
                               ((uint64_t)(view->sample_count ? view->sample_count : 4u) << 32) | (view->depth_device_ptr ? 1u : 0u),
                               ((uint64_t)s->rasterize_mode << 32) | (s->draw_mode << 1) | (s->visualize_bounding_box ? 1u : 0u)};
    uint64_t hsh = 0xcbf29ce484222325ull;   // FNV-1a over the words
    for (uint64_t w : words)
        for (int b = 0; b < 8; ++b) { hsh ^= (w >> (8 * b)) & 0xFFu; hsh *= 0x100000001b3ull; }
    return hsh ? hsh : 1ull;
}

// The frame about to be enqueued is of `kind`: ctx->sup_level becomes that kind's level (if it has one).
void switch_kind(bgs_ctx* ctx, uint64_t kind) {
    if (kind == ctx->cur_kind) return;
    const auto it = ctx->kinds.find(kind);
    if (it != ctx->kinds.end()) ctx->sup_level = it->second.sup_level;
    ctx->cur_kind = kind;
}

// Complete the frame pending on a lane: wait for it, check the watchdog word of the Control copy that
// travelled with the frame, and RE-RUN the frame on its lane if a data-dependent capacity turned out too
// small (a supertile list, the bucket sort's geometry, the tile-instance buffer): nobody has seen the
// frame's output yet, so the caller just gets the correct frame a little later. Fills the lane's stats.
int finish_lane(bgs_ctx* ctx, Lane& L) {
    L.ready = false;
    for (int attempt = 0; L.pending; ++attempt) {
        hipStream_t st = L.stream;
        HIP_TRY(ctx, hipEventSynchronize(L.done));  // not the stream: a sibling lane's frame may be queued behind
        L.pending = false;
        const bool render = L.pending_render, scan = L.pending_scan;
        const uint32_t n = L.pending_n, places = L.pending_places, num_st = L.pending_num_st;
        const size_t rec_bytes = L.pending_rec_bytes;

        const Control& h = *L.h_ctl;
        if (h.error) {
            // nothing a tripped frame left behind is trusted: not its counters, not the scratch region
            L.scratch_clean = false;
            ctx->draw_hint_valid = false;
            for (auto& sl : ctx->split_slots) sl.epoch = 0;
            return fail(ctx, BGS_EINTERNAL,
                        "device watchdog tripped (look-back spin bound), code " + std::to_string(h.error));
        }
        // ---- capacities that depend on the data ----
        bool rerun = false, sort_gave_up = false;
        if (L.pending_bucket && h.sort_overflow) {
            sort_gave_up = true;
            // A bucket over capacity (1): the view changed faster than the splitters follow; the table is dropped
            // and the re-run below (always on the digit passes) delivers a fresh one. Frames already in flight
            // with the same stale table fail for the same reason, so only a table NEWER than the last failed one
            // counts towards the back-off (three such tables in a row: 7, 15, ... 255 frames on the passes).
            // One key value far too often (2): no table can split that; 256 frames on the passes.
            if (L.pending_split_slot >= 0 && ctx->split_slots[L.pending_split_slot].epoch == L.pending_split_epoch)
                ctx->split_slots[L.pending_split_slot].epoch = 0;  // drop the table
            if (h.sort_overflow & 2u) {
                ctx->bucket_block = 256u;
            } else if (L.pending_split_epoch > ctx->split_failed_epoch) {
                ctx->bucket_fail_streak = std::min(ctx->bucket_fail_streak + 1u, 8u);
                if (ctx->bucket_fail_streak >= 3u) ctx->bucket_block = (1u << ctx->bucket_fail_streak) - 1u;
            }
            ctx->split_failed_epoch = std::max(ctx->split_failed_epoch, L.pending_split_epoch);
            ctx->reruns_sort += 1;
            rerun = true;
        } else if (L.pending_bucket) {
            ctx->bucket_fail_streak = 0;
        }
        uint64_t total = (uint64_t)h.instance_total_lo | ((uint64_t)h.instance_total_hi << 32);
        uint32_t pending_longest = 0;
        if (render && scan) {
            total = 0;
            uint32_t longest = 0;
            for (uint32_t i = 0; i < num_st; ++i) {
                total += h.coarse_total[i];
                longest = std::max(longest, h.coarse_total[i]);
            }
            // the capacity the next allocations aim at follows the longest list SEEN (25 % head-room, power of
            // two): up at once, down only after 64 completed frames in a row that would fit an eighth of it
            const uint32_t want = std::max<uint32_t>(pow2_ceil((uint64_t)longest + longest / 4), 4096u);
            if (L.pending_level == ctx->sup_level) {
                if (want > ctx->coarse_cap_hint) {
                    ctx->coarse_cap_hint = want;
                    ctx->list_shrink_votes = 0;
                } else if ((uint64_t)want * 8u <= ctx->coarse_cap_hint) {
                    if (++ctx->list_shrink_votes >= 64u) { ctx->coarse_cap_hint = want * 2u; ctx->list_shrink_votes = 0; }
                } else {
                    ctx->list_shrink_votes = 0;
                }
            }
            if (longest > L.pending_coarse_cap) {  // this frame dropped entries
                if (want > ctx->coarse_cap_hint) ctx->coarse_cap_hint = want;
                rerun = true;
                ctx->reruns_lists += 1;
            }
            pending_longest = longest;
        }
        if (render && !scan && h.overflow) {
            // BINNING_SORT overflow: grow to the next power of two with 25 % headroom
            if (total > MAX_INSTANCE_CAPACITY)
                return fail(ctx, BGS_ECAPACITY,
                            "frame needs " + std::to_string(total) + " tile instances, above the 2^30 limit");
            uint64_t cap = MIN_INSTANCE_CAPACITY;
            while (cap < total + total / 4) cap <<= 1;
            cap = std::min(cap, MAX_INSTANCE_CAPACITY);
            int rc = ensure_instances(ctx, L, cap);
            if (rc != BGS_OK) return rc;
            ctx->reruns_instances += 1;
            rerun = true;
        }
        // test hook (debug flag 0x8000000): every BINNING_SCAN frame is run twice, as if a capacity had been too small —
        // exercises the re-run path (same lane, same inputs, the buffers of the first attempt) under any pipeline state
        if ((L.in_debug_flags & 0x8000000u) && attempt == 0 && render && scan) rerun = true;
        if (rerun) {
            if (attempt >= 8) return fail(ctx, BGS_ECAPACITY, "frame kept overflowing its buffers");
            ctx->regrow_count += 1;
            uint32_t* const next_target = ctx->next_srgb8_target;  // belongs to a frame not enqueued yet
            ctx->next_srgb8_target = L.in_srgb8_target;
            if (sort_gave_up) L.force_onesweep = true;  // stays for every further attempt of this frame
            ctx->rerun_onesweep = L.force_onesweep;
            // the re-run sees the output state the frame was ENQUEUED with, not whatever the setters say by now
            const bool now_srgb8 = ctx->output_srgb8, now_f16 = ctx->output_rgba16f, now_packed = ctx->packed_only;
            const uint32_t now_flags = ctx->debug_flags;
            ctx->output_srgb8 = L.in_output_srgb8;
            ctx->output_rgba16f = L.in_output_rgba16f;
            ctx->packed_only = L.in_packed_only;
            ctx->debug_flags = L.in_debug_flags;
            int rc = enqueue_frame(ctx, L, L.in_cloud, &L.in_view, &L.in_settings, render, L.in_allow_graph);
            ctx->output_srgb8 = now_srgb8;
            ctx->output_rgba16f = now_f16;
            ctx->packed_only = now_packed;
            ctx->debug_flags = now_flags;
            ctx->rerun_onesweep = false;
            ctx->next_srgb8_target = next_target;
            if (rc != BGS_OK) return rc;
            continue;
        }

        L.force_onesweep = false;
        bool level_moved = false;
        if (render && scan) {   // the completed frame's heavy-tile feedback (null after a frame that left none) is what the next dense frames read
            L.heavy_done = L.pending_heavy_out;
            L.heavy_done_grid = L.pending_tx | (L.pending_ty << 16);
            // only now does the lane's next frame write the OTHER buffer: a re-run (above) wrote the one its failed
            // attempt wrote, never the completed frame's list it was reading
            if (L.pending_heavy_out) L.heavy_parity ^= 1u;
            L.cost_done = L.pending_cost_out;
            L.cost_done_kind = L.in_kind;
            L.cost_done_grid = L.pending_tx | (L.pending_ty << 16);
            // what the cost plane behind the lane's tile order said: the share of the frame's tile work that was in tiles which
            // ended saturated (tile_order_kernel sums, this frame's clean-up block reports; x 0x7FFF)
            if (L.pending_sat_kind && h.saturated_tiles_prev != 0xFFFFFFFFu) {
                const auto kit = ctx->kinds.find(L.pending_sat_kind);
                if (kit != ctx->kinds.end()) {
                    const double share = (double)h.saturated_tiles_prev / (double)0x7FFF;
                    if (share >= bgs_ctx::MIDROUND_ON) kit->second.midround = true;
                    else if (share <= bgs_ctx::MIDROUND_OFF) kit->second.midround = false;
                }
            }
            if (L.pending_cost_out) L.cost_parity ^= 1u;
        }
        // (up at once; down only after 64 completed frames in a row at under a quarter of it: a context that
        // cycles through cameras seeing different shares of the cloud keeps one hint — and one captured graph
        // per lane)
        if (!ctx->draw_hint_valid || h.draw_count > ctx->draw_hint) {
            ctx->draw_hint = (uint32_t)std::min<uint64_t>((uint64_t)h.draw_count + h.draw_count / 8 + 1024, 0xFFFFFFFFull);
            ctx->draw_hint_valid = true;
            ctx->draw_shrink_votes = 0;
        } else if ((uint64_t)h.draw_count * 4 < ctx->draw_hint) {
            if (++ctx->draw_shrink_votes >= 64u) {
                ctx->draw_hint = (uint32_t)std::min<uint64_t>((uint64_t)h.draw_count * 2 + 1024, 0xFFFFFFFFull);
                ctx->draw_shrink_votes = 0;
            }
        } else {
            ctx->draw_shrink_votes = 0;
        }
        if (places == 4 && h.draw_count >= BUCKET_COUNT) {
            // the frame's sorted list is good: its quantile keys balance the buckets of the next frames.
            // bucket() is only monotone for an ascending table, so that is checked, not assumed
            if (splitters_ascending(h.splitters, BUCKET_COUNT * L.pending_split_sub - 1u) && L.in_cloud) {
                int slot = find_splitter_slot(ctx, L.in_cloud, &L.in_view, &L.in_settings);
                if (slot < 0) {  // a view not seen lately: take an empty slot, else the least recently used one
                    slot = 0;
                    for (int i = 0; i < bgs_ctx::SPLITTER_SLOTS; ++i) {
                        if (!ctx->split_slots[i].epoch) { slot = i; break; }
                        if (ctx->split_slots[i].last_used < ctx->split_slots[slot].last_used) slot = i;
                    }
                }
                auto& sl = ctx->split_slots[slot];
                std::memcpy(sl.table.key, h.splitters, (BUCKET_COUNT * L.pending_split_sub - 1u) * sizeof(uint32_t));   // what the clean-up wrote
                sl.table.sub = L.pending_split_sub;   // (256 * sub - 1 quantile keys: what the frame's clean-up was asked for)
                sl.cloud = L.in_cloud;
                sl.n = n;
                sl.sort_mode = L.in_settings.sort_mode;
                std::memcpy(sl.transform, L.in_settings.transform, sizeof sl.transform);
                view_pose(&L.in_view, sl.pos, sl.fwd);
                std::memcpy(sl.clip_from_view, L.in_view.clip_from_view, sizeof sl.clip_from_view);
                sl.viewport_wh[0] = L.in_view.viewport[2];
                sl.viewport_wh[1] = L.in_view.viewport[3];
                // the median key is ~bits(dist^2) of the median drawable splat (keys are 0xFFFFFFFF - bits)
                const uint32_t mid_bits = 0xFFFFFFFFu - h.splitters[BUCKET_COUNT * L.pending_split_sub / 2 - 1];
                float d2;
                std::memcpy(&d2, &mid_bits, 4);
                sl.reach = (d2 > 0.0f && d2 < 3.0e38f) ? std::sqrt(d2) : 1.0f;
                sl.epoch = ++ctx->split_epoch;
                sl.last_used = ctx->seq;
            }
        }
        // after a render only the drawable prefix of the list is materialised (the culled tail stays
        // in its side buffer); bgs_sort appends it so that callers get the reference's full list
        L.last_sorted_n = render ? h.draw_count : n;
        if (!render && h.draw_count < n) {
            // bgs_sort contract: one contiguous list, culled entries last (ascending index)
            HIP_TRY(ctx, hipMemcpyAsync(const_cast<uint2*>(L.last_sorted) + h.draw_count, L.culled,
                                        (size_t)(n - h.draw_count) * sizeof(uint2), hipMemcpyDeviceToDevice, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
        }
        if (render && scan && h.visible_count > 0) {
            // list entries per visible splat: ~1.2 when splats are smaller than a supertile, 15-20 when they
            // span many -> the supertile level of the next frames (next_supertile_level, frame_params.h; relative
            // to the level THIS frame ran at, not to ctx->sup_level, which frames completed in the meantime may
            // already have moved)
            const uint32_t lv = L.pending_level;
            double longer = 1.0;
            uint32_t target = next_supertile_level((double)total / (double)h.visible_count, lv, L.pending_edges, &longer);
            // a level whose edge equals a lower level's is that lower level (enqueue_frame canonicalises the same way):
            // moving between them is not a change — no new capacity prediction, no vote reset, no level_changes
            while (target > 1 && L.pending_edges[target - 1] == L.pending_edges[target]) --target;
            if (L.in_kind != ctx->cur_kind) {
                // a frame of ANOTHER kind than the one the context is on by now (kinds alternate while frames are in
                // flight): its verdict belongs to its own kind, not to ctx->sup_level
                const auto kit = ctx->kinds.find(L.in_kind);
                if (kit != ctx->kinds.end()) kit->second.sup_level = target;
                level_moved = target != lv;
            } else {
            if (target != lv && ctx->sup_level != target) {
                // lists of another level: predicted from THIS frame's longest list (coarser supertiles hold
                // longer lists: entries scale with the ratio, lists with the area), never from the old hint
                const double predicted = (double)pending_longest * (target > lv ? longer : 1.0) * 1.25;
                ctx->coarse_cap_hint = std::max<uint32_t>(pow2_ceil((uint64_t)std::min(predicted, 1.0e9)), 4096u);
                ctx->list_shrink_votes = 0;
            }
            if (target != lv) { if (ctx->sup_level != target) ctx->level_changes += 1; ctx->sup_level = target; level_moved = true; }
            // (a settled kind remembers its level for the next time the context comes back to it)
            const auto kit = ctx->kinds.find(L.in_kind);
            if (kit != ctx->kinds.end()) kit->second.sup_level = target;
            }
        }

        // the kind is settled: a frame of it ran with everything it needed
        if (render && scan && attempt == 0 && !level_moved && L.in_kind) {
            if (ctx->kinds.size() >= (1u << 20)) ctx->kinds.clear();   // (12 MB of kinds: a host that hashes noise into its settings)
            ctx->kinds[L.in_kind].sup_level = L.pending_level;
        }

        bgs_stats& stt = L.result;
        std::memset(&stt, 0, sizeof stt);
        stt.regrow_count = ctx->regrow_count;
        stt.splat_count = n;
        stt.visible_count = render ? h.visible_count : h.draw_count;
        stt.draw_count = h.draw_count;
        stt.sort_path = L.pending_bucket ? 1u : 0u;
        stt.list_capacity = (render && scan) ? L.pending_coarse_cap : 0u;
        stt.instance_count = render ? total : 0;
        stt.instance_capacity = L.inst_cap;
        stt.list_entries_allocated = (render && scan) ? (uint64_t)L.coarse_entries : 0;
        stt.strip_tiles = (render && scan) ? h.strip_tiles : 0u;
        stt.tile_saturation = (render && scan && L.pending_sat_kind && h.saturated_tiles_prev != 0xFFFFFFFFu)
                                  ? (0x10000u | (h.saturated_tiles_prev & 0x7FFFu) | (L.pending_midround ? 0x80000000u : 0u)) : 0u;
        stt.tiles_x = render ? L.pending_tx : 0;
        stt.tiles_y = render ? L.pending_ty : 0;
        stt.depth_passes = places;
        stt.tile_passes = (render && !scan) ? 2 : 0;
        stt.binning_mode = scan ? BINNING_SCAN : BINNING_SORT;
        {
            // SURVEY 8(d) algorithmic bytes. SURVEY's bytes_sort is N*16 + N*8 + k*N*16; the partition
            // in keygen means only the D drawable pairs go through the k passes, so that is counted
            // (the bucket sort moves each drawable pair twice: scatter + gather, sorted write: k = 1.5).
            const uint64_t N = n, k = places, D = h.draw_count;
            // (keygen reads N positions and writes the D drawable pairs; the N - D culled pairs only in frames that
            // have a reader for them)
            uint64_t bytes = N * 16 + D * 8 + (L.pending_culled_written ? (N - D) * 8 : 0) + (L.pending_bucket ? D * 24 : k * D * 16);
            if (render) {
                const uint64_t B = L.pending_cloud_format == CLOUD_F16 ? 128 : 240, R = rec_bytes, V = h.visible_count, I = total;
                const uint64_t P = (uint64_t)L.pending_w * L.pending_h;
                if (scan)  // coarse entries (rank + tile rect, 8 B): written once, read by the tiles of their supertile
                    bytes += V * (B - 16) + V * R + V * 8 + V * 8 + I * 8 + I * 8 + P * 16;   // (+ the 4-byte tile rect per rank, written by project_kernel and read by bin_kernel)
                else
                    bytes += V * (B - 16) + V * R + I * 8 + 2 * I * 16 + I * (4 + R) + P * 16;
            }
            stt.algorithmic_bytes = bytes;
        }
        L.has_result = true;
        L.result_kind = (uint8_t)(!render ? 1 : (scan ? 2 : 3));
    }
    return BGS_OK;
}

int finish_all(bgs_ctx* ctx) {
    // oldest first, so that the stats left behind are those of the most recent frame
    for (;;) {
        int best = -1;
        for (int i = 0; i < MAX_LANES; ++i)
            if (ctx->lanes[i].pending && (best < 0 || ctx->lanes[i].seq < ctx->lanes[best].seq)) best = i;
        if (best < 0) break;
        int rc = finish_lane(ctx, ctx->lanes[best]);
        if (rc != BGS_OK) return rc;
    }
    for (auto& L : ctx->lanes) L.ready = false;
    return BGS_OK;
}

// Build ctx->stats: counters of the most recent frame + per-stage times averaged over every timed
// frame (all lanes) of the same pipeline since the previous call. All lanes must be complete.
int collect_stats(bgs_ctx* ctx) {
    Lane& R = ctx->lanes[ctx->recent];
    if (!R.has_result) return fail(ctx, BGS_EINVAL, "no frame has been run yet");
    ctx->stats = R.result;
    bgs_stats& stt = ctx->stats;
    if (ctx->profiling >= 1) {
        const uint8_t kind = R.result_kind;
        const bool render = kind != 1, scan = kind == 2;
        const int last = render ? 6 : 2;
        uint32_t used = 0;
        float acc[BGS_STAGE_COUNT] = {0, 0, 0, 0, 0, 0}, acc_total = 0.0f;
        for (auto& L : ctx->lanes) {
            const uint32_t frames = std::min<uint32_t>(L.frames_timed, EV_RING);
            for (uint32_t f = 0; f < frames; ++f) {
                const uint32_t slot = (L.ev_head + EV_RING - f) % EV_RING;
                if (L.ev_kind[slot] != kind) continue;
                hipEvent_t* const ev = L.ev_ring[slot];
                auto ms = [&](int a, int b) { float t = 0; (void)hipEventElapsedTime(&t, ev[a], ev[b]); return t; };
                if (ctx->profiling >= 2) {
                    acc[BGS_STAGE_KEYGEN] += ms(0, 1);
                    acc[BGS_STAGE_DEPTH_SORT] += ms(1, 2);
                    if (render && scan) {
                        acc[BGS_STAGE_PROJECT] += ms(2, 3);
                        acc[BGS_STAGE_RASTER] += ms(3, 6);
                    } else if (render) {
                        acc[BGS_STAGE_PROJECT] += ms(2, 3);
                        acc[BGS_STAGE_TILE_SORT] += ms(3, 4);
                        acc[BGS_STAGE_RANGES] += ms(4, 5);
                        acc[BGS_STAGE_RASTER] += ms(5, 6);
                    }
                }
                acc_total += ms(0, last);
                ++used;
            }
            L.frames_timed = 0;
        }
        if (used) {
            for (int i = 0; i < BGS_STAGE_COUNT; ++i) stt.stage_ms[i] = acc[i] / (float)used;
            stt.total_ms = acc_total / (float)used;
        }
        stt.frames_averaged = used;
    }
    ctx->have_stats = true;
    return BGS_OK;
}

// Enqueue one frame on lane L. Returns without waiting; the caller decides when to finish the lane.
int enqueue_frame(bgs_ctx* ctx, Lane& L, const bgs_cloud* cloud, const bgs_view* view, const bgs_settings* s,
                  bool render, bool allow_graph) {
    FrameParams fp{};
    fill_frame_params(cloud->ptrs.n, view, s, fp);
    fp.debug = ctx->debug_flags;
    fp.srgb8_target = render ? (uint64_t)(uintptr_t)ctx->next_srgb8_target : 0;
    const uint32_t n = fp.n;
    const uint32_t places = depth_places(s);
    const bool surfel = render && fp.gaussian_mode == 0u && fp.aabb != 0u;
    const size_t rec_bytes = surfel ? sizeof(RecordSurfel) : sizeof(Record);

    int rc;
    if ((rc = lane_create(ctx, L)) != BGS_OK) return rc;
    if ((rc = ensure_entries(ctx, L, n)) != BGS_OK) return rc;
    const bool scan = ctx->binning == BINNING_SCAN;
    // what finish_lane re-runs the frame with (view / settings may already live in the lane: a re-run)
    L.in_cloud = cloud;
    if (view != &L.in_view) L.in_view = *view;
    if (s != &L.in_settings) L.in_settings = *s;
    L.in_srgb8_target = render ? ctx->next_srgb8_target : nullptr;
    L.in_allow_graph = allow_graph;
    L.in_output_srgb8 = ctx->output_srgb8;
    L.in_output_rgba16f = ctx->output_rgba16f;
    L.in_packed_only = ctx->packed_only;
    L.in_debug_flags = ctx->debug_flags;

    // Depth-sort path. The bucket sort needs 32-bit keys (shorter keys are mostly ties, which it ranks
    // quadratically), a draw count that fits its geometry (a bucket holds <= BUCKET_CAP pairs) and the key
    // range of a recent frame; it is checked on the device and the frame re-run with the digit passes when
    // it does not work out (then bucket_block keeps the following frames on the passes for a while).
    // Debug flags: 0x80000 never, 0x200000 also with a guessed range (no completed frame yet).
    const bool guess = (ctx->debug_flags & 0x200000u) != 0u;
    const int split_slot = (places == 4 && n > 0) ? find_splitter_slot(ctx, cloud, view, s) : -1;
    // Geometry: NARROW buckets (BUCKET_CAP pairs, 256-thread workgroups) while 256 * BUCKET_SUB_KERNARG of them at BUCKET_TARGET
    // pairs hold the list (1.57 M pairs: every frame of the headline's kind), WIDE ones (BUCKET_CAP_WIDE, 1024 threads, 128 KB
    // of LDS) past that: a 5 M-pair list is 768 wide buckets instead of 2816 narrow ones, which keygen's scatter reached with
    // 1.5 pairs per (tile, bucket). Debug flag 0x100: narrow whatever the length (round 5's geometry, A/B), 0x800: wide
    // whatever the length (tests).
    const uint32_t narrow_max = BUCKET_COUNT * BUCKET_SUB_KERNARG * BUCKET_TARGET;
    const bool wide_out = ((ctx->draw_hint_valid && ctx->draw_hint > narrow_max && !(ctx->debug_flags & 0x100u)) || (ctx->debug_flags & 0x800u));
    const uint32_t cap_out = wide_out ? BUCKET_CAP_WIDE : BUCKET_CAP, target_out = wide_out ? BUCKET_TARGET_WIDE : BUCKET_TARGET;
    bool bucket = places == 4 && n > 0 && !(ctx->debug_flags & 0x80000u) && ctx->bucket_block == 0 && !ctx->rerun_onesweep &&
                  ((split_slot >= 0 && ctx->draw_hint_valid) || guess) &&
                  (!ctx->draw_hint_valid || ctx->draw_hint <= BUCKET_MAX * (cap_out / 4u) * 3u);
    if (ctx->bucket_block > 0 && places == 4) ctx->bucket_block -= 1;
    // Buckets: 256 * sub, as many as keep a bucket near its target (narrow: sub = 1 up to 524 k drawable pairs — the
    // headline's 120 k —, 2 at 1 M, 3 at 1.5 M; wide: 1 up to 2.1 M, 3 at 5 M, 16 up to 50 M; debug flag 0x200: at least 3
    // whatever the length, 0x400: at least 5 — the device-table path — to exercise the finer tables on small lists). A frame
    // sorts with the table its slot HOLDS (table.sub), in the geometry the hint asks for: a table that is too coarse for
    // the list in that geometry is not used, and every completed frame leaves a table of the sub the current hint asks for
    // (split_sub_out).
    uint32_t split_sub_out = 1u;
    if (ctx->draw_hint_valid)
        split_sub_out = std::min<uint32_t>(std::max<uint32_t>((ctx->draw_hint + BUCKET_COUNT * target_out - 1u) / (BUCKET_COUNT * target_out), 1u), BUCKET_SUB_MAX);
    if (ctx->debug_flags & 0x200u) split_sub_out = std::max<uint32_t>(split_sub_out, BUCKET_SUB_KERNARG);
    if (ctx->debug_flags & 0x400u) split_sub_out = std::max<uint32_t>(split_sub_out, 5u);
    uint32_t bucket_sub = 1u;
    if (bucket && split_slot >= 0) {
        bucket_sub = std::min<uint32_t>(std::max<uint32_t>(ctx->split_slots[split_slot].table.sub, 1u), BUCKET_SUB_MAX);
        if (ctx->draw_hint_valid && ctx->draw_hint > BUCKET_COUNT * bucket_sub * (cap_out / 4u) * 3u) bucket = false;   // too coarse a table
    }
    if (bucket && bucket_sub > BUCKET_SUB_KERNARG && !L.d_split_keys) {   // the lane's device table + its pinned staging
        L.d_split_keys = dev_alloc<uint32_t>(BUCKET_MAX);
        void* hp = nullptr;
        if (!L.d_split_keys || hipHostMalloc(&hp, BUCKET_MAX * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, BGS_ENOMEM, "hipMalloc / hipHostMalloc(splitter table) failed");
        }
        L.h_split_keys = (uint32_t*)hp;
    }
    if (bucket) {
        fp.sort_path = 1u;
        if ((rc = ensure_bucket_slots(ctx, L, bucket_sub, wide_out)) != BGS_OK) return rc;
        ctx->bucket_frames += 1;
    } else if (places > 0) {
        ctx->onesweep_frames += 1;
    }
    // Supertile edge (in tiles): four levels. Level 1 is the smallest power of two >= 8 that keeps the coarse
    // bins <= 256 and <= 32 per axis (8 at 1080p: 135 bins); level 0 the smallest edge >= 3/4 of it that does
    // (6 at 1080p: 240 bins); levels 2 and 3 are 2x and 4x level 1 (16 and 32 at 1080p: 40 and 12 bins).
    // Every tile scans its supertile's whole list, so small splats want short lists (level 0: scene-like frame
    // 91.7 -> 87.9 us against level 1); a splat that spans many supertiles costs one list entry, one append and
    // a share of the ballots in each, while a tile that saturates after ~60 hits does not mind scanning three
    // times as many candidates (dense frame, 6 lanes on 3 streams: 13.3 k frames/s at level 1, 14.6 k at level
    // 2, 15.3 k at level 3). Images do not depend on the level; it follows the entries-per-visible-splat ratio
    // of the completed frames (finish_lane). Debug flags force a level: 0x10000 -> 0, 0x8000 -> 1,
    // 0x400000 -> 2, 0x800000 -> 3.
    auto bins = [&](uint32_t e, uint32_t& bx, uint32_t& by) {
        bx = ((uint32_t)fp.tiles_x + e - 1) / e;
        by = ((uint32_t)fp.tiles_y + e - 1) / e;
        return bx * by <= MAX_SUPERTILES && bx <= MAX_SUPERTILES_PER_AXIS && by <= MAX_SUPERTILES_PER_AXIS;
    };
    uint32_t edge_c = 8, cbx = 0, cby = 0, edge_f = 1, fbx = 0, fby = 0;
    while (!bins(edge_c, cbx, cby)) edge_c *= 2;
    // the fine edge stays within 3/4 of the coarse one (no flip-flop between the two rules)
    edge_f = (3 * edge_c + 3) / 4;
    while (!bins(edge_f, fbx, fby)) ++edge_f;
    if (edge_f >= edge_c) { edge_f = edge_c; fbx = cbx; fby = cby; }
    uint32_t level = ctx->sup_level;
    if (ctx->debug_flags & 0x10000u) level = 0;
    else if (ctx->debug_flags & 0x8000u) level = 1;
    else if (ctx->debug_flags & 0x400000u) level = 2;
    else if (ctx->debug_flags & 0x800000u) level = 3;
    // levels whose edges coincide (edge_c >= 16, i.e. targets of ~2048 px and up: levels 2 and 3 both clamp to 32
    // tiles) are ONE level: the frame runs, and is accounted, at the lowest level with that edge
    auto level_edge = [&](uint32_t lv) { return lv == 0 ? edge_f : std::min<uint32_t>(edge_c << (lv - 1u), 32u); };
    while (level > 1 && level_edge(level - 1u) == level_edge(level)) --level;
    // tile / edge by reciprocal multiply is exact for edges <= 32 (supertile_div)
    uint32_t sup_edge = level == 0 ? edge_f : std::min<uint32_t>(edge_c << (level - 1u), 32u), sup_bx = 0, sup_by = 0;
    if (!bins(sup_edge, sup_bx, sup_by)) { sup_edge = edge_c; sup_bx = cbx; sup_by = cby; }
    const uint32_t num_st = sup_bx * sup_by;
    uint32_t coarse_cap = 1;  // entries per supertile list
    if (render) {
        if (scan) {
            if ((rc = ensure_coarse(ctx, L, n, num_st, &coarse_cap)) != BGS_OK) return rc;
            if ((rc = ensure_rects(ctx, L)) != BGS_OK) return rc;
        } else {
            if ((rc = ensure_instances(ctx, L, std::max<uint64_t>(L.inst_cap, MIN_INSTANCE_CAPACITY))) != BGS_OK) return rc;
        }
        if ((rc = ensure_records(ctx, L, (size_t)n * rec_bytes)) != BGS_OK) return rc;
        if ((rc = ensure_framebuffer(ctx, L, (uint32_t)fp.width, (uint32_t)fp.height, ctx->output_srgb8 || ctx->output_rgba16f)) != BGS_OK) return rc;
    }
    if ((rc = ensure_scratch(ctx, L, n, L.inst_cap)) != BGS_OK) return rc;

    hipStream_t st = L.stream;
    const bool need_memset = !L.scratch_clean;  // else the previous frame's rasteriser left it zeroed
    if (need_memset) L.ctl_parity = 0;
    Control* ctl = (Control*)(L.scratch + (L.ctl_parity ? L.off_ctl1 : 0));
    uint32_t* depth_status = (uint32_t*)(L.scratch + L.off_depth_status);
    unsigned long long* scan_status = (unsigned long long*)(L.scratch + L.off_scan_status);
    uint32_t* tile_status = (uint32_t*)(L.scratch + L.off_tile_status);
    uint2* ranges = (uint2*)(L.scratch + L.off_ranges);
    uint32_t* bin_status = (uint32_t*)(L.scratch + L.off_bin_status);
    uint32_t* part_status = (uint32_t*)(L.scratch + L.off_part_status);

    const bool timed_frame = (ctx->frame_counter++ % ctx->profiling_stride) == 0;
    const int prof = timed_frame ? ctx->profiling : 0;
    const int last_mark = render ? 6 : 2;
    if (prof) {  // untimed frames do not consume a ring slot
        L.ev_head = (L.ev_head + 1) % EV_RING;
        L.ev_kind[L.ev_head] = (uint8_t)(!render ? 1 : (scan ? 2 : 3));
        L.frames_timed += 1;
    }
    hipEvent_t* const ev = L.ev_ring[L.ev_head];
    auto mark = [&](int i) {
        if (prof >= 2 || (prof == 1 && (i == 0 || i == last_mark))) (void)hipEventRecord(ev[i], st);
    };

    // ---- what will be launched -------------------------------------------------------------------
    KeygenLaunch kg{};
    kg.fp = fp;
    kg.pos = cloud->ptrs.position_visibility;
    kg.entries = L.entries[0];
    // The culled tail (index order, 8 B per culled splat: 7 of the 24 MB keygen moves on the headline frame) has two
    // readers: bgs_sort's full list and RasterizeMode::Depth (sorted[N-1] of the full list, gaussian.wgsl:331-340).
    // Every other rendered frame skips the writes; bgs_sorted_entries_device_ptr after a render has always meant the
    // drawable prefix only.
    kg.culled = (render && s->rasterize_mode != BGS_RASTERIZE_DEPTH) ? nullptr : L.culled;
    kg.ctl = ctl;
    kg.part_status = part_status;
    kg.places = places;
    kg.ticket_slot = 7;
    kg.fp_out = L.d_fp;
    kg.zero_word = nullptr;   // set below, once the frame's heavy-tile feedback buffer is known
    kg.bucket_slots = L.bucket_slots;
    kg.bucket_status = depth_status;  // the depth passes' look-back words are free in a bucket-sort frame
    L.pending_split_slot = -1;
    L.pending_split_epoch = 0;
    L.pending_split_sub = split_sub_out;
    if (bucket) {
        if (split_slot >= 0) {
            const SplitterKeys& tk = ctx->split_slots[split_slot].table;
            const uint32_t nkeys = BUCKET_COUNT * bucket_sub - 1u;
            if (bucket_sub <= BUCKET_SUB_KERNARG) std::memcpy(kg.split.key, tk.key, nkeys * sizeof(uint32_t));
            else std::memcpy(L.h_split_keys, tk.key, nkeys * sizeof(uint32_t));   // (the lane's previous frame is complete: nobody reads the staging)
            kg.split.device_keys = L.d_split_keys;
            ctx->split_slots[split_slot].last_used = ctx->seq + 1;
            L.pending_split_slot = split_slot;
            L.pending_split_epoch = ctx->split_slots[split_slot].epoch;
        } else {  // debug flag 0x200000: a guessed table (equal steps over the 32-bit range: badly balanced)
            for (uint32_t i = 0; i < BUCKET_COUNT; ++i) kg.split.key[i] = (i + 1u) << 24;
        }
        kg.split.sub = bucket_sub;
        kg.split.wide = wide_out ? 1u : 0u;
    }
    kg.wide = ctx->depth == 1;
    const bool have_keygen = kg.prepare(ctx->num_cus * 4);
    const bool large = n > (4u << 20);
    const size_t depth_tiles = ((size_t)L.scratch_n + sort_tile_size(false) - 1) / sort_tile_size(false) + 1;
    const bool hinted = ctx->draw_hint_valid && !(ctx->debug_flags & 0x2000u);
    int sort_blocks = ctx->num_cus * 4;
    if (hinted) {
        // only the D drawable entries are sorted, and D is known on the device only; launching a block
        // per N/tile would start ~6x more blocks than tiles, each queueing for a ticket just to leave
        const uint64_t want = (uint64_t)ctx->draw_hint / sort_tile_size(large) + 8;
        sort_blocks = (int)std::min<uint64_t>((uint64_t)sort_blocks, std::max<uint64_t>(want, 32));
    }
    // project grid: one block per 256 ranks of the D drawable entries when that fits the chip (two 170-190-VGPR blocks
    // are resident per CU; the kernel strides over the rest); bin grid: one block per 1024 ranks (every block then
    // takes exactly one ticket)
    int bin_blocks = ctx->num_cus * 3, binning_blocks = ctx->num_cus * 4;
    if (hinted) {
        bin_blocks = (int)std::min<uint64_t>((uint64_t)bin_blocks, std::max<uint64_t>((uint64_t)ctx->draw_hint / 256 + 8, 32));
        binning_blocks = (int)std::min<uint64_t>((uint64_t)binning_blocks, std::max<uint64_t>((uint64_t)ctx->draw_hint / 1024 + 4, 16));
    }
    const bool want_srgb8 = render && (ctx->output_srgb8 || ctx->output_rgba16f || ctx->next_srgb8_target);
    const uint32_t out_format = !want_srgb8 ? 0u : ((ctx->output_rgba16f ? OUT_RGBA16F : OUT_SRGB8) |
                                                    ((ctx->packed_only && scan) ? OUT_SKIP_F32 : 0u));
    uint2* const draw_list = L.entries[places & 1u];  // the passes ping-pong from entries[0]
    // SortMode::Rayon / Std sort ascending on the inverted key; the last step of either path un-inverts it
    const uint32_t final_xor = (s->sort_mode == BGS_SORT_RAYON || s->sort_mode == BGS_SORT_STD) ? 0xFFFFFFFFu : 0u;
    FrameCleanup cl{};
    if (render && scan) {
        cl.part_status = part_status;
        cl.depth_status = depth_status;
        cl.bin_status = bin_status;
        cl.other_ctl = (Control*)(L.scratch + (L.ctl_parity ? 0 : L.off_ctl1));
        cl.host_ctl = L.h_ctl_dev;
        cl.pass_stride = (uint32_t)(depth_tiles * RADIX_BASE);
        cl.places = bucket ? 0u : places;
        cl.depth_tile = sort_tile_size(large);
        cl.sorted = draw_list;
        cl.key_xor = final_xor;
        cl.split_sub = split_sub_out;
        if (ctx->debug_flags & 0x1000u) cl = FrameCleanup{};  // experiment: classic memset + copy path
    }
    const bool raster_cleans = render && scan && fp.tiles_x > 0 && fp.tiles_y > 0 && cl.other_ctl != nullptr;
    // Dense frames (supertile level >= 2: the rasteriser's mid-round-exit instantiation) leave, and use, the heavy-tile
    // feedback (kernels.h HeavyFeedback) — when ONE frame is in flight (pipeline depth 1): the strip workgroups cut the
    // launch's tail (dense 1 M frame: raster 49.4 -> 45.3 us, the heaviest tiles' serial chains split four ways), but
    // with several frames in flight that tail is filled by the other lanes' kernels anyway and the extra workgroups
    // and the flag load only cost (20.6 -> 20.0 k frames/s with 8 lanes; profiles/r3_notes.md). Not under frame graphs
    // (the consumed buffer changes with every completed frame), not with the tile trace, not for the surfel variant.
    // Debug flag 0x2000000 switches it off, 0x4000000 forces it on at any depth (A/B).
    // (Sample2 / Sample8 and the bounding-box overlay have no mid-round-exit instantiation: launch_raster_scan)
    // ... and frames of a kind whose saturating tiles hold a good share of the work (KindState::midround, from the cost planes)
    // when several frames are in flight: the trained-like 1 M frame 5.75 -> 6.31 k frames/s with 8 lanes, but ALONE on the chip
    // its launch ends with its longest lists' serial chains, which do not saturate and only pay the checks (231 -> 272 us)
    bool kind_midround = false;
    if (ctx->depth > 1) { const auto kit = ctx->kinds.find(L.in_kind); if (kit != ctx->kinds.end()) kind_midround = kit->second.midround; }
    const bool midround_exit = (level >= 2u || kind_midround || (ctx->debug_flags & 0x20000u)) && !(ctx->debug_flags & 0x1000000u) && (fp.sample_count == 1u || fp.sample_count == 4u) &&   // (0x20000: at any level, A/B)
                               fp.visualize_bbox == 0u;
    const uint32_t ntiles_frame = (uint32_t)(fp.tiles_x * fp.tiles_y);
    const bool heavy_ok = render && scan && raster_cleans && midround_exit && level >= 2u && !surfel && !ctx->tile_trace &&
                          (ctx->depth == 1 || (ctx->debug_flags & 0x4000000u)) &&
                          !(ctx->debug_flags & 0x2000000u) && !(allow_graph && ctx->use_graphs) && ntiles_frame <= 65535u;
    uint8_t* heavy_out = nullptr;
    const uint8_t* heavy_in = nullptr;
    if (heavy_ok) {
        if ((rc = ensure_heavy(ctx, L, ntiles_frame)) != BGS_OK) return rc;
        heavy_out = L.heavy[L.heavy_parity];
        if (L.heavy_done && L.heavy_done != heavy_out && L.heavy_done_grid == ((uint32_t)fp.tiles_x | ((uint32_t)fp.tiles_y << 16))) heavy_in = L.heavy_done;
        kg.zero_word = reinterpret_cast<uint32_t*>(heavy_out);   // keygen, the frame's first kernel, zeroes the list's count
    }

    // Tile costs (kernels.h TileCost): every BINNING_SCAN frame leaves them, and a frame with more tile waves than the
    // chip holds at once draws its raster workgroups in the order made of a completed frame's costs. Not under frame
    // graphs (the buffers alternate with every completed frame); the tile trace shows it. Unlike the heavy-tile strips it
    // pays with frames in flight too, if little (+0.6 % dense, +0.9 % surfel frames/s; alone on the chip 6-21 % of the
    // rasteriser's time). Debug flag 0x10000000 switches it off, 0x20000000 off for pipeline depths > 1, 0x40000000
    // makes the order anew with every frame.
    const bool cost_ok = render && scan && raster_cleans &&
                         (ctx->depth == 1 || !(ctx->debug_flags & 0x20000000u)) && !(ctx->debug_flags & 0x10000000u) &&
                         !(allow_graph && ctx->use_graphs) && ntiles_frame <= 65535u;
    uint16_t* cost_out = nullptr;
    const uint16_t* cost_in = nullptr;
    uint16_t* tile_order = nullptr;
    if (cost_ok) {
        if ((rc = ensure_cost(ctx, L, ntiles_frame)) != BGS_OK) return rc;
        cost_out = L.cost[L.cost_parity];
        const uint32_t grid_now = (uint32_t)fp.tiles_x | ((uint32_t)fp.tiles_y << 16);
        if (ntiles_frame > (uint32_t)(ctx->num_cus * 4 * raster_scan_waves_per_simd(fp))) {
            // L.order holds a permutation of this grid's workgroups from the moment it was first made for the grid
            // (order_grid); it is made again from the newest completed costs every TILE_ORDER_REFRESH-th frame
            const bool have_costs = L.cost_done && L.cost_done != cost_out && L.cost_done_grid == grid_now;
            const bool have_order = L.order_grid == grid_now;
            if (have_costs && (!have_order || L.order_age + 1u >= TILE_ORDER_REFRESH || (ctx->debug_flags & 0x40000000u)))
                cost_in = L.cost_done;
            if (cost_in || have_order) tile_order = L.order;
        }
    }
    if (cost_out) ctx->cost_frames += 1;
    if (tile_order) ctx->ordered_frames += 1;
    if (cost_in) ctx->order_refreshes += 1;
    if (cost_in) L.order_kind = L.cost_done_kind;   // (this frame makes the order — and its saturation counts — anew, ahead of its own kernels)
    L.pending_sat_kind = 0;
    L.pending_midround = midround_exit;
    if (tile_order && cl.other_ctl) {
        cl.order_stats = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(tile_order) + tile_order_stats_offset(ntiles_frame));
        L.pending_sat_kind = L.order_kind;
    }

    // the launches of one frame, in stream order (issued directly, or once into a stream capture)
    auto issue = [&]() -> hipError_t {
        mark(0);
        if (cost_in) launch_tile_order(st, cost_in, tile_order, ntiles_frame, fp, midround_exit);   // (counted with the frame's first stage)
        if (have_keygen) {
            if (bucket && bucket_sub > BUCKET_SUB_KERNARG) {
                const hipError_t ce = hipMemcpyAsync(L.d_split_keys, L.h_split_keys, (BUCKET_COUNT * bucket_sub - 1u) * sizeof(uint32_t),
                                                     hipMemcpyHostToDevice, st);
                if (ce != hipSuccess) return ce;
            }
            hipError_t e = kg.launch(st);
            if (e != hipSuccess) return e;
        }
        mark(1);
        int cur = 0;
        if (bucket) {
            const hipError_t e = launch_bucket_sort(st, L.bucket_slots, draw_list, ctl, final_xor, BUCKET_COUNT * bucket_sub, wide_out);
            if (e != hipSuccess) return e;
        }
        for (uint32_t p = 0; p < (bucket ? 0u : places); ++p) {
            const uint32_t key_xor =
                (p + 1 == places && (s->sort_mode == BGS_SORT_RAYON || s->sort_mode == BGS_SORT_STD)) ? 0xFFFFFFFFu : 0u;
            // only the V' drawable entries are sorted; the culled tail is already in its final order
            launch_onesweep_pass(st, L.entries[cur], L.entries[cur ^ 1], &ctl->draw_count, n, ctl->hist_depth[p],
                                 depth_status + (size_t)p * depth_tiles * RADIX_BASE, &ctl->ticket[p][0], &ctl->error,
                                 p * RADIX_BITS, key_xor, large, sort_blocks);
            cur ^= 1;
        }
        mark(2);
        if (render && scan) {
            launch_project_bin(st, fp, L.d_fp, cloud->ptrs, draw_list, L.culled, ctl, bin_status, L.records, L.rects, L.coarse,
                               coarse_cap, sup_edge, /*ticket_slot=*/4, bin_blocks, binning_blocks, /*wide_bin=*/ctx->depth == 1);
            mark(3);
            launch_raster_scan(st, fp, L.d_fp, L.records, L.coarse, coarse_cap, sup_edge, ctl, L.fb, L.fb8,
                               (ctx->debug_flags & 0x40000u) ? 0u : out_format, cl, ctx->tile_trace,
                               midround_exit ? (level >= 2u ? 1 : 2) : 0, heavy_in, heavy_out, tile_order, cost_out);
            mark(6);
        } else if (render) {
            const uint32_t capacity = (uint32_t)std::min<uint64_t>(L.inst_cap, MAX_INSTANCE_CAPACITY);
            launch_project_emit(st, fp, cloud->ptrs, draw_list, L.culled, ctl, scan_status, L.records, L.inst[0], capacity,
                                /*ticket_slot=*/4, ctx->num_cus * 3);
            mark(3);
            const size_t inst_tiles = (L.scratch_inst_cap + sort_tile_size(true) - 1) / sort_tile_size(true) + 1;
            for (uint32_t p = 0; p < 2; ++p)
                launch_onesweep_pass(st, L.inst[p], L.inst[p ^ 1], &ctl->instance_count, capacity, ctl->hist_tile[p],
                                     tile_status + (size_t)p * inst_tiles * RADIX_BASE, &ctl->ticket[5 + p][0],
                                     &ctl->error, p * RADIX_BITS, 0u, true, ctx->num_cus * 4);
            mark(4);
            launch_tile_ranges(st, L.inst[0], ctl, ranges);
            mark(5);
            launch_raster(st, fp, L.records, L.inst[0], ranges, L.fb, view->clear_color, ctl);
            mark(6);
        }
        // BINNING_SCAN frames get their sRGB8 image from the rasteriser itself (debug flag 0x40000: from the
        // separate encode pass, for A/B runs)
        if (want_srgb8 && !(render && scan && !(ctx->debug_flags & 0x40000u)))
            launch_encode_srgb8(st, L.fb, L.fb8, (uint32_t)fp.width * (uint32_t)fp.height, L.d_fp, out_format);
        return hipGetLastError();
    };

    // ---- opt-in (bgs_set_graphs): a steady-state BINNING_SCAN frame as a hipGraph, captured once per
    // (lane, Control parity), then replayed with ONE node update — keygen's arguments carry the new
    // FrameParams, every other kernel reads them from the copy keygen leaves in device memory.
    // Measured: 7 launches cost 19 us of host time (30 us with stage events), a replay 10 us; on the
    // GPU a replayed frame is ~5 % SLOWER than the same launches issued directly (178 vs 171 us per
    // frame back to back on one stream), so it is for hosts that cannot spare the CPU time.
    const bool use_graph = allow_graph && ctx->use_graphs && render && scan && raster_cleans && !need_memset && bucket_sub <= BUCKET_SUB_KERNARG &&
                           prof == 0 && have_keygen && !(ctx->debug_flags & 0x4000u) && !ctx->tile_trace;
    if (use_graph) {
        GraphKey key;
        std::memset(&key, 0, sizeof key);
        const void* planes[6] = {cloud->ptrs.position_visibility, cloud->ptrs.packed, nullptr, nullptr, nullptr, nullptr};
        std::memcpy(key.cloud, planes, sizeof planes);
        const void* bufs[11] = {L.entries[0], L.entries[1], L.culled, L.records, L.coarse, L.fb, L.fb8, L.scratch, L.d_fp,
                                L.h_ctl_dev, L.bucket_slots};   // (L.rects lives and dies with L.entries)
        std::memcpy(key.bufs, bufs, sizeof bufs);
        key.n = n;
        key.format = cloud->ptrs.format;
        key.places = places;
        key.sort_mode = s->sort_mode;
        key.gaussian_mode = fp.gaussian_mode;
        key.aabb = fp.aabb;
        key.any_mode = (fp.rasterize_mode != RASTERIZE_COLOR || fp.draw_mode != 0u) ? 1u : 0u;
        key.srgb8 = out_format;
        key.debug_flags = ctx->debug_flags;
        key.width = fp.width;
        key.height = fp.height;
        key.sort_blocks = bucket ? 0 : sort_blocks;  // the bucket sort's grid is fixed
        key.bin_blocks = bin_blocks * 4096 + binning_blocks;   // (the bin kernel's shape follows ctx->depth, as keygen's does: key.keygen_threads)
        key.keygen_blocks = (int32_t)kg.blocks;
        key.keygen_func = kg.func;
        key.keygen_threads = kg.threads;
        key.wide_bin = ctx->depth == 1 ? 1u : 0u;
        key.raster_variant = fp.sample_count | (fp.depth_ptr ? 0x100u : 0u) | (fp.visualize_bbox ? 0x200u : 0u) | (midround_exit ? (level >= 2u ? 0x400u : 0x800u) : 0u);
        key.split_sub = split_sub_out;   // (round 5's advisor: a replay across a 524 k-pair step of the hint left a table of the captured sub under the new sub's label)
        key.sup_edge = sup_edge;
        key.scratch_bytes = L.scratch_bytes;
        key.scratch_inst_cap = L.scratch_inst_cap;
        key.scratch_n = L.scratch_n;
        key.coarse_cap = coarse_cap;
        key.sort_path = bucket ? (bucket_sub | (wide_out ? 0x100u : 0u)) : 0u;   // (the bucket sort's grid and instantiation and keygen's dynamic LDS follow it)
        FrameGraph& G = L.graph[L.ctl_parity];
        if (G.exec && std::memcmp(&G.key, &key, sizeof key) == 0) {
            HIP_TRY(ctx, kg.update_node(G.exec, G.keygen_node));
            ctx->graph_replays += 1;
        } else {
            graph_destroy(G);
            HIP_TRY(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const hipError_t ie = issue();
            const hipError_t ce = hipStreamEndCapture(st, &G.graph);
            size_t roots = 1;
            if (ie != hipSuccess || ce != hipSuccess || !G.graph ||
                hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0) != hipSuccess ||
                hipGraphGetRootNodes(G.graph, &G.keygen_node, &roots) != hipSuccess || roots != 1) {
                graph_destroy(G);
                (void)hipGetLastError();
                return fail(ctx, BGS_EHIP, "capturing the frame into a hipGraph failed");
            }
            G.key = key;
            ctx->graph_captures += 1;
        }
        HIP_TRY(ctx, hipGraphLaunch(G.exec, st));
    } else {
        if (need_memset) HIP_TRY(ctx, hipMemsetAsync(L.scratch, 0, L.scratch_bytes, st));
        // no keygen (empty cloud): the kernels behind it still read the frame's parameters
        if (!have_keygen) HIP_TRY(ctx, hipMemcpyAsync(L.d_fp, &fp, sizeof fp, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, issue());
    }
    L.scratch_clean = false;
    L.last_sorted = draw_list;
    L.last_sorted_n = n;
    L.fb8_valid = false;
    L.fb_valid = !(out_format & OUT_SKIP_F32) || (ctx->debug_flags & 0x40000u);
    L.fb8_is_f16 = (out_format & OUT_RGBA16F) != 0u;
    if (want_srgb8) {
        L.fb8_out = ctx->next_srgb8_target ? ctx->next_srgb8_target : L.fb8;
        L.fb8_valid = true;
    }
    ctx->next_srgb8_target = nullptr;
    // the Control block travels back with the frame; it is looked at when the lane is completed.
    // A BINNING_SCAN frame's rasteriser has already written the counters to L.h_ctl and left the
    // scratch region zeroed for the next frame.
    if (raster_cleans) { L.scratch_clean = true; L.ctl_parity ^= 1u; }
    else {
        if (places == 4 && n > 0) launch_splitters(st, draw_list, ctl, final_xor, split_sub_out);
        HIP_TRY(ctx, hipMemcpyAsync(L.h_ctl, ctl, sizeof(Control), hipMemcpyDeviceToHost, st));
    }

    HIP_TRY(ctx, hipEventRecord(L.done, st));
    L.pending = true;
    L.pending_render = render;
    L.pending_scan = scan;
    L.pending_bucket = bucket;
    L.pending_culled_written = kg.culled != nullptr;
    L.pending_heavy_out = heavy_out;
    L.pending_cost_out = cost_out;
    if (cost_in) { L.order_grid = (uint32_t)fp.tiles_x | ((uint32_t)fp.tiles_y << 16); L.order_age = 0; }
    else if (tile_order) L.order_age += 1u;
    L.pending_coarse_cap = coarse_cap;
    L.pending_level = level;
    L.pending_edges[0] = edge_f;
    for (uint32_t k = 1; k < 4; ++k) L.pending_edges[k] = std::min<uint32_t>(edge_c << (k - 1u), 32u);
    L.pending_n = n;
    L.pending_places = places;
    L.pending_num_st = num_st;
    L.pending_rec_bytes = (uint32_t)rec_bytes;
    L.pending_cloud_format = cloud->ptrs.format;
    L.pending_w = (uint32_t)fp.width;
    L.pending_h = (uint32_t)fp.height;
    L.pending_tx = (uint32_t)fp.tiles_x;
    L.pending_ty = (uint32_t)fp.tiles_y;
    L.seq = ++ctx->seq;
    return BGS_OK;
}

int run(bgs_ctx* ctx, const bgs_cloud* cloud, const bgs_view* view, const bgs_settings* s, bool render) {
    int rc = validate(ctx, cloud, view, s, render);
    if (rc != BGS_OK) return rc;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    const bool will_be_async = ctx->async_frames && render && ctx->binning == BINNING_SCAN;
    if (!will_be_async) {
        // a blocking call: complete whatever is queued first (surfaces its watchdog state), use lane 0
        if ((rc = finish_all(ctx)) != BGS_OK) return rc;
        Lane& L = ctx->lanes[0];
        ctx->recent = 0;
        ctx->regrow_count = 0;
        L.in_kind = render ? frame_kind(cloud, view, s) : 0ull;
        if (render) switch_kind(ctx, L.in_kind);
        if ((rc = enqueue_frame(ctx, L, cloud, view, s, render, false)) != BGS_OK) return rc;
        return finish_lane(ctx, L);  // re-runs the frame itself if a capacity was too small
    }
    // async frame: next lane of the ring; completing its previous occupant first
    Lane& L = ctx->lanes[ctx->next];
    if (L.pending && (rc = finish_lane(ctx, L)) != BGS_OK) return rc;
    L.ready = false;
    const uint64_t kind = frame_kind(cloud, view, s);
    const bool learn = ctx->kinds.find(kind) == ctx->kinds.end();
    L.in_kind = kind;
    switch_kind(ctx, kind);
    if ((rc = enqueue_frame(ctx, L, cloud, view, s, render, /*allow_graph=*/true)) != BGS_OK) return rc;
    ctx->recent = ctx->next;
    ctx->next = (ctx->next + 1) % ctx->depth;
    if (learn) {
        // complete it now (re-running it if a first guess was too small): the frames behind it start from what it learnt.
        // It stays in the ring for bgs_pipeline_pop like any other frame.
        if ((rc = finish_lane(ctx, L)) != BGS_OK) return rc;
        L.ready = true;
        ctx->early_frames += 1;
        // bounded: a kind whose frames never run clean (every frame re-run, a level that oscillates) is settled on anyway
        if (ctx->kinds.find(kind) != ctx->kinds.end()) {
            ctx->learning.erase(kind);   // (it ran clean: finish_lane settled it)
        } else if (++ctx->learning[kind] >= bgs_ctx::LEARN_MAX) {
            ctx->kinds[kind].sup_level = ctx->sup_level;
            ctx->learning.erase(kind);
        }
        if (ctx->learning.size() >= (1u << 16)) ctx->learning.clear();   // (a host that hashes noise into its settings)
    }
    return BGS_OK;
}

}  // namespace bgs_host
