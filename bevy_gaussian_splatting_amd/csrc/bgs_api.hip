// bgs_api.hip — libbgs host side: context, frame lanes, device buffers, orchestration, C ABI.
//
// One frame (bgs_render, default BGS_BINNING_SCAN) is 4 kernel launches in the steady state (7 on a frame
// without a splitter table) on ONE lane's stream with no host round trip in between; every size that
// depends on the data (drawable count V', list lengths) stays on the device in the lane's Control block
// and is consumed by persistent, ticket-driven kernels:
//
//   [memset]                      only when the scratch region is not known to be clean
//   keygen                        N x 16 B read; stable partition; drawable pairs -> key-range buckets
//                                 (bucket path) or -> an index-ordered list + digit histograms (onesweep path)
//   bucket_sort | onesweep x places   the V' drawable depth keys: one launch, or one per digit place
//   project_bin                   V' splats -> records (front-to-back) + ordered coarse lists of (rank, tile rect)
//   raster_scan                   one wave per 16x16 tile, lazy binning, saturation exit; writes the f32 target
//                                 and / or the packed image, zeroes the scratch the frame used, reports the
//                                 frame's counters and quantile keys to pinned host memory
//
// ADAPTIVE STATE. What a completed frame teaches the context sizes the next ones: the draw count (grid
// sizes), the sorted list's quantile keys (bucket splitters, kept per view slot), the longest supertile
// list (list capacity) and the entries-per-splat ratio (supertile level). Each is only a hint: kernels
// guard every write and count true totals, finish_lane compares them with what the frame ran with and
// RE-RUNS the frame on its lane when a capacity was too small — before anyone has seen its output.
//
// FRAME LANES (bgs_set_pipeline_depth): a single stream of these latency-bound kernels leaves most
// of the chip idle (1.9k waves of projection work, a 256-workgroup sort), so the context owns K
// lanes, each with its own per-frame buffers, multiplexed onto a few streams; async frames go round-robin
// over the lanes and the GPU overlaps one frame's sort with another frame's rasteriser (measured 2.0x with
// 6 lanes on 3 streams). A lane is completed (event wait + watchdog check + capacity check) when it is
// reused, popped, or on any blocking call.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/bgs.h"
#include "bgs_device.h"
#include "frame_params.h"
#include "kernels.h"

using namespace bgs;

struct bgs_cloud {
    CloudPtrs ptrs{};
    void* allocs[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t bytes = 0;
};

namespace {

thread_local std::string g_error;

constexpr uint64_t MIN_INSTANCE_CAPACITY = 1ull << 22;  // 4M instances (32 MB per buffer)
constexpr uint64_t MAX_INSTANCE_CAPACITY = 1ull << 30;  // look-back words carry 30-bit values
constexpr uint32_t MAX_SPLATS = (1u << 30) - 1u;
constexpr int EV_COUNT = BGS_STAGE_COUNT + 1;
constexpr int EV_RING = 64;   // per-stage timings are averaged over up to this many frames per lane
constexpr int MAX_LANES = 8;

template <class T>
T* dev_alloc(size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return nullptr;
    return (T*)p;
}

// What the launches of a captured frame depend on besides FrameParams (which one node carries, see
// KeygenLaunch): compared bytewise, any difference rebuilds the graph.
struct GraphKey {
    const void* cloud[6];
    const void* bufs[11];
    uint32_t n, format, places, sort_mode, gaussian_mode, aabb, any_mode, srgb8, debug_flags;
    int32_t width, height;
    int32_t sort_blocks, bin_blocks, keygen_blocks;
    uint32_t sup_edge;
    // the offsets baked into the nodes depend on the scratch layout and the list capacity, not only on
    // the base pointers (a re-allocation may return the same address)
    uint64_t scratch_bytes, scratch_inst_cap;
    uint32_t scratch_n, coarse_cap, sort_path;
};
struct FrameGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t keygen_node = nullptr;
    GraphKey key{};
};

// Everything one in-flight frame owns.
struct Lane {
    hipStream_t stream = nullptr;  // owned by the context; lanes may share one (bgs_set_pipeline_streams)
    hipEvent_t done = nullptr;     // recorded behind the lane's frame: what completing the lane waits for
    FrameParams* d_fp = nullptr;  // the frame's FrameParams as the kernels behind keygen read them
    FrameGraph graph[2];          // the captured BINNING_SCAN frame, one per Control parity

    // zeroed-every-frame scratch: [Control | depth status | scan status | tile status | ranges |
    //                              bin status | partition status]
    uint8_t* scratch = nullptr;
    size_t scratch_bytes = 0;
    size_t off_depth_status = 0, off_scan_status = 0, off_tile_status = 0, off_ranges = 0, off_bin_status = 0,
           off_part_status = 0, off_ctl1 = 0;
    uint32_t ctl_parity = 0;  // which of the lane's two Control blocks the next frame uses
    // true while the scratch region is known to be all zero without a memset: the rasteriser of a
    // BINNING_SCAN frame zeroes what the frame used (FrameCleanup, kernels.h)
    bool scratch_clean = false;
    uint32_t scratch_n = 0;         // splat capacity the scratch was laid out for
    uint64_t scratch_inst_cap = 0;  // instance capacity the scratch was laid out for

    uint2* entries[2] = {nullptr, nullptr};
    uint2* culled = nullptr;  // entries with the culled sentinel key, in index order
    uint32_t entries_cap = 0;
    void* records = nullptr;
    size_t records_bytes = 0;
    uint2* inst[2] = {nullptr, nullptr};  // BINNING_SORT only
    uint64_t inst_cap = 0;
    uint32_t* coarse = nullptr;  // BINNING_SCAN: [num_supertiles][coarse_cap] ordered (rank, tile rect) lists
    size_t coarse_entries = 0;   // 8-byte entries allocated (all lists together)
    uint2* bucket_slots = nullptr;  // bucket sort: [BUCKET_COUNT][BUCKET_CAP] pairs (64 MB), allocated on first use
    float4* fb = nullptr;
    size_t fb_pixels = 0;
    uint32_t* fb8 = nullptr;     // Rgba8UnormSrgb image (optional)
    uint32_t* fb8_out = nullptr; // where the last frame's sRGB8 image went (fb8 or a caller's target)
    size_t fb8_pixels = 0;
    uint32_t fb_w = 0, fb_h = 0;
    bool fb8_valid = false, fb8_is_f16 = false;
    bool fb_valid = true;  // false after a packed-only frame (the f32 target was not written)

    Control* h_ctl = nullptr;  // pinned; filled by a copy enqueued with the frame, or by the rasteriser
    Control* h_ctl_dev = nullptr;  // the same memory as the device sees it
    hipEvent_t ev_ring[EV_RING][EV_COUNT] = {};
    uint8_t ev_kind[EV_RING] = {};  // 0 unused, 1 sort-only frame, 2 render/scan, 3 render/sort-binning
    uint32_t ev_head = 0;
    uint32_t frames_timed = 0;  // timed frames since the last stats read-back

    bool pending = false;  // a frame is enqueued whose Control block has not been checked yet
    bool pending_render = false, pending_scan = false, pending_bucket = false;
    uint32_t pending_coarse_cap = 0, pending_level = 1, pending_edges[4] = {6, 8, 16, 32};
    bool force_onesweep = false;       // the pending frame is a re-run of one whose bucket sort gave up
    int pending_split_slot = -1;       // splitter slot the pending bucket-sort frame used (-1: a guessed table)
    uint64_t pending_split_epoch = 0;
    // what the pending frame was enqueued with: a frame whose data-dependent capacities turn out too
    // small (coarse lists, bucket sort, tile instances) is re-run on its lane when it is completed
    const bgs_cloud* in_cloud = nullptr;
    bgs_view in_view{};
    bgs_settings in_settings{};
    uint32_t* in_srgb8_target = nullptr;
    bool in_allow_graph = false;
    uint32_t pending_n = 0, pending_places = 0, pending_num_st = 0, pending_rec_bytes = 0, pending_cloud_format = 0;
    uint32_t pending_w = 0, pending_h = 0, pending_tx = 0, pending_ty = 0;
    uint64_t seq = 0;  // enqueue sequence number (to find the oldest pending lane)

    const uint2* last_sorted = nullptr;
    uint32_t last_sorted_n = 0;

    bgs_stats result{};      // counters of the last completed frame of this lane (no timings)
    bool has_result = false;
    uint8_t result_kind = 0; // ev_kind of that frame
};

}  // namespace

struct bgs_ctx {
    int device = 0;
    int num_cus = 256;
    std::string error;

    Lane lanes[MAX_LANES];
    hipStream_t streams[MAX_LANES] = {};
    hipStream_t queue_holders[3] = {};  // idle streams that make the runtime spread ours over its hardware queues (assign_streams)
    int num_streams = 4;  // streams the lanes are multiplexed onto, 0 = one per lane (one per hardware queue: include/bgs.h)
    int depth = 1;    // lanes in use
    int next = 0;     // lane the next frame goes to
    int recent = 0;   // lane of the most recently enqueued frame
    uint64_t seq = 0;

    uint32_t binning = BINNING_SCAN;
    uint32_t debug_flags = 0;
    int profiling = 2;              // 0 = no events, 1 = frame start/end only, 2 = every stage
    uint32_t profiling_stride = 1;  // record events only on every Nth frame
    uint32_t frame_counter = 0;
    bool async_frames = false;
    bool output_srgb8 = false;
    bool output_rgba16f = false;   // the packed image is Rgba16Float (8 B per pixel) instead of Rgba8UnormSrgb
    bool packed_only = false;      // frames with a packed image do not write the f32 target
    uint32_t* next_srgb8_target = nullptr;  // bgs_set_srgb8_target: one-shot destination of the next frame

    // Sizes the grids of the next frames' sort and projection launches: the draw_count of a completed
    // frame plus head-room, raised at once and lowered only after 64 frames at under a quarter of it, so that a
    // captured frame graph (whose grids are frozen) survives a moving camera. A hint only — the
    // kernels read the real count on the device and loop over tickets if the grid is short.
    uint32_t draw_hint = 0;
    bool draw_hint_valid = false;
    uint32_t draw_shrink_votes = 0;
    uint32_t sup_level = 1;  // supertile edge level of the next frames (see enqueue_frame)
    // Bucket sort (one launch instead of four digit passes) is used while a completed frame's quantile keys
    // are known, the draw count fits the bucket geometry, and it has not just failed.
    // Splitter tables: the quantile keys of completed frames' sorted lists, kept per "view slot" — a context that
    // alternates between cameras (the reference's multi_camera example), clouds or model transforms would
    // otherwise hand every frame the table of the wrong view. A table is used for a frame of the same cloud and
    // transform whose camera is near the pose it was measured at; it is dropped when a frame it served overflows.
    struct SplitterSlot {
        SplitterTable table{};
        const bgs_cloud* cloud = nullptr;
        uint32_t n = 0;
        uint32_t sort_mode = 0;   // Radix culls (a frustum's worth of keys), Rayon / Std keep every splat
        float transform[16] = {};
        float pos[3] = {}, fwd[3] = {};
        float reach = 0.0f;       // median view distance of the list the table came from (scale of "near")
        uint64_t epoch = 0;       // 0 = empty
        uint64_t last_used = 0;
    };
    static constexpr int SPLITTER_SLOTS = 8;
    SplitterSlot split_slots[SPLITTER_SLOTS];
    uint64_t split_epoch = 0;         // epochs handed out so far
    uint64_t split_failed_epoch = 0;  // newest epoch whose table overflowed (escalation looks at newer ones only)
    uint32_t bucket_block = 0;        // frames to stay on the onesweep passes after a bucket-sort overflow
    uint32_t bucket_fail_streak = 0;  // tables in a row that overflowed on their first use
    uint32_t list_shrink_votes = 0;   // completed frames in a row whose lists would fit a much smaller capacity
    bool rerun_onesweep = false;      // set while finish_lane re-enqueues a frame whose bucket sort gave up
    uint64_t bucket_frames = 0, onesweep_frames = 0;  // frames enqueued on either sort path (incl. re-runs)
    uint64_t reruns_sort = 0, reruns_lists = 0, reruns_instances = 0, level_changes = 0;
    // entries per supertile list the next frames allocate (grown from the longest list seen; a frame whose
    // lists overflow is re-run): the worst case is n entries in each of up to 256 lists (1.9 GB per lane at
    // 1 M splats), the real lists of a frame hold ~1 % of that
    uint32_t coarse_cap_hint = 0;
    bool use_graphs = false;  // async BINNING_SCAN frames replay a captured hipGraph (bgs_set_graphs)
    uint64_t graph_captures = 0, graph_replays = 0;

    bool have_stats = false;
    bgs_stats stats{};
    uint32_t regrow_count = 0;
};

namespace {

int fail(bgs_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->error = msg;
    g_error = msg;
    return status;
}

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ctx, BGS_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

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
            // BGS_QUEUE_HOLDERS=0 in the environment switches this off: a process whose other streams already hold
            // the queues (RCCL's, after a process group was initialised: bench.py's gather path sets it) is better
            // off without three more co-tenants on them (that path: 18.0 k frames/s without, 13.6 k with).
            const char* qh_env = std::getenv("BGS_QUEUE_HOLDERS");
            const bool park = !(qh_env && qh_env[0] == '0');
            if (park && !ctx->queue_holders[0])
                for (auto& qh : ctx->queue_holders) HIP_TRY(ctx, hipStreamCreateWithFlags(&qh, hipStreamNonBlocking));
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
    if (L.coarse) (void)hipFree(L.coarse);
    if (L.bucket_slots) (void)hipFree(L.bucket_slots);
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
    for (auto& e : L.entries) {
        e = dev_alloc<uint2>(n);
        if (!e) return fail(ctx, BGS_ENOMEM, "hipMalloc(sort entries) failed");
    }
    L.culled = dev_alloc<uint2>(n);
    if (!L.culled) return fail(ctx, BGS_ENOMEM, "hipMalloc(culled entries) failed");
    L.entries_cap = n;
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

int ensure_bucket_slots(bgs_ctx* ctx, Lane& L) {
    if (L.bucket_slots) return BGS_OK;
    L.bucket_slots = dev_alloc<uint2>((size_t)BUCKET_COUNT * BUCKET_CAP);
    if (!L.bucket_slots) return fail(ctx, BGS_ENOMEM, "hipMalloc(bucket sort slots) failed");
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
        const float dx = pos[0] - sl.pos[0], dy = pos[1] - sl.pos[1], dz = pos[2] - sl.pos[2];
        const float d = std::sqrt(dx * dx + dy * dy + dz * dz);
        const float c = fwd[0] * sl.fwd[0] + fwd[1] * sl.fwd[1] + fwd[2] * sl.fwd[2];
        if (!(d <= 0.05f * sl.reach) || !(c >= 0.9848f)) continue;
        const float score = d / std::max(sl.reach, 1e-30f) + (1.0f - c);
        if (best < 0 || score < best_d) { best = i; best_d = score; }
    }
    return best;
}

// Complete the frame pending on a lane: wait for it, check the watchdog word of the Control copy that
// travelled with the frame, and RE-RUN the frame on its lane if a data-dependent capacity turned out too
// small (a supertile list, the bucket sort's geometry, the tile-instance buffer): nobody has seen the
// frame's output yet, so the caller just gets the correct frame a little later. Fills the lane's stats.
int finish_lane(bgs_ctx* ctx, Lane& L) {
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
        if (rerun) {
            if (attempt >= 8) return fail(ctx, BGS_ECAPACITY, "frame kept overflowing its buffers");
            ctx->regrow_count += 1;
            uint32_t* const next_target = ctx->next_srgb8_target;  // belongs to a frame not enqueued yet
            ctx->next_srgb8_target = L.in_srgb8_target;
            if (sort_gave_up) L.force_onesweep = true;  // stays for every further attempt of this frame
            ctx->rerun_onesweep = L.force_onesweep;
            int rc = enqueue_frame(ctx, L, L.in_cloud, &L.in_view, &L.in_settings, render, L.in_allow_graph);
            ctx->rerun_onesweep = false;
            ctx->next_srgb8_target = next_target;
            if (rc != BGS_OK) return rc;
            continue;
        }

        L.force_onesweep = false;
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
            if (splitters_ascending(h.splitters, BUCKET_COUNT - 1u) && L.in_cloud) {
                int slot = find_splitter_slot(ctx, L.in_cloud, &L.in_view, &L.in_settings);
                if (slot < 0) {  // a view not seen lately: take an empty slot, else the least recently used one
                    slot = 0;
                    for (int i = 0; i < bgs_ctx::SPLITTER_SLOTS; ++i) {
                        if (!ctx->split_slots[i].epoch) { slot = i; break; }
                        if (ctx->split_slots[i].last_used < ctx->split_slots[slot].last_used) slot = i;
                    }
                }
                auto& sl = ctx->split_slots[slot];
                std::memcpy(sl.table.key, h.splitters, sizeof sl.table.key);
                sl.cloud = L.in_cloud;
                sl.n = n;
                sl.sort_mode = L.in_settings.sort_mode;
                std::memcpy(sl.transform, L.in_settings.transform, sizeof sl.transform);
                view_pose(&L.in_view, sl.pos, sl.fwd);
                // the median key is ~bits(dist^2) of the median drawable splat (keys are 0xFFFFFFFF - bits)
                const uint32_t mid_bits = 0xFFFFFFFFu - h.splitters[BUCKET_COUNT / 2 - 1];
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
            const uint32_t target = next_supertile_level((double)total / (double)h.visible_count, lv, L.pending_edges, &longer);
            if (target != lv && ctx->sup_level != target) {
                // lists of another level: predicted from THIS frame's longest list (coarser supertiles hold
                // longer lists: entries scale with the ratio, lists with the area), never from the old hint
                const double predicted = (double)pending_longest * (target > lv ? longer : 1.0) * 1.25;
                ctx->coarse_cap_hint = std::max<uint32_t>(pow2_ceil((uint64_t)std::min(predicted, 1.0e9)), 4096u);
                ctx->list_shrink_votes = 0;
            }
            if (target != lv) { if (ctx->sup_level != target) ctx->level_changes += 1; ctx->sup_level = target; }
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
        stt.instance_capacity = (render && scan) ? (uint64_t)L.coarse_entries : L.inst_cap;
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
            uint64_t bytes = N * 16 + N * 8 + (L.pending_bucket ? D * 24 : k * D * 16);
            if (render) {
                const uint64_t B = L.pending_cloud_format == CLOUD_F16 ? 128 : 240, R = rec_bytes, V = h.visible_count, I = total;
                const uint64_t P = (uint64_t)L.pending_w * L.pending_h;
                if (scan)  // coarse entries (rank + tile rect, 8 B): written once, read by the tiles of their supertile
                    bytes += V * (B - 16) + V * R + V * 8 + I * 8 + I * 8 + P * 16;
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
        if (best < 0) return BGS_OK;
        int rc = finish_lane(ctx, ctx->lanes[best]);
        if (rc != BGS_OK) return rc;
    }
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

    // Depth-sort path. The bucket sort needs 32-bit keys (shorter keys are mostly ties, which it ranks
    // quadratically), a draw count that fits its geometry (a bucket holds <= BUCKET_CAP pairs) and the key
    // range of a recent frame; it is checked on the device and the frame re-run with the digit passes when
    // it does not work out (then bucket_block keeps the following frames on the passes for a while).
    // Debug flags: 0x80000 never, 0x200000 also with a guessed range (no completed frame yet).
    const bool guess = (ctx->debug_flags & 0x200000u) != 0u;
    const int split_slot = (places == 4 && n > 0) ? find_splitter_slot(ctx, cloud, view, s) : -1;
    bool bucket = places == 4 && n > 0 && !(ctx->debug_flags & 0x80000u) && ctx->bucket_block == 0 && !ctx->rerun_onesweep &&
                  ((split_slot >= 0 && ctx->draw_hint_valid) || guess) &&
                  (!ctx->draw_hint_valid || ctx->draw_hint <= BUCKET_COUNT * (BUCKET_CAP / 4u) * 3u);
    if (ctx->bucket_block > 0 && places == 4) ctx->bucket_block -= 1;
    if (bucket) {
        fp.sort_path = 1u;
        if ((rc = ensure_bucket_slots(ctx, L)) != BGS_OK) return rc;
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
    // tile / edge by reciprocal multiply is exact for edges <= 32 (supertile_div)
    uint32_t sup_edge = level == 0 ? edge_f : std::min<uint32_t>(edge_c << (level - 1u), 32u), sup_bx = 0, sup_by = 0;
    if (!bins(sup_edge, sup_bx, sup_by)) { sup_edge = edge_c; sup_bx = cbx; sup_by = cby; }
    const uint32_t num_st = sup_bx * sup_by;
    uint32_t coarse_cap = 1;  // entries per supertile list
    if (render) {
        if (scan) {
            if ((rc = ensure_coarse(ctx, L, n, num_st, &coarse_cap)) != BGS_OK) return rc;
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
    kg.culled = L.culled;
    kg.ctl = ctl;
    kg.part_status = part_status;
    kg.places = places;
    kg.ticket_slot = 7;
    kg.fp_out = L.d_fp;
    kg.bucket_slots = L.bucket_slots;
    kg.bucket_status = depth_status;  // the depth passes' look-back words are free in a bucket-sort frame
    L.pending_split_slot = -1;
    L.pending_split_epoch = 0;
    if (bucket) {
        if (split_slot >= 0) {
            kg.split = ctx->split_slots[split_slot].table;
            ctx->split_slots[split_slot].last_used = ctx->seq + 1;
            L.pending_split_slot = split_slot;
            L.pending_split_epoch = ctx->split_slots[split_slot].epoch;
        } else {  // debug flag 0x200000: a guessed table (equal steps over the 32-bit range: badly balanced)
            for (uint32_t i = 0; i < BUCKET_COUNT; ++i) kg.split.key[i] = (i + 1u) << 24;
        }
    }
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
    // project+bin grid: one block per 256-rank tile of the D drawable entries when that fits the chip
    // (every block then takes exactly one ticket; two 172-VGPR blocks are resident per CU, a third of
    // the grid may queue behind them)
    int bin_blocks = ctx->num_cus * 3;
    if (hinted)
        bin_blocks = (int)std::min<uint64_t>((uint64_t)bin_blocks, std::max<uint64_t>((uint64_t)ctx->draw_hint / 256 + 8, 32));
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
        cl.bucket_chain_words = bucket ? (uint32_t)((n + keygen_tile_splats(n) - 1u) / keygen_tile_splats(n)) * BUCKET_COUNT : 0u;
        cl.sorted = draw_list;
        cl.key_xor = final_xor;
        if (ctx->debug_flags & 0x1000u) cl = FrameCleanup{};  // experiment: classic memset + copy path
    }
    const bool raster_cleans = render && scan && fp.tiles_x > 0 && fp.tiles_y > 0 && cl.other_ctl != nullptr;

    // the launches of one frame, in stream order (issued directly, or once into a stream capture)
    auto issue = [&]() -> hipError_t {
        mark(0);
        if (have_keygen) {
            hipError_t e = kg.launch(st);
            if (e != hipSuccess) return e;
        }
        mark(1);
        int cur = 0;
        if (bucket)
            launch_bucket_sort(st, L.bucket_slots, draw_list, ctl, final_xor);
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
            launch_project_bin(st, fp, L.d_fp, cloud->ptrs, draw_list, L.culled, ctl, bin_status, L.records, L.coarse,
                               coarse_cap, sup_edge, /*ticket_slot=*/4, bin_blocks);
            mark(3);
            launch_raster_scan(st, fp, L.d_fp, L.records, L.coarse, coarse_cap, sup_edge, ctl, L.fb, L.fb8,
                               (ctx->debug_flags & 0x40000u) ? 0u : out_format, cl);
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
    const bool use_graph = allow_graph && ctx->use_graphs && render && scan && raster_cleans && !need_memset &&
                           prof == 0 && have_keygen && !(ctx->debug_flags & 0x4000u);
    if (use_graph) {
        GraphKey key;
        std::memset(&key, 0, sizeof key);
        const void* planes[6] = {cloud->ptrs.position_visibility, cloud->ptrs.sh_f32, cloud->ptrs.rot_scale,
                                 cloud->ptrs.cov3d_opacity, cloud->ptrs.sh_f16, cloud->ptrs.rot_scale_opacity_f16};
        std::memcpy(key.cloud, planes, sizeof planes);
        const void* bufs[11] = {L.entries[0], L.entries[1], L.culled, L.records, L.coarse, L.fb, L.fb8, L.scratch, L.d_fp,
                                L.h_ctl_dev, L.bucket_slots};
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
        key.bin_blocks = bin_blocks;
        key.keygen_blocks = (int32_t)kg.blocks;
        key.sup_edge = sup_edge;
        key.scratch_bytes = L.scratch_bytes;
        key.scratch_inst_cap = L.scratch_inst_cap;
        key.scratch_n = L.scratch_n;
        key.coarse_cap = coarse_cap;
        key.sort_path = bucket ? 1u : 0u;
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
        if (places == 4 && n > 0) launch_splitters(st, draw_list, ctl, final_xor);
        HIP_TRY(ctx, hipMemcpyAsync(L.h_ctl, ctl, sizeof(Control), hipMemcpyDeviceToHost, st));
    }

    HIP_TRY(ctx, hipEventRecord(L.done, st));
    L.pending = true;
    L.pending_render = render;
    L.pending_scan = scan;
    L.pending_bucket = bucket;
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
        if ((rc = enqueue_frame(ctx, L, cloud, view, s, render, false)) != BGS_OK) return rc;
        return finish_lane(ctx, L);  // re-runs the frame itself if a capacity was too small
    }
    // async frame: next lane of the ring; completing its previous occupant first
    Lane& L = ctx->lanes[ctx->next];
    if (L.pending && (rc = finish_lane(ctx, L)) != BGS_OK) return rc;
    if ((rc = enqueue_frame(ctx, L, cloud, view, s, render, /*allow_graph=*/true)) != BGS_OK) return rc;
    ctx->recent = ctx->next;
    ctx->next = (ctx->next + 1) % ctx->depth;
    return BGS_OK;
}

void mat4_mul(const float* a, const float* b, float* out) {  // column-major out = a * b
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc += a[4 * k + r] * b[4 * c + k];
            out[4 * c + r] = acc;
        }
}

bool mat4_inverse(const float* m, float* out) {
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = m[4 * c + r];
            a[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int i = 0; i < 4; ++i) {
        int piv = i;
        for (int r = i + 1; r < 4; ++r)
            if (std::fabs(a[r][i]) > std::fabs(a[piv][i])) piv = r;
        if (a[piv][i] == 0.0) return false;
        if (piv != i)
            for (int c = 0; c < 8; ++c) std::swap(a[i][c], a[piv][c]);
        const double d = a[i][i];
        for (int c = 0; c < 8; ++c) a[i][c] /= d;
        for (int r = 0; r < 4; ++r)
            if (r != i) {
                const double f = a[r][i];
                for (int c = 0; c < 8; ++c) a[r][c] -= f * a[i][c];
            }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[4 * c + r] = (float)a[r][4 + c];
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

uint32_t bgs_version(void) { return (BGS_VERSION_MAJOR << 16) | BGS_VERSION_MINOR; }

const char* bgs_last_error(const bgs_ctx* ctx) { return ctx ? ctx->error.c_str() : g_error.c_str(); }

int bgs_create(int hip_device, bgs_ctx** out) {
    if (!out) return fail(nullptr, BGS_EINVAL, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, BGS_EHIP, std::string("no usable HIP device (there is no CPU fallback): ") +
                                           hipGetErrorString(e));
    if (hip_device < 0 || hip_device >= count) return fail(nullptr, BGS_EINVAL, "hip_device out of range");
    bgs_ctx* ctx = new (std::nothrow) bgs_ctx();
    if (!ctx) return fail(nullptr, BGS_ENOMEM, "out of host memory");
    ctx->device = hip_device;
    auto bail = [&](const std::string& msg) {
        bgs_destroy(ctx);
        return fail(nullptr, BGS_EHIP, msg);
    };
    if ((e = hipSetDevice(hip_device)) != hipSuccess) return bail(std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, hip_device)) != hipSuccess)
        return bail(std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (lane_create(ctx, ctx->lanes[0]) != BGS_OK) return bail(ctx->error);
    *out = ctx;
    return BGS_OK;
}

void bgs_destroy(bgs_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto& L : ctx->lanes) lane_destroy(L);
    for (auto st : ctx->streams) if (st) (void)hipStreamDestroy(st);
    for (auto st : ctx->queue_holders) if (st) (void)hipStreamDestroy(st);
    delete ctx;
}

void bgs_settings_default(bgs_settings* out) {
    if (!out) return;
    std::memset(out, 0, sizeof *out);
    for (int i = 0; i < 4; ++i) out->transform[5 * i] = 1.0f;
    out->global_opacity = 1.0f;             // src/gaussian/settings.rs:114
    out->global_scale = 1.0f;               // :115
    out->gaussian_mode = BGS_GAUSSIAN_3D;   // :17-22 default
    out->aabb = 0;                          // :113
    out->opacity_adaptive_radius = 1;       // :116
    out->color_space = BGS_COLOR_SRGB;      // :79-84 default
    out->radix_depth_bits = 32;             // :52-57 default
    out->sh_degree = 3;                     // Cargo.toml default feature sh3
    out->sort_mode = BGS_SORT_RADIX;        // src/sort/mod.rs:60-74
    out->rasterize_mode = BGS_RASTERIZE_COLOR;  // src/gaussian/settings.rs:40-41
    out->num_classes = 1;                   // :124
    for (int i = 0; i < 3; ++i) out->position_max[i] = 1.0f;
    out->position_min[3] = out->position_max[3] = 1.0f;  // aabb.min().extend(1.0) src/render/mod.rs:1070
}

void bgs_view_perspective(const float world_from_view[16], float fov_y_radians, float near_plane,
                          uint32_t width, uint32_t height, bgs_view* out) {
    if (!out || !world_from_view) return;
    std::memset(out, 0, sizeof *out);
    std::memcpy(out->world_from_view, world_from_view, 16 * sizeof(float));
    if (!mat4_inverse(world_from_view, out->view_from_world))
        for (int i = 0; i < 4; ++i) out->view_from_world[5 * i] = 1.0f;
    // glam Mat4::perspective_infinite_reverse_rh (bevy PerspectiveProjection)
    const float f = 1.0f / std::tan(0.5f * fov_y_radians);
    const float aspect = (float)width / (float)height;
    out->clip_from_view[0] = f / aspect;
    out->clip_from_view[5] = f;
    out->clip_from_view[11] = -1.0f;
    out->clip_from_view[14] = near_plane;
    mat4_mul(out->clip_from_view, out->view_from_world, out->clip_from_world);
    out->viewport[0] = 0.0f;
    out->viewport[1] = 0.0f;
    out->viewport[2] = (float)width;
    out->viewport[3] = (float)height;
    out->clear_color[3] = 1.0f;  // opaque black, examples/headless.rs:70
    std::memcpy(out->previous_clip_from_world, out->clip_from_world, sizeof out->clip_from_world);
    out->delta_time = 1.0f / 60.0f;
}

static int upload_plane(bgs_ctx* ctx, const void* host, size_t bytes, void** dev) {
    *dev = nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess)
        return fail(ctx, BGS_ENOMEM, "hipMalloc(cloud plane) failed");
    if (bytes && hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p);
        return fail(ctx, BGS_EHIP, "hipMemcpy(cloud plane) failed");
    }
    *dev = p;
    return BGS_OK;
}

int bgs_cloud_upload_f32(bgs_ctx* ctx, uint32_t n, const float* pv, const float* sh, const float* rot,
                         const float* so, bgs_cloud** out) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (!out) return fail(ctx, BGS_EINVAL, "out is NULL");
    *out = nullptr;
    if (n > MAX_SPLATS) return fail(ctx, BGS_EINVAL, "too many splats");
    if (n && (!pv || !sh || !rot || !so)) return fail(ctx, BGS_EINVAL, "NULL plane pointer");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    bgs_cloud* c = new (std::nothrow) bgs_cloud();
    if (!c) return fail(ctx, BGS_ENOMEM, "out of host memory");
    // rotation and scale_opacity are read together, by visible splats only, at sorted (random) indices:
    // interleaved into one 32-byte record per splat they cost one cache line instead of two
    std::vector<float> rs;
    try {
        rs.resize((size_t)n * 8);
    } catch (const std::bad_alloc&) {
        delete c;
        return fail(ctx, BGS_ENOMEM, "out of host memory");
    }
    for (size_t i = 0; i < n; ++i) {
        std::memcpy(&rs[8 * i], rot + 4 * i, 16);
        std::memcpy(&rs[8 * i + 4], so + 4 * i, 16);
    }
    const void* src[3] = {pv, sh, rs.data()};
    const size_t bytes[3] = {(size_t)n * 16, (size_t)n * 192, (size_t)n * 32};
    for (int i = 0; i < 3; ++i) {
        int rc = upload_plane(ctx, src[i], bytes[i], &c->allocs[i]);
        if (rc != BGS_OK) { bgs_cloud_free(ctx, c); return rc; }
        c->bytes += bytes[i];
    }
    c->ptrs.position_visibility = (const float4*)c->allocs[0];
    c->ptrs.sh_f32 = (const float*)c->allocs[1];
    c->ptrs.rot_scale = (const float4*)c->allocs[2];
    c->ptrs.n = n;
    c->ptrs.format = CLOUD_F32;
    *out = c;
    return BGS_OK;
}

int bgs_cloud_upload_f16(bgs_ctx* ctx, uint32_t n, const float* pv, const uint32_t* sh_h2,
                         const uint32_t* rso, bgs_cloud** out) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (!out) return fail(ctx, BGS_EINVAL, "out is NULL");
    *out = nullptr;
    if (n > MAX_SPLATS) return fail(ctx, BGS_EINVAL, "too many splats");
    if (n && (!pv || !sh_h2 || !rso)) return fail(ctx, BGS_EINVAL, "NULL plane pointer");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    bgs_cloud* c = new (std::nothrow) bgs_cloud();
    if (!c) return fail(ctx, BGS_ENOMEM, "out of host memory");
    const void* src[3] = {pv, sh_h2, rso};
    const size_t bytes[3] = {(size_t)n * 16, (size_t)n * 96, (size_t)n * 16};
    for (int i = 0; i < 3; ++i) {
        int rc = upload_plane(ctx, src[i], bytes[i], &c->allocs[i]);
        if (rc != BGS_OK) { bgs_cloud_free(ctx, c); return rc; }
        c->bytes += bytes[i];
    }
    c->ptrs.position_visibility = (const float4*)c->allocs[0];
    c->ptrs.sh_f16 = (const uint32_t*)c->allocs[1];
    c->ptrs.rot_scale_opacity_f16 = (const uint4*)c->allocs[2];
    c->ptrs.n = n;
    c->ptrs.format = CLOUD_F16;
    *out = c;
    return BGS_OK;
}

int bgs_cloud_upload_cov3d_f32(bgs_ctx* ctx, uint32_t n, const float* pv, const float* sh, const float* cov3d_opacity,
                               bgs_cloud** out) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (!out) return fail(ctx, BGS_EINVAL, "out is NULL");
    *out = nullptr;
    if (n > MAX_SPLATS) return fail(ctx, BGS_EINVAL, "too many splats");
    if (n && (!pv || !sh || !cov3d_opacity)) return fail(ctx, BGS_EINVAL, "NULL plane pointer");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    bgs_cloud* c = new (std::nothrow) bgs_cloud();
    if (!c) return fail(ctx, BGS_ENOMEM, "out of host memory");
    const void* src[3] = {pv, sh, cov3d_opacity};
    const size_t bytes[3] = {(size_t)n * 16, (size_t)n * 192, (size_t)n * 32};
    for (int i = 0; i < 3; ++i) {
        int rc = upload_plane(ctx, src[i], bytes[i], &c->allocs[i]);
        if (rc != BGS_OK) { bgs_cloud_free(ctx, c); return rc; }
        c->bytes += bytes[i];
    }
    c->ptrs.position_visibility = (const float4*)c->allocs[0];
    c->ptrs.sh_f32 = (const float*)c->allocs[1];
    c->ptrs.cov3d_opacity = (const float4*)c->allocs[2];
    c->ptrs.n = n;
    c->ptrs.format = CLOUD_COV3D;
    *out = c;
    return BGS_OK;
}

void bgs_cloud_free(bgs_ctx* ctx, bgs_cloud* cloud) {
    if (!cloud) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)finish_all(ctx);  // frames in flight may still be re-run with this cloud (finish_lane)
        for (auto& L : ctx->lanes) {
            if (L.stream) (void)hipStreamSynchronize(L.stream);
            if (L.in_cloud == cloud) L.in_cloud = nullptr;
        }
        for (auto& sl : ctx->split_slots)
            if (sl.cloud == cloud) sl.epoch = 0;  // a later cloud may get the same address
    }
    for (auto p : cloud->allocs) if (p) (void)hipFree(p);
    delete cloud;
}

uint32_t bgs_cloud_len(const bgs_cloud* cloud) { return cloud ? cloud->ptrs.n : 0; }

int bgs_sort(bgs_ctx* ctx, const bgs_cloud* cloud, const bgs_view* view, const bgs_settings* settings,
             bgs_sort_entry* host_out) {
    int rc = run(ctx, cloud, view, settings, /*render=*/false);
    if (rc != BGS_OK) return rc;
    Lane& L = ctx->lanes[0];
    if (host_out && L.last_sorted_n)
        HIP_TRY(ctx, hipMemcpy(host_out, L.last_sorted, (size_t)L.last_sorted_n * sizeof(uint2), hipMemcpyDeviceToHost));
    return BGS_OK;
}

int bgs_render(bgs_ctx* ctx, const bgs_cloud* cloud, const bgs_view* view, const bgs_settings* settings,
               float* rgba_host_out) {
    int rc = run(ctx, cloud, view, settings, /*render=*/true);
    if (rc != BGS_OK) return rc;
    if (rgba_host_out) {
        Lane& L = ctx->lanes[ctx->recent];
        if (L.pending && (rc = finish_lane(ctx, L)) != BGS_OK) return rc;
        if (!L.fb_valid) return fail(ctx, BGS_EINVAL, "the frame wrote its packed image only (bgs_set_packed_only): no f32 target to copy");
        HIP_TRY(ctx, hipMemcpy(rgba_host_out, L.fb, (size_t)L.fb_w * L.fb_h * sizeof(float4), hipMemcpyDeviceToHost));
    }
    return BGS_OK;
}

int bgs_framebuffer_device_ptr(bgs_ctx* ctx, void** dptr, uint64_t* bytes) {
    if (!ctx || !dptr) return fail(ctx, BGS_EINVAL, "NULL argument");
    Lane& L = ctx->lanes[ctx->recent];
    if (L.pending) {
        int rc = finish_lane(ctx, L);
        if (rc != BGS_OK) return rc;
    }
    if (!L.fb) return fail(ctx, BGS_EINVAL, "no frame has been rendered yet");
    if (!L.fb_valid) return fail(ctx, BGS_EINVAL, "the last frame wrote its packed image only (bgs_set_packed_only)");
    *dptr = L.fb;
    if (bytes) *bytes = (uint64_t)L.fb_w * L.fb_h * sizeof(float4);
    return BGS_OK;
}

int bgs_framebuffer_srgb8_device_ptr(bgs_ctx* ctx, void** dptr, uint64_t* bytes) {
    if (!ctx || !dptr) return fail(ctx, BGS_EINVAL, "NULL argument");
    Lane& L = ctx->lanes[ctx->recent];
    if (L.pending) {
        int rc = finish_lane(ctx, L);
        if (rc != BGS_OK) return rc;
    }
    if (!L.fb8_out || !L.fb8_valid || L.fb8_is_f16)
        return fail(ctx, BGS_EINVAL, "no Rgba8UnormSrgb frame: call bgs_set_output_srgb8(ctx, 1) before rendering");
    *dptr = L.fb8_out;
    if (bytes) *bytes = (uint64_t)L.fb_w * L.fb_h * 4u;
    return BGS_OK;
}

int bgs_pipeline_pop(bgs_ctx* ctx, void** rgba_f32, void** rgba8) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    int best = -1;
    for (int i = 0; i < MAX_LANES; ++i)
        if (ctx->lanes[i].pending && (best < 0 || ctx->lanes[i].seq < ctx->lanes[best].seq)) best = i;
    if (best < 0) return fail(ctx, BGS_EINVAL, "no frame in flight");
    Lane& L = ctx->lanes[best];
    int rc = finish_lane(ctx, L);
    if (rc != BGS_OK) return rc;
    if (rgba_f32) *rgba_f32 = L.fb_valid ? L.fb : nullptr;
    if (rgba8) *rgba8 = L.fb8_valid ? L.fb8_out : nullptr;
    return BGS_OK;
}

int bgs_frames_in_flight(bgs_ctx* ctx, uint32_t* count) {
    if (!ctx || !count) return fail(ctx, BGS_EINVAL, "NULL argument");
    uint32_t c = 0;
    for (auto& L : ctx->lanes) c += L.pending ? 1u : 0u;
    *count = c;
    return BGS_OK;
}

int bgs_sorted_entries_device_ptr(bgs_ctx* ctx, void** dptr, uint32_t* n) {
    if (!ctx || !dptr) return fail(ctx, BGS_EINVAL, "NULL argument");
    Lane& L = ctx->lanes[ctx->recent];
    if (L.pending) {
        int rc = finish_lane(ctx, L);
        if (rc != BGS_OK) return rc;
    }
    if (!L.last_sorted) return fail(ctx, BGS_EINVAL, "no sort has been run yet");
    *dptr = (void*)L.last_sorted;
    if (n) *n = L.last_sorted_n;
    return BGS_OK;
}

int bgs_synchronize(bgs_ctx* ctx) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    for (auto& L : ctx->lanes)
        if (L.stream) HIP_TRY(ctx, hipStreamSynchronize(L.stream));
    return BGS_OK;
}

int bgs_set_async(bgs_ctx* ctx, int enabled) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (!enabled) {
        int rc = finish_all(ctx);
        if (rc != BGS_OK) return rc;
    }
    ctx->async_frames = enabled != 0;
    return BGS_OK;
}

int bgs_set_pipeline_depth(bgs_ctx* ctx, uint32_t lanes) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (lanes < 1 || lanes > (uint32_t)MAX_LANES) return fail(ctx, BGS_EINVAL, "pipeline depth must be 1..8");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    ctx->depth = (int)lanes;
    ctx->next = 0;
    ctx->recent = 0;
    for (uint32_t i = 0; i < lanes; ++i)
        if ((rc = lane_create(ctx, ctx->lanes[i])) != BGS_OK) return rc;
    return assign_streams(ctx);
}

int bgs_set_pipeline_streams(bgs_ctx* ctx, uint32_t streams) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (streams > (uint32_t)MAX_LANES) return fail(ctx, BGS_EINVAL, "pipeline streams must be 0..8");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    for (auto st : ctx->streams)
        if (st) HIP_TRY(ctx, hipStreamSynchronize(st));
    ctx->num_streams = (int)streams;
    return assign_streams(ctx);
}

int bgs_set_srgb8_target(bgs_ctx* ctx, void* device_ptr) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    ctx->next_srgb8_target = (uint32_t*)device_ptr;
    return BGS_OK;
}

int bgs_download(bgs_ctx* ctx, const void* device_ptr, void* host_out, uint64_t bytes) {
    if (!ctx || !device_ptr || !host_out) return fail(ctx, BGS_EINVAL, "NULL argument");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    if (bytes) HIP_TRY(ctx, hipMemcpy(host_out, device_ptr, (size_t)bytes, hipMemcpyDeviceToHost));
    return BGS_OK;
}

int bgs_set_output_srgb8(bgs_ctx* ctx, int enabled) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    ctx->output_srgb8 = enabled != 0;
    if (enabled) ctx->output_rgba16f = false;
    return BGS_OK;
}

int bgs_set_output_rgba16f(bgs_ctx* ctx, int enabled) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    ctx->output_rgba16f = enabled != 0;
    if (enabled) ctx->output_srgb8 = false;
    return BGS_OK;
}

int bgs_set_packed_only(bgs_ctx* ctx, int enabled) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    ctx->packed_only = enabled != 0;
    return BGS_OK;
}

int bgs_framebuffer_rgba16f_device_ptr(bgs_ctx* ctx, void** dptr, uint64_t* bytes) {
    if (!ctx || !dptr) return fail(ctx, BGS_EINVAL, "NULL argument");
    Lane& L = ctx->lanes[ctx->recent];
    if (L.pending) {
        int rc = finish_lane(ctx, L);
        if (rc != BGS_OK) return rc;
    }
    if (!L.fb8_out || !L.fb8_valid || !L.fb8_is_f16)
        return fail(ctx, BGS_EINVAL, "no Rgba16Float frame: call bgs_set_output_rgba16f(ctx, 1) before rendering");
    *dptr = L.fb8_out;
    if (bytes) *bytes = (uint64_t)L.fb_w * L.fb_h * 8u;
    return BGS_OK;
}

int bgs_stream(bgs_ctx* ctx, void** hip_stream) {
    if (!ctx || !hip_stream) return fail(ctx, BGS_EINVAL, "NULL argument");
    *hip_stream = (void*)ctx->lanes[ctx->recent].stream;
    return BGS_OK;
}

int bgs_set_profiling(bgs_ctx* ctx, int enabled) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (enabled < 0 || enabled > 2) return fail(ctx, BGS_EINVAL, "profiling level must be 0, 1 or 2");
    ctx->profiling = enabled;
    return BGS_OK;
}

int bgs_set_profiling_stride(bgs_ctx* ctx, uint32_t every_nth_frame) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (every_nth_frame == 0) return fail(ctx, BGS_EINVAL, "stride must be >= 1");
    ctx->profiling_stride = every_nth_frame;
    ctx->frame_counter = 0;
    return BGS_OK;
}

int bgs_set_binning(bgs_ctx* ctx, uint32_t mode) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (mode > BINNING_SORT) return fail(ctx, BGS_EINVAL, "binning mode must be 0 (scan) or 1 (sort)");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    ctx->binning = mode;
    return BGS_OK;
}

int bgs_set_debug_flags(bgs_ctx* ctx, uint32_t flags) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    ctx->debug_flags = flags;
    return BGS_OK;
}

int bgs_adaptive_counters(bgs_ctx* ctx, uint64_t out[8]) {
    if (!ctx || !out) return fail(ctx, BGS_EINVAL, "NULL argument");
    out[0] = ctx->bucket_frames;
    out[1] = ctx->onesweep_frames;
    out[2] = ctx->reruns_sort;
    out[3] = ctx->reruns_lists;
    out[4] = ctx->reruns_instances;
    out[5] = ctx->level_changes;
    out[6] = ctx->sup_level;
    out[7] = ctx->coarse_cap_hint;
    return BGS_OK;
}

int bgs_reset_adaptive_state(bgs_ctx* ctx) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    ctx->draw_hint_valid = false;
    for (auto& sl : ctx->split_slots) sl.epoch = 0;
    ctx->split_failed_epoch = ctx->split_epoch;
    ctx->bucket_block = 0;
    ctx->bucket_fail_streak = 0;
    ctx->list_shrink_votes = 0;
    ctx->sup_level = 1;
    ctx->coarse_cap_hint = 0;
    return BGS_OK;
}

int bgs_get_stats(bgs_ctx* ctx, bgs_stats* out) {
    if (!ctx || !out) return fail(ctx, BGS_EINVAL, "NULL argument");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    if ((rc = collect_stats(ctx)) != BGS_OK) return rc;
    *out = ctx->stats;
    return BGS_OK;
}

int bgs_radix_sort_pairs(bgs_ctx* ctx, bgs_sort_entry* entries, uint32_t n, uint32_t passes) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (passes < 1 || passes > 4) return fail(ctx, BGS_EINVAL, "passes must be 1..4");
    if (n > MAX_SPLATS) return fail(ctx, BGS_EINVAL, "too many pairs");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    if (n == 0) return BGS_OK;
    if (!entries) return fail(ctx, BGS_EINVAL, "entries is NULL");
    Lane& L = ctx->lanes[0];
    if ((rc = ensure_entries(ctx, L, n)) != BGS_OK) return rc;
    if ((rc = ensure_scratch(ctx, L, n, L.inst_cap)) != BGS_OK) return rc;
    hipStream_t st = L.stream;
    Control* ctl = (Control*)L.scratch;
    uint32_t* depth_status = (uint32_t*)(L.scratch + L.off_depth_status);
    HIP_TRY(ctx, hipMemsetAsync(L.scratch, 0, L.scratch_bytes, st));
    L.scratch_clean = false;
    HIP_TRY(ctx, hipMemcpyAsync(L.entries[0], entries, (size_t)n * sizeof(uint2), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)&ctl->splat_count, (int)n, 1, st));
    launch_histogram(st, L.entries[0], n, &ctl->hist_depth[0][0], passes);
    const bool large = n > (4u << 20);
    const size_t depth_tiles = ((size_t)L.scratch_n + sort_tile_size(false) - 1) / sort_tile_size(false) + 1;
    int cur = 0;
    for (uint32_t p = 0; p < passes; ++p) {
        launch_onesweep_pass(st, L.entries[cur], L.entries[cur ^ 1], &ctl->splat_count, n, ctl->hist_depth[p],
                             depth_status + (size_t)p * depth_tiles * RADIX_BASE, &ctl->ticket[p][0], &ctl->error,
                             p * RADIX_BITS, 0u, large, ctx->num_cus * 4);
        cur ^= 1;
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(entries, L.entries[cur], (size_t)n * sizeof(uint2), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(L.h_ctl, ctl, sizeof(Control), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (L.h_ctl->error) return fail(ctx, BGS_EINTERNAL, "device watchdog tripped in radix sort");
    return BGS_OK;
}

int bgs_hbm_probe(bgs_ctx* ctx, uint64_t bytes, uint32_t iters, float* copy_gbs, float* triad_gbs) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    bytes &= ~(uint64_t)15;
    if (bytes < 4096 || iters == 0 || iters > 10000) return fail(ctx, BGS_EINVAL, "bytes >= 4096 and 1 <= iters <= 10000");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    char* buf = nullptr;
    if (hipMalloc((void**)&buf, (size_t)bytes * 3) != hipSuccess) {
        (void)hipGetLastError();
        return fail(ctx, BGS_ENOMEM, "hipMalloc(probe buffers) failed");
    }
    hipStream_t st = ctx->lanes[0].stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms_copy = 0.0f, ms_triad = 0.0f;
    bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    float4 *a = (float4*)buf, *b = (float4*)(buf + bytes), *c = (float4*)(buf + 2 * bytes);
    const size_t n4 = (size_t)bytes / 16;
    const int blocks = ctx->num_cus * 16;
    ok = ok && hipMemsetAsync(buf, 0, (size_t)bytes * 3, st) == hipSuccess;
    if (ok) {  // warm-up, then `iters` back-to-back repetitions between two events
        ok = hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
        ok = ok && hipEventRecord(e0, st) == hipSuccess;
        for (uint32_t i = 0; ok && i < iters; ++i)
            ok = hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
        ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
             hipEventElapsedTime(&ms_copy, e0, e1) == hipSuccess;
    }
    if (ok) {
        launch_triad(st, a, b, c, 0.5f, n4, blocks);
        ok = hipEventRecord(e0, st) == hipSuccess;
        for (uint32_t i = 0; ok && i < iters; ++i) launch_triad(st, a, b, c, 0.5f, n4, blocks);
        ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(e1, st) == hipSuccess &&
             hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms_triad, e0, e1) == hipSuccess;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (!ok) { (void)hipGetLastError(); return fail(ctx, BGS_EHIP, "HBM probe failed"); }
    if (copy_gbs) *copy_gbs = ms_copy > 0.0f ? (float)(2.0 * (double)bytes * iters / (ms_copy * 1e6)) : 0.0f;
    if (triad_gbs) *triad_gbs = ms_triad > 0.0f ? (float)(3.0 * (double)bytes * iters / (ms_triad * 1e6)) : 0.0f;
    return BGS_OK;
}

int bgs_set_graphs(bgs_ctx* ctx, int enabled) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    ctx->use_graphs = enabled != 0;
    return BGS_OK;
}

int bgs_graph_counters(bgs_ctx* ctx, uint64_t* captures, uint64_t* replays) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (captures) *captures = ctx->graph_captures;
    if (replays) *replays = ctx->graph_replays;
    return BGS_OK;
}

}  // extern "C"
