// bgs_context.h — what the translation units of libbgs' host side share: the context, its frame lanes and captured frame
// graphs (bgs_frame.hip owns their logic), and the functions the C ABI (bgs_api.hip), the diagnostics (bgs_diag.hip)
// and the frame gather (bgs_comm.hip) call. Internal: nothing here is part of the C ABI (include/bgs.h).
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/bgs.h"
#include "../../include/bgs_diag.h"
#include "bgs_device.h"
#include "frame_params.h"
#include "kernels.h"

using namespace bgs;

struct bgs_cloud {
    CloudPtrs ptrs{};
    void* allocs[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t bytes = 0;
};

namespace bgs_host {

extern thread_local std::string g_error;   // (bgs_frame.hip) the message of the calling thread's last failure: bgs_last_error(NULL)

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
    // keygen has several instantiations per size (with / without chains, 256- or 1024-thread tiles: KeygenLaunch): a
    // captured node is only ever UPDATED to the kernel and block size it was captured with
    const void* keygen_func;
    uint32_t keygen_threads;
    uint32_t wide_bin;         // bin_kernel<16> (1024 threads) or <4>: follows the pipeline depth, not the grid sizes
    uint32_t raster_variant;   // sample_count | (a depth buffer is bound) << 8: which rasteriser instantiation the graph holds
    uint32_t split_sub;        // FrameCleanup::split_sub, a rasteriser argument: how many quantile keys the captured clean-up leaves
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
    uint32_t* rects = nullptr;   // BINNING_SCAN: packed tile rectangle per rank (project_kernel -> bin_kernel), entries_cap words
    uint2* inst[2] = {nullptr, nullptr};  // BINNING_SORT only
    uint64_t inst_cap = 0;
    uint32_t* coarse = nullptr;  // BINNING_SCAN: [num_supertiles][coarse_cap] ordered (rank, tile rect) lists
    size_t coarse_entries = 0;   // 8-byte entries allocated (all lists together)
    uint2* bucket_slots = nullptr;  // bucket sort: [256 * bucket_sub_cap][BUCKET_CAP] pairs (8 MB per narrow sub, 32 MB per wide one), allocated on first use, grown with sub
    uint32_t bucket_sub_cap = 0;
    uint32_t* d_split_keys = nullptr;   // a long splitter table (sub > BUCKET_SUB_KERNARG) as keygen reads it: BUCKET_MAX words
    uint32_t* h_split_keys = nullptr;   // ... and the pinned staging it is copied from, ahead of keygen, on the frame's stream
    // heavy-tile feedback of the rasteriser (kernels.h HeavyFeedback): two buffers per lane, written alternately, so that
    // frames of other lanes can still be reading the one this lane's previous frame completed
    uint8_t* heavy[2] = {nullptr, nullptr};
    uint32_t heavy_tiles = 0, heavy_parity = 0;  // heavy[heavy_parity] is what the lane's NEXT frame writes; flipped when a frame COMPLETES
    uint8_t* pending_heavy_out = nullptr;  // what the pending frame's rasteriser writes (null: no feedback from this frame)
    // The feedback the lane's next dense frame consumes: the buffer its most recently COMPLETED dense frame wrote (any
    // view: a stale list costs balance, never pixels — every tile is drawn exactly once, by its regular wave or by a
    // strip workgroup, whichever the list it reads says). Per LANE: a lane's frames run one after the other, so the
    // buffer a frame reads (heavy[parity ^ 1], complete) is never the one it — or a re-run of it — writes
    // (heavy[parity]), and no frame of another lane ever reads either.
    const uint8_t* heavy_done = nullptr;
    uint32_t heavy_done_grid = 0;   // tiles_x | tiles_y << 16 of that frame
    // per-tile cost feedback (kernels.h TileCost), the same life cycle as the heavy-tile lists: cost[cost_parity] is what
    // the lane's next frame writes, cost_done what its most recently completed frame wrote; `order` is made of
    // cost_done at the start of a frame and read by that frame's rasteriser only
    uint64_t in_kind = 0;  // frame_kind() of the lane's frame
    bool ready = false;   // the lane's frame is complete but nobody has taken it yet (bgs_pipeline_pop): a frame finished early, see bgs_ctx::kinds
    uint16_t* cost[2] = {nullptr, nullptr};
    uint16_t* order = nullptr;
    uint32_t cost_tiles = 0, cost_parity = 0;
    uint16_t* pending_cost_out = nullptr;
    const uint16_t* cost_done = nullptr;
    uint32_t cost_done_grid = 0;
    uint32_t order_grid = 0xFFFFFFFFu, order_age = 0;   // the grid `order` is a permutation for; frames drawn with it since it was made
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
    bool pending_culled_written = true;  // keygen wrote the culled tail (bgs_sort, RasterizeMode::Depth)
    uint32_t pending_coarse_cap = 0, pending_level = 1, pending_edges[4] = {6, 8, 16, 32};
    bool force_onesweep = false;       // the pending frame is a re-run of one whose bucket sort gave up
    int pending_split_slot = -1;       // splitter slot the pending bucket-sort frame used (-1: a guessed table)
    uint32_t pending_split_sub = 1;    // the quantile table the pending frame LEAVES has 256 * sub - 1 keys
    uint64_t cost_done_kind = 0;       // kind of the frame that left cost_done
    uint64_t order_kind = 0;           // kind of the frame whose cost plane the lane's order (and its saturation counts) was made of
    bool pending_midround = false;     // the pending frame ran the mid-round-exit rasteriser
    uint64_t pending_sat_kind = 0;     // = order_kind when the pending frame's clean-up reports those counts (0: it does not)
    uint64_t pending_split_epoch = 0;
    // what the pending frame was enqueued with: a frame whose data-dependent capacities turn out too
    // small (coarse lists, bucket sort, tile instances) is re-run on its lane when it is completed
    const bgs_cloud* in_cloud = nullptr;
    bgs_view in_view{};
    bgs_settings in_settings{};
    uint32_t* in_srgb8_target = nullptr;
    bool in_allow_graph = false;
    // ... and the context state it was enqueued under: a re-run must produce the SAME outputs even if a setter
    // (bgs_set_output_srgb8 / _rgba16f / _packed_only / _debug_flags) was called while the frame was in flight —
    // a caller's bgs_set_srgb8_target buffer is sized for the format the frame was enqueued with
    bool in_output_srgb8 = false, in_output_rgba16f = false, in_packed_only = false;
    uint32_t in_debug_flags = 0;
    uint32_t pending_n = 0, pending_places = 0, pending_num_st = 0, pending_rec_bytes = 0, pending_cloud_format = 0;
    uint32_t pending_w = 0, pending_h = 0, pending_tx = 0, pending_ty = 0;
    uint64_t seq = 0;  // enqueue sequence number (to find the oldest pending lane)

    const uint2* last_sorted = nullptr;
    uint32_t last_sorted_n = 0;

    bgs_stats result{};      // counters of the last completed frame of this lane (no timings)
    bool has_result = false;
    uint8_t result_kind = 0; // ev_kind of that frame
};

}  // namespace bgs_host
using namespace bgs_host;

struct bgs_ctx {
    int device = 0;
    int num_cus = 256;
    std::string error;

    Lane lanes[MAX_LANES];
    hipStream_t streams[MAX_LANES] = {};
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
        SplitterKeys table{};
        const bgs_cloud* cloud = nullptr;
        uint32_t n = 0;
        uint32_t sort_mode = 0;   // Radix culls (a frustum's worth of keys), Rayon / Std keep every splat
        float transform[16] = {};
        float pos[3] = {}, fwd[3] = {};
        // SORT_RADIX keys only what the frustum keeps, so the key set also depends on the projection and viewport:
        // two cameras at one pose with different fov / zoom / target size must not share (and overwrite) a table
        float clip_from_view[16] = {};
        float viewport_wh[2] = {};
        float reach = 0.0f;       // median view distance of the list the table came from (scale of "near")
        uint64_t epoch = 0;       // 0 = empty
        uint64_t last_used = 0;
    };
    static constexpr int SPLITTER_SLOTS = 16;  // (1 KB each; the key includes the projection since round 3, so zooms / resizes take slots too)
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
    // KINDS OF FRAME. What a completed frame teaches (list capacity, supertile level) depends on the kind of frame it was:
    // (splat count and storage format of the cloud, mode, quad shape, global scale to half an octave, viewport size,
    // sample count, depth buffer or not) — frame_kind(); poses do not count, a moving camera stays pipelined. An async
    // frame of a kind the context has not settled on is completed at once (re-running it if a first guess was too
    // small) instead of sending pipeline-depth frames out on first guesses and re-running every one of them (round 3's
    // verdict: "reruns: 8" on every first use); the frames behind it start from what it learnt. A kind is settled once
    // a frame of it has run with everything it needed (no re-run, no change of supertile level) — or, at the latest,
    // after LEARN_MAX frames of it IN TOTAL were completed early (a frame that always re-runs, a level that oscillates:
    // bounded, not a phase the context can stay in — counted per kind since round 6: round 5 counted frames "in a row",
    // and a host that alternated two kinds that never ran clean reset the streak with every frame and stayed blocking). The set NEVER FORGETS (round 4 kept the last 16 kinds FIFO and
    // hashed the cloud's ADDRESS and the raw scale bits into the kind: a host with more than 16 clouds or settings, a new
    // cloud handle per frame or an animated global_scale lost its pipelining for good, silently), and every kind keeps
    // its own supertile level, so a context that alternates between kinds does not run each at the other's level.
    // midround: frames of the kind run the rasteriser's mid-round-exit instantiation whatever their supertile level — set
    // (with hysteresis) from the share of a completed frame's tiles that ended SATURATED rather than at the end of their
    // lists (round 6: a cloud of opaque surfaces saturates its tiles inside a staging round like a dense one does)
    struct KindState { uint32_t sup_level = 1; bool midround = false; };
    std::unordered_map<uint64_t, KindState> kinds;   // the kinds settled on (8 + 4 bytes each; reset by bgs_reset_adaptive_state)
    static constexpr uint32_t LEARN_MAX = 3;
    // share of a frame's tile work in tiles that ended saturated at/above which frames of its kind run the mid-round-exit
    // rasteriser, and at/below which they stop (hysteresis; measured: profiles/r6_notes.md §9)
    static constexpr double MIDROUND_ON = 0.30, MIDROUND_OFF = 0.15;
    uint64_t cur_kind = 0;          // kind of the most recently enqueued frame (ctx->sup_level is that kind's level)
    std::unordered_map<uint64_t, uint32_t> learning;   // kinds being learnt: frames of each completed early so far
    uint64_t early_frames = 0;      // async frames completed inside their bgs_render call (bgs_learning_counters)
    uint4* tile_trace = nullptr;  // bgs_set_tile_trace: caller-owned device buffer the rasteriser's TRACE instantiation fills
    bool use_graphs = false;  // async BINNING_SCAN frames replay a captured hipGraph (bgs_set_graphs)
    uint64_t graph_captures = 0, graph_replays = 0;
    uint64_t cost_frames = 0, ordered_frames = 0, order_refreshes = 0;   // frames that left tile costs / drew their raster workgroups in cost order / made the order anew

    bool have_stats = false;
    bgs_stats stats{};
    uint32_t regrow_count = 0;
};

namespace bgs_host {
// (bgs_frame.hip)
int fail(bgs_ctx* ctx, int status, const std::string& msg);   // records the message, returns the status
int assign_streams(bgs_ctx* ctx);
int lane_create(bgs_ctx* ctx, Lane& L);
void lane_destroy(Lane& L);
int finish_lane(bgs_ctx* ctx, Lane& L);                        // completes the lane's frame in flight (re-runs it if a capacity was short)
int finish_all(bgs_ctx* ctx);                                  // ... every lane's
int collect_stats(bgs_ctx* ctx);
int ensure_scratch(bgs_ctx* ctx, Lane& L, uint32_t n, uint64_t inst_cap);
int ensure_entries(bgs_ctx* ctx, Lane& L, uint32_t n);
int run(bgs_ctx* ctx, const bgs_cloud* cloud, const bgs_view* view, const bgs_settings* s, bool render);   // bgs_sort / bgs_render
extern int g_queue_holders_mode;           // bgs_set_queue_holders
extern std::mutex g_queue_holders_mutex;
}  // namespace bgs_host

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(ctx, BGS_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)
