// kernels.h — host-callable launchers of the HIP kernels (one per stage of the hot path).
// Every launcher only enqueues work on `stream`; none synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgs_device.h"

namespace bgs {

// Device image of an uploaded planar cloud (src/gaussian/formats/planar_3d.rs:45-54). At the boundary the cloud is
// the reference's planes; in HBM it is TWO arrays, laid out for who reads what:
//   position_visibility[n]   16 B per splat, read by keygen for ALL n splats, coalesced (the only plane the sort touches)
//   packed[n][stride]        everything the vertex stage needs for ONE splat in one aligned record, read by project_*
//                            for the VISIBLE splats only, at sorted (= random) indices. The memory side fetches whole
//                            128-byte lines (round 3's gather calibration, profiles/r3_*/gather_fetch_calibration.txt:
//                            a 16-byte gather moves 128 B, FETCH_SIZE tallies it as 64), so three gathers from three
//                            planes (position 16 B, rotation + scale 32 B, SH 192 / 96 B) cost 4 lines per f32 splat and
//                            3-4 per f16 splat — 2.1x / 3.1x the bytes asked for (measured: 5 M-splat f16 frame, 288 MB
//                            fetched for 93 MB of gathers, project+bin at 3.2 TB/s). One record per splat is
//                              f32:   pos_vis | rotation | scale_opacity | sh[48] | pad   = 256 B = 2 lines
//                              cov3d: pos_vis | cov3d[6] opacity pad     | sh[48] | pad   = 256 B = 2 lines
//                              f16:   pos_vis | rot scale opacity (8 x f16) | sh[48] f16  = 128 B = 1 line
//                            (the position is duplicated into the record so the vertex stage needs no second line).
// Memory: 272 B per f32 splat instead of 240, 144 instead of 128 for f16 — on a 288 GB device the price of the
// duplicate position plane is nothing next to a halved (f32) or thirded (f16) gather traffic.
struct CloudPtrs {
    const float4* position_visibility;  // n
    const uint4* packed;                // n * packed_v4 16-byte words, record i at packed + i * packed_v4 (aligned to its size)
    uint32_t packed_v4;                 // 16 (f32, cov3d: 256-byte records) or 8 (f16: 128-byte records)
    uint32_t n;
    uint32_t format;                    // cloud format: 0 = f32 planes, 1 = f16 planes, 2 = f32 with precomputed covariance
};
// 16-byte word offsets inside a packed record
constexpr uint32_t PACK_ROT = 1u, PACK_SCALE_OPACITY = 2u, PACK_SH_F32 = 3u;   // f32
constexpr uint32_t PACK_COV3D = 1u;                                            // cov3d (2 words), SH at PACK_SH_F32
constexpr uint32_t PACK_RSO_F16 = 1u, PACK_SH_F16 = 2u;                        // f16
constexpr uint32_t CLOUD_F32 = 0, CLOUD_F16 = 1, CLOUD_COV3D = 2;

// What the rasteriser of a BINNING_SCAN frame tidies up so that the NEXT frame of the same lane needs
// neither a memset of the scratch region nor a device-to-host copy of the Control block (each a
// separate blit kernel with its own launch gap, ~13 us per frame on one stream). Every status word the
// frame's chained scans used is dead once project_bin has finished, so the raster threads zero them
// at their start; the lane alternates between two Control blocks, and this frame's rasteriser zeroes
// the OTHER one (last used by the lane's previous frame) and copies the header + coarse totals of its
// own, final since project_bin, to pinned host memory. No counters, no atomics (a "last wave done"
// counter was tried first: 8160 same-address atomics cost 65 us). All-null = no cleanup.
struct FrameCleanup {
    uint32_t* part_status;    // keygen chain: ceil(n / KEYGEN_TILE) words used
    uint32_t* depth_status;   // depth passes: places x [pass_stride words], ceil(D / depth_tile) * 256 used each
    uint32_t* bin_status;     // bin_kernel's chains: ceil(D / 1024) * MAX_SUPERTILES words used
    Control* other_ctl;       // the lane's other Control block: zeroed here
    Control* host_ctl;        // pinned host Control (device-visible)
    uint32_t pass_stride;     // words between the status arrays of consecutive depth passes
    uint32_t places;
    uint32_t depth_tile;      // keys per onesweep tile of the depth passes
    // quantile keys of this frame's sorted list -> host_ctl->splitters (keygen key space = key ^ key_xor)
    const uint2* sorted;
    uint32_t key_xor;
    uint32_t split_sub;       // the table this frame leaves has 256 * split_sub - 1 quantile keys (1 .. BUCKET_SUB_MAX)
    // what tile_order_kernel counted in the completed frame's cost plane the lane's order was made of (tile_order_stats_offset;
    // null: no order): the clean-up block hands the sums to the host — what decides whether frames of this kind run the
    // mid-round-exit rasteriser
    const uint32_t* order_stats;
};

// Onesweep geometry: 256 threads x KPT keys per tile.
constexpr int SORT_THREADS = 256;
constexpr int SORT_KPT_SMALL = 8;   // 2048-pair tiles: more tiles for <= ~4M keys
constexpr int SORT_KPT_LARGE = 16;  // 4096-pair tiles
inline uint32_t sort_tile_size(bool large) { return SORT_THREADS * (large ? SORT_KPT_LARGE : SORT_KPT_SMALL); }

// keygen + fused global digit histograms (radix_sort_a, src/sort/radix.wgsl:71-107) + stable
// partition: drawable entries -> `entries` (index order), culled-sentinel entries -> `culled`
// (index order); ctl->draw_count = number of drawable entries. part_status: zeroed chain words,
// one per 2048-splat tile.
// fp_out (device): the kernel also leaves a copy of `fp` there for the kernels behind it, which take
// FrameParams by pointer — so one node of a captured frame carries everything that changes per frame.
struct KeygenLaunch {
    FrameParams fp;
    const float4* pos;
    uint2* entries;
    uint2* culled;
    Control* ctl;
    uint32_t* part_status;
    uint32_t places;
    uint32_t ticket_slot;
    FrameParams* fp_out;
    uint2* bucket_slots;  // fp.sort_path == 1: [BUCKET_COUNT * split.sub][BUCKET_CAP] slot regions
    uint32_t* bucket_status;  // (unused since round 5: bucket placement needs no chains)
    SplitterTable split;      // fp.sort_path == 1: bucket = number of key[0 .. 256 * sub - 2] <= key
    uint32_t* zero_word;      // a word the frame needs zeroed before its later kernels run (the rasteriser's heavy-tile count), or null
    bool wide = false;        // the frame is alone on the chip (pipeline depth 1): chainless tiles as 1024-thread workgroups
    // filled by prepare(): the launch geometry and the argument vector (points into this object)
    const void* func;
    uint32_t blocks, threads;
    void* argv[13];
    uint32_t lds_bytes;   // dynamic LDS of the launch (the per-bucket counters of a bucket frame)
    bool prepare(int max_blocks);  // false: nothing to launch (n == 0)
    hipError_t launch(hipStream_t stream);
    hipError_t update_node(hipGraphExec_t exec, hipGraphNode_t node);  // same launch, as a graph node update
};
constexpr uint32_t KEYGEN_TILE = 2048;  // the smallest tile: what the chain words are allocated and zeroed for
// Splats per keygen tile (= per block and ticket) for a cloud of n splats. The chained scans behind a tile cost
// per TILE, so tiles grow with the cloud: 2048 (256 threads x 8) for small clouds, 4096 (256 x 16) from 2^19 splats.
// BGS_KEYGEN_WIDE_THREADS = 1024 builds the wide tiles as 1024 threads x 4 (and 8192-splat tiles, 1024 x 8, from
// 2^22 splats): alone on the chip that keygen is faster (1 M splats 28.6 -> 24.3 us, 5 M 66 -> 62 us), but a
// 16-wave workgroup needs a whole CU's worth of free wave slots and LDS at once, and with the frames of six lanes
// in flight it — like a 1024-thread bucket sort, 27.9 -> 15.3 us alone at 600 k pairs — costs throughput:
// 16.3 k (256 / 256) vs 15.8 k (1024 / 1024) frames/s on the headline frame, 6.9 k vs 6.7 k on the 5 M cloud
// (same-box A/B, profiles/r2_notes.md). The defaults are what the pipelined frame rate wants.
#ifndef BGS_KEYGEN_WIDE_THREADS
#define BGS_KEYGEN_WIDE_THREADS 256
#endif
#ifndef BGS_KEYGEN_HUGE_LOG2
#define BGS_KEYGEN_HUGE_LOG2 (BGS_KEYGEN_WIDE_THREADS == 256 ? 31 : 22)
#endif
inline uint32_t keygen_tile_splats(uint32_t n) { return n >= (1u << BGS_KEYGEN_HUGE_LOG2) ? 8192u : (n >= (1u << 19) ? 4096u : 2048u); }

// Standalone digit histograms of existing pairs (used by bgs_radix_sort_pairs).
void launch_histogram(hipStream_t stream, const uint2* pairs, uint32_t n, uint32_t* hist /*[4][256]*/,
                      uint32_t passes);

// One stable 8-bit LSD pass (Onesweep: chained-scan look-back, LDS-ranked scatter).
//   n_ptr      device word holding the pair count
//   hist       256 global digit counts for this pass (un-scanned)
//   status     zeroed look-back words [ceil(max_n / tile)][256]
//   ticket     zeroed dynamic tile counter
//   key_xor    XOR-ed into the key on output (un-inverts SORT_RAYON keys in the final pass)
void launch_onesweep_pass(hipStream_t stream, const uint2* in, uint2* out, const uint32_t* n_ptr,
                          uint32_t max_n, const uint32_t* hist, uint32_t* status, uint32_t* ticket,
                          uint32_t* error_flag, uint32_t shift, uint32_t key_xor, bool large_tiles,
                          int max_blocks);

// Bucket sort of the drawable pairs keygen placed (fp.sort_path == 1): one launch instead of the digit
// passes, workgroup = bucket: each bucket (<= BUCKET_CAP pairs) is sorted by (key, index) in LDS and written
// to out[] at its offset; out[0 .. draw_count) is then bit-identical to what the stable LSD passes
// produce. Sets ctl->sort_overflow instead when a bucket holds more than BUCKET_CAP pairs or one key value
// repeats more than BUCKET_FINE_MAX times (the host re-runs the frame with the onesweep passes).
// wide (round 6, SplitterTable::wide): the buckets are slot regions of BUCKET_CAP_WIDE pairs (1024-thread workgroups, 128 KB of LDS).
hipError_t launch_bucket_sort(hipStream_t stream, const uint2* bucket_slots, uint2* out, Control* ctl, uint32_t key_xor,
                              uint32_t buckets = BUCKET_COUNT /* 256 * sub */, bool wide = false);
// ctl->splitters = the 255 quantile keys of the sorted draw list (for frames without a cleaning rasteriser).
void launch_splitters(hipStream_t stream, const uint2* sorted, Control* ctl, uint32_t key_xor, uint32_t sub = 1u);

// Vertex stage in front-to-back order + ordered tile-instance emission
// (vs_points once per splat, src/render/gaussian.wgsl:184-436).
void launch_project_emit(hipStream_t stream, const FrameParams& fp, const CloudPtrs& cloud,
                         const uint2* draw_list, const uint2* culled, Control* ctl, unsigned long long* scan_status,
                         void* records, uint2* instances, uint32_t capacity, uint32_t ticket_slot,
                         int max_blocks);

// BINNING_SCAN: the vertex stage in front-to-back order (project_kernel: rank -> record + packed tile rectangle, no
// ordering between ranks) and the ORDERED coarse binning (bin_kernel: every rank is appended, in rank order, to the list
// of each supertile (sup_edge x sup_edge tiles) its tile rectangle overlaps; one pass, no sort, no atomics on the data
// path: the <= 256 supertiles are the "digits" of the same chained-scan look-back the radix sort uses), two launches.
// d_fp: the device copy of `fp` the kernels read (written by this frame's keygen). rects: 4 bytes per rank.
// wide_bin: the frame is alone on the chip (pipeline depth 1): bin_kernel as 1024-thread workgroups (else 256).
void launch_project_bin(hipStream_t stream, const FrameParams& fp, const FrameParams* d_fp, const CloudPtrs& cloud,
                        const uint2* draw_list, const uint2* culled, Control* ctl, uint32_t* bin_status, void* records,
                        uint32_t* rects, uint32_t* coarse, uint32_t coarse_cap, uint32_t sup_edge,
                        uint32_t ticket_slot, int project_blocks, int bin_blocks, bool wide_bin);

// Tile rasteriser for BINNING_SCAN: walks the supertile's ordered list, keeps the ranks whose
// rectangle contains this tile (order-preserving ballot compaction), stages their records in LDS
// and composites front-to-back until the tile saturates. With want_srgb8 it also writes the frame as
// Rgba8UnormSrgb (to d_fp->srgb8_target, else srgb8_default): no separate encode pass for this path.
void launch_raster_scan(hipStream_t stream, const FrameParams& fp, const FrameParams* d_fp, const void* records,
                        const uint32_t* coarse, uint32_t coarse_cap,
                        uint32_t sup_edge, Control* ctl, float4* framebuffer, uint32_t* srgb8_default,
                        uint32_t out_format, const FrameCleanup& cleanup, uint4* tile_trace = nullptr,
                        int mode = 0 /* raster_scan_kernel's MODE: 0 plain, 1 dense + mid-round exit, 2 sparse + mid-round exit */,
                        const uint8_t* heavy_in = nullptr, uint8_t* heavy_out = nullptr,
                        const uint16_t* order = nullptr, uint16_t* cost_out = nullptr);
// TileCost: what every tile of a frame cost its wave — records blended, records staged, staging rounds and candidate
// groups scanned, weighted into eighths of a blended record (render_kernels.hip WORK_*; u16 per tile) — left by the
// rasteriser (cost_out) for the frames behind it, and the order made of it for a frame's raster workgroups
// (launch_tile_order: u16 per workgroup of four tiles, a permutation whatever the costs hold — see tile_order_kernel).
// Work counts, not the waves' lifetimes: the same frame leaves the same costs whatever else runs on the chip, and the
// order made of lifetimes was the worse one (a wave of the launch's first round shares its SIMD with four others and
// looks heavier than it is). Frames with more tile waves than the chip holds at once (4x multisampling) end earlier
// for it when they are alone on the chip: the rasteriser of the dense / scene-like 1 M frame 68.4 -> 61.0 / 128 ->
// 116 us, of the 1 M-surfel frame 662 -> 522 us, of the scene-like 5 M frame 538 -> 504 us
// (profiles/r4_experiments/tile_order.txt); with frames in flight it is worth under 1 %. `fp` and `midround_exit`
// pick the share each XCD has in the frame's instantiation (raster_scan_kernel: RUNS). An order stays valid — a
// permutation of the workgroups — for as long as the tile grid does, so it is made anew every TILE_ORDER_REFRESH-th
// frame only (the kernel is a chain of ~40 barriers, ~7 us).
constexpr uint32_t TILE_ORDER_REFRESH = 8u;
inline size_t tile_cost_bytes(uint32_t tiles) { return ((size_t)tiles + 4u) * 2u; }
// the order (u16 per workgroup) and, behind it, what tile_order_kernel summed over the cost plane it read: per XCD share x
// the pair (work of all tiles, work of the tiles that ended saturated) at stats[2 x], stats[2 x + 1] (TileCost units)
__host__ __device__ inline size_t tile_order_stats_offset(uint32_t tiles) { return (((((size_t)tiles + 3u) / 4u + 8u) * 2u + 63u) / 64u) * 64u; }
inline size_t tile_order_bytes(uint32_t tiles) { return tile_order_stats_offset(tiles) + 16u * sizeof(uint32_t); }
void launch_tile_order(hipStream_t stream, const uint16_t* cost, uint16_t* order, uint32_t ntiles, const FrameParams& fp,
                       bool midround_exit);
// the same with the runs per XCD given (1, 2 or 4; bgs_selftest_tile_order)
void launch_tile_order_runs(hipStream_t stream, const uint16_t* cost, uint16_t* order, uint32_t ntiles, uint32_t runs);
// tile waves of raster_scan_kernel a SIMD holds at once for this frame's instantiation (its __launch_bounds__)
int raster_scan_waves_per_simd(const FrameParams& fp);
// HeavyFeedback: what the rasteriser of a dense frame (midround_exit) leaves for the frames behind it — the tiles that
// did not saturate inside their first staging round, as a list (for the strip workgroups) and as a flag per tile (for the
// regular waves, which step aside). One buffer: [count (own 128-byte line) | list u16[HEAVY_CAP] | flag u8[tiles]].
// The count is zeroed by the producing frame's keygen; a frame only ever reads the buffer of a COMPLETED frame
// (bgs_frame.hip hands out the pointer in finish_lane), so list, flags and count are consistent by construction.
constexpr uint32_t HEAVY_CAP = 256u, HEAVY_LIST_OFFSET = 128u, HEAVY_FLAGS_OFFSET = HEAVY_LIST_OFFSET + 2u * HEAVY_CAP;
inline size_t heavy_feedback_bytes(uint32_t tiles) { return (size_t)HEAVY_FLAGS_OFFSET + tiles; }
// out_format bits: which packed image the rasteriser writes next to (or instead of) the f32 target
constexpr uint32_t OUT_SRGB8 = 1u;     // Rgba8UnormSrgb, 4 B per pixel
constexpr uint32_t OUT_RGBA16F = 2u;   // Rgba16Float, 8 B per pixel (the reference's hdr target)
constexpr uint32_t OUT_SKIP_F32 = 4u;  // do not write the f32 target (bgs_set_packed_only)

// Per-tile [start, end) over the tile-sorted instances; ranges indexed by (ty << 8 | tx).
void launch_tile_ranges(hipStream_t stream, const uint2* instances, const Control* ctl, uint2* ranges);

// Per-16x16-tile front-to-back blend (fs_main + blend, src/render/gaussian.wgsl:438-505,
// src/render/mod.rs:944-948).
void launch_raster(hipStream_t stream, const FrameParams& fp, const void* records,
                   const uint2* instances, const uint2* ranges, float4* framebuffer,
                   const float clear_color[4], const Control* ctl);

// Rgba8UnormSrgb image of the f32 framebuffer (the reference's render-target format).
// The destination is d_fp->srgb8_target when that is non-zero, else default_out.
void launch_encode_srgb8(hipStream_t stream, const float4* framebuffer, uint32_t* default_out, uint32_t pixels,
                         const FrameParams* d_fp, uint32_t out_format = 1u);

// One-time re-layout at upload: the boundary's planes (device copies) -> packed records (see CloudPtrs). Word w of
// record i is pos[i] for w = 0, then the v4_a / v4_b / v4_c 16-byte words of splat i from planes a, b, c in that order,
// then zero padding up to stride_v4. Coalesced reads and writes.
void launch_pack_cloud(hipStream_t stream, const uint4* pos, const uint4* a, uint32_t v4_a, const uint4* b, uint32_t v4_b,
                       const uint4* c, uint32_t v4_c, uint4* out, uint32_t stride_v4, uint32_t n);

// STREAM-triad on float4: a = b + s * c (HBM ceiling probe, bgs_hbm_probe).
void launch_triad(hipStream_t stream, float4* a, const float4* b, const float4* c, float s, size_t n4,
                  int blocks);

// Device self-test of ln_f32_cr (exact_log.h) over the binary32 bit patterns first_bits .. first_bits + count - 1:
// out[i] (may be null) = ln of pattern i; *sum += sum_i mix(bits(in_i), bits(out_i)) (ln_selftest_mix, wrap-around).
void launch_selftest_ln(hipStream_t stream, uint32_t first_bits, uint32_t count, float* out, unsigned long long* sum,
                        int blocks);

}  // namespace bgs
