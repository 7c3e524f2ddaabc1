/*
 * bgs_diag.h — diagnostics, test hooks and experiment switches of libbgs.
 *
 * NOT part of the drop-in seam (include/bgs.h: context, cloud upload, bgs_sort, bgs_render, targets, frame
 * pipeline). Nothing here is needed to replace the reference's sort + rasterize path; these entry points exist
 * for the parity tests, bench.py's roofline legs and the profiling scripts under scripts/. Same conventions as
 * bgs.h (plain C, negative bgs_status on failure, bgs_last_error).
 */
#ifndef BGS_DIAG_H
#define BGS_DIAG_H

#include "bgs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel-ablation switches for performance experiments (scripts/ablate.py). Bits 1..64 switch parts
 * of kernels off and produce WRONG images; 0x1000 (per-frame memset + Control copy instead of the
 * rasteriser's in-kernel clean-up), 0x2000 (no draw-count hint for the sort grids), 0x4000 (no
 * hipGraph replay even when bgs_set_graphs is on), 0x10000 / 0x8000 / 0x400000 / 0x800000 (force supertile
 * level 0 / 1 / 2 / 3 instead of choosing by the completed frames' list statistics), 0x40000 (sRGB8 image from the
 * separate encode pass instead of the rasteriser's fused output), 0x80000 (depth sort always by the
 * onesweep digit passes, never the bucket sort), 0x200000 (bucket sort even before a completed frame has
 * told the key range: full 32-bit range guessed), 0x100000 (supertile lists start at 64 entries, to
 * exercise the overflow -> re-run path), 0x200 / 0x400 (the bucket sort with at least 768 / 1280 buckets whatever the list's
 * length: the finer splitter tables, in keygen's arguments / in the lane's device table) keep images correct and exist for
 * A/B timing and tests.
 * 0x8000000: every BINNING_SCAN frame is run twice, as if a data-dependent capacity had been too small (exercises the
 * re-run path). 0x10000000: no tile-cost feedback / cost-ordered raster workgroups; 0x20000000: none at pipeline
 * depths > 1; 0x40000000: the order is made anew with every frame (default: every 8th).
 * 0x100 / 0x800: the bucket sort with narrow (4096-pair) / wide (16 384-pair) buckets whatever the list's length (default: wide
 * past 1.57 M drawable pairs). 0x1000000: never the rasteriser's mid-round-exit instantiations; 0x20000: always one of them,
 * whatever the supertile level and the kind's saturation share (the dense frames' at level >= 2, the sparse frames' below);
 * 0x2000000 / 0x4000000: the heavy-tile strip workgroups never / at any pipeline depth (default: depth 1 only).
 * Bits 1..64 exist only in libraries built with -DBGS_ABLATION=1 (scripts/build_variant.sh); the production library
 * ignores them. Production code leaves this at 0. */
int bgs_set_debug_flags(bgs_ctx* ctx, uint32_t flags);

/* What the adaptive machinery has done since bgs_create: out[0] frames enqueued on the bucket sort path,
 * [1] on the onesweep passes (both counts include re-runs), [2] frames re-run because the bucket sort gave
 * up, [3] because a supertile list overflowed, [4] because the tile-instance buffer was too small,
 * [5] supertile level changes, [6] the current level, [7] the current list-capacity hint (entries). */
int bgs_adaptive_counters(bgs_ctx* ctx, uint64_t out[8]);

/* The learning phase (bgs.h "Async frames"): how many async frames were completed inside their bgs_render
 * call because the context had not settled on their kind of frame yet, and how many kinds it has settled on. A host
 * whose frames/s fall to the blocking rate sees it here: early_frames grows with every frame. */
int bgs_learning_counters(bgs_ctx* ctx, uint64_t* early_frames, uint64_t* kinds_settled);

/* Stable LSD radix sort of n (key,index) pairs on the device, `passes` 8-bit digit
 * places starting at bit 0 (the Onesweep kernel used for both the depth and the tile
 * sort). entries_inout is a HOST buffer; used by the parity tests to exercise the sort
 * kernel on arbitrary keys (ties, all-equal, ragged sizes). */
int bgs_radix_sort_pairs(bgs_ctx* ctx, bgs_sort_entry* entries_inout, uint32_t n,
                         uint32_t passes);

/* Measured HBM ceiling of this device, for the roofline ("state both" the nominal and
 * the measured peak): `bytes` per buffer (rounded down to 16), `iters` timed repetitions.
 * copy_gbs = hipMemcpyDtoD rate counting read + write; triad_gbs = a[i] = b[i] + s * c[i] with
 * float4 accesses, counting 2 reads + 1 write. Allocates 3 * bytes for the call. */
int bgs_hbm_probe(bgs_ctx* ctx, uint64_t bytes, uint32_t iters, float* copy_gbs, float* triad_gbs);

/* Device self-test of the correctly rounded natural logarithm behind the adaptive cutoff
 * (src/render/gaussian.wgsl:229-235; csrc/exact_log.h: the one transcendental that reaches a cull decision).
 * Evaluates it ON THE DEVICE for the `count` binary32 bit patterns first_bits, first_bits + 1, ...:
 * host_out (may be NULL) receives the results; checksum_out (may be NULL) the wrap-around sum over the inputs of
 * mix((in_bits << 32 | out_bits)) with mix(v) = (v * 0x9E3779B97F4A7C15, v ^= v >> 29, v * 0xBF58476D1CE4E5B9),
 * so that a caller can compare 2^31 results with its own without moving them. */
int bgs_selftest_ln_f32(bgs_ctx* ctx, uint32_t first_bits, uint32_t count, float* host_out, uint64_t* checksum_out);

/* The library parks three idle "queue holder" streams per DEVICE (process-global, created once, however many
 * contexts the process has) before a context creates its own streams, so that the HIP runtime deals those out one per
 * hardware queue (see above). 0 switches that off for contexts that have not created their streams yet — for a
 * process whose other streams already hold the queues (e.g. RCCL's after a process group was initialised); 1 forces
 * it on; -1 (default) follows the environment variable BGS_QUEUE_HOLDERS (unset or non-zero: on). Never fails. */
int bgs_set_queue_holders(int enabled);

/* Diagnostics: per-tile trace of the default (BGS_BINNING_SCAN) rasteriser. With a non-NULL device buffer of
 * tiles_x * tiles_y * 32 bytes, every following frame runs the rasteriser's instrumented instantiation, in which each
 * tile's wave writes 8 uint32: s_memtime at its start (lo, hi) and end (lo, hi), the HW_ID and XCC_ID registers (which
 * XCD / SE / CU / SIMD / wave slot it ran on), the list candidates it scanned, and records blended | staged << 16.
 * scripts/tile_trace.py turns that into the launch's per-SIMD occupancy picture (the "tail"). NULL switches it off.
 * Completes the frames in flight; the buffer stays the caller's. Costs ~10 % of the rasteriser's time while on.
 * There is no traced instantiation with a depth buffer: bgs_render with bgs_view.depth_device_ptr set while a trace
 * buffer is set fails with BGS_EINVAL (it used to run untraced and leave the buffer's old contents). */
int bgs_set_tile_trace(bgs_ctx* ctx, void* device_ptr);

/* How many frames were captured into a graph / replayed from one since bgs_create. */
int bgs_graph_counters(bgs_ctx* ctx, uint64_t* captures, uint64_t* replays);

/* How many frames (re-runs included) left per-tile costs for the frames behind them / drew their raster workgroups in the
 * order made of a completed frame's costs (frames with more tile waves than the chip holds at once) /
 * made that order anew (tile_order_kernel launches). */
int bgs_tile_order_counters(bgs_ctx* ctx, uint64_t* cost_frames, uint64_t* ordered_frames, uint64_t* refreshes);

/* tile_order_kernel on caller-supplied per-tile costs (host_cost[ntiles], u16: work in bits 0-14, bit 15 = the tile ended
 * saturated; 1 <= ntiles <= 65535; runs per XCD 1, 2 or 4): host_order[(ntiles + 3) / 4] receives the raster workgroups'
 * order — a permutation of 0 .. (ntiles + 3) / 4 - 1 whatever the costs hold, heaviest workgroup (its heaviest tile) first
 * inside every XCD's share — and host_sums[2] (may be NULL) the kernel's sums: the work of all tiles, the work of the tiles
 * that ended saturated. Test hook. */
int bgs_selftest_tile_order(bgs_ctx* ctx, const uint16_t* host_cost, uint32_t ntiles, uint32_t runs, uint16_t* host_order,
                            uint32_t* host_sums);

#ifdef __cplusplus
}
#endif
#endif /* BGS_DIAG_H */
