// sort_kernels.hip — depth keys + device radix sort (gfx950, wave64).
//
// Replaces the reference's GPU radix sort (src/sort/radix.wgsl, host src/sort/radix.rs):
//   radix_reset + radix_sort_a   -> keygen_kernel (keys, pairs, 4 digit histograms, LDS-privatised)
//   radix_sort_b                 -> folded into every onesweep block (256-bin exclusive scan)
//   radix_sort_c_{count,scan,scatter} x places
//                                -> onesweep_kernel x places: ONE kernel per digit place, each
//                                   pair read once and written once (16 B/pair/pass).
// Output contract is the reference's: ascending key, ties by ascending input position
// (stable LSD, 8-bit digits), so the final order is bit-identical to radix.wgsl's.
//
// Onesweep pass, per 256-thread block and per tile of 256*KPT pairs (tiles are handed out by
// an atomic ticket, so a tile's predecessors have always started — no dispatch-order assumption):
//   1. coalesced load, wave-striped (wave w owns KPT consecutive 64-pair rows)
//   2. per row: 8 ballots -> same-digit lane mask -> rank = popc(mask & lanes_below); the row
//      leader bumps the wave's private LDS digit counter (no LDS atomics)
//   3. per digit (thread = digit): scan over the 4 waves, publish the tile aggregate, decoupled
//      look-back over predecessor tiles (relaxed agent-scope 4-byte words: the data is the flag)
//   4. scatter pairs into LDS in digit order, then write runs to HBM coalesced.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "lookback.h"
#include "splat_math.h"

namespace bgs {

namespace {

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// Exclusive scan of one value per thread over a 256-thread block. s_tot: 4 LDS words.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* s_tot,
                                                             uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) s_tot[wave] = inc;
    __syncthreads();
    const uint32_t w0 = s_tot[0], w1 = s_tot[1], w2 = s_tot[2], w3 = s_tot[3];
    const uint32_t woff = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return woff + inc - v;
}


// key-range bucket of a depth key: monotone (non-decreasing) in the key for every lo / shift
__device__ __forceinline__ uint32_t bucket_of(uint32_t key, uint32_t lo, uint32_t shift) {
    const uint32_t d = key > lo ? key - lo : 0u;
    return min(d >> shift, BUCKET_COUNT - 1u);
}

}  // namespace

// ---------------------------------------------------------------------------------------
// keygen + stable partition: one thread per splat, 16 B read + 8 B write per splat, coalesced.
//
// Entries whose key is the "culled" sentinel (all ones >> shift) all compare equal and sort to
// the END of the reference's output in ascending index order. So instead of dragging them through
// every radix pass, they are split off here, in index order, into `culled`, and only the V'
// drawable entries (also kept in index order, which the stable LSD passes need for ties) go to
// `entries` and into the digit histograms. sorted(drawable) ++ culled is bit-identical to the
// reference's full stable sort. The ordered split is a chained scan over 2048-splat tiles.
// ---------------------------------------------------------------------------------------
// BUCKET (fp.sort_path == 1): instead of the index-ordered list + digit histograms that the onesweep
// passes need, the drawable pairs are scattered into BUCKET_COUNT key-range buckets (fixed slot regions of
// BUCKET_CAP pairs, a returning atomic per pair on the bucket's counter; order inside a bucket is
// arbitrary — bucket_sort_kernel orders by (key, index), which is what "stable" means for these pairs).
template <int KG_ITEMS, bool BUCKET>  // splats per thread
__global__ __launch_bounds__(256) void keygen_kernel(FrameParams fp, const float4* __restrict__ pos,
                                                     uint2* __restrict__ entries,
                                                     uint2* __restrict__ culled, Control* ctl,
                                                     uint32_t* part_status, uint32_t places,
                                                     uint32_t ticket_slot, FrameParams* fp_out,
                                                     uint2* __restrict__ bucket_slots) {
    __shared__ uint32_t s_hist[BUCKET ? 1 : 4][RADIX_BASE];
    __shared__ uint32_t s_cnt[KG_ITEMS][4];  // drawable per (row, wave)
    __shared__ uint32_t s_keys[256 * KG_ITEMS];  // the tile's drawable keys, compacted (for the histograms / the scatter)
    __shared__ uint32_t s_idx[BUCKET ? 256 * KG_ITEMS : 1];  // their splat indices (BUCKET)
    __shared__ uint32_t s_minmax[2][4];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The frame's first kernel is the only one that receives FrameParams by value: it leaves a copy
    // in device memory for the kernels behind it, so that a captured frame (hipGraph) is re-aimed at
    // a new view by updating this one node's arguments.
    if (fp_out && blockIdx.x == 0 && (uint32_t)tid < (uint32_t)(sizeof(FrameParams) / 4u))
        reinterpret_cast<uint32_t*>(fp_out)[tid] = reinterpret_cast<const uint32_t*>(&fp)[tid];
    if constexpr (!BUCKET) {
#pragma unroll
        for (int p = 0; p < 4; ++p) s_hist[p][tid] = 0u;
    }
    uint32_t kmin_inv = 0u, kmax = 0u;  // over this block's drawable keys (~min so that 0 = nothing seen)
    const uint32_t sentinel = KEY_CULLED >> fp.key_shift;
    const uint32_t per_tile = 256u * KG_ITEMS;
    const uint32_t num_tiles = (fp.n + per_tile - 1u) / per_tile;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    // Tiles are handed out by ticket so that a tile's predecessors in the chained scan are always owned
    // by running blocks (kernels of several lanes share the chip). Same-address device-scope atomics
    // retire at ~8 ns each, so when the grid covers every tile a block takes ONE ticket and leaves
    // after its tile instead of queueing for a second ticket only to be told there is nothing left.
    const bool single_shot = gridDim.x >= num_tiles;

    for (;;) {
        if (tid == 0) s_tile = atomicAdd(&ctl->ticket[ticket_slot][0], 1u);
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= num_tiles) break;
        const uint32_t base = tile * per_tile;
        uint32_t key[KG_ITEMS], below[KG_ITEMS];
        bool draw[KG_ITEMS];
        // all of the tile's position loads are issued before the first key is computed: with one
        // 4-wave block per CU nothing else hides their latency
        float4 pin[KG_ITEMS];
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
            const uint32_t i = base + (uint32_t)k * 256u + (uint32_t)tid;
            pin[k] = i < fp.n ? pos[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
            const uint32_t i = base + (uint32_t)k * 256u + (uint32_t)tid;
            key[k] = sentinel;
            draw[k] = false;
            if (i < fp.n) {
                const float4 p = pin[k];
                key[k] = sort_key(fp, V3{p.x, p.y, p.z});
                // entries that reach the vertex stage: everything unless the radix key is "culled"
                draw[k] = fp.sort_mode != SORT_RADIX || key[k] != sentinel;
            }
            const unsigned long long b = __ballot(draw[k]);
            below[k] = (uint32_t)__popcll(b & lanes_below);
            if (lane == 0) s_cnt[k][wave] = (uint32_t)__popcll(b);
        }
        __syncthreads();
        // exclusive offsets in (row, wave, lane) = index order
        uint32_t off[KG_ITEMS];
        uint32_t run = 0u;
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w == wave) off[k] = run;
                run += s_cnt[k][w];
            }
        }
        const uint32_t total = run;
        // Digit histograms from the COMPACTED keys: with the usual ~12 % of a tile drawable, counting
        // in place costs 4 LDS atomics on each of the 16 rows of every wave (3.7 us of the kernel);
        // compacted, a tile's ~600 keys are 3 rows.
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k)
            if (draw[k]) {
                s_keys[off[k] + below[k]] = key[k];
                if constexpr (BUCKET) s_idx[off[k] + below[k]] = base + (uint32_t)k * 256u + (uint32_t)tid;
                kmin_inv = max(kmin_inv, ~key[k]);
                kmax = max(kmax, key[k]);
            }
        if (wave == 0) {  // one chain per block: the whole wave walks it, 64 predecessors per hop
            uint32_t* const my_status = part_status + tile;
            uint32_t excl = 0u;
            if (tile > 0u) {
                if (lane == 0) st_agent(my_status, STATUS_AGGREGATE | total);
                excl = lookback_wave(part_status, tile, lane, &ctl->error, 8u);
            }
            if (lane == 0) {
                st_agent(my_status, STATUS_PREFIX | ((excl + total) & STATUS_VALUE_MASK));
                s_base = excl;
                if (tile == num_tiles - 1u) {
                    ctl->draw_count = excl + total;
                    ctl->splat_count = fp.n;
                }
            }
        }
        __syncthreads();
        if constexpr (BUCKET) {
            // the tile's ~12 % drawable pairs, compacted: full waves of returning atomics, all of a thread's
            // atomics in flight before the first slot is written
            constexpr int ROUNDS = KG_ITEMS;
            uint32_t bk[ROUNDS], at[ROUNDS];
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const uint32_t j = (uint32_t)r * 256u + (uint32_t)tid;
                if (j < total) {
                    bk[r] = bucket_of(s_keys[j], fp.bucket_lo, fp.bucket_shift);
                    at[r] = atomicAdd(&ctl->bucket_count[bk[r]], 1u);
                }
            }
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const uint32_t j = (uint32_t)r * 256u + (uint32_t)tid;
                if (j < total && at[r] < BUCKET_CAP)  // a full bucket is seen by bucket_sort_kernel (count > cap)
                    bucket_slots[(size_t)bk[r] * BUCKET_CAP + at[r]] = make_uint2(s_keys[j], s_idx[j]);
            }
        } else {
            for (uint32_t j = (uint32_t)tid; j < total; j += 256u) {
                const uint32_t kk = s_keys[j];
                for (uint32_t pl = 0; pl < places; ++pl)
                    atomicAdd(&s_hist[pl][(kk >> (RADIX_BITS * pl)) & (RADIX_BASE - 1u)], 1u);
            }
        }
        const uint32_t vis_base = s_base;
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
            const uint32_t i = base + (uint32_t)k * 256u + (uint32_t)tid;
            if (i < fp.n) {
                const uint32_t before = vis_base + off[k] + below[k];  // drawable entries before i
                if (draw[k]) { if constexpr (!BUCKET) entries[before] = make_uint2(key[k], i); }
                else culled[i - before] = make_uint2(key[k], i);
            }
        }
        if (single_shot) break;
        __syncthreads();
    }
    if constexpr (!BUCKET) {
        for (uint32_t pl = 0; pl < places; ++pl) {
            const uint32_t v = s_hist[pl][tid];
            if (v) atomicAdd(&ctl->hist_depth[pl][tid], v);
        }
    }
    // range of the drawable keys: the bucket range of the NEXT frames (two atomics per block)
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) {
        kmin_inv = max(kmin_inv, (uint32_t)__shfl_xor((int)kmin_inv, off2, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off2, 64));
    }
    if (lane == 0) { s_minmax[0][wave] = kmin_inv; s_minmax[1][wave] = kmax; }
    __syncthreads();
    if (tid == 0) {
        const uint32_t a = max(max(s_minmax[0][0], s_minmax[0][1]), max(s_minmax[0][2], s_minmax[0][3]));
        const uint32_t b = max(max(s_minmax[1][0], s_minmax[1][1]), max(s_minmax[1][2], s_minmax[1][3]));
        if (a) atomicMax(&ctl->key_min_inv, a);
        if (b) atomicMax(&ctl->key_max, b);
    }
}

bool KeygenLaunch::prepare(int max_blocks) {
    if (fp.n == 0) return false;
    // 4096-splat tiles once there are enough splats to fill the chip with them: half the tickets and
    // chain hops (measured at 1 M splats: 30.8 -> 27.3 us)
    const bool wide = fp.n >= (1u << 19);
    const uint32_t per_block = 256u * (wide ? 16u : 8u);
    blocks = (fp.n + per_block - 1) / per_block;
    if (blocks > (uint32_t)max_blocks) blocks = (uint32_t)max_blocks;
    if (fp.sort_path == 1u)
        func = wide ? reinterpret_cast<const void*>(&keygen_kernel<16, true>) : reinterpret_cast<const void*>(&keygen_kernel<8, true>);
    else
        func = wide ? reinterpret_cast<const void*>(&keygen_kernel<16, false>) : reinterpret_cast<const void*>(&keygen_kernel<8, false>);
    argv[0] = &fp; argv[1] = &pos; argv[2] = &entries; argv[3] = &culled; argv[4] = &ctl;
    argv[5] = &part_status; argv[6] = &places; argv[7] = &ticket_slot; argv[8] = &fp_out; argv[9] = &bucket_slots;
    return true;
}

hipError_t KeygenLaunch::launch(hipStream_t stream) {
    return hipLaunchKernel(func, dim3(blocks), dim3(256), argv, 0, stream);
}

hipError_t KeygenLaunch::update_node(hipGraphExec_t exec, hipGraphNode_t node) {
    hipKernelNodeParams np{};
    np.func = const_cast<void*>(func);
    np.gridDim = dim3(blocks);
    np.blockDim = dim3(256);
    np.sharedMemBytes = 0;
    np.kernelParams = argv;
    np.extra = nullptr;
    return hipGraphExecKernelNodeSetParams(exec, node, &np);
}

// ---------------------------------------------------------------------------------------
// standalone histogram (test entry point bgs_radix_sort_pairs)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void histogram_kernel(const uint2* __restrict__ pairs, uint32_t n,
                                                        uint32_t* hist, uint32_t passes) {
    __shared__ uint32_t s_hist[4][RADIX_BASE];
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) s_hist[p][tid] = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256u + tid; i < n; i += gridDim.x * 256u) {
        const uint32_t key = pairs[i].x;
        for (uint32_t pl = 0; pl < passes; ++pl)
            atomicAdd(&s_hist[pl][(key >> (RADIX_BITS * pl)) & (RADIX_BASE - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t pl = 0; pl < passes; ++pl) {
        const uint32_t v = s_hist[pl][tid];
        if (v) atomicAdd(&hist[pl * RADIX_BASE + tid], v);
    }
}

void launch_histogram(hipStream_t stream, const uint2* pairs, uint32_t n, uint32_t* hist,
                      uint32_t passes) {
    if (n == 0) return;
    uint32_t blocks = (n + 256u * 16u - 1) / (256u * 16u);
    if (blocks > 1024u) blocks = 1024u;
    hipLaunchKernelGGL(histogram_kernel, dim3(blocks), dim3(256), 0, stream, pairs, n, hist, passes);
}

// ---------------------------------------------------------------------------------------
// Onesweep digit pass
// ---------------------------------------------------------------------------------------
template <int KPT>
__global__ __launch_bounds__(256) void onesweep_kernel(const uint2* __restrict__ in,
                                                       uint2* __restrict__ out,
                                                       const uint32_t* __restrict__ n_ptr,
                                                       const uint32_t* __restrict__ hist,
                                                       uint32_t* status, uint32_t* ticket,
                                                       uint32_t* error_flag, uint32_t shift,
                                                       uint32_t key_xor) {
    constexpr uint32_t TILE = 256u * KPT;
    __shared__ uint2 s_pairs[TILE];
    __shared__ uint32_t s_wave_hist[4][RADIX_BASE];
    __shared__ uint32_t s_block_excl[RADIX_BASE];   // first slot of each digit in the LDS order
    __shared__ uint32_t s_global_base[RADIX_BASE];  // dst = s_global_base[d] + slot
    __shared__ uint32_t s_hist_excl[RADIX_BASE];
    __shared__ uint32_t s_tot[4];
    __shared__ uint32_t s_tile;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the first ticket, the key count and the global histogram are three independent L2/HBM round
    // trips: issue them together (taken one after the other they were ~4 us of an ~12 us pass)
    const uint32_t h = hist[tid];
    const uint32_t n = *n_ptr;
    const uint32_t num_tiles = (n + TILE - 1u) / TILE;
    if (num_tiles == 0u) return;

    {   // radix_sort_b: exclusive scan of the global digit histogram, once per block
        uint32_t total;
        s_hist_excl[tid] = block_exclusive_scan_256(h, s_tot, total);
    }
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const bool single_shot = gridDim.x >= num_tiles;  // see keygen_kernel

    for (;;) {
        if (tid == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
        for (int w = 0; w < 4; ++w) s_wave_hist[w][tid] = 0u;
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= num_tiles) break;
        const uint32_t tile_base = tile * TILE;
        const uint32_t row_base = tile_base + (uint32_t)wave * (uint32_t)(KPT * 64) + (uint32_t)lane;

        uint2 kv[KPT];
        uint32_t rank[KPT];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t idx = row_base + (uint32_t)k * 64u;
            kv[k] = idx < n ? in[idx] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        }
        // wave-private ranking, one 64-pair row at a time (rows in input order => stable).
        // DS operations of one wave execute in issue order, so the row leader's counter update
        // is seen by the next row's read; volatile keeps the compiler from caching the counters.
        volatile uint32_t* const wh = s_wave_hist[wave];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t idx = row_base + (uint32_t)k * 64u;
            const bool valid = idx < n;
            const uint32_t d = (kv[k].x >> shift) & (RADIX_BASE - 1u);
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const uint32_t below = (uint32_t)__popcll(m & lanes_below);
            const uint32_t prev = wh[d];
            __builtin_amdgcn_wave_barrier();
            if (valid && below == 0u) wh[d] = prev + (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
            rank[k] = prev + below;
        }
        __syncthreads();

        // thread = digit: scan over the 4 waves
        uint32_t total;
        {
            const uint32_t c0 = s_wave_hist[0][tid], c1 = s_wave_hist[1][tid];
            const uint32_t c2 = s_wave_hist[2][tid], c3 = s_wave_hist[3][tid];
            s_wave_hist[0][tid] = 0u;
            s_wave_hist[1][tid] = c0;
            s_wave_hist[2][tid] = c0 + c1;
            s_wave_hist[3][tid] = c0 + c1 + c2;
            total = c0 + c1 + c2 + c3;
        }
        // chained scan with decoupled look-back, one chain per digit
        uint32_t* const my_status = status + (size_t)tile * RADIX_BASE + tid;
        uint32_t excl = 0u;
        if (tile > 0u) {
            st_agent(my_status, STATUS_AGGREGATE | total);
            // 4 predecessors per round trip up to ~1000 tiles (measured at 74 tiles, 4 passes: 46.6 us
            // with 4, 48.3 with 2, 49.7 with 16, 52.6 with 32, 65.6 with 64: a prefix is usually met
            // within the first few words, and every extra word polled is fabric traffic), 16 beyond
            excl = num_tiles > 1024u ? lookback_u32<16>(status + tid, tile, RADIX_BASE, error_flag, 1u)
                                     : lookback_u32<4>(status + tid, tile, RADIX_BASE, error_flag, 1u);
        }
        st_agent(my_status, STATUS_PREFIX | ((excl + total) & STATUS_VALUE_MASK));

        uint32_t blk_total;
        const uint32_t bexcl = block_exclusive_scan_256(total, s_tot, blk_total);
        s_block_excl[tid] = bexcl;
        s_global_base[tid] = s_hist_excl[tid] + excl - bexcl;
        __syncthreads();

        // scatter into LDS in digit order
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t idx = row_base + (uint32_t)k * 64u;
            if (idx < n) {
                const uint32_t d = (kv[k].x >> shift) & (RADIX_BASE - 1u);
                const uint32_t slot = s_block_excl[d] + s_wave_hist[wave][d] + rank[k];
                s_pairs[slot] = kv[k];
            }
        }
        __syncthreads();
        const uint32_t count = min(TILE, n - tile_base);
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const uint32_t slot = (uint32_t)k * 256u + (uint32_t)tid;
            if (slot < count) {
                uint2 e = s_pairs[slot];
                const uint32_t d = (e.x >> shift) & (RADIX_BASE - 1u);
                e.x ^= key_xor;
                const uint32_t dst = s_global_base[d] + slot;
                if (dst < n) out[dst] = e;  // always true unless the watchdog tripped
            }
        }
        if (single_shot) break;
        __syncthreads();
    }
}

void launch_onesweep_pass(hipStream_t stream, const uint2* in, uint2* out, const uint32_t* n_ptr,
                          uint32_t max_n, const uint32_t* hist, uint32_t* status, uint32_t* ticket,
                          uint32_t* error_flag, uint32_t shift, uint32_t key_xor, bool large_tiles,
                          int max_blocks) {
    if (max_n == 0) return;
    const uint32_t tile = sort_tile_size(large_tiles);
    uint32_t blocks = (max_n + tile - 1u) / tile;
    if (blocks > (uint32_t)max_blocks) blocks = (uint32_t)max_blocks;
    if (large_tiles)
        hipLaunchKernelGGL(onesweep_kernel<SORT_KPT_LARGE>, dim3(blocks), dim3(256), 0, stream, in, out,
                           n_ptr, hist, status, ticket, error_flag, shift, key_xor);
    else
        hipLaunchKernelGGL(onesweep_kernel<SORT_KPT_SMALL>, dim3(blocks), dim3(256), 0, stream, in, out,
                           n_ptr, hist, status, ticket, error_flag, shift, key_xor);
}

// ---------------------------------------------------------------------------------------
// Bucket sort: ONE launch instead of the digit passes, for draw lists that fit the bucket geometry
// (bgs_device.h). A digit pass over ~10^5 pairs is a chain of dependent L2 round trips (ticket -> load ->
// look-back -> scatter, ~11 us whatever its bandwidth) and 32-bit keys need four of them; here keygen has
// already scattered the pairs into BUCKET_COUNT key-range buckets, so the rest of the sort is local:
//   0. every workgroup scans the 4096 bucket counts (16 KB out of L2): exclusive prefix P[b]
//   1. chunk c = the non-empty buckets with floor(P[b] / BUCKET_HALF) == c: consecutive buckets holding at
//      most BUCKET_HALF + BUCKET_CAP pairs; its output offset is P[first bucket]. No workgroup waits for
//      another one.
//   2. gather the chunk's pairs (bucket regions are contiguous: coalesced per bucket)
//   3. counting sort in LDS on BUCKET_FINE fine key ranges of the chunk's own [min, max] (returning LDS
//      atomics: order inside a fine bucket is arbitrary)
//   4. rank every pair among the handful that share its fine bucket by (key, index) and write it to
//      out[P[first] + rank] — ascending key, ties by ascending index: the order of the stable LSD passes.
// Order never depends on the bucket range, balance does: a bucket over capacity, or a key value repeated
// more than BUCKET_FINE_MAX times (step 4 is quadratic in the ties), sets ctl->sort_overflow and the host
// re-runs the frame with the onesweep passes.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bucket_sort_kernel(const uint2* __restrict__ slots,
                                                          uint2* __restrict__ out, Control* ctl,
                                                          uint32_t key_xor) {
    constexpr uint32_t NB = BUCKET_COUNT, BPT = NB / 256u;   // buckets per thread (contiguous)
    constexpr uint32_t EPT = BUCKET_CHUNK / 256u;            // pairs per thread (strided)
    constexpr uint32_t NF = BUCKET_FINE, FPT = NF / 256u;    // fine buckets per thread (contiguous)
    static_assert(BPT == 16 && (NF & (NF - 1u)) == 0u, "bucket geometry");
    __shared__ uint32_t s_P[NB + 1];
    __shared__ uint2 s_el[BUCKET_CHUNK];
    __shared__ uint32_t s_f[NF + 1];
    __shared__ uint32_t s_tot[4];
    __shared__ uint32_t s_red[4][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t cnt[BPT], pre[BPT];
    {
        const uint4* src = reinterpret_cast<const uint4*>(ctl->bucket_count) + (size_t)tid * (BPT / 4u);
#pragma unroll
        for (uint32_t q = 0; q < BPT / 4u; ++q) {
            const uint4 v = src[q];
            cnt[4 * q] = v.x; cnt[4 * q + 1] = v.y; cnt[4 * q + 2] = v.z; cnt[4 * q + 3] = v.w;
        }
    }
    uint32_t local = 0u, mx = 0u;
#pragma unroll
    for (uint32_t j = 0; j < BPT; ++j) {
        mx = max(mx, cnt[j]);
        cnt[j] = min(cnt[j], BUCKET_CAP);  // slots past the capacity were never written
        pre[j] = local;
        local += cnt[j];
    }
    uint32_t total;
    const uint32_t excl0 = block_exclusive_scan_256(local, s_tot, total);
#pragma unroll
    for (uint32_t j = 0; j < BPT; ++j) {
        pre[j] += excl0;
        s_P[(uint32_t)tid * BPT + j] = pre[j];
    }
    if (tid == 0) s_P[NB] = total;
    if (blockIdx.x == 0u) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
        if (lane == 0) {
            atomicMax(&ctl->bucket_max, mx);
            if (mx > BUCKET_CAP) atomicOr(&ctl->sort_overflow, 1u);
        }
    }
    __syncthreads();
    if (total == 0u) return;
    const uint32_t num_chunks = (total - 1u) / BUCKET_HALF + 1u;

    for (uint32_t c = blockIdx.x; c < num_chunks; c += gridDim.x) {
        // ---- 1. the chunk's bucket range [b0, b1) ----
        uint32_t b0 = NB, b1 = 0u;
#pragma unroll
        for (uint32_t j = 0; j < BPT; ++j)
            if (cnt[j] != 0u && pre[j] / BUCKET_HALF == c) {
                b0 = min(b0, (uint32_t)tid * BPT + j);
                b1 = max(b1, (uint32_t)tid * BPT + j + 1u);
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            b0 = min(b0, (uint32_t)__shfl_xor((int)b0, off, 64));
            b1 = max(b1, (uint32_t)__shfl_xor((int)b1, off, 64));
        }
        if (lane == 0) { s_red[wave][0] = b0; s_red[wave][1] = b1; }
        __syncthreads();
        b0 = min(min(s_red[0][0], s_red[1][0]), min(s_red[2][0], s_red[3][0]));
        b1 = max(max(s_red[0][1], s_red[1][1]), max(s_red[2][1], s_red[3][1]));
        __syncthreads();
        if (b1 == 0u) continue;  // (every chunk id below num_chunks owns a bucket; defensive)
        const uint32_t base = s_P[b0], m = s_P[b1] - base;

        // ---- 2. gather ----
        uint2 kv[EPT];
        uint32_t kmn = 0xFFFFFFFFu, kmx = 0u;
#pragma unroll
        for (uint32_t k = 0; k < EPT; ++k) {
            const uint32_t e = k * 256u + (uint32_t)tid;
            kv[k] = make_uint2(0u, 0u);
            if (e < m) {
                const uint32_t x = base + e;
                uint32_t lo = b0, hi = b1 - 1u;  // largest b with P[b] <= x (empty buckets share their successor's P)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (s_P[mid] <= x) lo = mid; else hi = mid - 1u;
                }
                kv[k] = slots[(size_t)lo * BUCKET_CAP + (x - s_P[lo])];
                kmn = min(kmn, kv[k].x);
                kmx = max(kmx, kv[k].x);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, off, 64));
            kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, off, 64));
        }
        if (lane == 0) { s_red[wave][2] = kmn; s_red[wave][3] = kmx; }
#pragma unroll
        for (uint32_t j = 0; j < FPT; ++j) s_f[j * 256u + (uint32_t)tid] = 0u;
        __syncthreads();
        kmn = min(min(s_red[0][2], s_red[1][2]), min(s_red[2][2], s_red[3][2]));
        kmx = max(max(s_red[0][3], s_red[1][3]), max(s_red[2][3], s_red[3][3]));
        const uint32_t span = kmx - kmn;
        const uint32_t bits = span ? 32u - (uint32_t)__builtin_clz(span) : 0u;
        const uint32_t fshift = bits > 11u ? bits - 11u : 0u;  // (span >> fshift) < NF = 2^11
        static_assert(NF == 2048u, "fshift assumes 2^11 fine buckets");

        // ---- 3. counting sort on the fine buckets ----
        uint32_t slot[EPT];
#pragma unroll
        for (uint32_t k = 0; k < EPT; ++k) {
            const uint32_t e = k * 256u + (uint32_t)tid;
            if (e < m) slot[k] = atomicAdd(&s_f[(kv[k].x - kmn) >> fshift], 1u);
        }
        __syncthreads();
        uint32_t f[FPT], fsum = 0u, fmax = 0u;
#pragma unroll
        for (uint32_t j = 0; j < FPT; ++j) {
            f[j] = s_f[(uint32_t)tid * FPT + j];
            fmax = max(fmax, f[j]);
            fsum += f[j];
        }
        uint32_t ftotal;
        uint32_t fexcl = block_exclusive_scan_256(fsum, s_tot, ftotal);
#pragma unroll
        for (uint32_t j = 0; j < FPT; ++j) {
            s_f[(uint32_t)tid * FPT + j] = fexcl;
            fexcl += f[j];
        }
        if (tid == 0) s_f[NF] = m;
        if (__syncthreads_or(fmax > BUCKET_FINE_MAX ? 1 : 0)) {  // step 4 is quadratic in equal keys: give up
            if (tid == 0) atomicOr(&ctl->sort_overflow, 2u);
            continue;
        }
#pragma unroll
        for (uint32_t k = 0; k < EPT; ++k) {
            const uint32_t e = k * 256u + (uint32_t)tid;
            if (e < m) s_el[s_f[(kv[k].x - kmn) >> fshift] + slot[k]] = kv[k];
        }
        __syncthreads();

        // ---- 4. rank among the fine bucket's pairs by (key, index), write out ----
#pragma unroll 4
        for (uint32_t k = 0; k < EPT; ++k) {
            const uint32_t p = k * 256u + (uint32_t)tid;
            if (p < m) {
                const uint2 el = s_el[p];
                const uint32_t fb = (el.x - kmn) >> fshift;
                const uint32_t fs = s_f[fb], fe = s_f[fb + 1u];
                uint32_t r = 0u;
                for (uint32_t j = fs; j < fe; ++j) {
                    const uint2 o = s_el[j];
                    r += (o.x < el.x || (o.x == el.x && o.y < el.y)) ? 1u : 0u;
                }
                out[base + fs + r] = make_uint2(el.x ^ key_xor, el.y);
            }
        }
        __syncthreads();  // s_el / s_f are rewritten by the next chunk
    }
}

void launch_bucket_sort(hipStream_t stream, const uint2* bucket_slots, uint2* out, Control* ctl, uint32_t key_xor,
                        int blocks) {
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bucket_sort_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, bucket_slots, out, ctl,
                       key_xor);
}

// ---------------------------------------------------------------------------------------
// HBM ceiling probe: STREAM triad, 16-byte accesses, grid-stride
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void triad_kernel(float4* __restrict__ a, const float4* __restrict__ b,
                                                    const float4* __restrict__ c, float s, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256u;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += stride) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}

void launch_triad(hipStream_t stream, float4* a, const float4* b, const float4* c, float s, size_t n4,
                  int blocks) {
    if (n4 == 0) return;
    hipLaunchKernelGGL(triad_kernel, dim3(blocks), dim3(256), 0, stream, a, b, c, s, n4);
}

}  // namespace bgs
