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
// Two paths produce that order (the host picks one per frame, bgs_frame.hip):
//   onesweep   keygen (partition + digit histograms) + one onesweep_kernel per digit place — any key width,
//              any key distribution, any size; ~11 us per place at 10^5 pairs (a chain of L2 round trips)
//   bucket     keygen places the drawable pairs into 256 key-range buckets (splitters = quantile keys of a
//              completed frame's sorted list) + ONE bucket_sort_kernel that sorts every bucket in LDS by
//              (key, index): 6.5 us at 10^5 pairs. Falls back (re-run) when a bucket overflows.
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
#include <type_traits>

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
// passes need, the drawable pairs go straight into BUCKET_COUNT key-range buckets (fixed slot regions of
// BUCKET_CAP pairs): bucket = number of splitters <= key (binary search of a 255-entry table in LDS),
// slot = [pairs of the bucket in earlier tiles: chained-scan look-back, one chain per bucket, exactly the
// radix pass's machinery] + [arrival order within the tile: a returning LDS atomic]. The sum of the 256
// exclusive prefixes is the tile's drawable-entry offset, so the partition needs no chain of its own.
// Order inside a bucket is arbitrary; bucket_sort_kernel orders by (key, index).
// ORDERED = false (BUCKET frames that write no culled tail, i.e. every rendered frame): nothing downstream needs the
// drawable pairs or the culled ones in index order — bucket_sort_kernel orders every bucket by (key, index) whatever
// the order inside its slots — so the tile needs no ticket, no row scan and, above all, NO CHAIN: the 256 per-bucket
// look-backs are replaced by one returning atomicAdd per (tile, bucket) on the bucket's running count. With the
// chains, all 245 tiles of a 1 M-splat cloud reach the look-back within a microsecond of each other and the
// inclusive prefixes then travel tile by tile: s_memtime stamps put the wait at 50 ns per predecessor — 6 us for
// tile 120, 12 us for tile 244, half of the kernel (profiles/r3_notes.md section 10).
#ifndef BGS_KG_GROUP
#define BGS_KG_GROUP 2   // splats of a thread whose keys are computed together (divides every KG_ITEMS): 2 / 4 / 8 / 16 ->
                         // 1 M keygen 20.6 / 21.2 / 22.2 / 23.5 us, 5 M 51 / 53 / 64 / 88 us (registers: 153 / 164 / 206 / 256)
#endif
template <int KG_ITEMS, bool BUCKET, int THREADS, bool ORDERED = true>  // splats per thread; 256 or 1024 threads per block
// (the ordered 1024-thread tile is held to 64 registers — 8 waves per SIMD, two tiles per CU —: with the all-drawable path it came
// to 65 and one tile per CU; the chainless one has 61 of its own accord and is left alone)
__global__ __launch_bounds__(THREADS, (THREADS == 1024 && ORDERED) ? 8 : 1) void keygen_kernel(FrameParams fp, const float4* __restrict__ pos,
                                                         uint2* __restrict__ entries,
                                                         uint2* __restrict__ culled, Control* ctl,
                                                         uint32_t* part_status, uint32_t places,
                                                         uint32_t ticket_slot, FrameParams* fp_out,
                                                         uint2* __restrict__ bucket_slots, uint32_t* bucket_status,
                                                         SplitterTable split, uint32_t* zero_word) {
    constexpr int WAVES = THREADS / 64;
    constexpr int ROWS = KG_ITEMS * WAVES;  // 64-splat rows of a tile, in index order (item, wave)
    static_assert((THREADS == 256 || THREADS == 512 || THREADS == 1024) && ROWS <= 128, "tile geometry");
    __shared__ uint32_t s_hist[BUCKET ? 1 : 4][RADIX_BASE];
    __shared__ uint32_t s_cnt[ROWS];   // drawable splats of each row ...
    __shared__ uint32_t s_off[ROWS];   // ... and of the rows before it
    __shared__ uint32_t s_keys[THREADS * KG_ITEMS];  // the tile's drawable keys, compacted (for the histograms / the buckets)
    __shared__ uint32_t s_idx[BUCKET ? THREADS * KG_ITEMS : 1];  // their splat indices (BUCKET)
    // BUCKET: dynamic LDS = nb counters (nb = 256 * split.sub buckets) + the nb - 1 splitters padded with ~0 to a power of two
    extern __shared__ uint32_t s_dyn[];
    uint32_t* const s_bcnt = s_dyn;     // pairs of this tile per bucket; then, IN PLACE, the bucket's pairs of earlier tiles
    uint32_t* const s_bexcl = s_dyn;    // (thread f reads its count and leaves the returning atomic's value in the same word)
    uint32_t* const s_split = s_dyn + (BUCKET ? BUCKET_COUNT * split.sub : 0u);
    __shared__ uint16_t s_at[BUCKET ? THREADS * KG_ITEMS : 1];  // arrival slot of compacted pair j inside its bucket (this tile)
    __shared__ uint16_t s_bk[BUCKET ? THREADS * KG_ITEMS : 1];  // its bucket
    __shared__ uint32_t s_total;
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The frame's first kernel is the only one that receives FrameParams by value: it leaves a copy
    // in device memory for the kernels behind it, so that a captured frame (hipGraph) is re-aimed at
    // a new view by updating this one node's arguments.
    if (fp_out && blockIdx.x == 0 && (uint32_t)tid < (uint32_t)(sizeof(FrameParams) / 4u))
        reinterpret_cast<uint32_t*>(fp_out)[tid] = reinterpret_cast<const uint32_t*>(&fp)[tid];
    if (zero_word && blockIdx.x == 0 && tid == 0) *zero_word = 0u;
    // nb = 256 * sub buckets of BUCKET_CAP pairs — or, for the long lists (SplitterTable::wide), of BUCKET_CAP_WIDE
    const uint32_t nsub = BUCKET ? split.sub : 1u, nb = BUCKET_COUNT * nsub;
    const uint32_t bucket_cap = (BUCKET && split.wide != 0u) ? BUCKET_CAP_WIDE : BUCKET_CAP;   // pairs per slot region
    uint32_t p2 = 256u;                 // the splitter table's padded length: the power of two >= nb
    while (p2 < nb) p2 <<= 1;
    if constexpr (BUCKET) {
        const uint32_t* __restrict__ keys = nsub <= BUCKET_SUB_KERNARG ? split.key : split.device_keys;
        for (uint32_t t = (uint32_t)tid; t < p2; t += (uint32_t)THREADS)
            s_split[t] = t < nb - 1u ? keys[t] : 0xFFFFFFFFu;
    } else if (tid < 256) {
#pragma unroll
        for (int p = 0; p < 4; ++p) s_hist[p][tid] = 0u;
    }
    const uint32_t sentinel = KEY_CULLED >> fp.key_shift;
    constexpr uint32_t per_tile = (uint32_t)(THREADS * KG_ITEMS);
    const uint32_t num_tiles = (fp.n + per_tile - 1u) / per_tile;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    // Tiles are handed out by ticket so that a tile's predecessors in the chained scan are always owned
    // by running blocks (kernels of several lanes share the chip). Same-address device-scope atomics
    // retire at ~8 ns each, so when the grid covers every tile a block takes ONE ticket and leaves
    // after its tile instead of queueing for a second ticket only to be told there is nothing left.
    const bool single_shot = gridDim.x >= num_tiles;

    // bucket of a key = the number of splitters <= key (the entries behind the table's end are ~0: never counted unless the
    // key is ~0 itself, hence the min): 8 steps over the 255 splitters of the usual frame, up to 12 over 4095
    auto bucket_of = [&](const uint32_t kk) -> uint32_t {
        uint32_t lo = 0u;
        if (nsub == 1u) {
#pragma unroll
            for (uint32_t step = 128u; step > 0u; step >>= 1)
                if (s_split[lo + step - 1u] <= kk) lo += step;
        } else if (p2 == 512u) {
#pragma unroll
            for (uint32_t step = 256u; step > 0u; step >>= 1)
                if (s_split[lo + step - 1u] <= kk) lo += step;
        } else if (p2 == 1024u) {
#pragma unroll
            for (uint32_t step = 512u; step > 0u; step >>= 1)
                if (s_split[lo + step - 1u] <= kk) lo += step;
        } else {
            for (uint32_t step = p2 >> 1; step > 0u; step >>= 1)
                if (s_split[lo + step - 1u] <= kk) lo += step;
        }
        return min(lo, nb - 1u);
    };
    // thread = bucket(s): the tile's pairs of a bucket take the next `mine` slots of the bucket, whoever comes first — one
    // returning atomic per (tile, bucket); the word that held the tile's count then holds the bucket's pairs of earlier
    // tiles. With more buckets than threads a thread's atomics are issued eight at a time (each is a round trip to L2).
    auto claim_bucket_slots = [&]() {
        if (nb <= (uint32_t)THREADS) {
            if ((uint32_t)tid < nb) {
                const uint32_t mine = s_bcnt[tid];
                s_bexcl[tid] = mine ? atomicAdd(&ctl->bucket_count[tid], mine) : 0u;
            }
            return;
        }
        for (uint32_t f0 = (uint32_t)tid; f0 < nb; f0 += 8u * (uint32_t)THREADS) {
            uint32_t ex[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) {
                const uint32_t f = f0 + k * (uint32_t)THREADS;
                const uint32_t mine = f < nb ? s_bcnt[f] : 0u;
                ex[k] = mine ? atomicAdd(&ctl->bucket_count[f], mine) : 0u;
            }
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) {
                const uint32_t f = f0 + k * (uint32_t)THREADS;
                if (f < nb) s_bexcl[f] = ex[k];
            }
        }
    };
    if constexpr (BUCKET && !ORDERED) {
        if (blockIdx.x == 0 && tid == 0) ctl->splat_count = fp.n;
        for (uint32_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
            for (uint32_t f = (uint32_t)tid; f < nb; f += (uint32_t)THREADS) s_bcnt[f] = 0u;
            if (tid == 0) s_total = 0u;
            __syncthreads();   // (first tile: s_split is in place as well)
            const uint32_t base = tile * per_tile;
            float4 pin[KG_ITEMS];
#pragma unroll
            for (int k = 0; k < KG_ITEMS; ++k) {
                const uint32_t i = base + (uint32_t)(k * THREADS) + (uint32_t)tid;
                pin[k] = i < fp.n ? pos[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            uint32_t key[KG_ITEMS], below[KG_ITEMS], rowoff[KG_ITEMS];
            const bool all_draw = fp.sort_mode != SORT_RADIX;
#define BGS_KG_KEYS(KIND)                                                                      \
    _Pragma("unroll") for (int k0 = 0; k0 < KG_ITEMS; k0 += BGS_KG_GROUP) {                    \
        /* BGS_KG_GROUP splats at a time: straight-line keys (independent chains the compiler interleaves; all 16 at once \
           cost 256 registers), then ONE rare branch for the splats whose frustum verdict needs the divisions */ \
        uint32_t unsure_mask = 0u;                                                             \
        _Pragma("unroll") for (int k = k0; k < k0 + BGS_KG_GROUP; ++k) {                       \
            const uint32_t i = base + (uint32_t)(k * THREADS) + (uint32_t)tid;                 \
            const float4 p = pin[k];                                                           \
            bool unsure;                                                                       \
            const uint32_t kf = sort_key_fast<KIND>(fp, V3{p.x, p.y, p.z}, unsure);            \
            key[k] = i < fp.n ? kf : sentinel;                                                 \
            unsure_mask |= (unsure && i < fp.n) ? (1u << (k - k0)) : 0u;                       \
        }                                                                                      \
        if (unsure_mask) {                                                                     \
            _Pragma("unroll") for (int k = k0; k < k0 + BGS_KG_GROUP; ++k)                     \
                if ((unsure_mask >> (k - k0)) & 1u) key[k] = sort_key_kind<KIND>(fp, V3{pin[k].x, pin[k].y, pin[k].z}); \
        }                                                                                      \
    }
            if (fp.sort_mode == SORT_RADIX) { BGS_KG_KEYS(1) }
            else if (fp.sort_mode == SORT_NONE) { BGS_KG_KEYS(0) }
            else { BGS_KG_KEYS(2) }
#undef BGS_KG_KEYS
#define BGS_KG_DRAWN(k) (all_draw ? (base + (uint32_t)((k) * THREADS) + (uint32_t)tid < fp.n) : (key[k] != sentinel))
            // the wave's drawable pairs go to one contiguous piece of the tile's compacted arrays: ONE LDS atomic per wave
            uint32_t run = 0u;
#pragma unroll
            for (int k = 0; k < KG_ITEMS; ++k) {
                const unsigned long long b = __ballot(BGS_KG_DRAWN(k));
                below[k] = (uint32_t)__popcll(b & lanes_below);
                rowoff[k] = run;
                run += (uint32_t)__popcll(b);
            }
            uint32_t wbase = 0u;
            if (lane == 0 && run) wbase = atomicAdd(&s_total, run);
            wbase = (uint32_t)__shfl((int)wbase, 0, 64);
#pragma unroll
            for (int k = 0; k < KG_ITEMS; ++k)
                if (BGS_KG_DRAWN(k)) {
                    const uint32_t j = wbase + rowoff[k] + below[k];
                    s_keys[j] = key[k];
                    s_idx[j] = base + (uint32_t)(k * THREADS) + (uint32_t)tid;
                }
#undef BGS_KG_DRAWN
            __syncthreads();
            const uint32_t total = s_total;
#pragma unroll
            for (int r = 0; r < KG_ITEMS; ++r) {
                const uint32_t j = (uint32_t)(r * THREADS) + (uint32_t)tid;
                if (j < total) {
                    const uint32_t bkt = bucket_of(s_keys[j]);
                    s_bk[j] = (uint16_t)bkt;
                    s_at[j] = (uint16_t)min(atomicAdd(&s_bcnt[bkt], 1u), 0xFFFFu);
                }
            }
            __syncthreads();
            claim_bucket_slots();
            __syncthreads();
#pragma unroll
            for (int r = 0; r < KG_ITEMS; ++r) {
                const uint32_t j = (uint32_t)(r * THREADS) + (uint32_t)tid;
                if (j < total) {
                    const uint32_t bkt = s_bk[j];
                    const uint32_t slot = s_bexcl[bkt] + (uint32_t)s_at[j];
                    if (slot < bucket_cap)  // a bucket over capacity is seen by bucket_sort_kernel (count > cap)
                        bucket_slots[(size_t)bkt * bucket_cap + slot] = make_uint2(s_keys[j], s_idx[j]);
                }
            }
        }
        return;
    }

    for (;;) {
        if (tid == 0) s_tile = atomicAdd(&ctl->ticket[ticket_slot][0], 1u);
        if constexpr (BUCKET) { for (uint32_t f = (uint32_t)tid; f < nb; f += (uint32_t)THREADS) s_bcnt[f] = 0u; }
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= num_tiles) break;
        const uint32_t base = tile * per_tile;
        uint32_t key[KG_ITEMS], below[KG_ITEMS];
        // which entries reach the vertex stage: everything unless the radix key is "culled" (out-of-range slots
        // carry the sentinel too). Re-derived from the key where it is needed: a lane mask per item kept alive
        // across the tile costs two scalar registers each, and this kernel spills them
        const bool all_draw = fp.sort_mode != SORT_RADIX;
#define BGS_KG_DRAWN(k) (all_draw ? (base + (uint32_t)((k) * THREADS) + (uint32_t)tid < fp.n) : (key[k] != sentinel))
        // all of the tile's position loads are issued before the first key is computed
        float4 pin[KG_ITEMS];
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
            const uint32_t i = base + (uint32_t)(k * THREADS) + (uint32_t)tid;
            pin[k] = i < fp.n ? pos[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        // the sort mode is the launch's, not the splat's: one branch around the keys instead of one inside
        // each (every copy re-reading its matrices from spilled scalar registers)
#define BGS_KG_KEYS(KIND)                                                                      \
    _Pragma("unroll") for (int k = 0; k < KG_ITEMS; ++k) {                                     \
        /* the tiles with chains keep the reference's own key (three divisions per splat, no branch: the compiler    \
           interleaves a thread's splats): with the two-step key of the chainless tiles this kernel was 3 us slower */ \
        const uint32_t i = base + (uint32_t)(k * THREADS) + (uint32_t)tid;                     \
        const float4 p = pin[k];                                                               \
        key[k] = i < fp.n ? sort_key_kind<KIND>(fp, V3{p.x, p.y, p.z}) : sentinel;             \
    }
        if (fp.sort_mode == SORT_RADIX) { BGS_KG_KEYS(1) }
        else if (fp.sort_mode == SORT_NONE) { BGS_KG_KEYS(0) }
        else { BGS_KG_KEYS(2) }
#undef BGS_KG_KEYS
        if constexpr (BUCKET && KG_ITEMS <= 8) {
            // A FULL tile of a sort mode that draws every splat (Rayon / Std / None; round 6): nothing to compact and no
            // order to keep — the keys go from the registers straight to their buckets, without the row counts, their scan,
            // the compacted LDS arrays and two of the barriers (5 M splats, SortMode::Rayon: keygen 76.7 -> 60.3 us)
            if (all_draw && base + per_tile <= fp.n) {
                uint32_t bkat[KG_ITEMS];   // bucket (< 4096) | arrival order in it (< tile) << 16: one register per splat
#pragma unroll
                for (int k = 0; k < KG_ITEMS; ++k) {
                    const uint32_t bkt = bucket_of(key[k]);
                    bkat[k] = bkt | (atomicAdd(&s_bcnt[bkt], 1u) << 16);
                }
                if (tid == 0 && tile == num_tiles - 1u) { ctl->draw_count = fp.n; ctl->splat_count = fp.n; }
                __syncthreads();
                claim_bucket_slots();
                __syncthreads();
#pragma unroll
                for (int k = 0; k < KG_ITEMS; ++k) {
                    const uint32_t bkt = bkat[k] & 0xFFFFu, slot = s_bexcl[bkt] + (bkat[k] >> 16);
                    if (slot < bucket_cap)  // a bucket over capacity is seen by bucket_sort_kernel (count > cap)
                        bucket_slots[(size_t)bkt * bucket_cap + slot] = make_uint2(key[k], base + (uint32_t)(k * THREADS) + (uint32_t)tid);
                }
                if (single_shot) break;
                __syncthreads();
                continue;
            }
        }
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
            const unsigned long long b = __ballot(BGS_KG_DRAWN(k));
            below[k] = (uint32_t)__popcll(b & lanes_below);
            if (lane == 0) s_cnt[k * WAVES + wave] = (uint32_t)__popcll(b);
        }
        __syncthreads();
        // exclusive offsets of the rows in (item, wave) = index order: one wave scans the <= 128 row counts
        if (wave == 0) {
            const uint32_t c0 = lane < ROWS ? s_cnt[lane] : 0u;
            const uint32_t c1 = ROWS > 64 && lane + 64 < ROWS ? s_cnt[lane + 64] : 0u;
            const uint32_t i0 = wave_inclusive_scan(c0, lane);
            const uint32_t t0 = (uint32_t)__shfl((int)i0, 63, 64);
            if (lane < ROWS) s_off[lane] = i0 - c0;
            uint32_t t1 = 0u;
            if constexpr (ROWS > 64) {
                const uint32_t i1 = wave_inclusive_scan(c1, lane);
                t1 = (uint32_t)__shfl((int)i1, 63, 64);
                if (lane + 64 < ROWS) s_off[lane + 64] = t0 + i1 - c1;
            }
            if (lane == 0) s_total = t0 + t1;
        }
        __syncthreads();
        uint32_t off[KG_ITEMS];
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) off[k] = s_off[k * WAVES + wave];
        const uint32_t total = s_total;
        // Digit histograms from the COMPACTED keys: with the usual ~12 % of a tile drawable, counting
        // in place costs 4 LDS atomics on each row of every wave (3.7 us of the kernel);
        // compacted, a tile's ~600 keys are 3 rows.
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k)
            if (BGS_KG_DRAWN(k)) {
                s_keys[off[k] + below[k]] = key[k];
                if constexpr (BUCKET) s_idx[off[k] + below[k]] = base + (uint32_t)(k * THREADS) + (uint32_t)tid;
            }
        {
            if (wave == 0) {  // one chain per block: the whole wave walks it, 64 predecessors per hop
                uint32_t* const my_status = part_status + tile;
                uint32_t excl = 0u;
                if (tile > 0u) {
                    if (lane == 0) st_agent(my_status, STATUS_AGGREGATE | total);
                    // (a sort mode that draws every splat — Rayon / Std / None — has full tiles before this one: no chain to
                    // walk, 12 of the 83 us of a 5 M-splat keygen; the status words are written all the same)
                    excl = all_draw ? tile * per_tile : lookback_wave(part_status, tile, lane, &ctl->error, 8u);
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
            if constexpr (!BUCKET)
                for (uint32_t j = (uint32_t)tid; j < total; j += (uint32_t)THREADS) {
                    const uint32_t kk = s_keys[j];
                    for (uint32_t pl = 0; pl < places; ++pl)
                        atomicAdd(&s_hist[pl][(kk >> (RADIX_BITS * pl)) & (RADIX_BASE - 1u)], 1u);
                }
        }
        if constexpr (BUCKET) {
            // Bucket placement as in the chainless tiles above: a returning atomic per (tile, bucket). The tile's place
            // in the index order — what the culled tail needs — comes from the block's ONE chain above (round 4 walked
            // 256 per-bucket chains here and took their sum).
#pragma unroll
            for (int r = 0; r < KG_ITEMS; ++r) {
                const uint32_t j = (uint32_t)(r * THREADS) + (uint32_t)tid;
                if (j < total) {
                    const uint32_t bkt = bucket_of(s_keys[j]);
                    s_bk[j] = (uint16_t)bkt;
                    s_at[j] = (uint16_t)min(atomicAdd(&s_bcnt[bkt], 1u), 0xFFFFu);
                }
            }
            __syncthreads();
            claim_bucket_slots();
            __syncthreads();
#pragma unroll
            for (int r = 0; r < KG_ITEMS; ++r) {
                const uint32_t j = (uint32_t)(r * THREADS) + (uint32_t)tid;
                if (j < total) {
                    const uint32_t bkt = s_bk[j];
                    const uint32_t slot = s_bexcl[bkt] + (uint32_t)s_at[j];
                    if (slot < bucket_cap)  // a bucket over capacity is seen by bucket_sort_kernel (count > cap)
                        bucket_slots[(size_t)bkt * bucket_cap + slot] = make_uint2(s_keys[j], s_idx[j]);
                }
            }
        }
        {
        const uint32_t vis_base = s_base;
#pragma unroll
        for (int k = 0; k < KG_ITEMS; ++k) {
            const uint32_t i = base + (uint32_t)(k * THREADS) + (uint32_t)tid;
            if (i < fp.n) {
                const uint32_t before = vis_base + off[k] + below[k];  // drawable entries before i
                if (BGS_KG_DRAWN(k)) { if constexpr (!BUCKET) entries[before] = make_uint2(key[k], i); }
                else if (culled) {  // null in frames nobody reads the tail of (see enqueue_frame): 8 B per culled splat saved
                    // streaming store: the culled tail is only read when somebody asks for the whole list
                    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store((v2u){key[k], i}, reinterpret_cast<v2u*>(culled + (i - before)));
                }
            }
        }
        }
#undef BGS_KG_DRAWN
        if (single_shot) break;
        __syncthreads();
    }
    if constexpr (!BUCKET) {
        __syncthreads();  // a block that leaves after its one tile comes here straight from the LDS atomics above
        if (tid < 256)
            for (uint32_t pl = 0; pl < places; ++pl) {
                const uint32_t v = s_hist[pl][tid];
                if (v) atomicAdd(&ctl->hist_depth[pl][tid], v);
            }
    }
}

bool KeygenLaunch::prepare(int max_blocks) {
    if (fp.n == 0) return false;
    // Tile size by cloud size (keygen_tile_splats): every tile is a ticket, a hop in 256 look-back chains and a
    // round of barriers
    const bool bucket = fp.sort_path == 1u;
    const bool unordered = bucket && culled == nullptr;   // no culled tail wanted: bucket placement without chains
#ifndef BGS_KG_UNORDERED_TILE
#define BGS_KG_UNORDERED_TILE 0   // 0: the size rule of the chained tiles
#endif
    const uint32_t per_block = (unordered && BGS_KG_UNORDERED_TILE) ? (uint32_t)BGS_KG_UNORDERED_TILE : keygen_tile_splats(fp.n);
#define BGS_KG_PICK(ITEMS, THR) \
    (bucket ? (unordered ? reinterpret_cast<const void*>(&keygen_kernel<ITEMS, true, THR, false>)                       \
                         : reinterpret_cast<const void*>(&keygen_kernel<ITEMS, true, THR, true>))                       \
            : reinterpret_cast<const void*>(&keygen_kernel<ITEMS, false, THR, true>))
#if BGS_KEYGEN_WIDE_THREADS == 256
    threads = 256u;
    func = per_block >= 4096u ? BGS_KG_PICK(16, 256) : BGS_KG_PICK(8, 256);
#else
    threads = per_block >= 4096u ? 1024u : 256u;
    func = per_block == 8192u ? BGS_KG_PICK(8, 1024) : (per_block == 4096u ? BGS_KG_PICK(4, 1024) : BGS_KG_PICK(8, 256));
#endif
    // A frame that is ALONE on the chip (pipeline depth 1: a blocking caller) runs its chainless 4096-splat tiles as
    // 1024 threads x 4 splats instead of 256 x 16: four waves per SIMD instead of one on the tile's CU, a quarter of
    // every thread's chain (keygen 17.6 -> 12.5 us at 1 M). With frames in flight the narrow tile wins (21.0 vs 20.2 k
    // frames/s: a 16-wave workgroup needs a CU's worth of free slots at once, and there the other frames' kernels are
    // what hides a tile's latency) — the same trade as in round 2, now decided per frame.
    // (round 6: the tiles with a chain — bgs_sort's bucket frames — likewise)
    if (bucket && wide && per_block == 4096u) {
        threads = 1024u;
        func = unordered ? reinterpret_cast<const void*>(&keygen_kernel<4, true, 1024, false>)
                         : reinterpret_cast<const void*>(&keygen_kernel<4, true, 1024, true>);
    }
#undef BGS_KG_PICK
    blocks = (fp.n + per_block - 1) / per_block;
    // (tiles without chains depend on nobody: one workgroup per tile, dispatched as slots come free)
    if (!unordered && blocks > (uint32_t)max_blocks) blocks = (uint32_t)max_blocks;
    argv[0] = &fp; argv[1] = &pos; argv[2] = &entries; argv[3] = &culled; argv[4] = &ctl;
    argv[5] = &part_status; argv[6] = &places; argv[7] = &ticket_slot; argv[8] = &fp_out; argv[9] = &bucket_slots;
    argv[10] = &bucket_status; argv[11] = &split; argv[12] = &zero_word;
    if (split.sub < 1u || split.sub > BUCKET_SUB_MAX) split.sub = 1u;
    {   // the per-bucket counters + the padded splitter table
        uint32_t nbk = BUCKET_COUNT * split.sub, pad2 = 256u;
        while (pad2 < nbk) pad2 <<= 1;
        lds_bytes = bucket ? (nbk + pad2) * (uint32_t)sizeof(uint32_t) : 0u;
    }
    return true;
}

hipError_t KeygenLaunch::launch(hipStream_t stream) {
    // (a long splitter table takes static + dynamic LDS past the runtime's default 64 KB per workgroup; the chip allows 160 KB)
    if (lds_bytes > 12288u) {
        const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    return hipLaunchKernel(func, dim3(blocks), dim3(threads), argv, lds_bytes, stream);
}

hipError_t KeygenLaunch::update_node(hipGraphExec_t exec, hipGraphNode_t node) {
    hipKernelNodeParams np{};
    np.func = const_cast<void*>(func);
    np.gridDim = dim3(blocks);
    np.blockDim = dim3(threads);
    np.sharedMemBytes = lds_bytes;
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
#ifndef BGS_ONESWEEP_WAVES
#define BGS_ONESWEEP_WAVES 0
#endif
template <int KPT>
#if BGS_ONESWEEP_WAVES
__global__ __launch_bounds__(256, BGS_ONESWEEP_WAVES) void onesweep_kernel(
#else
__global__ __launch_bounds__(256) void onesweep_kernel(
#endif
const uint2* __restrict__ in,
                                                       uint2* __restrict__ out,
                                                       const uint32_t* __restrict__ n_ptr,
                                                       const uint32_t* __restrict__ hist,
                                                       uint32_t* status, uint32_t* ticket,
                                                       uint32_t* error_flag, uint32_t shift,
                                                       uint32_t key_xor) {
    constexpr uint32_t TILE = 256u * KPT;
    __shared__ uint2 s_pairs[TILE];
    __shared__ uint32_t s_wave_hist[4][RADIX_BASE];
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
    const bool single_shot = gridDim.x >= num_tiles;  // see keygen_kernel

    for (;;) {
        if (tid == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
        for (int w = 0; w < 4; ++w) s_wave_hist[w][tid] = 0u;
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= num_tiles) break;
        // a tile that lies wholly inside the list (all but the last) runs without the per-key bounds tests: 16 exec-mask
        // regions in the loads, 16 ballots of `valid` and 32 compares in the two scatters less (round 6)
        auto process = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
            const uint32_t tile_base = tile * TILE;
            const uint32_t row_base = tile_base + (uint32_t)wave * (uint32_t)(KPT * 64) + (uint32_t)lane;

            uint2 kv[KPT];
            uint32_t rank[KPT];
    #pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const uint32_t idx = row_base + (uint32_t)k * 64u;
                kv[k] = (FULL || idx < n) ? in[idx] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            }
            // wave-private ranking, one 64-pair row at a time (rows in input order => stable).
            // DS operations of one wave execute in issue order, so the row leader's counter update
            // is seen by the next row's read; volatile keeps the compiler from caching the counters.
            // (round 6: the lanes with this lane's digit as two 32-bit words narrowed by v_xnor + v_and against each bit's
            // ballot — six vector instructions per bit where the 64-bit select form compiled to nine — and the wave's
            // counters through relaxed LDS atomics: `volatile` made every access a FLAT load with system scope)
            uint32_t* const wh = s_wave_hist[wave];
    #pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const uint32_t idx = row_base + (uint32_t)k * 64u;
                const bool valid = FULL || idx < n;
                const uint32_t d = (kv[k].x >> shift) & (RADIX_BASE - 1u);
                const unsigned long long mv = FULL ? ~0ull : __ballot(valid);
                uint32_t m_lo = (uint32_t)mv, m_hi = (uint32_t)(mv >> 32);
    #pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int32_t seli = __builtin_amdgcn_sbfe((int32_t)d, (uint32_t)b, 1u);   // all ones where the bit is set (v_bfe_i32)
                    const uint32_t sel = (uint32_t)seli;
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(seli < 0);
                    m_lo &= ~((uint32_t)bal ^ sel);                            // same-bit lanes: bal where set, ~bal where clear
                    m_hi &= ~((uint32_t)(bal >> 32) ^ sel);
                }
                const uint32_t below = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
                const uint32_t prev = __atomic_load_n(&wh[d], __ATOMIC_RELAXED);
                __builtin_amdgcn_wave_barrier();
                if (valid && below == 0u) __atomic_store_n(&wh[d], prev + (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi), __ATOMIC_RELAXED);
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
            s_global_base[tid] = s_hist_excl[tid] + excl - bexcl;
            // (the digit's first slot folded into the waves' offsets here, four additions per digit, instead of a second LDS
            // read and an addition per KEY in the scatter below)
    #pragma unroll
            for (int w = 0; w < 4; ++w) s_wave_hist[w][tid] += bexcl;
            __syncthreads();

            // scatter into LDS in digit order
    #pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const uint32_t idx = row_base + (uint32_t)k * 64u;
                if (FULL || idx < n) {
                    const uint32_t d = (kv[k].x >> shift) & (RADIX_BASE - 1u);
                    const uint32_t slot = s_wave_hist[wave][d] + rank[k];
                    s_pairs[slot] = kv[k];
                }
            }
            __syncthreads();
            const uint32_t count = FULL ? TILE : min(TILE, n - tile_base);
    #pragma unroll
            for (int k = 0; k < KPT; ++k) {
                const uint32_t slot = (uint32_t)k * 256u + (uint32_t)tid;
                if (FULL || slot < count) {
                    uint2 e = s_pairs[slot];
                    const uint32_t d = (e.x >> shift) & (RADIX_BASE - 1u);
                    e.x ^= key_xor;
                    const uint32_t dst = s_global_base[d] + slot;
                    if (dst < n) out[dst] = e;  // always true unless the watchdog tripped
                }
            }
        };
        if ((tile + 1u) * TILE <= n) process(std::true_type{}); else process(std::false_type{});
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
// already placed the pairs into 256 key-range buckets, so the rest of the sort is local. Workgroup = bucket:
//   1. output offset = sum of the counts of the buckets before it (256 words out of L2)
//   2. load the bucket's pairs (one contiguous region: coalesced)
//   3. counting sort in LDS on BUCKET_FINE fine key ranges of the bucket's own [min, max] (returning LDS
//      atomics: order inside a fine range is arbitrary)
//   4. rank every pair among the handful that share its fine range by (key, index) and write it to
//      out[offset + rank] — ascending key, ties by ascending index: the order of the stable LSD passes.
// No workgroup waits for another one. Order never depends on the splitters, balance does: a bucket over
// capacity, or a key value repeated more than BUCKET_FINE_MAX times (step 4 is quadratic in the ties), sets
// ctl->sort_overflow and the host re-runs the frame with the onesweep passes.
// ---------------------------------------------------------------------------------------
// A bucket is one workgroup. With 1024 threads (4 pairs and 2 fine ranges per thread) instead of 256 a launch puts
// 16 waves on every CU instead of 4 and is faster alone (600 k pairs of a 5 M-splat cloud 27.9 -> 15.3 us,
// 120 k pairs 9.9 -> 9.5 us), but slower where it matters, with other frames' kernels sharing the chip
// (kernels.h, BGS_KEYGEN_WIDE_THREADS): 256 is the default.
// WIDE buckets (round 6, lists past ~1.5 M pairs; bgs_device.h BUCKET_CAP_WIDE): 16 384 pairs = 128 KB of the CU's 160 KB
// of LDS, 1024 threads x 16 pairs, 4096 fine ranges. A 5 M-pair list is 768 of them instead of 2816 narrow ones: keygen's
// scatter leaves runs of ~5 pairs per (tile, bucket) instead of 1.5, and a quarter of the workgroups start up.
#ifndef BGS_BUCKET_SORT_THREADS
#define BGS_BUCKET_SORT_THREADS 256
#endif
constexpr uint32_t BUCKET_SORT_THREADS = BGS_BUCKET_SORT_THREADS;
template <uint32_t THREADS, uint32_t CAP, uint32_t NF>  // 256 or 1024 threads; pairs a bucket holds; fine ranges
__global__ __launch_bounds__(THREADS) void bucket_sort_kernel(const uint2* __restrict__ slots,
                                                              uint2* __restrict__ out, Control* ctl,
                                                              uint32_t key_xor, uint32_t nb /* buckets of the frame: 256 * sub */) {
    constexpr uint32_t WAVES = THREADS / 64u;
    constexpr uint32_t EPT = CAP / THREADS;                     // pairs per thread (strided)
    constexpr uint32_t FPT = NF / THREADS;                      // fine ranges per thread (contiguous)
    // the fine ranges' counts, then offsets: a word each — or, for the wide buckets, HALF a word (PACK: 8192 ranges in the 16 KB
    // that 128 KB of pairs leave; counts and offsets are <= CAP = 2^14, and a count that would carry into its neighbour is
    // impossible for the same reason)
    constexpr bool PACK = NF > 4096u;
    constexpr uint32_t FWORDS = PACK ? NF / 2u + 1u : NF + 1u;
    static_assert((NF == 2048u || NF == 4096u || NF == 8192u) && BUCKET_COUNT == 256u && (THREADS == 256u || THREADS == 1024u) &&
                  EPT >= 1u && FPT >= 2u && FPT % 2u == 0u && CAP <= 65535u, "bucket geometry");
    extern __shared__ uint2 s_el[];                              // CAP pairs (dynamic: a wide bucket's 128 KB are past the 64 KB default)
    __shared__ uint32_t s_f[FWORDS];
    __shared__ uint32_t s_tot[WAVES];
    __shared__ uint32_t s_red[WAVES][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x;
    // offset of this bucket in the sorted list, its own count, the fullest bucket (every thread takes a stride of the counts)
    uint32_t before = 0u, mine = 0u, mx = 0u;
    for (uint32_t i = (uint32_t)tid; i < nb; i += THREADS) {
        const uint32_t cnt = ctl->bucket_count[i];
        before += i < b ? cnt : 0u;
        mine += i == b ? cnt : 0u;
        mx = max(mx, cnt);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        before += (uint32_t)__shfl_xor((int)before, off, 64);
        mine += (uint32_t)__shfl_xor((int)mine, off, 64);
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
    }
    if (lane == 0) { s_red[wave][0] = before; s_red[wave][1] = mine; s_tot[wave] = mx; }
    for (uint32_t j = (uint32_t)tid; j < FWORDS; j += THREADS) s_f[j] = 0u;
    __syncthreads();
    uint32_t base = 0u, m = 0u;
    mx = 0u;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; ++w) { base += s_red[w][0]; m += s_red[w][1]; mx = max(mx, s_tot[w]); }
    if (b == 0u && tid == 0) {
        ctl->bucket_max = mx;
        if (mx > CAP) {
            // some bucket lost pairs: the list is void. draw_count = 0 keeps the kernels behind this one
            // (project, raster) away from the unwritten entries; the host re-runs the frame.
            ctl->sort_overflow = 1u;
            ctl->draw_count = 0u;
        }
    }
    // the length of the list: the last bucket's offset + its pairs (a keygen without chains leaves only the per-bucket
    // counts; one with chains has written the same number already)
    if (b == nb - 1u && tid == 0 && mx <= CAP) ctl->draw_count = base + m;
    if (m == 0u || mx > CAP) return;
    __syncthreads();  // s_red / s_tot are reused below

    // ---- 2. load ----
    const uint2* __restrict__ src = slots + (size_t)b * CAP;
    uint2 kv[EPT];
    uint32_t kmn = 0xFFFFFFFFu, kmx = 0u;
#pragma unroll
    for (uint32_t k = 0; k < EPT; ++k) {
        const uint32_t e = k * THREADS + (uint32_t)tid;
        kv[k] = make_uint2(0u, 0u);
        if (e < m) {
            kv[k] = src[e];
            kmn = min(kmn, kv[k].x);
            kmx = max(kmx, kv[k].x);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, off, 64));
        kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, off, 64));
    }
    if (lane == 0) { s_red[wave][0] = kmn; s_red[wave][1] = kmx; }
    __syncthreads();
    kmn = 0xFFFFFFFFu; kmx = 0u;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; ++w) { kmn = min(kmn, s_red[w][0]); kmx = max(kmx, s_red[w][1]); }
    // fine range of a key: floor((key - kmn) * NF / (span + 1)) — monotone in the key and ALL NF ranges in use whatever the
    // span (a shift — round 5 — left between NF / 2 and NF of them in use: up to twice the pairs per range, four times the
    // compares of step 4, which is 17 of a 5 M-pair sort's 51 us). As a multiplication by floor(2^32 NF / (span + 1)) < 2^32.
    const uint32_t span = kmx - kmn;
    const bool direct = span < NF;   // (every key value its own range)
    const uint32_t scale = direct ? 0u : (uint32_t)(((unsigned long long)NF << 32) / ((unsigned long long)span + 1ull));
    auto fine_of = [&](const uint32_t key) -> uint32_t { const uint32_t d = key - kmn; return direct ? d : __umulhi(d, scale); };

    // ---- 3. counting sort on the fine ranges ----
    uint32_t slot[EPT];
#pragma unroll
    for (uint32_t k = 0; k < EPT; ++k) {
        const uint32_t e = k * THREADS + (uint32_t)tid;
        if (e < m) {   // (slot: the fine range in the low half, the arrival order inside it in the high half)
            const uint32_t fb = fine_of(kv[k].x);
            if constexpr (PACK) slot[k] = fb | (((atomicAdd(&s_f[fb >> 1], 1u << ((fb & 1u) << 4)) >> ((fb & 1u) << 4)) & 0xFFFFu) << 16);
            else slot[k] = fb | (atomicAdd(&s_f[fb], 1u) << 16);
        }
    }
    __syncthreads();
    uint32_t f[FPT], fsum = 0u, fmax = 0u;
#pragma unroll
    for (uint32_t j = 0; j < FPT; ++j) {
        if constexpr (PACK) f[j] = (s_f[((uint32_t)tid * FPT + j) >> 1] >> ((j & 1u) << 4)) & 0xFFFFu;   // (FPT is even)
        else f[j] = s_f[(uint32_t)tid * FPT + j];
        fmax = max(fmax, f[j]);
        fsum += f[j];
    }
    // exclusive scan of fsum over the block (wave scan + the waves' totals through LDS)
    const uint32_t finc = wave_inclusive_scan(fsum, lane);
    if (lane == 63) s_tot[wave] = finc;
    __syncthreads();
    uint32_t fexcl = finc - fsum;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; ++w) fexcl += w < (uint32_t)wave ? s_tot[w] : 0u;
    if constexpr (PACK) {
#pragma unroll
        for (uint32_t j = 0; j < FPT; j += 2u) {
            s_f[((uint32_t)tid * FPT + j) >> 1] = fexcl | ((fexcl + f[j]) << 16);
            fexcl += f[j] + f[j + 1u];
        }
        if (tid == 0) s_f[NF / 2u] = m;
    } else {
#pragma unroll
        for (uint32_t j = 0; j < FPT; ++j) {
            s_f[(uint32_t)tid * FPT + j] = fexcl;
            fexcl += f[j];
        }
        if (tid == 0) s_f[NF] = m;
    }
    // first pair of fine range fb in s_el (fb = NF: the bucket's pair count)
    auto fine_start = [&](const uint32_t fb) -> uint32_t {
        if constexpr (PACK) return (uint32_t)reinterpret_cast<const uint16_t*>(s_f)[fb];   // (ds_read_u16)
        else return s_f[fb];
    };
    if (__syncthreads_or(fmax > BUCKET_FINE_MAX ? 1 : 0)) {  // step 4 is quadratic in equal keys: give up
        if (tid == 0) {
            atomicOr(&ctl->sort_overflow, 2u);
            // this bucket's stretch of the list stays unwritten: the kernels behind this one read sort_overflow != 0
            // as "no draw list" (block 255 may publish draw_count AFTER this store: the flag is what is sticky)
            ctl->draw_count = 0u;
        }
        return;
    }
#pragma unroll
    for (uint32_t k = 0; k < EPT; ++k) {
        const uint32_t e = k * THREADS + (uint32_t)tid;
        if (e < m) s_el[fine_start(slot[k] & 0xFFFFu) + (slot[k] >> 16)] = make_uint2(kv[k].y, kv[k].x);   // (index, key): as ONE 64-bit number key-major
    }
    __syncthreads();

    // ---- 4. rank among the fine range's pairs by (key, index), write out ----
    // Eight pairs at a time, their reads issued together (pair -> fine range -> the range's bounds -> the range's pairs is a
    // chain of dependent LDS round trips). What this step costs is its instruction count (16 waves on 4 SIMDs, ~16 pairs a
    // thread): a pair is ONE 64-bit number in LDS (key in the high word), a compare is v_cmp_lt_u64 + an add with carry.
    const unsigned long long* const s_el64 = reinterpret_cast<const unsigned long long*>(s_el);
    constexpr uint32_t G = EPT >= 8u ? 8u : EPT;
#pragma unroll 1
    for (uint32_t k0 = 0; k0 < EPT; k0 += G) {
        if (k0 * THREADS >= m) break;   // (uniform)
        unsigned long long el[G];
        uint32_t fs[G], fe[G];
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            const uint32_t p = (k0 + k) * THREADS + (uint32_t)tid;
            el[k] = s_el64[min(p, m - 1u)];
        }
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            const uint32_t fb = fine_of((uint32_t)(el[k] >> 32));
            fs[k] = fine_start(fb);
            fe[k] = fine_start(fb + 1u);
        }
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            const uint32_t p = (k0 + k) * THREADS + (uint32_t)tid;
            if (p < m) {
                uint32_t r = 0u;
                for (uint32_t j = fs[k]; j < fe[k]; ++j) r += s_el64[j] < el[k] ? 1u : 0u;
                out[base + fs[k] + r] = make_uint2((uint32_t)(el[k] >> 32) ^ key_xor, (uint32_t)el[k]);
            }
        }
    }
}

hipError_t launch_bucket_sort(hipStream_t stream, const uint2* bucket_slots, uint2* out, Control* ctl, uint32_t key_xor, uint32_t buckets,
                              bool wide) {
    if (wide) {
        const auto k = &bucket_sort_kernel<1024u, BUCKET_CAP_WIDE, BUCKET_FINE_WIDE>;
        constexpr int lds = (int)(BUCKET_CAP_WIDE * sizeof(uint2));
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(buckets), dim3(1024), lds, stream, bucket_slots, out, ctl, key_xor, buckets);
    } else {
        hipLaunchKernelGGL((bucket_sort_kernel<BUCKET_SORT_THREADS, BUCKET_CAP, BUCKET_FINE>), dim3(buckets), dim3(BUCKET_SORT_THREADS),
                           BUCKET_CAP * sizeof(uint2), stream, bucket_slots, out, ctl, key_xor, buckets);
    }
    return hipSuccess;
}

// The 255 keys at the 1/256-quantiles of a sorted draw list, in keygen's key space (key ^ key_xor): the
// SplitterTable of later frames. Frames whose rasteriser does the clean-up (BINNING_SCAN) get them from
// there; this one-block kernel serves the others (bgs_sort, BINNING_SORT).
__global__ __launch_bounds__(256) void splitter_kernel(const uint2* __restrict__ sorted, Control* ctl, uint32_t key_xor, uint32_t sub) {
    const uint32_t d = ctl->draw_count, nbk = BUCKET_COUNT * min(max(sub, 1u), BUCKET_SUB_MAX);
    for (uint32_t t = threadIdx.x; t < BUCKET_MAX; t += 256u)
        ctl->splitters[t] = (d != 0u && t < nbk - 1u) ? (sorted[(uint32_t)(((unsigned long long)(t + 1u) * d) / nbk)].x ^ key_xor) : 0xFFFFFFFFu;
}

void launch_splitters(hipStream_t stream, const uint2* sorted, Control* ctl, uint32_t key_xor, uint32_t sub) {
    hipLaunchKernelGGL(splitter_kernel, dim3(1), dim3(256), 0, stream, sorted, ctl, key_xor, sub);
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

// ln_f32_cr on the device over a range of binary32 bit patterns (bgs_selftest_ln_f32: the parity tests hold the
// device build of exact_log.h to the host build and to the oracle's x87 logl on EVERY positive input)
__global__ void __launch_bounds__(256) selftest_ln_kernel(uint32_t first_bits, uint32_t count, float* __restrict__ out,
                                                          unsigned long long* __restrict__ sum) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        const uint32_t in_bits = first_bits + (uint32_t)i;
        const float r = ln_f32_cr(__uint_as_float(in_bits));
        if (out) out[i] = r;
        acc += ln_selftest_mix(in_bits, __float_as_uint(r));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
}

void launch_selftest_ln(hipStream_t stream, uint32_t first_bits, uint32_t count, float* out, unsigned long long* sum,
                        int blocks) {
    hipLaunchKernelGGL(selftest_ln_kernel, dim3(blocks), dim3(256), 0, stream, first_bits, count, out, sum);
}

void launch_triad(hipStream_t stream, float4* a, const float4* b, const float4* c, float s, size_t n4,
                  int blocks) {
    if (n4 == 0) return;
    hipLaunchKernelGGL(triad_kernel, dim3(blocks), dim3(256), 0, stream, a, b, c, s, n4);
}

}  // namespace bgs
