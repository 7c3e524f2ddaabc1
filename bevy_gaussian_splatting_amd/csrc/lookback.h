// lookback.h — decoupled look-back over the status words of a chained scan (device only).
//
// Every tile publishes STATUS_AGGREGATE | total as soon as it knows its local total, then sums
// its predecessors' words until it meets a STATUS_PREFIX (inclusive prefix). With every tile of a
// launch resident at once all aggregates appear at about the same time, and a walker consumes B
// predecessors per L2 round trip: a whole wave per hop (64) where the block has one chain (keygen: one
// poller wave per block), 4 where each thread owns a chain (one per digit / bucket / supertile). Wider is NOT better for the per-thread chains: a prefix is usually met within a few words
// and every extra word polled by 256 threads x hundreds of blocks is fabric traffic (one depth pass at
// 74 tiles: 11.6 us with 4 words per hop, 12.4 with 16, 16.4 with 64).
// What matters most is how waiting is done: see lb_backoff.
#pragma once
#include <hip/hip_runtime.h>

#include "bgs_device.h"

namespace bgs {

constexpr uint32_t LOOKBACK_SPIN_LIMIT = 1u << 21;  // bounded spin (watchdog, never expected): >= 2 s with the back-off

__device__ __forceinline__ uint32_t lb_load(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Spin back-off while a predecessor has not published: s_sleep units are 64 clocks. Thousands of
// threads poll at once when several chained-scan kernels share the chip (3 lanes x 512 blocks x 135
// chains on a 5 M-splat frame); without a back-off the device-scope polling loads alone saturate the
// fabric and starve the very blocks everyone is waiting for (observed: the watchdog bound was reached).
__device__ __forceinline__ void lb_backoff(uint32_t spins) {
    if (spins < 8u) __builtin_amdgcn_s_sleep(1);
    else if (spins < 64u) __builtin_amdgcn_s_sleep(8);
    else __builtin_amdgcn_s_sleep(32);
}

// One chain per THREAD (status words `stride` apart): B predecessors per round trip, consumed in
// order up to the first unpublished word or the first inclusive prefix. While the NEAREST predecessor
// is unpublished only that one word is polled.
template <int B>
__device__ __forceinline__ uint32_t lookback_u32(const uint32_t* chain, uint32_t tile, uint32_t stride,
                                                 uint32_t* error_flag, uint32_t error_code) {
    uint32_t excl = 0u, spins = 0u;
    int p = (int)tile - 1;
    while (p >= 0) {
        uint32_t v[B];
#pragma unroll
        for (int b = 0; b < B; ++b) v[b] = p - b >= 0 ? lb_load(chain + (size_t)(p - b) * stride) : STATUS_PREFIX;
        while ((v[0] >> STATUS_FLAG_SHIFT) == 0u) {  // nearest predecessor not there yet: poll it alone
            if (++spins > LOOKBACK_SPIN_LIMIT) { atomicOr(error_flag, error_code); return excl; }
            lb_backoff(spins);
            v[0] = lb_load(chain + (size_t)p * stride);
        }
        int used = 0;
        bool finished = false;
#pragma unroll
        for (int b = 0; b < B; ++b) {
            if (finished || used != b) continue;
            const uint32_t flag = v[b] >> STATUS_FLAG_SHIFT;
            if (flag == 0u) continue;           // not published yet: retry from p - b
            excl += v[b] & STATUS_VALUE_MASK;
            used = b + 1;
            if (flag == 2u) finished = true;    // inclusive prefix: done
        }
        if (finished) break;
        p -= used;
    }
    return excl;
}

// One chain per BLOCK (contiguous status words): called by all 64 lanes of ONE wave; lane l reads
// predecessor p - l, so a hop covers 64 tiles. Returns the exclusive prefix in every lane.
__device__ __forceinline__ uint32_t lookback_wave(const uint32_t* chain, uint32_t tile, int lane,
                                                  uint32_t* error_flag, uint32_t error_code) {
    uint32_t excl = 0u, spins = 0u;
    int p = (int)tile - 1;
    while (p >= 0) {
        const int idx = p - lane;
        const uint32_t v = idx >= 0 ? lb_load(chain + idx) : STATUS_PREFIX;
        const uint32_t flag = v >> STATUS_FLAG_SHIFT;
        const unsigned long long unpub = __ballot(flag == 0u), pref = __ballot(flag == 2u);
        const int first_unpub = unpub ? (int)__builtin_ctzll(unpub) : 64;
        const int first_pref = pref ? (int)__builtin_ctzll(pref) : 64;
        const bool done = first_pref < first_unpub;
        const int take = done ? first_pref + 1 : first_unpub;  // lanes [0, take) are consumed
        uint32_t c = lane < take ? (v & STATUS_VALUE_MASK) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        excl += c;
        if (done) break;
        if (take == 0) {
            if (++spins > LOOKBACK_SPIN_LIMIT) { if (lane == 0) atomicOr(error_flag, error_code); break; }
            lb_backoff(spins);
        }
        p -= take;
    }
    return excl;
}

}  // namespace bgs
