// bgs_diag.hip — libbgs: the test hooks and probes of include/bgs_diag.h that launch kernels of their own (the digit
// passes on caller-supplied pairs, the HBM probe, the device self-tests). Counters and switches stay with the context
// (bgs_api.hip / bgs_frame.hip).
#include "bgs_context.h"

extern "C" {

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

int bgs_selftest_ln_f32(bgs_ctx* ctx, uint32_t first_bits, uint32_t count, float* host_out, uint64_t* checksum_out) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (count == 0 || (!host_out && !checksum_out)) return fail(ctx, BGS_EINVAL, "count >= 1 and one of host_out / checksum_out");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    float* d_out = nullptr;
    unsigned long long* d_sum = dev_alloc<unsigned long long>(1);
    if (!d_sum) return fail(ctx, BGS_ENOMEM, "hipMalloc(selftest) failed");
    if (host_out && !(d_out = dev_alloc<float>(count))) {
        (void)hipFree(d_sum);
        return fail(ctx, BGS_ENOMEM, "hipMalloc(selftest output) failed");
    }
    hipStream_t st = ctx->lanes[0].stream;
    unsigned long long sum = 0;
    bool ok = hipMemsetAsync(d_sum, 0, sizeof sum, st) == hipSuccess;
    if (ok) {
        launch_selftest_ln(st, first_bits, count, d_out, d_sum, ctx->num_cus * 8);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(&sum, d_sum, sizeof sum, hipMemcpyDeviceToHost, st) == hipSuccess;
    }
    if (ok && host_out) ok = hipMemcpyAsync(host_out, d_out, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    (void)hipFree(d_sum);
    if (d_out) (void)hipFree(d_out);
    if (!ok) { (void)hipGetLastError(); return fail(ctx, BGS_EHIP, "ln self-test failed on the device"); }
    if (checksum_out) *checksum_out = (uint64_t)sum;
    return BGS_OK;
}

int bgs_selftest_tile_order(bgs_ctx* ctx, const uint16_t* host_cost, uint32_t ntiles, uint32_t runs, uint16_t* host_order, uint32_t* host_sums) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (!host_cost || !host_order || ntiles == 0 || ntiles > 65535u || (runs != 1u && runs != 2u && runs != 4u))
        return fail(ctx, BGS_EINVAL, "tile-order self-test: 1 <= ntiles <= 65535, runs 1 / 2 / 4, both host buffers");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    int rc = finish_all(ctx);
    if (rc != BGS_OK) return rc;
    const uint32_t nblocks = (ntiles + 3u) / 4u;
    uint16_t* d_cost = dev_alloc<uint16_t>(tile_cost_bytes(ntiles) / 2u);
    uint16_t* d_order = dev_alloc<uint16_t>(tile_order_bytes(ntiles) / 2u);
    hipStream_t st = ctx->lanes[0].stream;
    uint32_t pairs[16];
    bool ok = d_cost && d_order &&
              hipMemsetAsync(d_order, 0xFF, tile_order_bytes(ntiles), st) == hipSuccess &&
              hipMemcpyAsync(d_cost, host_cost, (size_t)ntiles * 2u, hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok) {
        launch_tile_order_runs(st, d_cost, d_order, ntiles, runs);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync(host_order, d_order, (size_t)nblocks * 2u, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipMemcpyAsync(pairs, reinterpret_cast<const uint8_t*>(d_order) + tile_order_stats_offset(ntiles), sizeof pairs,
                            hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    if (ok && host_sums) {
        host_sums[0] = host_sums[1] = 0u;
        for (uint32_t x = 0; x < 8u; ++x) { host_sums[0] += pairs[2u * x]; host_sums[1] += pairs[2u * x + 1u]; }
    }
    if (d_cost) (void)hipFree(d_cost);
    if (d_order) (void)hipFree(d_order);
    if (!ok) { (void)hipGetLastError(); return fail(ctx, BGS_EHIP, "tile-order self-test failed on the device"); }
    return BGS_OK;
}

}  // extern "C"
