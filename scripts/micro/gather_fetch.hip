// gather_fetch.hip — what rocprofv3's FETCH_SIZE reports on gfx950 for the access patterns of project_bin_kernel:
// gathers of 16 / 32 / 96 / 192-byte records at random record indices of a buffer far larger than the 256 MB Infinity
// Cache, next to the wide coalesced stream the guide's "x2" correction was calibrated on (MI355X_MICROARCH.md, HBM).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_fetch.hip -o /tmp/gather_fetch
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/gf -o gf -- /tmp/gather_fetch
// prints the bytes each kernel asks for per launch; compare with FETCH_SIZE (KB) per kernel name:
//   factor = requested bytes / (FETCH_SIZE * 1024)   (2.0 = the guide's correction applies, 1.0 = counted in full)
// Every kernel touches each record once (a random PERMUTATION of the record indices), so nothing is re-read; records start
// 256 B apart over the whole 1 GiB buffer, so FETCH_SIZE / gathers is the memory-side bytes ONE gather of that size costs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int V4>  // record = V4 float4s
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ buf, const uint32_t* __restrict__ idx,
                                                     uint32_t n, float* __restrict__ out) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4* p = buf + (size_t)idx[i] * 16u;   // records start 256 B apart: no two share a 128-B line
#pragma unroll
        for (int k = 0; k < V4; ++k) { const float4 v = p[k]; acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 123.456f) out[0] = acc;  // never true: keeps the loads
}
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ buf, size_t n4, float* __restrict__ out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256u) {
        const float4 v = buf[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int V4>
void run(const char* name, const float4* buf, size_t buf_bytes, float* out) {
    // one gather per record of the buffer's first `n` records, in a random order (LCG permutation over a power of two)
    const uint32_t n = 1u << 22;  // 4 M gathers
    if ((size_t)n * 256 > buf_bytes) { printf("%s: buffer too small\n", name); return; }
    std::vector<uint32_t> h(n);
    uint32_t x = 12345u;
    for (uint32_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = x & (n - 1u); }  // full-period LCG mod 2^22: a permutation
    uint32_t* d;
    hipMalloc(&d, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gather_kernel<V4>, dim3(4096), dim3(256), 0, 0, buf, d, n, out);
    hipDeviceSynchronize();
    printf("%-28s requests %10.1f KB per launch (%u records of %d B) + %u KB of indices (coalesced)\n", name,
           (double)n * V4 * 16 / 1024.0, n, V4 * 16, n * 4 / 1024);
    hipFree(d);
}

int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB: 4 x the Infinity Cache
    float4* buf;
    float* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream_kernel, dim3(8192), dim3(256), 0, 0, buf, bytes / 16, out);
    hipDeviceSynchronize();
    printf("%-28s requests %10.1f KB per launch (coalesced 16 B / lane)\n", "stream_kernel", (double)bytes / 1024.0);
    run<1>("gather_kernel<1> (16 B)", buf, bytes, out);
    run<2>("gather_kernel<2> (32 B)", buf, bytes, out);
    run<6>("gather_kernel<6> (96 B)", buf, bytes, out);
    run<12>("gather_kernel<12> (192 B)", buf, bytes, out);
    return 0;
}
