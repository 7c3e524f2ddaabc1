// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 (same number of FMAs per thread):
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/pk_fma_rate.hip -o /tmp/pk_fma_rate && /tmp/pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_scalar(float* out, float b, float c, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i);
#pragma unroll 16
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));  // (plain C gets SLP-packed)
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_packed(float* out, float b, float c, int iters) {
    float2v a[4];
    for (int i = 0; i < 4; ++i) a[i] = float2v{(float)(threadIdx.x + 2 * i), (float)(threadIdx.x + 2 * i + 1)};
    const float2v bb = {b, b}, cc = {c, c};
#pragma unroll 16
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = __builtin_elementwise_fma(a[i], bb, cc);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    const int blocks = 256 * 8, iters = 64000;   // 8 blocks (32 waves) per CU
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best_s = 1e30f, best_p = 1e30f;
    for (int rep = 0; rep < 8; ++rep) {   // alternate; the first repetitions also warm the clocks up
        float ms_s = 0, ms_p = 0;
        hipEventRecord(e0); k_scalar<<<blocks, 256>>>(out, 0.999f, 0.5f, iters); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms_s, e0, e1);
        hipEventRecord(e0); k_packed<<<blocks, 256>>>(out, 0.999f, 0.5f, iters); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms_p, e0, e1);
        printf("rep %d: v_fma_f32 %.3f ms, v_pk_fma_f32 %.3f ms\n", rep, ms_s, ms_p);
        if (ms_s < best_s) best_s = ms_s;
        if (ms_p < best_p) best_p = ms_p;
    }
    const double fmas = (double)blocks * 256 * 8 * iters;
    printf("best: v_fma_f32 %.3f ms = %.1f TFLOP/s   v_pk_fma_f32 %.3f ms = %.1f TFLOP/s   ratio %.2f\n", best_s,
           2 * fmas / best_s / 1e9, best_p, 2 * fmas / best_p / 1e9, best_s / best_p);
    return 0;
}
