// Issue cost of the vector instructions the rasteriser's inner loop is made of, measured on gfx950 in SHADER CYCLES
// (s_memtime ticks: no clock assumption), at 1 / 2 / 4 / 8 waves per SIMD:
//
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_issue.hip -o scripts/micro/valu_issue && scripts/micro/valu_issue
//
// Every kernel runs ITERS x 16 instructions of ONE opcode per wave on 8 (or 16) independent registers, between two
// s_memtime reads; the grid is 256 CUs x W blocks of 256 threads, so W waves share each SIMD (the dispatcher deals the
// blocks of a launch round-robin over the CUs; the residency is checked through HW_ID and printed). With W waves
// issuing the same stream, a wave's elapsed cycles are W x instructions x (issue cycles per wave-instruction) once
// the SIMD is issue bound; at W = 1 the figure is max(issue, latency / independent chains). The wall-clock time of the
// launch gives the clock the chip sustained (cycles / time).
// Printed per opcode: cycles per wave-instruction and SIMD (= elapsed / (W x instructions)), for each W.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

enum Op { FMA, PK_FMA, MUL, PK_MUL, ADD, PK_ADD, MAX, MAX_ABS, EXP, RCP, LOG, CNDMASK, CMP, FMA_DEP, MIX_BLEND,
          FMAC, MIN, MED3, MOV, CVT, ADD_U32, AND, LSHL, SQRT, CMP_SGPR, FMA_ABS, MAX3, COUNT };
static const char* const NAMES[COUNT] = {
    "v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_add_f32", "v_pk_add_f32", "v_max_f32",
    "v_max_f32 |a|,|b|", "v_exp_f32", "v_rcp_f32", "v_log_f32", "v_cndmask_b32", "v_cmp_le_f32 (vcc)",
    "v_fma_f32 dependent chain", "blend mix (2 fma, max, cmp, exp, 5 fma)",
    "v_fmac_f32 (VOP2)", "v_min_f32", "v_med3_f32", "v_mov_b32", "v_cvt_f32_i32", "v_add_u32", "v_and_b32", "v_lshlrev_b32",
    "v_sqrt_f32", "v_cmp_le_f32 (sgpr pair, VOP3)", "v_fma_f32 |a|,b,c", "v_max3_f32"};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float b, float c, int iters) {
    float a[16];
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = v2f{a[2 * i], a[2 * i + 1]};
    const v2f bb = {b, b}, cc = {c, c};
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(a[0] > b);
    unsigned long long sg = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == MAX_ABS) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if constexpr (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if constexpr (OP == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(mask));
            if constexpr (OP == CMP) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
            if constexpr (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c));
            if constexpr (OP == CVT) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
            if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            if constexpr (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if constexpr (OP == CMP_SGPR) asm volatile("v_cmp_le_f32 %0, %1, %2" : "=s"(sg) : "v"(a[i]), "v"(c));
            if constexpr (OP == FMA_ABS) asm volatile("v_fma_f32 %0, |%0|, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
            if constexpr (OP == PK_FMA) if (i < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(bb), "v"(cc));
            if constexpr (OP == PK_MUL) if (i < 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(bb));
            if constexpr (OP == PK_ADD) if (i < 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cc));
        }
        if constexpr (OP == MIX_BLEND) {
            // the OBB blend of one record and four pixels as the rasteriser issues it (no exec masking): per pixel
            // 2 fma (u, v), max |u| |v|, compare, then u*u, fma, exp2, mul, min, mul, pk_fma, fma, sub
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u, v, g, e;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(u) : "v"(b), "v"(a[r]), "v"(a[4]));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(c), "v"(a[r]), "v"(a[5]));
                asm volatile("v_max_f32 %0, |%1|, |%2|" : "=v"(g) : "v"(u), "v"(v));
                asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(g), "v"(c) : "vcc");
                asm volatile("v_mul_f32 %0, %1, %1" : "=v"(e) : "v"(u));
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(e) : "v"(v));
                asm volatile("v_exp_f32 %0, -%0" : "+v"(e));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e) : "v"(b));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(e) : "v"(c));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e) : "v"(a[8 + r]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[r]) : "v"(v2f{e, e}), "v"(bb));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[12 + r]) : "v"(e), "v"(b));
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[8 + r]) : "v"(e));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 123.456f || sg == 12345ull) out[0] = 0;  // keeps the registers alive
    if ((threadIdx.x & 63) == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID
        out[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = t1 - t0;
        out[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = hw;
    }
}

template <int OP>
static void run(unsigned long long* d_out, std::vector<unsigned long long>& h, int cus) {
    const int iters = 20000;
    const double per_wave = OP == MIX_BLEND ? 4.0 * 13 * iters : ((OP == PK_FMA || OP == PK_MUL || OP == PK_ADD) ? 8.0 : 16.0) * iters;
    printf("%-42s", NAMES[OP]);
    for (int W : {1, 2, 4, 8}) {
        const int blocks = cus * W, launches = 4;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        double best_ms = 1e30, ticks = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            for (int j = 0; j < launches; ++j) k<OP><<<blocks, 256>>>(d_out, 0.999f, 0.5f, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best_ms) {
                best_ms = ms;
                (void)hipMemcpy(h.data(), d_out, (size_t)blocks * 4 * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                std::vector<unsigned long long> t(blocks * 4);
                for (int i = 0; i < blocks * 4; ++i) t[i] = h[2 * i];
                std::nth_element(t.begin(), t.begin() + t.size() / 2, t.end());
                ticks = (double)t[t.size() / 2];
            }
        }
        // wall clock: the SIMD issued W x per_wave instructions per launch
        const double ns = best_ms * 1e6 / (launches * per_wave * W);
        printf("  W=%d %5.2f ns (%5.2f tk)", W, ns, ticks / (per_wave * W));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    printf("ns = wall-clock time of the launches / (W x instructions per wave): time per wave64 instruction and SIMD with W waves per SIMD\n"
           "(2 cycles of a 2.4 GHz clock = 0.83 ns, 4 cycles = 1.67 ns); (tk) = the same from the median wave's s_memtime ticks\n");
    unsigned long long* d_out;
    const size_t words = (size_t)cus * 8 * 4 * 2;
    hipMalloc(&d_out, words * sizeof(unsigned long long));
    std::vector<unsigned long long> h(words);
    run<FMA>(d_out, h, cus);
    run<PK_FMA>(d_out, h, cus);
    run<MUL>(d_out, h, cus);
    run<PK_MUL>(d_out, h, cus);
    run<ADD>(d_out, h, cus);
    run<PK_ADD>(d_out, h, cus);
    run<MAX>(d_out, h, cus);
    run<MAX_ABS>(d_out, h, cus);
    run<EXP>(d_out, h, cus);
    run<RCP>(d_out, h, cus);
    run<LOG>(d_out, h, cus);
    run<CNDMASK>(d_out, h, cus);
    run<CMP>(d_out, h, cus);
    run<FMA_DEP>(d_out, h, cus);
    run<MIX_BLEND>(d_out, h, cus);
    run<FMAC>(d_out, h, cus);
    run<FMA_ABS>(d_out, h, cus);
    run<MIN>(d_out, h, cus);
    run<MED3>(d_out, h, cus);
    run<MAX3>(d_out, h, cus);
    run<MOV>(d_out, h, cus);
    run<CVT>(d_out, h, cus);
    run<ADD_U32>(d_out, h, cus);
    run<AND>(d_out, h, cus);
    run<LSHL>(d_out, h, cus);
    run<SQRT>(d_out, h, cus);
    run<CMP_SGPR>(d_out, h, cus);
    return 0;
}
