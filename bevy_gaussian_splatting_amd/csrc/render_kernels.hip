// render_kernels.hip — projection + tile binning + tile rasteriser (gfx950, wave64).
//
// The reference rasterises one instanced quad per splat through the fixed-function pipeline
// with premultiplied-alpha blending in back-to-front draw order
// (src/render/mod.rs:925-983, src/render/gaussian.wgsl:184-505). This file produces the same
// image with a compute pipeline:
//
//   project_emit_kernel  vs_points ONCE per splat (the reference runs it per quad vertex, 4x),
//                        walking the sorted draw list from its END, i.e. front-to-back, so
//                        projected records land in HBM in traversal order; then an ordered
//                        (chained-scan) expansion into (tile, rank) instances. Because instances
//                        are emitted in rank order, a STABLE sort on the tile id alone yields
//                        per-tile lists that are already front-to-back (== sorting on
//                        tile-major|depth keys, with 2 instead of 6 digit passes).
//   tile_ranges_kernel   [start, end) of every tile in the tile-sorted instance list.
//   raster_kernel        one 256-thread workgroup per 16x16 tile; batches of 256 records are
//                        staged in LDS (coalesced index read, 16-byte gathers), every thread
//                        owns one pixel and composites front-to-back:
//                            C += T*alpha*c, T *= 1-alpha        (fs_main + blend, reordered)
//                        and stops when every pixel of the tile has T < T_EPS (2^-13).
//
// Front-to-back vs the reference's back-to-front "over" is the same polynomial evaluated in
// the opposite association order; the difference is f32 rounding (<< 1e-3).
#include <algorithm>
#include <type_traits>

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "lookback.h"
#include "splat_math.h"

// Kernel-ablation bits (fp.debug & 1 .. 64: parts of kernels switched off, WRONG images; scripts/ablate.py) exist only in
// builds with -DBGS_ABLATION=1 (scripts/ab_variants.sh builds those into gpurun_variants/); the production
// instantiations carry none of them.
#ifndef BGS_ABLATION
#define BGS_ABLATION 0
#endif

namespace bgs {

namespace {

__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr unsigned long long S64_FLAG_SHIFT = 62;
constexpr unsigned long long S64_VALUE_MASK = (1ull << S64_FLAG_SHIFT) - 1ull;
constexpr unsigned long long S64_AGGREGATE = 1ull << S64_FLAG_SHIFT;
constexpr unsigned long long S64_PREFIX = 2ull << S64_FLAG_SHIFT;
constexpr uint32_t SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// IEEE binary16 -> binary32 (exact; v_cvt_f32_f16 keeps subnormals), like unpack2x16float
__device__ __forceinline__ float half_bits_to_float(unsigned short h) {
    _Float16 x;
    __builtin_memcpy(&x, &h, 2);
    return (float)x;
}
__device__ __forceinline__ float half_lo(uint32_t v) { return half_bits_to_float((unsigned short)(v & 0xFFFFu)); }
__device__ __forceinline__ float half_hi(uint32_t v) { return half_bits_to_float((unsigned short)(v >> 16)); }

// SH coefficient fetchers: all 48 coefficients of one splat, as 16-byte loads.
struct ShF32 {
    const float* base;  // 192-byte records, 16-byte aligned
    __device__ __forceinline__ void load_all(float* c) const {
        const float4* p = reinterpret_cast<const float4*>(base);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float4 v = p[i];
            c[4 * i] = v.x; c[4 * i + 1] = v.y; c[4 * i + 2] = v.z; c[4 * i + 3] = v.w;
        }
    }
};
// f16 plane: u32 word i holds coefficient 2i in the low half, 2i+1 in the high half
// (src/render/planar.wgsl:117-130); 96-byte records, 16-byte aligned.
struct ShF16 {
    const uint32_t* base;
    __device__ __forceinline__ void load_all(float* c) const {
        const uint4* p = reinterpret_cast<const uint4*>(base);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const uint4 v = p[i];
            c[8 * i] = half_lo(v.x); c[8 * i + 1] = half_hi(v.x);
            c[8 * i + 2] = half_lo(v.y); c[8 * i + 3] = half_hi(v.y);
            c[8 * i + 4] = half_lo(v.z); c[8 * i + 5] = half_hi(v.z);
            c[8 * i + 6] = half_lo(v.w); c[8 * i + 7] = half_hi(v.w);
        }
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------
// vertex stage for one front-to-back rank: load the splat, project, write its record.
// Returns the packed tile rectangle (RECT_EMPTY if nothing is to be drawn).
// ---------------------------------------------------------------------------------------
// RASTERIZE_DEPTH normalises by the distances of sorted[count-1] and sorted[1]
// (src/render/gaussian.wgsl:331-340; count = gaussian_uniforms.count = N). The full sorted list is
// drawable-prefix ++ culled-tail (see keygen_kernel), so sorted[j] lives in one of the two buffers.
// Uniform over the launch; every thread reads the same two positions (scalar-cache hits).
__device__ __forceinline__ ColorInputs frame_color_inputs(const FrameParams& fp, const CloudPtrs& cloud,
                                                          const uint2* draw_list, const uint2* culled,
                                                          uint32_t draw_count) {
    ColorInputs ci{0.0f, 0.0f, 0.0f};
    if (fp.rasterize_mode == RASTERIZE_DEPTH && fp.n > 0u) {
        const uint32_t j_first = fp.n > 1u ? 1u : 0u, j_last = fp.n - 1u;
        const uint32_t i_first = j_first < draw_count ? draw_list[j_first].y : culled[j_first - draw_count].y;
        const uint32_t i_last = j_last < draw_count ? draw_list[j_last].y : culled[j_last - draw_count].y;
        const float4 pf = cloud.position_visibility[i_first], pl = cloud.position_visibility[i_last];
        ci.max_distance = distance_to_camera(fp, V3{pf.x, pf.y, pf.z});
        ci.min_distance = distance_to_camera(fp, V3{pl.x, pl.y, pl.z});
    }
    return ci;
}

// The frame's largest colour magnitude (see T_EPS): non-negative floats order like their bits, and after the
// first few waves of a launch the word already holds a value no later wave exceeds, so most skip the atomic.
__device__ __forceinline__ void publish_color_max(Control* ctl, const float mag) {
    const uint32_t bits = __float_as_uint(mag);
    if (bits > __hip_atomic_load(&ctl->color_max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&ctl->color_max_bits, bits);
}

template <int FMT, bool SURFEL, bool ANY_MODE>
__device__ __forceinline__ uint32_t project_rank(const FrameParams& fp, const CloudPtrs& cloud,
                                                 const uint2 entry, const uint32_t j,
                                                 float4* __restrict__ records, ColorInputs ci,
                                                 bool& visible, float& color_mag) {
    const uint32_t si = entry.y;
    // ONE aligned record per splat (CloudPtrs): every load below falls into the same one or two 128-byte lines
    const uint4* __restrict__ rec = cloud.packed + (size_t)si * cloud.packed_v4;
    const float4 pv = *reinterpret_cast<const float4*>(rec);
    float rot[4], so[4], cov[6];
    if constexpr (FMT == (int)CLOUD_COV3D) {
        // Covariance3dOpacity (src/gaussian/f32.rs:218-251): cov3d[6], opacity, pad
        const float4 c0 = *reinterpret_cast<const float4*>(rec + PACK_COV3D), c1 = *reinterpret_cast<const float4*>(rec + PACK_COV3D + 1u);
        cov[0] = c0.x; cov[1] = c0.y; cov[2] = c0.z; cov[3] = c0.w; cov[4] = c1.x; cov[5] = c1.y;
        rot[0] = 1.0f; rot[1] = rot[2] = rot[3] = 0.0f;
        so[0] = so[1] = so[2] = 0.0f; so[3] = c1.z;
    } else if constexpr (FMT == (int)CLOUD_F16) {
        // src/render/planar.wgsl:154-176: first value of each pair in the HIGH half
        const uint4 raw = rec[PACK_RSO_F16];
        rot[0] = half_hi(raw.x); rot[1] = half_lo(raw.x);
        rot[2] = half_hi(raw.y); rot[3] = half_lo(raw.y);
        so[0] = half_hi(raw.z); so[1] = half_lo(raw.z);
        so[2] = half_hi(raw.w); so[3] = half_lo(raw.w);
    } else {
        const float4 r4 = *reinterpret_cast<const float4*>(rec + PACK_ROT);
        const float4 s4 = *reinterpret_cast<const float4*>(rec + PACK_SCALE_OPACITY);
        rot[0] = r4.x; rot[1] = r4.y; rot[2] = r4.z; rot[3] = r4.w;
        so[0] = s4.x; so[1] = s4.y; so[2] = s4.z; so[3] = s4.w;
    }
    Projected pr;
    ci.visibility = pv.w;
    if constexpr (FMT == (int)CLOUD_F16)
        project_splat<ANY_MODE>(fp, entry.x, V3{pv.x, pv.y, pv.z}, rot, so, ShF16{reinterpret_cast<const uint32_t*>(rec + PACK_SH_F16)}, ci, pr);
    else if constexpr (FMT == (int)CLOUD_COV3D)
        project_splat<ANY_MODE>(fp, entry.x, V3{pv.x, pv.y, pv.z}, rot, so, ShF32{reinterpret_cast<const float*>(rec + PACK_SH_F32)}, ci, pr, cov);
    else
        project_splat<ANY_MODE>(fp, entry.x, V3{pv.x, pv.y, pv.z}, rot, so, ShF32{reinterpret_cast<const float*>(rec + PACK_SH_F32)}, ci, pr);
    visible = pr.visible;
    if (!pr.draw) return RECT_EMPTY;
    // fmaxf drops a NaN: a NaN colour poisons its pixels whatever the cut-off is
    color_mag = fmaxf(color_mag, fmaxf(fabsf(pr.color[0]), fmaxf(fabsf(pr.color[1]), fabsf(pr.color[2]))));
    const uint32_t rect = (uint32_t)pr.tx0 | ((uint32_t)pr.tx1 << 8) | ((uint32_t)pr.ty0 << 16) |
                          ((uint32_t)pr.ty1 << 24);
    if constexpr (SURFEL) {
        float4* dst = records + (size_t)j * 6u;
        dst[0] = make_float4(pr.quad.cx, pr.quad.cy, pr.p[0], pr.p[1]);
        // the fragment stage's ray-surfel intersection p = (pcx T2 - T0) x (pcy T2 - T1) is affine in the
        // pixel: p = pcx (T1 x T2) + pcy (T2 x T0) + T0 x T1 (stage_surfel). The three cross products are
        // formed here, once per splat and in double (they cancel for surfels seen edge-on), and travel in
        // the record in place of local_to_pixel.
        const double T0x = pr.surfel.T[0], T0y = pr.surfel.T[1], T0z = pr.surfel.T[2];
        const double T1x = pr.surfel.T[3], T1y = pr.surfel.T[4], T1z = pr.surfel.T[5];
        const double T2x = pr.surfel.T[6], T2y = pr.surfel.T[7], T2z = pr.surfel.T[8];
        dst[1] = make_float4(pr.radius, pr.surfel.mean_x, pr.surfel.mean_y, (float)(T1y * T2z - T1z * T2y));
        dst[2] = make_float4((float)(T1z * T2x - T1x * T2z), (float)(T1x * T2y - T1y * T2x),
                             (float)(T2y * T0z - T2z * T0y), (float)(T2z * T0x - T2x * T0z));
        dst[3] = make_float4((float)(T2x * T0y - T2y * T0x), (float)(T0y * T1z - T0z * T1y),
                             (float)(T0z * T1x - T0x * T1z), (float)(T0x * T1y - T0y * T1x));
        dst[4] = make_float4(pr.color[0], pr.color[1], pr.color[2], pr.color[3]);
        dst[5] = make_float4(pr.ndc_z, 0.0f, 0.0f, 0.0f);
    } else {
        float4* dst = records + (size_t)j * 3u;
        dst[0] = make_float4(pr.quad.cx, pr.quad.cy, pr.p[0], pr.p[1]);
        dst[1] = make_float4(pr.p[2], pr.p[3], pr.p[4], pr.color[0]);
        dst[2] = make_float4(pr.color[1], pr.color[2], pr.color[3], pr.ndc_z);
    }
    return rect;
}

// ---------------------------------------------------------------------------------------
// BINNING_SORT: project + ordered instance emission
// ---------------------------------------------------------------------------------------
template <int FMT, bool SURFEL, bool ANY_MODE>
__global__ __launch_bounds__(256) void project_emit_kernel(FrameParams fp, CloudPtrs cloud,
                                                           const uint2* __restrict__ draw_list,
                                                           const uint2* __restrict__ culled,
                                                           Control* ctl,
                                                           unsigned long long* scan_status,
                                                           float4* __restrict__ records,
                                                           uint2* __restrict__ instances,
                                                           uint32_t capacity, uint32_t ticket_slot) {
    __shared__ uint32_t s_prefix[256];  // exclusive tile-count prefix of the block's splats
    __shared__ uint32_t s_rect[256];
    __shared__ uint32_t s_histx[RADIX_BASE];
    __shared__ uint32_t s_histy[RADIX_BASE];
    __shared__ uint32_t s_tot[4];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_tile;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a bucket sort that gave up has voided the list, whatever a later block of it wrote to draw_count (sticky)
    const uint32_t count = ctl->sort_overflow ? 0u : ctl->draw_count;
    const uint32_t num_tiles = (count + 255u) / 256u;
    if (num_tiles == 0u) return;
    s_histx[tid] = 0u;
    s_histy[tid] = 0u;
    uint32_t visible_acc = 0u;
    float color_mag = 0.0f;  // max |r|, |g|, |b| of the records this thread wrote
    const ColorInputs ci = ANY_MODE ? frame_color_inputs(fp, cloud, draw_list, culled, count)
                                    : ColorInputs{0.0f, 0.0f, 0.0f};

    for (;;) {
        if (tid == 0) s_tile = atomicAdd(&ctl->ticket[ticket_slot][0], 1u);
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= num_tiles) break;
        const uint32_t j = tile * 256u + (uint32_t)tid;  // front-to-back rank
        uint32_t ntiles = 0u, rect = 0u;
        if (j < count) {
            // the LAST entry of the draw list is drawn on top => it is the front-most
            bool vis;
            const uint32_t r = project_rank<FMT, SURFEL, ANY_MODE>(fp, cloud, draw_list[count - 1u - j], j, records, ci, vis, color_mag);
            visible_acc += vis ? 1u : 0u;
            if (r != RECT_EMPTY) {
                rect = r;
                ntiles = (((r >> 8) & 255u) - (r & 255u) + 1u) * ((r >> 24) - ((r >> 16) & 255u) + 1u);
            }
        }
        // block exclusive scan of the tile counts
        const uint32_t inc = wave_inclusive_scan(ntiles, lane);
        if (lane == 63) s_tot[wave] = inc;
        s_rect[tid] = rect;
        __syncthreads();
        const uint32_t w0 = s_tot[0], w1 = s_tot[1], w2 = s_tot[2], w3 = s_tot[3];
        const uint32_t woff = wave == 0 ? 0u : (wave == 1 ? w0 : (wave == 2 ? w0 + w1 : w0 + w1 + w2));
        const uint32_t block_total = w0 + w1 + w2 + w3;
        s_prefix[tid] = woff + inc - ntiles;

        // chained scan over blocks (64-bit words: totals may exceed 2^32 before clamping)
        if (tid == 0) {
            unsigned long long excl = 0ull;
            if (tile > 0u) {
                st_agent64(scan_status + tile, S64_AGGREGATE | (unsigned long long)block_total);
                uint32_t p = tile - 1u, spins = 0u;
                for (;;) {
                    const unsigned long long v = ld_agent64(scan_status + p);
                    const unsigned long long flag = v >> S64_FLAG_SHIFT;
                    if (flag == 0ull) {
                        if (++spins > SPIN_LIMIT) { atomicOr(&ctl->error, 2u); break; }
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                    excl += v & S64_VALUE_MASK;
                    if (flag == 2ull || p == 0u) break;
                    --p;
                }
            }
            const unsigned long long incl = excl + (unsigned long long)block_total;
            st_agent64(scan_status + tile, S64_PREFIX | (incl & S64_VALUE_MASK));
            s_base = excl;
            if (tile == num_tiles - 1u) {  // this block owns the grand total
                const bool overflow = incl > (unsigned long long)capacity;
                ctl->instance_total_lo = (uint32_t)incl;
                ctl->instance_total_hi = (uint32_t)(incl >> 32);
                ctl->overflow = overflow ? 1u : 0u;
                // on overflow the later stages see an empty list; the host grows and re-runs
                ctl->instance_count = overflow ? 0u : (uint32_t)incl;
            }
        }
        // per-digit histograms of the tile keys (digit 0 = tile x, digit 1 = tile y): a splat
        // with a w x h tile rectangle adds h to each of its w columns and w to each of its h rows
        if (ntiles) {
            const uint32_t tx0 = rect & 255u, tx1 = (rect >> 8) & 255u;
            const uint32_t ty0 = (rect >> 16) & 255u, ty1 = rect >> 24;
            const uint32_t w = tx1 - tx0 + 1u, h = ty1 - ty0 + 1u;
            for (uint32_t x = tx0; x <= tx1; ++x) atomicAdd(&s_histx[x], h);
            for (uint32_t y = ty0; y <= ty1; ++y) atomicAdd(&s_histy[y], w);
        }
        __syncthreads();

        // cooperative, coalesced expansion: instance i of the block belongs to the splat t with
        // s_prefix[t] <= i < s_prefix[t+1]
        const unsigned long long base = s_base;
        for (uint32_t i = (uint32_t)tid; i < block_total; i += 256u) {
            uint32_t lo = 0u, hi = 255u;  // largest t with s_prefix[t] <= i
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if (s_prefix[mid] <= i) lo = mid; else hi = mid - 1u;
            }
            const uint32_t r = s_rect[lo];
            const uint32_t tx0 = r & 255u, tx1 = (r >> 8) & 255u, ty0 = (r >> 16) & 255u;
            const uint32_t w = tx1 - tx0 + 1u;
            const uint32_t k = i - s_prefix[lo];
            const uint32_t row = k / w, col = k - row * w;
            const unsigned long long g = base + (unsigned long long)i;
            if (g < (unsigned long long)capacity)
                instances[g] = make_uint2(((ty0 + row) << 8) | (tx0 + col), tile * 256u + lo);
        }
        __syncthreads();
    }
    // flush block-private histograms
    {
        const uint32_t hx = s_histx[tid], hy = s_histy[tid];
        if (hx) atomicAdd(&ctl->hist_tile[0][tid], hx);
        if (hy) atomicAdd(&ctl->hist_tile[1][tid], hy);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        visible_acc += __shfl_down(visible_acc, off, 64);
        color_mag = fmaxf(color_mag, __shfl_down(color_mag, off, 64));
    }
    if (lane == 0 && visible_acc) atomicAdd(&ctl->visible_count, visible_acc);
    __shared__ float s_cmag[4];
    if (lane == 0) s_cmag[wave] = color_mag;
    __syncthreads();
    if (tid == 0) publish_color_max(ctl, fmaxf(fmaxf(s_cmag[0], s_cmag[1]), fmaxf(s_cmag[2], s_cmag[3])));
}

void launch_project_emit(hipStream_t stream, const FrameParams& fp, const CloudPtrs& cloud,
                         const uint2* draw_list, const uint2* culled, Control* ctl,
                         unsigned long long* scan_status,
                         void* records, uint2* instances, uint32_t capacity, uint32_t ticket_slot,
                         int max_blocks) {
    if (fp.n == 0) return;
    uint32_t blocks = (fp.n + 255u) / 256u;
    if (blocks > (uint32_t)max_blocks) blocks = (uint32_t)max_blocks;
    const bool surfel = fp.gaussian_mode == 0u && fp.aabb != 0u;
    float4* rec = (float4*)records;
    const bool any_mode = fp.rasterize_mode != RASTERIZE_COLOR || fp.draw_mode != 0u;
#define BGS_LAUNCH_PE(F16, SURFEL, ANY)                                                             \
    hipLaunchKernelGGL((project_emit_kernel<F16, SURFEL, ANY>), dim3(blocks), dim3(256), 0, stream, \
                       fp, cloud, draw_list, culled, ctl, scan_status, rec, instances, capacity,    \
                       ticket_slot)
#define BGS_LAUNCH_PE2(F16, SURFEL) \
    do { if (any_mode) BGS_LAUNCH_PE(F16, SURFEL, true); else BGS_LAUNCH_PE(F16, SURFEL, false); } while (0)
    if (cloud.format == CLOUD_F16) {
        if (surfel) BGS_LAUNCH_PE2(1, true); else BGS_LAUNCH_PE2(1, false);
    } else if (cloud.format == CLOUD_COV3D) {
        BGS_LAUNCH_PE2(2, false);  // 2DGS needs rotation and scale: refused for these clouds (validate)
    } else {
        if (surfel) BGS_LAUNCH_PE2(0, true); else BGS_LAUNCH_PE2(0, false);
    }
#undef BGS_LAUNCH_PE2
#undef BGS_LAUNCH_PE
}

// ---------------------------------------------------------------------------------------
// BINNING_SCAN: the vertex stage (project_kernel) and the ordered coarse binning (bin_kernel)
//
// Rounds 1-3 ran both in one kernel: 256 ranks per workgroup, a ticket and a 256-wide look-back chain per workgroup
// wrapped around a 160-190-VGPR vertex stage at 2 waves per SIMD — 470 tiles for the headline frame's 120 k ranks, the
// last of which waited 6 us for its ticket and 10 us in the chains (profiles/r3_notes.md section 11) while holding those
// registers. The two halves want opposite things, so they are two kernels since round 4:
//   project_kernel   rank -> record + packed tile rectangle. No order between ranks: no ticket, no chain, a static grid.
//   bin_kernel       reads 4 bytes per rank, 1024 ranks per workgroup (118 tiles for the headline frame, so the chains
//                    are a quarter as long), ~30 VGPRs: its waves cost the other lanes' kernels nothing while they spin.
// ---------------------------------------------------------------------------------------
template <int FMT, bool SURFEL, bool ANY_MODE>
__global__ __launch_bounds__(256) void project_kernel(const FrameParams* __restrict__ fpp, CloudPtrs cloud,
                                                      const uint2* __restrict__ draw_list,
                                                      const uint2* __restrict__ culled, Control* ctl,
                                                      float4* __restrict__ records, uint32_t* __restrict__ rects) {
    // left in device memory by the frame's keygen (kernels.h, KeygenLaunch). Read THROUGH the pointer, where it is used:
    // a by-value copy parks ~110 dwords in scalar registers and the compiler spills them to vector lanes
    // (v_writelane / v_readlane: 220 of the kernel's 1700 vector instructions)
    const FrameParams& fp = *fpp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a bucket sort that gave up has voided the list, whatever a later block of it wrote to draw_count (sticky)
    const uint32_t count = ctl->sort_overflow ? 0u : ctl->draw_count;
    const uint32_t num_tiles = (count + 255u) / 256u;
    if (num_tiles == 0u) return;
    uint32_t visible_acc = 0u;
    float color_mag = 0.0f;  // max |r|, |g|, |b| of the records this thread wrote
    const ColorInputs ci = ANY_MODE ? frame_color_inputs(fp, cloud, draw_list, culled, count)
                                    : ColorInputs{0.0f, 0.0f, 0.0f};
    for (uint32_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const uint32_t j = tile * 256u + (uint32_t)tid;  // front-to-back rank
        if (j < count) {
            bool vis = false;
            // the LAST entry of the draw list is drawn on top => it is the front-most
            rects[j] = project_rank<FMT, SURFEL, ANY_MODE>(fp, cloud, draw_list[count - 1u - j], j, records, ci, vis, color_mag);
            visible_acc += vis ? 1u : 0u;
        }
    }
    // one atomic per BLOCK: same-address atomics retire one at a time (~8 ns), and a launch's last
    // waves all arrive here together
    __shared__ uint32_t s_vis[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        visible_acc += __shfl_down(visible_acc, off, 64);
        color_mag = fmaxf(color_mag, __shfl_down(color_mag, off, 64));
    }
    __shared__ float s_cmag[4];
    if (lane == 0) {
        s_vis[wave] = visible_acc;
        s_cmag[wave] = color_mag;
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t v = s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3];
        if (v) atomicAdd(&ctl->visible_count, v);
        publish_color_max(ctl, fmaxf(fmaxf(s_cmag[0], s_cmag[1]), fmaxf(s_cmag[2], s_cmag[3])));
    }
}

// ORDERED coarse binning without a sort. A workgroup of BIN_GROUPS waves takes BIN_RANKS consecutive ranks, wave q the
// group of 64 ranks [64 q, 64 q + 64) of the tile. A rank overlaps supertile (sx, sy) iff sx is in its x-range AND sy
// is in its y-range, so the per-supertile lane masks factor into sup_x column masks and sup_y row masks per group:
// sup_x + sup_y ballots instead of sup_x * sup_y. Thread = supertile (the first 256 threads) then counts its hits per
// group, runs the chained-scan look-back of the radix sort over the workgroups (<= 256 supertiles ride the 256-wide
// chain) and the block appends its hits to every list in rank order: lists are front-to-back by construction.
// WAVES = 16 (a 1024-thread workgroup, one wave per group): the phases are as parallel as with one workgroup per 256
// ranks and the chain is a quarter as long — the fastest shape for a frame that is alone on the chip (project + bin
// 27.6 us against the fused kernel's 29.2 on the headline frame, 78 against 90 on the 5 M-splat scene-like frame). But a
// 16-wave workgroup needs a whole CU's worth of wave slots at once, and with the frames of eight lanes in flight that
// costs throughput (like the wide keygen and bucket sort, kernels.h): WAVES = 4 (256 threads, four groups per wave:
// slower alone — 31 / 130 us on those two frames — and +15 % / +11 % frames per second in flight on the 5 M-splat
// frames, +1-2 % at 1 M) is what pipelined frames run. Same-box A/B: profiles/r4_experiments/project_bin_split.txt.
#ifndef BGS_BIN_LOOKBACK
#define BGS_BIN_LOOKBACK 4   // status words per look-back hop (profiles/r5_experiments/bin_lookback.txt)
#endif
constexpr uint32_t BIN_GROUPS = 16u, BIN_RANKS = 64u * BIN_GROUPS;
template <uint32_t WAVES>
__global__ __launch_bounds__(64u * WAVES) void bin_kernel(const uint32_t* __restrict__ rects, Control* ctl, uint32_t* bin_status,
                                                          uint32_t* __restrict__ coarse, uint32_t coarse_cap, uint32_t sup_mul,
                                                          uint32_t sup_x, uint32_t sup_y, uint32_t ticket_slot) {
    constexpr uint32_t GPW = BIN_GROUPS / WAVES;          // groups per wave
    constexpr uint32_t TPS = 64u * WAVES / MAX_SUPERTILES; // threads per supertile in the sparse append (1 or 4)
    static_assert(WAVES == 4u || WAVES == 16u, "256 or 1024 threads");
    __shared__ unsigned long long s_xmask[BIN_GROUPS][32];
    __shared__ unsigned long long s_ymask[BIN_GROUPS][32];
    __shared__ uint32_t s_rect[BIN_RANKS];       // packed tile rectangle of each of the block's ranks
    __shared__ uint32_t s_excl[MAX_SUPERTILES];  // list offset of the block's first hit, per supertile
    __shared__ uint16_t s_off[BIN_GROUPS][MAX_SUPERTILES];  // ... and of every group's first hit behind it
    __shared__ uint32_t s_block_hits;            // list entries this block appends (picks the append strategy)
    __shared__ uint32_t s_tile;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t count = ctl->sort_overflow ? 0u : ctl->draw_count;
    const uint32_t num_tiles = (count + BIN_RANKS - 1u) / BIN_RANKS;
    if (num_tiles == 0u) return;
    const uint32_t num_st = sup_x * sup_y;
    const uint32_t st_of_thread = (uint32_t)tid & (MAX_SUPERTILES - 1u);
    const uint32_t my_sy = st_of_thread / sup_x, my_sx = st_of_thread - my_sy * sup_x;
    const bool single_shot = gridDim.x >= num_tiles;  // one ticket per block (see keygen_kernel)

    for (;;) {
        // tiles are handed out by an atomic ticket, so a tile's predecessors have always started: nothing assumes
        // dispatch order
        if (tid == 0) { s_tile = atomicAdd(&ctl->ticket[ticket_slot][0], 1u); s_block_hits = 0u; }
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= num_tiles) break;
        const uint32_t rank0 = tile * BIN_RANKS;
        uint32_t rect_g[GPW];
#pragma unroll
        for (uint32_t g = 0u; g < GPW; ++g) {   // the wave's loads first, all in flight together
            const uint32_t j = rank0 + ((uint32_t)wave * GPW + g) * 64u + (uint32_t)lane;
            rect_g[g] = j < count ? rects[j] : RECT_EMPTY;
        }
#pragma unroll
        for (uint32_t g = 0u; g < GPW; ++g) {
            const uint32_t q = (uint32_t)wave * GPW + g, rect = rect_g[g];
            s_rect[q * 64u + (uint32_t)lane] = rect;
            // supertile bounds of the rectangle; an empty rect has x0 = 255 > x1 = 0, so sx0 > sx1: no column matches
            // tile / supertile edge by reciprocal multiply (exact for tiles < 256, supertile_div)
            const uint32_t sx0 = supertile_div(rect & 255u, sup_mul), sx1 = supertile_div((rect >> 8) & 255u, sup_mul);
            const uint32_t sy0 = supertile_div((rect >> 16) & 255u, sup_mul), sy1 = supertile_div(rect >> 24, sup_mul);
            for (uint32_t c = 0u; c < sup_x; ++c) {
                const unsigned long long b = __ballot(c >= sx0 && c <= sx1);
                if (lane == 0) s_xmask[q][c] = b;
            }
            for (uint32_t r = 0u; r < sup_y; ++r) {
                const unsigned long long b = __ballot(r >= sy0 && r <= sy1);
                if (lane == 0) s_ymask[q][r] = b;
            }
        }
        __syncthreads();
        // thread = supertile: hits of the block per group, chained scan over the blocks
        if ((uint32_t)tid < num_st) {
            uint32_t total = 0u;
#pragma unroll 4
            for (uint32_t g = 0u; g < BIN_GROUPS; ++g) {
                s_off[g][tid] = (uint16_t)total;
                total += (uint32_t)__popcll(s_xmask[g][my_sx] & s_ymask[g][my_sy]);
            }
            uint32_t* const my_status = bin_status + (size_t)tile * MAX_SUPERTILES + tid;
            uint32_t excl = 0u;
            if (tile > 0u) {
                __hip_atomic_store(my_status, STATUS_AGGREGATE | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // 4 words per hop at every size (measured on the fused kernel of rounds 1-3: 16 per hop 4 % slower, 32: 11 %)
                excl = lookback_u32<BGS_BIN_LOOKBACK>(bin_status + tid, tile, MAX_SUPERTILES, &ctl->error, 4u);
            }
            __hip_atomic_store(my_status, STATUS_PREFIX | ((excl + total) & STATUS_VALUE_MASK),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tile == num_tiles - 1u) ctl->coarse_total[tid] = excl + total;
            s_excl[tid] = excl;
            if (total) atomicAdd(&s_block_hits, total);
        }
        __syncthreads();
        // Append the block's hits to the supertile lists in rank order. On the dense workload the cost of this step is
        // the stores themselves (scattered 8-byte stores), so a block with many hits makes them contiguous: wave v takes
        // supertiles v, v + WAVES, ...; for each of the block's groups, lane l owns hit bit l and the set lanes store to
        // consecutive list slots. A block with few hits per list (the scene-like workload) lets the threads of a
        // supertile (one, or four with four groups each) walk the bits.
        // entry = (rank, its packed tile rectangle): the rasteriser's candidate scan then is one coalesced
        // 8-byte stream instead of a rank stream plus a 64-line gather of rects[rank].
        if (s_block_hits >= 32u * num_st) {
            for (uint32_t st = (uint32_t)wave; st < num_st; st += WAVES) {
                const uint32_t sy = st / sup_x, sx = st - sy * sup_x;
                uint32_t pos = s_excl[st];
                uint2* __restrict__ dst = reinterpret_cast<uint2*>(coarse) + (size_t)st * coarse_cap;
                for (uint32_t g = 0u; g < BIN_GROUPS; ++g) {
                    const unsigned long long m = s_xmask[g][sx] & s_ymask[g][sy];
                    const uint32_t at = pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (((m >> lane) & 1ull) && at < coarse_cap)
                        dst[at] = make_uint2(rank0 + g * 64u + (uint32_t)lane, s_rect[g * 64u + (uint32_t)lane]);
                    pos += (uint32_t)__popcll(m);
                }
            }
        } else if (st_of_thread < num_st) {
            uint2* __restrict__ dst = reinterpret_cast<uint2*>(coarse) + (size_t)st_of_thread * coarse_cap;
            const uint32_t g0 = ((uint32_t)tid >> 8) * (BIN_GROUPS / TPS);    // this thread's groups
            for (uint32_t g = g0; g < g0 + BIN_GROUPS / TPS; ++g) {
                unsigned long long bits = s_xmask[g][my_sx] & s_ymask[g][my_sy];
                uint32_t pos = s_excl[st_of_thread] + s_off[g][st_of_thread];
                while (bits) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(bits);
                    bits &= bits - 1ull;
                    if (pos < coarse_cap) dst[pos] = make_uint2(rank0 + g * 64u + l, s_rect[g * 64u + l]);
                    ++pos;
                }
            }
        }
        if (single_shot) break;
        __syncthreads();
    }
}

void launch_project_bin(hipStream_t stream, const FrameParams& fp, const FrameParams* d_fp, const CloudPtrs& cloud,
                        const uint2* draw_list, const uint2* culled, Control* ctl, uint32_t* bin_status,
                        void* records, uint32_t* rects, uint32_t* coarse, uint32_t coarse_cap, uint32_t sup_edge,
                        uint32_t ticket_slot, int project_blocks, int bin_blocks, bool wide_bin) {
    if (fp.n == 0) return;
    uint32_t blocks = (fp.n + 255u) / 256u;
    if (blocks > (uint32_t)project_blocks) blocks = (uint32_t)project_blocks;
    const uint32_t sup = sup_edge, sup_mul = supertile_mul(sup_edge);
    const uint32_t sup_x = ((uint32_t)fp.tiles_x + sup - 1u) / sup, sup_y = ((uint32_t)fp.tiles_y + sup - 1u) / sup;
    const bool surfel = fp.gaussian_mode == 0u && fp.aabb != 0u;
    float4* rec = (float4*)records;
    const bool any_mode = fp.rasterize_mode != RASTERIZE_COLOR || fp.draw_mode != 0u;
#define BGS_LAUNCH_PB(F16, SURFEL, ANY)                                                            \
    hipLaunchKernelGGL((project_kernel<F16, SURFEL, ANY>), dim3(blocks), dim3(256), 0, stream,     \
                       d_fp, cloud, draw_list, culled, ctl, rec, rects)
#define BGS_LAUNCH_PB2(F16, SURFEL) \
    do { if (any_mode) BGS_LAUNCH_PB(F16, SURFEL, true); else BGS_LAUNCH_PB(F16, SURFEL, false); } while (0)
    if (cloud.format == CLOUD_F16) {
        if (surfel) BGS_LAUNCH_PB2(1, true); else BGS_LAUNCH_PB2(1, false);
    } else if (cloud.format == CLOUD_COV3D) {
        BGS_LAUNCH_PB2(2, false);  // 2DGS needs rotation and scale: refused for these clouds (validate)
    } else {
        if (surfel) BGS_LAUNCH_PB2(0, true); else BGS_LAUNCH_PB2(0, false);
    }
#undef BGS_LAUNCH_PB2
#undef BGS_LAUNCH_PB
    uint32_t bblocks = (fp.n + BIN_RANKS - 1u) / BIN_RANKS;
    if (bblocks > (uint32_t)bin_blocks) bblocks = (uint32_t)bin_blocks;
    if (wide_bin)
        hipLaunchKernelGGL(bin_kernel<16u>, dim3(bblocks), dim3(1024), 0, stream, rects, ctl, bin_status, coarse, coarse_cap,
                           sup_mul, sup_x, sup_y, ticket_slot);
    else
        hipLaunchKernelGGL(bin_kernel<4u>, dim3(bblocks), dim3(256), 0, stream, rects, ctl, bin_status, coarse, coarse_cap,
                           sup_mul, sup_x, sup_y, ticket_slot);
}

// ---------------------------------------------------------------------------------------
// tile ranges
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint2* __restrict__ inst,
                                                          const Control* ctl, uint2* ranges) {
    const uint32_t n = ctl->instance_count;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t key = inst[i].x;
        if (i == 0u || inst[i - 1u].x != key) ranges[key].x = i;
        if (i == n - 1u || inst[i + 1u].x != key) ranges[key].y = i + 1u;
    }
}

void launch_tile_ranges(hipStream_t stream, const uint2* instances, const Control* ctl, uint2* ranges) {
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(2048), dim3(256), 0, stream, instances, ctl, ranges);
}

// ---------------------------------------------------------------------------------------
// tile rasteriser
// ---------------------------------------------------------------------------------------
constexpr int RV_OBB = 0, RV_AABB3D = 1, RV_SURFEL = 2;
// A pixel stops compositing once its transmittance is below the frame's cut-off
//     t_eps = min(T_EPS, T_BUDGET / cmax),   cmax = the largest colour magnitude among the frame's records
// (project kernels, Control::color_max_bits). What the splats behind could still add is at most t_eps * cmax
// (telescoping sum), so the dropped tail is <= T_BUDGET = 2^-11 = 4.9e-4, half the 1e-3 tolerance, whatever the
// colour range, and <= T_EPS = 2^-13 = 1.2e-4 for colours up to 4; T_EPS is the cut-off of the 3DGS reference
// rasteriser (T < 1e-4), which only ever sees colours in [0, 1]. SH colours are not clamped and an f32 / Rgba16Float
// target keeps them, so a fixed cut-off is no bound at all: with a fixed 2^-13 the 200-seed randomized sweep found two
// frames 3e-3 and 8e-3 off (colours of ~50 behind a stack of alpha-clamped splats). The synthetic benchmark clouds
// (SH ~ U(-1, 1)) reach |c| = 15: t_eps = 2^-14.9 there. Same-box A/B of FIXED cut-offs on the headline frame
// (profiles/r2_notes.md): 2^-16 15.0 k, 2^-13 15.9 k, 2^-12 16.5 k frames/s.
#ifndef BGS_T_EPS_LOG2
#define BGS_T_EPS_LOG2 13
#endif
constexpr float T_EPS = 1.0f / (float)(1u << BGS_T_EPS_LOG2);
constexpr float T_BUDGET = 1.0f / 2048.0f;
__device__ __forceinline__ float frame_t_eps(const uint32_t color_max_bits) {
    return fminf(T_EPS, T_BUDGET / __uint_as_float(color_max_bits));  // cmax = 0 -> T_EPS; inf -> 0: never cut
}

// One staged record, decoded once per splat and shared by every pixel a lane owns.
template <int VARIANT>
struct StagedRecord {
    float4 a0, a1, a2, a3, a4, a5;
    __device__ __forceinline__ void load(const float4* __restrict__ rec) {
        a0 = rec[0];
        a1 = rec[1];
        a2 = rec[2];
        if constexpr (VARIANT == 2) {
            a3 = rec[3];
            a4 = rec[4];
            a5 = rec[5];
        }
    }
};

// OBB records are re-expressed at staging time in the coordinates of the tile that blends them, with
// the falloff constant folded in, so that the per-pixel work is two fmas, a compare and exp2:
//   c  = sqrt(4.5 * log2(e))             exp(-4.5 (u^2 + v^2)) = exp2(-(u'^2 + v'^2)),  u' = c u
//   a0 = (U0', V0', m00', m01'),  a1 = (m10', m11', -, r)   with m' = c m and (U0', V0') = (u', v') at the
//   centre (ox, oy) of the tile's first pixel;  u' = fma(m01', yl, fma(m00', xl, U0')) for the pixel
//   (xl, yl) of the tile. Covered <=> max(|u'|, |v'|) <= c.
// Both rasterisers stage through this function, so their images stay bit-identical.
constexpr float OBB_C = 2.5479654147f;  // sqrt(4.5 * 1.4426950408889634)
__device__ __forceinline__ void stage_obb(float4& r0, float4& r1, const float ox, const float oy) {
    const float m00 = OBB_C * r0.z, m01 = OBB_C * r0.w, m10 = OBB_C * r1.x, m11 = OBB_C * r1.y;
    const float dx = ox - r0.x, dy = oy - r0.y;
    r0 = make_float4(fmaf(m01, dy, m00 * dx), fmaf(m11, dy, m10 * dx), m00, m01);
    r1.x = m10;
    r1.y = m11;
}

// 2DGS surfel records (AABB quad, gaussian.wgsl:440-455 + gaussian_2d.wgsl:134-156) are re-expressed at
// staging time too. The fragment stage intersects the pixel's ray with the surfel as
//     p = (pcx T2 - T0) x (pcy T2 - T1) = pcx (T1 x T2) + pcy (T2 x T0) + T0 x T1,   (us, vs) = p.xy / p.z
// and pcx = u radius + mean.x, pcy = v radius aspect + mean.y are affine in the pixel, so p is an affine
// vec3 function of the pixel's position (xl, yl) inside the tile: p = P0 + Px xl + Py yl. The cross products
// come with the record (formed once per splat, in double, by project_rank); P0, Px, Py are formed here ONCE per
// record and tile, and a pixel then costs three FMAs and one reciprocal instead of six FMAs and a cross
// product (where a ray grazes the surfel plane p.z cancels in either form; the oracle's ambiguity bound covers it). Constants are folded as for OBB:
//     exp(-0.5 min(s3, s2)) = exp2(-min(c^2 s3, c^2 s2)), c^2 = 0.5 log2(e): p.xy carry c, the deltas sqrt(2) c.
// Staged layout (6 x float4):
//   a0 = U0 du V0 dv          u = fma(du, xl, U0), v = fma(dv, yl, V0): the quad's own uv (coverage)
//   a1 = P0x P0y P0z Pxx | a2 = Pxy Pxz Pyx Pyy | a3 = Pyz Dx0 dDx Dy0 | a4 = dDy r g b | a5 = opacity keep - -
// Both rasterisers stage through this function, so their images stay bit-identical.
__device__ __forceinline__ void stage_surfel(const float4* __restrict__ src, const float ox, const float oy,
                                             const float aspect, float4 out[6]) {
    // record: cx cy m00 m11 | radius mean.x mean.y A.x | A.y A.z B.x B.y | B.z C.x C.y C.z | rgba
    // with A = T1 x T2, B = T2 x T0, C = T0 x T1 (project_rank)
    const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3], r4 = src[4];
    const float Ax = r1.w, Ay = r2.x, Az = r2.y, Bx = r2.z, By = r2.w, Bz = r3.x, Cx = r3.y, Cy = r3.z, Cz = r3.w;
    const float U0 = r0.z * (ox - r0.x), V0 = r0.w * (oy - r0.y);  // uv = (m00 dx, m11 dy) at the tile's first pixel
    const float radius = r1.x;
    const float ax = r0.z * radius, bx = fmaf(U0, radius, r1.y);                         // pcx = ax xl + bx
    const float ay = r0.w * radius * aspect, by = fmaf(V0 * radius, aspect, r1.z);       // pcy = ay yl + by
    constexpr float C1 = 0.84932180028801907f;   // sqrt(0.5 * log2(e))
    constexpr float C2 = 1.2011224087864498f;    // sqrt(2) * C1
    out[0] = make_float4(U0, r0.z, V0, r0.w);
    out[1] = make_float4(C1 * fmaf(Ax, bx, fmaf(Bx, by, Cx)), C1 * fmaf(Ay, bx, fmaf(By, by, Cy)),
                         fmaf(Az, bx, fmaf(Bz, by, Cz)), C1 * Ax * ax);
    out[2] = make_float4(C1 * Ay * ax, Az * ax, C1 * Bx * ay, C1 * By * ay);
    out[3] = make_float4(Bz * ay, C2 * (r1.y - bx), -C2 * ax, C2 * (r1.z - by));
    out[4] = make_float4(-C2 * ay, r4.x, r4.y, r4.z);
    out[5] = make_float4(r4.w, 0.0f, 0.0f, 0.0f);
}

// A surfel's AABB quad is the square around the LONGER side of its 3-sigma ellipse (bounding_box_cov2d:
// max(rx, ry)), so most of the tiles a quad touches see nothing of the ellipse: on the dense 1 M-surfel frame
// 54 % of the (record, tile) pairs reach no pixel with more than 2^-23 of the surfel's opacity (measured on the
// host with the exact per-pixel expression). Such a pair is dropped at staging time by a conservative bound:
// p is affine in the pixel, so over the tile's 16 x 16 pixel centres |p.x| and |p.y| are at least their value
// at the tile centre minus the half-range of the affine part, |p.z| at most centre plus half-range, the same
// for the screen-space deltas, and power = min(s3, s2) >= min of the two bounds (interval arithmetic; within
// 2 % of the exact count). What is dropped is at most 2^-23 / max(1, cmax) of a record's opacity (cmax: the
// frame's largest colour magnitude, as for the transmittance cut-off), i.e. <= 1.2e-7 of colour per record —
// 7e-5 if all of a saturated tile's ~600 records were dropped at the threshold, against the 1e-3 tolerance;
// DESIGN.md "Documented deviations". A NaN anywhere keeps the record.
#ifndef BGS_SURFEL_CULL_LOG2
#define BGS_SURFEL_CULL_LOG2 23
#endif
__device__ __forceinline__ bool surfel_negligible_in_tile(const float4 st[6], const float limit) {
    constexpr float H = 7.5f;  // pixel centres of a tile: xl, yl in [0, 15]
    const float pxc = fmaf(H, st[1].w + st[2].z, st[1].x), ex = H * (fabsf(st[1].w) + fabsf(st[2].z));
    const float pyc = fmaf(H, st[2].x + st[2].w, st[1].y), ey = H * (fabsf(st[2].x) + fabsf(st[2].w));
    const float pzc = fmaf(H, st[2].y + st[3].x, st[1].z), ez = H * (fabsf(st[2].y) + fabsf(st[3].x));
    const float lx = fmaxf(fabsf(pxc) - ex, 0.0f), ly = fmaxf(fabsf(pyc) - ey, 0.0f), hz = fabsf(pzc) + ez;
    const float s3 = fmaf(lx, lx, ly * ly) * __builtin_amdgcn_rcpf(hz * hz);  // 1 ulp is nothing to a bound with this margin
    const float dx = fmaxf(fabsf(fmaf(H, st[3].z, st[3].y)) - H * fabsf(st[3].z), 0.0f);
    const float dy = fmaxf(fabsf(fmaf(H, st[4].x, st[3].w)) - H * fabsf(st[4].x), 0.0f);
    const float s2 = fmaf(dx, dx, dy * dy);
    return s3 >= limit && s2 >= limit;  // staged powers are in exp2 units
}
// the threshold of the frame: 2^-23 of a record's opacity, lowered by the frame's largest colour magnitude
// like the transmittance cut-off (what is dropped is alpha * colour)
__device__ __forceinline__ float frame_surfel_limit(const uint32_t color_max_bits) {
    return (float)BGS_SURFEL_CULL_LOG2 + __builtin_amdgcn_logf(fmaxf(1.0f, __uint_as_float(color_max_bits)));  // log2
}

// fs_main + blend for ONE record and ONE pixel (src/render/gaussian.wgsl:438-505,
// src/render/mod.rs:944-948), front-to-back form, branch-free:
//     w = covered && T >= t_eps ? T * alpha : 0;   C += w * c;   T -= w
// (T - T*alpha == T*(1 - alpha); a pixel stops accumulating once T < t_eps (frame_t_eps), a per-pixel rule that
// does not depend on how splats are batched, so every rasteriser variant gives the same bits).
// Explicit fmaf so both rasterisers contract identically.
typedef float v2f __attribute__((ext_vector_type(2)));
// DEPTH: the fragment is also tested against the pixel's scene depth `dpx` (src/render/mod.rs:959-974: GreaterEqual on
// a reverse-Z Depth32Float attachment, no write); `z` is the quad's constant depth.
// BBOX: CloudSettings::visualize_bounding_box (src/gaussian/settings.rs:95, key bit src/render/mod.rs:418,824,
// src/render/gaussian.wgsl:486-495): a fragment in the outer 8 % of the quad's uv square — uv * 0.5 + 0.5 outside
// [0.08, 0.92], i.e. max(|u|, |v|) > 0.84 — is (0.3, 1, 0.1, 1) instead of the splat's colour: alpha 1, whatever is
// behind it is hidden (front-to-back: the pixel's transmittance drops to 0). fs_main's AABB discard (power > 0) comes
// first, as in the shader. A separate instantiation: frames without the switch carry none of it.
constexpr float BBOX_EDGE_WIDTH = 0.08f;                       // gaussian.wgsl:488
constexpr float BBOX_EDGE = 1.0f - 2.0f * BBOX_EDGE_WIDTH;      // uv * 0.5 + 0.5 outside [w, 1 - w]  <=>  |uv| > 1 - 2 w
template <int VARIANT, bool DEPTH = false, bool BBOX = false>
__device__ __forceinline__ void blend_px(const StagedRecord<VARIANT>& s, const float qx, const float qy,
                                         const float aspect, const float t_eps, float& T, v2f& crg, float& cb,
                                         const float z = 0.0f, const float dpx = 0.0f) {
    float alpha, r, g, b;
    bool hit;
    [[maybe_unused]] float bu = 0.0f, bv = 0.0f, blim = 1.0f;   // the quad's own uv and its limit (BBOX)
    if constexpr (VARIANT == RV_OBB) {
        // staged by stage_obb: a0 = U0' V0' m00' m01' | a1 = m10' m11' - r | a2 = g b a -; here
        // (qx, qy) is the pixel's position INSIDE the tile (0..15)
        const float u = fmaf(s.a0.w, qy, fmaf(s.a0.z, qx, s.a0.x));
        const float v = fmaf(s.a1.y, qy, fmaf(s.a1.x, qx, s.a0.y));
        hit = fmaxf(fabsf(u), fabsf(v)) <= OBB_C;
        if constexpr (BBOX) { bu = u; bv = v; blim = OBB_C; }
        // fs_main OBB: power = -dot(uv,uv) / (2 * (1/3)^2)  (gaussian.wgsl:474-480);
        // exp(power) = exp2(-(u'^2 + v'^2))
        const float e = __builtin_amdgcn_exp2f(-fmaf(u, u, v * v));
        alpha = fminf(e * s.a2.z, 0.999f);
        r = s.a1.w; g = s.a2.x; b = s.a2.y;
    } else if constexpr (VARIANT == RV_AABB3D) {
        const float dx = qx - s.a0.x, dy = qy - s.a0.y;
        // a0 = cx cy m00 m11 | a1 = A B C r | a2 = g b a rect
        const float u = s.a0.z * dx, v = s.a0.w * dy;
        if constexpr (BBOX) { bu = u; bv = v; }
        hit = fmaxf(fabsf(u), fabsf(v)) <= 1.0f;
        const float power = fmaf(s.a1.y * u, v, -0.5f * fmaf(s.a1.x * u, u, s.a1.z * v * v));
        hit = hit && !(power > 0.0f);
        alpha = fminf(__expf(power) * s.a2.z, 0.999f);
        r = s.a1.w; g = s.a2.x; b = s.a2.y;
    } else {
        // staged by stage_surfel; (qx, qy) is the pixel's position INSIDE the tile (0..15)
        const float u = fmaf(s.a0.y, qx, s.a0.x), v = fmaf(s.a0.w, qy, s.a0.z);
        if constexpr (BBOX) { bu = u; bv = v; }
        hit = fmaxf(fabsf(u), fabsf(v)) <= 1.0f;
        // surfel_fragment_power (gaussian_2d.wgsl:134-156): p = P0 + Px xl + Py yl, the x part is shared
        // by the pixels of a lane
        const float px = fmaf(s.a2.z, qy, fmaf(s.a1.w, qx, s.a1.x));
        const float py = fmaf(s.a2.w, qy, fmaf(s.a2.x, qx, s.a1.y));
        const float pz = fmaf(s.a3.x, qy, fmaf(s.a2.y, qx, s.a1.z));
        // one reciprocal instead of two IEEE divisions (v_rcp_f32 is good to 1 ulp, far inside the 1e-3
        // tolerance of the image)
        const float icz = __builtin_amdgcn_rcpf(pz);
        const float us = px * icz, vs = py * icz;
        const float ddx = fmaf(s.a3.z, qx, s.a3.y), ddy = fmaf(s.a4.x, qy, s.a3.w);
        const float s3 = fmaf(us, us, vs * vs);
        const float s2 = fmaf(ddx, ddx, ddy * ddy);
        // power = -0.5 min(sigmas_3d, sigmas_2d) <= 0 always (fs_main's `power > 0` discard never fires; a NaN
        // goes through min / exp as in the reference's expression)
        alpha = fminf(__builtin_amdgcn_exp2f(-fminf(s3, s2)) * s.a5.x, 0.999f);
        r = s.a4.y; g = s.a4.z; b = s.a4.w;
    }
    if constexpr (DEPTH) hit = hit && z >= dpx;
    if constexpr (BBOX) {
        const bool frame = fmaxf(fabsf(bu), fabsf(bv)) > BBOX_EDGE * blim;
        alpha = frame ? 1.0f : alpha; r = frame ? 0.3f : r; g = frame ? 1.0f : g; b = frame ? 0.1f : b;
    }
    // a real branch on purpose: it becomes an exec-mask region that a wave skips entirely when none
    // of its 64 pixels (a 16x4 strip in the wave-per-tile rasteriser) is covered
    if (hit && T >= t_eps) {
        asm volatile("");  // not speculatable: keeps this a branch (the compiler would turn it into selects)
        const float w = T * alpha;
        // red and green as one packed fma on the register pair the record's (r, g) arrive in: left to itself
        // the compiler pairs red with blue and spends two moves per pixel on assembling the operand
        crg = __builtin_elementwise_fma((v2f){w, w}, (v2f){r, g}, crg);
        cb = fmaf(w, b, cb);
        T -= w;
    }
}

// ---------------------------------------------------------------------------------------
// 4x multisampling: MultisampleState { count: key.sample_count } with sample_count = Msaa::samples() of the camera
// (src/render/mod.rs:357,412,422,975-979; Bevy's default is Msaa::Sample4 and nothing in the reference sets another).
// The fixed-function pipeline then decides COVERAGE (and the depth test) per sample, runs fs_main ONCE per pixel with
// the interpolants taken at the pixel centre (`@interpolate(linear)`: centre sampling, gaussian.wgsl:146-162;
// extrapolated when the centre itself is outside the quad), blends the same source colour into every covered sample,
// and resolves to the mean of the samples. So per pixel alpha and colour are shared and only the transmittance is per
// sample: C += alpha c mean_s(cov_s T_s);  T_s *= 1 - alpha for the covered samples; the resolved pixel is
// C + clear * mean_s(T_s). Sample positions: the standard 4x pattern (Vulkan / D3D / Metal), relative to the centre
//     s0 (-1/8, -3/8)   s1 (+3/8, -1/8)   s2 (-3/8, +1/8) = -s1   s3 (+1/8, +3/8) = -s0.
// State of a pixel: T_s = S * r[s]. A record that covers ALL samples of a pixel — nearly every (record, pixel) pair
// away from the quad edges — only scales S (one fma, like the single-sample path); only a pixel the quad's edge passes
// through scales its covered r[s]. rb = mean_s r[s] rides along so that the mean transmittance S * rb costs one
// multiplication. A pixel stops accumulating once S * rb < t_eps: what the splats behind could still add is at most
// mean_s(T_s) * cmax, the bound of the single-sample rule.
// Which of the two updates a pixel takes is decided per WAVE (the general one runs as soon as one active pixel of the
// wave needs it; it is correct for fully covered pixels as well): both rasterisers give a wave the same 16 x 4 pixels,
// so their images stay bit-identical.
// ---------------------------------------------------------------------------------------
struct PxMs { float S, rb, r0, r1, r2, r3; };
constexpr float MS_OX0 = -0.125f, MS_OY0 = -0.375f, MS_OX1 = 0.375f, MS_OY1 = -0.125f;
// largest |change of u| between the pixel centre and one of its samples, u = a x + b y (staged coefficients)
__device__ __forceinline__ float ms_margin(const float a, const float b) {
    return fmaxf(fabsf(fmaf(b, MS_OY0, a * MS_OX0)), fabsf(fmaf(b, MS_OY1, a * MS_OX1)));
}
// 1 where |x| <= lim, 0 where |x| > lim (exactly: clamp((lim - |x|) * 2^60) — 1 from lim - |x| >= 2^-60 on, 0 from
// lim - |x| <= 0 down; see blend_px_ms): ONE v_fma_f32 with the abs / neg input and the clamp output modifier
__device__ __forceinline__ float ms_inside(const float x, const float big, const float limbig) {
    float r;
    asm("v_fma_f32 %0, -|%1|, %2, %3 clamp" : "=v"(r) : "v"(x), "v"(big), "v"(limbig));
    return r;
}
// float4 per STAGED record in the wave-per-tile rasteriser's LDS: the 4x OBB record carries a fourth one, the four sample
// offsets of (u, v) — wave-uniform values every pixel's per-sample update needs, formed once by the staging lane instead of
// with eight vector instructions per (record, tile) in the record loop (round 6)
// (not under a depth buffer: with the tile's 16 KB of depth samples a fifth workgroup would no longer fit a CU's LDS)
// ... and not in the dense frames' instantiation (mid-round exit): their strips rarely take the per-sample update, the compiler
// had sunk the eight instructions into it, and the fourth LDS read per record only costs (dense 1 M -3.5 % frames/s; scene-like
// +1.5 %, trained-like +2.5 %, 5 M scene-like +2.1 %: profiles/r6_experiments/staged_sample_offsets_ab.txt)
__host__ __device__ constexpr int staged_v4(const int variant, const int msaa, const bool depth, const bool dense) {
    return variant == 2 ? 6 : (variant == 0 && msaa == 4 && !depth && !dense ? 4 : 3);
}
// acc = fma(-a, b, acc), the result in acc's own register
__device__ __forceinline__ void ms_fnma_in_place(float& acc, const float a, const float b) {
    asm("v_fma_f32 %0, -%1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int VARIANT, bool DEPTH, bool BBOX = false, bool PRE = false>   // PRE: s.a3 = (du0, du1, dv0, dv1), staged
__device__ __forceinline__ void blend_px_ms(const StagedRecord<VARIANT>& s, const float qx, const float qy,
                                            const float aspect, const float t_eps, PxMs& t, v2f& crg, float& cb,
                                            const float z, const bool zmixed, const float4 dpx) {
    float u, v, lim, m, alpha, r, g, b, du0, du1, dv0, dv1;
    bool ok = true;
    if constexpr (VARIANT == RV_OBB) {
        // staged by stage_obb: a0 = U0' V0' m00' m01' | a1 = m10' m11' margin r | a2 = g b a keepz
        u = fmaf(s.a0.w, qy, fmaf(s.a0.z, qx, s.a0.x));
        v = fmaf(s.a1.y, qy, fmaf(s.a1.x, qx, s.a0.y));
        lim = OBB_C;
        m = s.a1.z;
        const float e = __builtin_amdgcn_exp2f(-fmaf(u, u, v * v));
        alpha = fminf(e * s.a2.z, 0.999f);
        // fs_main's OBB discard (dot(uv, uv) > 9, gaussian.wgsl:481-483) can only fire for a fragment shaded at a pixel centre
        // OUTSIDE a small quad (a sample is covered, the centre extrapolates); its alpha is < e^-40 there, so dropping it
        // or not is the same image — except under the overlay, which would paint it opaque
        if constexpr (BBOX) ok = !(fmaf(u, u, v * v) > 9.0f * OBB_C * OBB_C);
        r = s.a1.w; g = s.a2.x; b = s.a2.y;
        if constexpr (PRE) { du0 = s.a3.x; du1 = s.a3.y; dv0 = s.a3.z; dv1 = s.a3.w; }
        else {
            du0 = fmaf(s.a0.w, MS_OY0, s.a0.z * MS_OX0); du1 = fmaf(s.a0.w, MS_OY1, s.a0.z * MS_OX1);
            dv0 = fmaf(s.a1.y, MS_OY0, s.a1.x * MS_OX0); dv1 = fmaf(s.a1.y, MS_OY1, s.a1.x * MS_OX1);
        }
    } else if constexpr (VARIANT == RV_AABB3D) {
        const float dx = qx - s.a0.x, dy = qy - s.a0.y;
        u = s.a0.z * dx; v = s.a0.w * dy;
        lim = 1.0f;
        m = 0.375f * fmaxf(fabsf(s.a0.z), fabsf(s.a0.w));
        const float power = fmaf(s.a1.y * u, v, -0.5f * fmaf(s.a1.x * u, u, s.a1.z * v * v));
        ok = !(power > 0.0f);   // fs_main's discard: the fragment — all its samples — is dropped
        alpha = fminf(__expf(power) * s.a2.z, 0.999f);
        r = s.a1.w; g = s.a2.x; b = s.a2.y;
        du0 = s.a0.z * MS_OX0; du1 = s.a0.z * MS_OX1; dv0 = s.a0.w * MS_OY0; dv1 = s.a0.w * MS_OY1;
    } else {
        u = fmaf(s.a0.y, qx, s.a0.x); v = fmaf(s.a0.w, qy, s.a0.z);
        lim = 1.0f;
        m = 0.375f * fmaxf(fabsf(s.a0.y), fabsf(s.a0.w));
        const float px = fmaf(s.a2.z, qy, fmaf(s.a1.w, qx, s.a1.x));
        const float py = fmaf(s.a2.w, qy, fmaf(s.a2.x, qx, s.a1.y));
        const float pz = fmaf(s.a3.x, qy, fmaf(s.a2.y, qx, s.a1.z));
        const float icz = __builtin_amdgcn_rcpf(pz);
        const float us = px * icz, vs = py * icz;
        const float ddx = fmaf(s.a3.z, qx, s.a3.y), ddy = fmaf(s.a4.x, qy, s.a3.w);
        const float s3 = fmaf(us, us, vs * vs);
        const float s2 = fmaf(ddx, ddx, ddy * ddy);
        alpha = fminf(__builtin_amdgcn_exp2f(-fminf(s3, s2)) * s.a5.x, 0.999f);
        r = s.a4.y; g = s.a4.z; b = s.a4.w;
        du0 = s.a0.y * MS_OX0; du1 = s.a0.y * MS_OX1; dv0 = s.a0.w * MS_OY0; dv1 = s.a0.w * MS_OY1;
    }
    const float gmax = fmaxf(fabsf(u), fabsf(v));
    if constexpr (BBOX) {   // the fragment at the pixel centre is the quad's frame (see blend_px)
        const bool frame = gmax > BBOX_EDGE * lim;
        alpha = frame ? 1.0f : alpha; r = frame ? 0.3f : r; g = frame ? 1.0f : g; b = frame ? 0.1f : b;
    }
    // beyond lim + m no sample of the pixel is covered; within lim - m all four are
    // (bitwise on purpose, round 6: both compares up front and ONE exec-mask region — the short-circuit form nests two:
    // three scalar instructions more per visited strip, for one multiplication and one compare less per missed one. What
    // the record loop costs is its instruction count, scalar ones included: scene-like 1 M +1.3 %, trained-like +2.5 %
    // frames/s, rasteriser alone -3 % / -8 %; profiles/r6_experiments/one_region_ab.txt)
    if ((gmax <= lim + m) & ok & (t.S * t.rb >= t_eps)) {
        asm volatile("");  // keeps this a branch (see blend_px)
        const bool full = gmax <= lim - m;
        // TWO one-armed regions, not an if / else: as one region the compiler gives the per-sample arm an edge into the
        // other one (its lowering of a branch on a ballot inside an exec-masked region) and keeps the old r0 .. r3, rb alive
        // along it — five v_mov per visited strip, ten with the updates forced in place. A one-armed region updates in place.
        // (The surfel variant takes the first region as a branch, not a select: dense 1 M surfels 3.02 -> 3.06 k frames/s, the
        // scene-like surfel frame 10.6 k either way, 10.35 k with the if / else — profiles/r6_experiments/surfel_form_ab.txt)
#ifndef BGS_SURFEL_FORM
#define BGS_SURFEL_FORM 2
#endif
        constexpr int FORM = VARIANT == RV_SURFEL ? BGS_SURFEL_FORM : 1;   // 0: if / else, 1: two regions, 2: two regions, the first a branch
        const bool all_full = !(DEPTH && zmixed) && __builtin_amdgcn_ballot_w64(!full) == 0ull;
        // Coverage of a sample as a 0 / 1 factor out of multiplications, additions and the clamp output modifier:
        // min / max, compares and selects issue at half the rate of those on this chip (wave64: 4 clocks against 2;
        // profiles/r4_micro/valu_issue.txt). clamp((lim - |x|) * 2^60) is 1 for |x| <= lim - 2^-60, 0 for |x| >= lim —
        // and something in between for a sample closer to the quad's edge than any rasteriser's rounding can place it.
        auto per_sample = [&](float& w) {
            const float big = 1.152921504606846976e18f, limbig = lim * 1.152921504606846976e18f;   // 2^60
            float t0 = ms_inside(u + du0, big, limbig) * ms_inside(v + dv0, big, limbig) * t.r0;
            float t1 = ms_inside(u + du1, big, limbig) * ms_inside(v + dv1, big, limbig) * t.r1;
            float t2 = ms_inside(u - du1, big, limbig) * ms_inside(v - dv1, big, limbig) * t.r2;
            float t3 = ms_inside(u - du0, big, limbig) * ms_inside(v - dv0, big, limbig) * t.r3;
            if constexpr (DEPTH) {
                t0 = z >= dpx.x ? t0 : 0.0f; t1 = z >= dpx.y ? t1 : 0.0f;
                t2 = z >= dpx.z ? t2 : 0.0f; t3 = z >= dpx.w ? t3 : 0.0f;
            }
            const float sum = (t0 + t1) + (t2 + t3), aq = 0.25f * alpha;
            w = (t.S * aq) * sum;
            if constexpr (FORM == 0) {
                t.r0 = fmaf(-alpha, t0, t.r0); t.r1 = fmaf(-alpha, t1, t.r1);
                t.r2 = fmaf(-alpha, t2, t.r2); t.r3 = fmaf(-alpha, t3, t.r3);
                t.rb = fmaf(-aq, sum, t.rb);
            } else {
                ms_fnma_in_place(t.r0, alpha, t0); ms_fnma_in_place(t.r1, alpha, t1);
                ms_fnma_in_place(t.r2, alpha, t2); ms_fnma_in_place(t.r3, alpha, t3);
                ms_fnma_in_place(t.rb, aq, sum);
            }
        };
        float w;
        if constexpr (FORM == 0) {
            if (all_full) {
                w = (t.S * t.rb) * alpha;
                t.S = fmaf(-alpha, t.S, t.S);
            } else {
                per_sample(w);
            }
        } else {
            w = (t.S * t.rb) * alpha;
            if constexpr (FORM == 1) { if (all_full) t.S = fmaf(-alpha, t.S, t.S); }   // (the compiler makes it a select; as a branch: the same within noise)
            else { if (all_full) { asm volatile(""); t.S = fmaf(-alpha, t.S, t.S); } }
            asm volatile("");   // (keeps the two regions apart)
            if (!all_full) per_sample(w);
        }
        crg = __builtin_elementwise_fma((v2f){w, w}, (v2f){r, g}, crg);
        cb = fmaf(w, b, cb);
    }
}

// ---------------------------------------------------------------------------------------
// Msaa::Sample2 and Msaa::Sample8 (Bevy's `Msaa` enum feeds MultisampleState.count unfiltered: src/render/mod.rs:357,
// 412-424, 975-979): the same scheme with N samples at the graphics APIs' standard positions, written out generally (no
// symmetric pairs to exploit at 8x). Offsets from the pixel centre:
//   2x  (+1/4, +1/4) (-1/4, -1/4)
//   8x  (+1, -3) (-1, +3) (+5, +1) (-3, -5) (-5, +5) (-7, -1) (+3, +7) (+7, -7)  / 16
// The 4x instantiations above stay as they are (the headline's inner loop); nothing in the reference selects 2 or 8.
// ---------------------------------------------------------------------------------------
template <int NS> struct PxMsN { float S, rb, r[NS]; };
template <int NS> __device__ constexpr float ms_ox(const int k) {
    constexpr float o2[2] = {0.25f, -0.25f};
    constexpr float o8[8] = {0.0625f, -0.0625f, 0.3125f, -0.1875f, -0.3125f, -0.4375f, 0.1875f, 0.4375f};
    return NS == 2 ? o2[k & 1] : o8[k & 7];
}
template <int NS> __device__ constexpr float ms_oy(const int k) {
    constexpr float o2[2] = {0.25f, -0.25f};
    constexpr float o8[8] = {-0.1875f, 0.1875f, 0.0625f, -0.3125f, 0.3125f, -0.0625f, 0.4375f, -0.4375f};
    return NS == 2 ? o2[k & 1] : o8[k & 7];
}
// largest |offset| component of the pattern (the axis-aligned variants' margin), and the half extent of the box a tile's
// sample positions span around the tile centre
constexpr float ms_reach(const int ns) { return ns == 1 ? 0.0f : ns == 2 ? 0.25f : ns == 4 ? 0.375f : 0.4375f; }
// largest |change of u| between the pixel centre and one of its samples, u = a x + b y
template <int NS> __device__ __forceinline__ float ms_margin_n(const float a, const float b) {
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < NS; ++k) m = fmaxf(m, fabsf(fmaf(b, ms_oy<NS>(k), a * ms_ox<NS>(k))));
    return m;
}
template <int VARIANT, bool DEPTH, int NS, bool BBOX>
__device__ __forceinline__ void blend_px_msn(const StagedRecord<VARIANT>& s, const float qx, const float qy,
                                             const float aspect, const float t_eps, PxMsN<NS>& t, v2f& crg, float& cb,
                                             const float z, const bool zmixed, const float* __restrict__ dpx) {
    float u, v, lim, m, alpha, r, g, b, au, bu_, av, bv_;   // u = au x + bu y (+ const), v = av x + bv y (+ const)
    bool ok = true;
    if constexpr (VARIANT == RV_OBB) {
        u = fmaf(s.a0.w, qy, fmaf(s.a0.z, qx, s.a0.x));
        v = fmaf(s.a1.y, qy, fmaf(s.a1.x, qx, s.a0.y));
        lim = OBB_C;
        m = s.a1.z;
        const float e = __builtin_amdgcn_exp2f(-fmaf(u, u, v * v));
        alpha = fminf(e * s.a2.z, 0.999f);
        if constexpr (BBOX) ok = !(fmaf(u, u, v * v) > 9.0f * OBB_C * OBB_C);   // fs_main's OBB discard (see blend_px_ms)
        r = s.a1.w; g = s.a2.x; b = s.a2.y;
        au = s.a0.z; bu_ = s.a0.w; av = s.a1.x; bv_ = s.a1.y;
    } else if constexpr (VARIANT == RV_AABB3D) {
        const float dx = qx - s.a0.x, dy = qy - s.a0.y;
        u = s.a0.z * dx; v = s.a0.w * dy;
        lim = 1.0f;
        m = ms_reach(NS) * fmaxf(fabsf(s.a0.z), fabsf(s.a0.w));
        const float power = fmaf(s.a1.y * u, v, -0.5f * fmaf(s.a1.x * u, u, s.a1.z * v * v));
        ok = !(power > 0.0f);
        alpha = fminf(__expf(power) * s.a2.z, 0.999f);
        r = s.a1.w; g = s.a2.x; b = s.a2.y;
        au = s.a0.z; bu_ = 0.0f; av = 0.0f; bv_ = s.a0.w;
    } else {
        u = fmaf(s.a0.y, qx, s.a0.x); v = fmaf(s.a0.w, qy, s.a0.z);
        lim = 1.0f;
        m = ms_reach(NS) * fmaxf(fabsf(s.a0.y), fabsf(s.a0.w));
        const float px = fmaf(s.a2.z, qy, fmaf(s.a1.w, qx, s.a1.x));
        const float py = fmaf(s.a2.w, qy, fmaf(s.a2.x, qx, s.a1.y));
        const float pz = fmaf(s.a3.x, qy, fmaf(s.a2.y, qx, s.a1.z));
        const float icz = __builtin_amdgcn_rcpf(pz);
        const float us = px * icz, vs = py * icz;
        const float ddx = fmaf(s.a3.z, qx, s.a3.y), ddy = fmaf(s.a4.x, qy, s.a3.w);
        const float s3 = fmaf(us, us, vs * vs);
        const float s2 = fmaf(ddx, ddx, ddy * ddy);
        alpha = fminf(__builtin_amdgcn_exp2f(-fminf(s3, s2)) * s.a5.x, 0.999f);
        r = s.a4.y; g = s.a4.z; b = s.a4.w;
        au = s.a0.y; bu_ = 0.0f; av = 0.0f; bv_ = s.a0.w;
    }
    const float gmax = fmaxf(fabsf(u), fabsf(v));
    if constexpr (BBOX) {
        const bool frame = gmax > BBOX_EDGE * lim;
        alpha = frame ? 1.0f : alpha; r = frame ? 0.3f : r; g = frame ? 1.0f : g; b = frame ? 0.1f : b;
    }
    if ((gmax <= lim + m) & ok & (t.S * t.rb >= t_eps)) {   // (one exec-mask region: see blend_px_ms)
        asm volatile("");  // keeps this a branch (see blend_px)
        const bool full = gmax <= lim - m;
        const bool all_full = !(DEPTH && zmixed) && __builtin_amdgcn_ballot_w64(!full) == 0ull;   // (two one-armed regions: see blend_px_ms)
        float w = (t.S * t.rb) * alpha;
        if (all_full) t.S = fmaf(-alpha, t.S, t.S);
        asm volatile("");
        if (!all_full) {
            const float big = 1.152921504606846976e18f, limbig = lim * 1.152921504606846976e18f;   // 2^60
            float ts[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const float du = fmaf(bu_, ms_oy<NS>(k), au * ms_ox<NS>(k)), dv = fmaf(bv_, ms_oy<NS>(k), av * ms_ox<NS>(k));
                float c = ms_inside(u + du, big, limbig) * ms_inside(v + dv, big, limbig) * t.r[k];
                if constexpr (DEPTH) c = z >= dpx[k] ? c : 0.0f;
                ts[k] = c;
            }
            float sum;
            if constexpr (NS == 2) sum = ts[0] + ts[1];
            else sum = ((ts[0] + ts[1]) + (ts[2] + ts[3])) + ((ts[4] + ts[5]) + (ts[6] + ts[7]));
            const float aq = (1.0f / (float)NS) * alpha;
            w = (t.S * aq) * sum;
#pragma unroll
            for (int k = 0; k < NS; ++k) ms_fnma_in_place(t.r[k], alpha, ts[k]);
            ms_fnma_in_place(t.rb, aq, sum);
        }
        crg = __builtin_elementwise_fma((v2f){w, w}, (v2f){r, g}, crg);
        cb = fmaf(w, b, cb);
    }
}
// the per-pixel transmittance state of a rasteriser instantiation, and one blend of it
template <int MSAA> struct TransOf { typedef PxMsN<MSAA> type; };
template <> struct TransOf<1> { typedef float type; };
template <> struct TransOf<4> { typedef PxMs type; };
template <int MSAA> __device__ __forceinline__ typename TransOf<MSAA>::type trans_init(const bool inside) {
    if constexpr (MSAA == 1) return inside ? 1.0f : 0.0f;
    else if constexpr (MSAA == 4) return PxMs{inside ? 1.0f : 0.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
    else {
        PxMsN<MSAA> t;
        t.S = inside ? 1.0f : 0.0f; t.rb = 1.0f;
#pragma unroll
        for (int k = 0; k < MSAA; ++k) t.r[k] = 1.0f;
        return t;
    }
}
__device__ __forceinline__ float trans_mean(const float T) { return T; }
__device__ __forceinline__ float trans_mean(const PxMs& T) { return T.S * T.rb; }
template <int NS> __device__ __forceinline__ float trans_mean(const PxMsN<NS>& T) { return T.S * T.rb; }

// Rgba8UnormSrgb packing of one premultiplied linear pixel (shared by the rasteriser's fused output and
// encode_srgb8_kernel, so both give the same bytes)
__device__ __forceinline__ uint32_t unorm8(float x) {
    x = fminf(fmaxf(x, 0.0f), 1.0f);  // NaN -> 0
    return (uint32_t)(x * 255.0f + 0.5f);
}
__device__ __forceinline__ float srgb_oetf(float x) {
    x = fminf(fmaxf(x, 0.0f), 1.0f);
    return x <= 0.0031308f ? 12.92f * x : fmaf(1.055f, __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 2.4f)), -0.055f);
}
// ---------------------------------------------------------------------------------------
// INTERIOR records (round 6). What the record loop of a tile wave costs is its instruction COUNT, scalar ones included:
// a SIMD retires one (record, tile) pair per ~296 clocks whether three or five tile waves share it (per-tile trace of the
// dense frame), 16 more scalar moves per record cost 7 us of the launch and 16 more vector moves 5 us
// (profiles/r6_notes.md) — and the compiled loop is ~87 vector + ~76 scalar instructions per pair: two exec-mask regions,
// a ballot and three branches per 16 x 4 strip around 16 vector instructions. Most of that decides things that are the
// same for all 256 pixels when the quad covers the tile with room to spare, which is what nearly every blended pair of a
// frame with large splats is: the lane that stages a record decides once per (record, tile) whether every SAMPLE of every
// pixel of the tile lies inside the quad — |u| + 7.5 (|m00| + |m01|) + margin <= 0.9999 lim on both axes, `margin` the one
// blend_px_ms compares with (the slack covers the rounding of the per-pixel fmas by three orders of magnitude) — and the
// verdicts of a staging round travel as two wave-wide bit masks in scalar registers (kept / interior). For an interior
// record every pixel passes blend_px_ms's `gmax <= lim - m`, every strip takes its fast update, and all that is left to
// decide per pixel is `T >= t_eps`: blend_interior_ms below is that update for two strips, written out — v_cmpx narrows
// exec (no scalar mask juggling), 13 vector + 3 scalar instructions per strip where the compiled path has 16 + ~19, the
// same operations in the same order on every active pixel, so the image keeps every bit (both rasterisers, every
// instantiation: test_binning_modes_give_bit_identical_images). Records that are not interior go through blend_px_ms as
// before; the two kinds run in separate inner loops (runs of a kind), so that neither loop's register assignment has to
// agree with the other's inside the loop (one loop with two bodies made the compiler copy the whole pixel state, 35
// moves, at every join).
// Hazards the assembler does not see inside an asm block (gfx940-class): one wait state between a transcendental and
// the use of its result (s_nop 0).
// ---------------------------------------------------------------------------------------
#ifndef BGS_INTERIOR_PATH
#define BGS_INTERIOR_PATH 1
#endif
#define BGS_IA_STRIP_MS(S, RB, CX, CY, CB, QY, L)                                  \
    "v_mul_f32 %[tm], " S ", " RB "\n\t"                                           \
    "v_cmpx_le_f32 vcc, %[te], %[tm]\n\t"                                          \
    "v_fma_f32 %[u], %[m01], " QY ", %[ux]\n\t"                                    \
    "v_fma_f32 %[v], %[m11], " QY ", %[vx]\n\t"                                    \
    "s_cbranch_execz " L "\n\t"                                                    \
    "v_mul_f32 %[v], %[v], %[v]\n\t"                                               \
    "v_fmac_f32 %[v], %[u], %[u]\n\t"                                              \
    "v_exp_f32_e64 %[v], -%[v]\n\t"                                                \
    "s_nop 0\n\t"                                                                  \
    "v_mul_f32 %[v], %[v], %[al]\n\t"                                              \
    "v_min_f32 %[v], 0x3f7fbe77, %[v]\n\t"                                         \
    "v_mul_f32 %[u], %[tm], %[v]\n\t"                                              \
    "v_fma_f32 " S ", -%[v], " S ", " S "\n\t"                                     \
    "v_fma_f32 " CX ", %[u], %[cr], " CX "\n\t"                                    \
    "v_fma_f32 " CY ", %[u], %[cg], " CY "\n\t"                                    \
    "v_fma_f32 " CB ", %[u], %[cb], " CB "\n"                                      \
    L ":\n\t"                                                                      \
    "s_mov_b64 exec, %[sv]\n\t"
// two strips (pixel rows) of a tile wave, one interior record; 4 samples per pixel: T_s = S r[s], only S moves
template <typename PX>   // PxMs (4 samples) or PxMsN<2 / 8>: T_s = S r[s] either way, an interior record moves S only
__device__ __forceinline__ void blend_interior_ms(const float ux, const float vx, const float m01, const float m11,
                                                  const float al, const float cr, const float cg, const float cbl,
                                                  const float t_eps, const float qy0, const float qy1,
                                                  PX& t0, v2f& c0, float& b0, PX& t1, v2f& c1, float& b1) {
    float tm, u, v;
    unsigned long long sv;
    float c0x = c0.x, c0y = c0.y, c1x = c1.x, c1y = c1.y;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 BGS_IA_STRIP_MS("%[s0]", "%[r0]", "%[c0x]", "%[c0y]", "%[b0]", "%[q0]", "BGS_IA0_%=")
                 BGS_IA_STRIP_MS("%[s1]", "%[r1]", "%[c1x]", "%[c1y]", "%[b1]", "%[q1]", "BGS_IA1_%=")
                 : [s0] "+v"(t0.S), [c0x] "+v"(c0x), [c0y] "+v"(c0y), [b0] "+v"(b0),
                   [s1] "+v"(t1.S), [c1x] "+v"(c1x), [c1y] "+v"(c1y), [b1] "+v"(b1),
                   [tm] "=&v"(tm), [u] "=&v"(u), [v] "=&v"(v), [sv] "=&s"(sv)
                 : [r0] "v"(t0.rb), [r1] "v"(t1.rb), [q0] "v"(qy0), [q1] "v"(qy1), [ux] "v"(ux), [vx] "v"(vx),
                   [m01] "v"(m01), [m11] "v"(m11), [al] "v"(al), [cr] "v"(cr), [cg] "v"(cg), [cb] "v"(cbl), [te] "s"(t_eps)
                 : "vcc");
    c0.x = c0x; c0.y = c0y; c1.x = c1x; c1.y = c1y;
}
// the single-sampled twin (blend_px with hit == true): T is the pixel's transmittance
#define BGS_IA_STRIP_1(T, CX, CY, CB, QY, L)                                       \
    "v_cmpx_le_f32 vcc, %[te], " T "\n\t"                                          \
    "v_fma_f32 %[u], %[m01], " QY ", %[ux]\n\t"                                    \
    "v_fma_f32 %[v], %[m11], " QY ", %[vx]\n\t"                                    \
    "s_cbranch_execz " L "\n\t"                                                    \
    "v_mul_f32 %[v], %[v], %[v]\n\t"                                               \
    "v_fmac_f32 %[v], %[u], %[u]\n\t"                                              \
    "v_exp_f32_e64 %[v], -%[v]\n\t"                                                \
    "s_nop 0\n\t"                                                                  \
    "v_mul_f32 %[v], %[v], %[al]\n\t"                                              \
    "v_min_f32 %[v], 0x3f7fbe77, %[v]\n\t"                                         \
    "v_mul_f32 %[u], " T ", %[v]\n\t"                                              \
    "v_fma_f32 " CX ", %[u], %[cr], " CX "\n\t"                                    \
    "v_fma_f32 " CY ", %[u], %[cg], " CY "\n\t"                                    \
    "v_fma_f32 " CB ", %[u], %[cb], " CB "\n\t"                                    \
    "v_sub_f32 " T ", " T ", %[u]\n"                                               \
    L ":\n\t"                                                                      \
    "s_mov_b64 exec, %[sv]\n\t"
__device__ __forceinline__ void blend_interior_1(const float ux, const float vx, const float m01, const float m11,
                                                 const float al, const float cr, const float cg, const float cbl,
                                                 const float t_eps, const float qy0, const float qy1,
                                                 float& t0, v2f& c0, float& b0, float& t1, v2f& c1, float& b1) {
    float u, v;
    unsigned long long sv;
    float c0x = c0.x, c0y = c0.y, c1x = c1.x, c1y = c1.y;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 BGS_IA_STRIP_1("%[s0]", "%[c0x]", "%[c0y]", "%[b0]", "%[q0]", "BGS_IB0_%=")
                 BGS_IA_STRIP_1("%[s1]", "%[c1x]", "%[c1y]", "%[b1]", "%[q1]", "BGS_IB1_%=")
                 : [s0] "+v"(t0), [c0x] "+v"(c0x), [c0y] "+v"(c0y), [b0] "+v"(b0),
                   [s1] "+v"(t1), [c1x] "+v"(c1x), [c1y] "+v"(c1y), [b1] "+v"(b1),
                   [u] "=&v"(u), [v] "=&v"(v), [sv] "=&s"(sv)
                 : [q0] "v"(qy0), [q1] "v"(qy1), [ux] "v"(ux), [vx] "v"(vx),
                   [m01] "v"(m01), [m11] "v"(m11), [al] "v"(al), [cr] "v"(cr), [cg] "v"(cg), [cb] "v"(cbl), [te] "s"(t_eps)
                 : "vcc");
    c0.x = c0x; c0.y = c0y; c1.x = c1x; c1.y = c1y;
}

// The 2DGS surfel twin (RV_SURFEL: the ray-splat intersection, gaussian_2d.wgsl:134-156 as staged by stage_surfel): ONE strip
// per block (the operands of two do not fit an asm statement). pxq / pyq / pzq = the parts of p that depend on the pixel's
// column only (formed once per record by the caller), ddx likewise. Same operations in the same order as blend_px_ms /
// blend_px's surfel branch on a pixel whose samples all lie inside the quad: u, v are not even formed.
#define BGS_IA_SURFEL_BODY(TM)                                                     \
    "v_fma_f32 %[a], %[pyx], %[qy], %[pxq]\n\t"                                    \
    "v_fma_f32 %[b], %[pyy], %[qy], %[pyq]\n\t"                                    \
    "v_fma_f32 %[c], %[pyz], %[qy], %[pzq]\n\t"                                    \
    "s_cbranch_execz BGS_IS_%=\n\t"                                                \
    "v_rcp_f32 %[c], %[c]\n\t"                                                     \
    "s_nop 0\n\t"                                                                  \
    "v_mul_f32 %[a], %[a], %[c]\n\t"                                               \
    "v_mul_f32 %[b], %[b], %[c]\n\t"                                               \
    "v_fma_f32 %[c], %[ddyk], %[qy], %[dy0]\n\t"                                   \
    "v_mul_f32 %[b], %[b], %[b]\n\t"                                               \
    "v_fmac_f32 %[b], %[a], %[a]\n\t"                                              \
    "v_mul_f32 %[c], %[c], %[c]\n\t"                                               \
    "v_fmac_f32 %[c], %[ddx], %[ddx]\n\t"                                          \
    "v_min_f32 %[b], %[b], %[c]\n\t"                                               \
    "v_exp_f32_e64 %[b], -%[b]\n\t"                                                \
    "s_nop 0\n\t"                                                                  \
    "v_mul_f32 %[b], %[b], %[al]\n\t"                                              \
    "v_min_f32 %[b], 0x3f7fbe77, %[b]\n\t"                                         \
    "v_mul_f32 %[a], " TM ", %[b]\n\t"
template <typename PX>
__device__ __forceinline__ void blend_interior_surfel_ms(const float pxq, const float pyq, const float pzq, const float ddx,
                                                         const float pyx, const float pyy, const float pyz, const float ddyk,
                                                         const float dy0, const float al, const float cr, const float cg,
                                                         const float cbl, const float t_eps, const float qy,
                                                         PX& t, v2f& c, float& bl) {
    float tm, a, b, cc;
    unsigned long long sv;
    float cx = c.x, cy = c.y;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_mul_f32 %[tm], %[s], %[rb]\n\t"
                 "v_cmpx_le_f32 vcc, %[te], %[tm]\n\t"
                 BGS_IA_SURFEL_BODY("%[tm]")
                 "v_fma_f32 %[s], -%[b], %[s], %[s]\n\t"
                 "v_fma_f32 %[cx], %[a], %[cr], %[cx]\n\t"
                 "v_fma_f32 %[cy], %[a], %[cg], %[cy]\n\t"
                 "v_fma_f32 %[bl], %[a], %[cb], %[bl]\n"
                 "BGS_IS_%=:\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 : [s] "+v"(t.S), [cx] "+v"(cx), [cy] "+v"(cy), [bl] "+v"(bl),
                   [tm] "=&v"(tm), [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(cc), [sv] "=&s"(sv)
                 : [rb] "v"(t.rb), [qy] "v"(qy), [pxq] "v"(pxq), [pyq] "v"(pyq), [pzq] "v"(pzq), [ddx] "v"(ddx),
                   [pyx] "v"(pyx), [pyy] "v"(pyy), [pyz] "v"(pyz), [ddyk] "v"(ddyk), [dy0] "v"(dy0), [al] "v"(al),
                   [cr] "v"(cr), [cg] "v"(cg), [cb] "v"(cbl), [te] "s"(t_eps)
                 : "vcc");
    c.x = cx; c.y = cy;
}
__device__ __forceinline__ void blend_interior_surfel_1(const float pxq, const float pyq, const float pzq, const float ddx,
                                                        const float pyx, const float pyy, const float pyz, const float ddyk,
                                                        const float dy0, const float al, const float cr, const float cg,
                                                        const float cbl, const float t_eps, const float qy,
                                                        float& t, v2f& c, float& bl) {
    float a, b, cc;
    unsigned long long sv;
    float cx = c.x, cy = c.y;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_le_f32 vcc, %[te], %[s]\n\t"
                 BGS_IA_SURFEL_BODY("%[s]")
                 "v_fma_f32 %[cx], %[a], %[cr], %[cx]\n\t"
                 "v_fma_f32 %[cy], %[a], %[cg], %[cy]\n\t"
                 "v_fma_f32 %[bl], %[a], %[cb], %[bl]\n\t"
                 "v_sub_f32 %[s], %[s], %[a]\n"
                 "BGS_IS_%=:\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 : [s] "+v"(t), [cx] "+v"(cx), [cy] "+v"(cy), [bl] "+v"(bl),
                   [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(cc), [sv] "=&s"(sv)
                 : [qy] "v"(qy), [pxq] "v"(pxq), [pyq] "v"(pyq), [pzq] "v"(pzq), [ddx] "v"(ddx),
                   [pyx] "v"(pyx), [pyy] "v"(pyy), [pyz] "v"(pyz), [ddyk] "v"(ddyk), [dy0] "v"(dy0), [al] "v"(al),
                   [cr] "v"(cr), [cg] "v"(cg), [cb] "v"(cbl), [te] "s"(t_eps)
                 : "vcc");
    c.x = cx; c.y = cy;
}

// Rgba16Float texel: IEEE binary16, round to nearest even, overflow to inf (what a float16 target stores)
__device__ __forceinline__ uint2 pack_rgba16f(const float4 c) {
    const _Float16 h[4] = {(_Float16)c.x, (_Float16)c.y, (_Float16)c.z, (_Float16)c.w};
    uint2 out;
    __builtin_memcpy(&out, h, 8);
    return out;
}
__device__ __forceinline__ uint32_t pack_srgb8(const float4 c) {
    return unorm8(srgb_oetf(c.x)) | (unorm8(srgb_oetf(c.y)) << 8) | (unorm8(srgb_oetf(c.z)) << 16) | (unorm8(c.w) << 24);
}

// XCD-aware work order: workgroup b runs on XCD b % 8 (observed dispatch, a speed assumption
// only); give each XCD a contiguous band of work items so neighbouring tiles, which share records
// and coarse lists, share an L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    const uint32_t q = n / 8u, r = n % 8u, xcd = b % 8u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + b / 8u;
}

// The same with S contiguous runs per XCD, dealt round-robin (run j of 8 S belongs to XCD j % 8): an XCD's tiles
// come from S places of the image instead of one band, so that a frame whose work varies over the image (edges of a
// dense cloud are heavier than its middle) loads the XCDs alike. Work item b / 8 of XCD b % 8 is found by walking the
// XCD's runs (scalar integer work, <= 8 S steps, once per workgroup). A bijection of [0, n) for every n
// (splat_math.h: xcd_runs_item, checked on the host by tests/test_device_math_host.py).
__device__ __forceinline__ uint32_t xcd_remap_runs(const uint32_t b, const uint32_t n, const uint32_t S) {
    return xcd_runs_item(b, n, S);
}

// Staging-time extras of a record for the tile that blends it (both rasterisers go through these, so that their
// staged records — and with them their images — stay bit-identical):
//   OBB, multisampled: the margin of blend_px_ms in the spare dword of the second vector;
//   keepz: the quad's depth, or -1 for a record the tile does not need to blend at all (the exact quad-vs-tile test
//   failed, a surfel that is negligible in this tile, a quad behind everything the tile's depth buffer holds).
//   A kept record's depth is > 0 (in_frustum), so the sign bit is the flag.
template <int MSAA>
__device__ __forceinline__ void stage_obb_margin(const float4& r0, float4& r1) {
    if constexpr (MSAA == 4) r1.z = fmaxf(ms_margin(r0.z, r0.w), ms_margin(r1.x, r1.y));
    else if constexpr (MSAA == 1) r1.z = 0.0f;
    else r1.z = fmaxf(ms_margin_n<MSAA>(r0.z, r0.w), ms_margin_n<MSAA>(r1.x, r1.y));
}
__device__ __forceinline__ float keepz_of(const bool keep, const float z) { return keep ? z : -1.0f; }
__device__ __forceinline__ bool keepz_keeps(const float keepz) { return (int32_t)__float_as_uint(keepz) >= 0; }

// min / max over a wave (all 64 lanes take part)
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// BINNING_SORT rasteriser: one workgroup per tile, one pixel per thread; the tile's instances are
// a contiguous range of the tile-sorted list, staged 256 records at a time in LDS.
// MSAA: samples per pixel (1 or 4, blend_px_ms); DEPTH: test against the view's depth buffer (fp.depth_ptr).
template <int VARIANT, int MSAA, bool DEPTH, bool BBOX>
__global__ __launch_bounds__(256) void raster_kernel(FrameParams fp, const float4* __restrict__ records,
                                                     const uint2* __restrict__ instances,
                                                     const uint2* __restrict__ ranges,
                                                     float4* __restrict__ fb, float4 clear,
                                                     const Control* __restrict__ ctl) {
    constexpr int REC_V4 = VARIANT == RV_SURFEL ? 6 : 3;
    __shared__ float4 s_rec[256 * REC_V4];
    __shared__ float s_zr[4][2];

    const uint32_t tile = xcd_remap(blockIdx.x, (uint32_t)(fp.tiles_x * fp.tiles_y));
    const uint32_t ty = tile / (uint32_t)fp.tiles_x, tx = tile - ty * (uint32_t)fp.tiles_x;
    const int tid = threadIdx.x;
    const int px = (int)tx * TILE_PX + (tid & 15), py = (int)ty * TILE_PX + (tid >> 4);
    const bool in_image = px < fp.width && py < fp.height;
    // OBB and surfel records are staged tile-local (stage_obb / stage_surfel): the pixel is addressed inside its tile
    const float qx = VARIANT != RV_AABB3D ? (float)(tid & 15) : (float)px + 0.5f;
    const float qy = VARIANT != RV_AABB3D ? (float)(tid >> 4) : (float)py + 0.5f;
    const float tile_ox = (float)((int)tx * TILE_PX) + 0.5f, tile_oy = (float)((int)ty * TILE_PX) + 0.5f;
    const float aspect = fp.viewport_w / fp.viewport_h;

    const uint2 range = ranges[(ty << 8) | tx];
    const float t_eps = frame_t_eps(ctl->color_max_bits);
    const float surfel_limit = frame_surfel_limit(ctl->color_max_bits);
    float cb = 0.0f;
    typename TransOf<MSAA>::type tm = trans_init<MSAA>(in_image);
    v2f crg = {0.0f, 0.0f};
    // the pixel's scene depth(s) and the tile's range of them (what a record's constant depth is compared with first)
    float4 dpx = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    [[maybe_unused]] float dpn[MSAA];   // Sample2 / Sample8
#pragma unroll
    for (int k = 0; k < MSAA; ++k) dpn[k] = 0.0f;
    float tile_dmin = 0.0f, tile_dmax = 0.0f;
    if constexpr (DEPTH) {
        float lo = INFINITY, hi = -INFINITY;
        if (in_image) {
            const float* dsrc = reinterpret_cast<const float*>(fp.depth_ptr) + ((size_t)py * (size_t)fp.width + (size_t)px) * MSAA;
            if constexpr (MSAA == 4) dpx = *reinterpret_cast<const float4*>(dsrc);
            else if constexpr (MSAA == 1) dpx.x = dpx.y = dpx.z = dpx.w = dsrc[0];
            if constexpr (MSAA == 1 || MSAA == 4) {
                lo = fminf(fminf(dpx.x, dpx.y), fminf(dpx.z, dpx.w));
                hi = fmaxf(fmaxf(dpx.x, dpx.y), fmaxf(dpx.z, dpx.w));
            } else {
#pragma unroll
                for (int k = 0; k < MSAA; ++k) { dpn[k] = dsrc[k]; lo = fminf(lo, dpn[k]); hi = fmaxf(hi, dpn[k]); }
            }
        }
        lo = wave_min(lo); hi = wave_max(hi);
        if ((tid & 63) == 0) { s_zr[tid >> 6][0] = lo; s_zr[tid >> 6][1] = hi; }
        __syncthreads();
        tile_dmin = fminf(fminf(s_zr[0][0], s_zr[1][0]), fminf(s_zr[2][0], s_zr[3][0]));
        tile_dmax = fmaxf(fmaxf(s_zr[0][1], s_zr[1][1]), fmaxf(s_zr[2][1], s_zr[3][1]));
    }
    auto saturated = [&]() { return trans_mean(tm) < t_eps; };

    for (uint32_t base = range.x; base < range.y; base += 256u) {
        const uint32_t cnt = min(256u, range.y - base);
        if ((uint32_t)tid < cnt) {
            const uint32_t rank = instances[base + (uint32_t)tid].y;
            const float4* src = records + (size_t)rank * REC_V4;
            if constexpr (VARIANT == RV_OBB) {
                float4 r0 = src[0], r1 = src[1], r2 = src[2];
                stage_obb(r0, r1, tile_ox, tile_oy);
                stage_obb_margin<MSAA>(r0, r1);
                r2.w = keepz_of(!DEPTH || r2.w >= tile_dmin, r2.w);
                s_rec[tid * REC_V4 + 0] = r0;
                s_rec[tid * REC_V4 + 1] = r1;
                s_rec[tid * REC_V4 + 2] = r2;
            } else if constexpr (VARIANT == RV_SURFEL) {
                float4 st[6];
                stage_surfel(src, tile_ox, tile_oy, aspect, st);
                const float z = src[5].x;
                // (the bounding-box overlay draws a quad's frame whatever its Gaussian is worth there: nothing is negligible)
                st[5].y = keepz_of((BBOX || !surfel_negligible_in_tile(st, surfel_limit)) && (!DEPTH || z >= tile_dmin), z);  // as raster_scan_kernel decides
#pragma unroll
                for (int v = 0; v < 6; ++v) s_rec[tid * REC_V4 + v] = st[v];
            } else {
                float4 r2 = src[2];
                r2.w = keepz_of(!DEPTH || r2.w >= tile_dmin, r2.w);
                s_rec[tid * REC_V4 + 0] = src[0];
                s_rec[tid * REC_V4 + 1] = src[1];
                s_rec[tid * REC_V4 + 2] = r2;
            }
        }
        __syncthreads();
        if (!__all(saturated()))
            for (uint32_t k = 0; k < cnt; ++k) {
                StagedRecord<VARIANT> sr;
                sr.load(s_rec + k * REC_V4);
                const float keepz = VARIANT == RV_SURFEL ? sr.a5.y : sr.a2.w;
                const float zr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(keepz)));
                if (!keepz_keeps(zr)) continue;
                if constexpr (MSAA == 4) blend_px_ms<VARIANT, DEPTH, BBOX>(sr, qx, qy, aspect, t_eps, tm, crg, cb, zr, DEPTH && zr < tile_dmax, dpx);
                else if constexpr (MSAA == 1) blend_px<VARIANT, DEPTH, BBOX>(sr, qx, qy, aspect, t_eps, tm, crg, cb, zr, dpx.x);
                else blend_px_msn<VARIANT, DEPTH, MSAA, BBOX>(sr, qx, qy, aspect, t_eps, tm, crg, cb, zr, DEPTH && zr < tile_dmax, dpn);
            }
        // also the barrier that protects s_rec before the next batch overwrites it
        if (__syncthreads_and(saturated() ? 1 : 0)) break;
    }
    if (in_image) {
        // dst = src + dst*(1-src.a) unrolled over the whole list, target cleared to `clear`; multisampled: the mean
        // of the samples, C + clear * mean_s(T_s)
        const float Tf = trans_mean(tm);
        fb[(size_t)py * (size_t)fp.width + (size_t)px] =
            make_float4(fmaf(Tf, clear.x, crg.x), fmaf(Tf, clear.y, crg.y), fmaf(Tf, clear.z, cb),
                        fmaf(Tf, clear.w, 1.0f - Tf));
    }
}

// BINNING_SCAN rasteriser: ONE WAVE per 16x16 tile (4 tiles per workgroup, no workgroup barriers),
// each lane owns 4 pixels (column lane&15, rows (lane>>4) + {0,4,8,12}), so one LDS broadcast read
// of a record feeds 256 pixel evaluations of a single wave. The wave bins lazily while it
// composites: per group of 64 candidates from its supertile's depth-ordered list it keeps those
// whose tile rectangle contains this tile (order-preserving ballot compaction), stages their
// records in its private LDS slice, blends, and stops at saturation. The candidate stream is
// prefetched two groups (ranks) / one group (rectangles) ahead of the blend.
// __launch_bounds__(256, 8): 8 waves/SIMD (<= 64 VGPRs). A 1080p frame is 8160 one-wave tiles for
// 1024 SIMDs x 8 slots, so at 7 waves/SIMD a second, nearly empty round of waves appears.
// The surfel variant carries a 24-dword staged record through a 30-slot blend: it spills at 64 registers
// (14 VGPRs) and at 72 / 80; same-box A/B of the dense 1 M-surfel frame: 5 waves/SIMD (90 VGPRs, no spill)
// 0.49 ms, 6: 0.54, 7: 0.55, 8: 0.51 (and 0.13 instead of 0.09 ms scene-like). Round 3: the kernel has since come
// down to 84 registers and fits 80 without a spill: 6 waves/SIMD (6 x 25.6 KB of LDS per CU) is +2 % frames/s with
// frames in flight on both the dense and the scene-like surfel frame and +-2 % on a frame alone on the chip.
// TRACE (bgs_set_tile_trace, diagnostics only): every tile's wave also writes two uint4 to trace[2 * tile]:
//   { s_memtime at wave start (lo, hi), s_memtime at wave end (lo, hi) },
//   { HW_ID register, XCC_ID register, candidates scanned, records blended | records staged << 16 }
// — where a tile ran (XCD / SE / CU / SIMD / wave slot), for how long, and on how much work: what the analysis of
// the launch's tail (scripts/tile_trace.py) is made from. The production instantiation carries none of it.
// MIDROUND_EXIT: look for saturation inside a staging round as well (every 4th record), not only at its end. A tile
// of a DENSE frame saturates somewhere inside a round of up to 64 records, and the records behind that point each still
// pay the four strips' reject path: leaving early changes no bit (saturated pixels never accumulate) and saves 7 % of
// the rasteriser's time on the dense 1 M frame, 18 % on the dense 5 M frame. On frames whose tiles rarely saturate
// (small splats; surfels, which need ~300 records per tile) the four compares, the ballot and the extra loop exit
// only cost: +18 % / +21 % / +14 % on the scene-like 1 M / 5 M and the dense surfel frame. So it is an instantiation
// the launcher picks per frame (supertile level >= 2, i.e. splats larger than a supertile, and not the surfel variant).
#ifndef BGS_MIDROUND_PERIOD
#define BGS_MIDROUND_PERIOD 4u
#endif
// What a tile cost its wave, in eighths of a blended record (kernels.h TileCost): a record that went through the four
// rows' blend, a record staged (and possibly skipped), a staging round (one dependent gather nothing hides, ~5 records'
// worth), a group of 64 candidates scanned. Scalar additions on wave-uniform counts; the same frame gives the same
// numbers whatever else runs on the chip, which a wave's lifetime does not.
constexpr uint32_t WORK_BLENDED = 8u, WORK_STAGED = 1u, WORK_ROUND = 40u, WORK_GROUP = 3u;
// -DBGS_PHASE_TRACE=1 (experiment builds, scripts/tile_phases.py): the traced instantiations also stamp where a tile
// wave's life goes — s_memtime at: entering raster_tile, first candidates tested, first staging round staged, last record
// blended — into a third uint4 per tile (trace[2 * ntiles + tile], ticks since the wave's start)
#ifndef BGS_PHASE_TRACE
#define BGS_PHASE_TRACE 0
#endif
#if BGS_PHASE_TRACE
#define BGS_PHASE(i) do { if constexpr (TRACE) { if (phase[i] == 0u) phase[i] = (uint32_t)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define BGS_PHASE(i) do { } while (0)
#endif
#ifndef BGS_DENSE_RUNS_MS
#define BGS_DENSE_RUNS_MS 2u
#endif
#ifndef BGS_DENSE_RUNS
#define BGS_DENSE_RUNS 1u
#endif
// contiguous runs of workgroups per XCD (raster_scan_kernel: RUNS)
constexpr uint32_t raster_runs(const int variant, const int msaa, const bool midround_exit) {
    return variant == RV_SURFEL ? 4u : (midround_exit ? (msaa == 4 ? BGS_DENSE_RUNS_MS : BGS_DENSE_RUNS) : 1u);
}
// Which group of four tiles workgroup b of n draws when nothing is known about the frame (RUNS: see raster_scan_kernel)
template <uint32_t RUNS>
__device__ __forceinline__ uint32_t raster_block_item(const uint32_t b, const uint32_t n) {
    return RUNS == 1u ? xcd_remap(b, n) : xcd_remap_runs(b, n, RUNS);
}
// One tile — or, with ROWS == 1, one 16 x 4 ROW STRIP of a tile (rows row0 .. row0 + 3, one pixel per lane) — by one wave.
// A wave that owns a whole tile has four pixels per lane (rows row0 + (lane >> 4) + {0, 4, 8, 12}, row0 = 0) and runs
// the four strips' chains one after the other, ~950 clocks per record whether it shares its SIMD or not; a strip wave
// runs one chain per record. The arithmetic per pixel is the same (records are staged relative to the TILE's first
// pixel either way), so which of the two shapes draws a pixel changes no bit. Returns the staging rounds it blended.
template <int ROWS>
__device__ __forceinline__ bool all_saturated(const float (&T)[ROWS], const float t_eps) {
    bool s = true;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) s = s && T[r] < t_eps;
    return s;
}
template <int ROWS>
__device__ __forceinline__ bool all_saturated(const PxMs (&T)[ROWS], const float t_eps) {
    bool s = true;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) s = s && T[r].S * T[r].rb < t_eps;
    return s;
}
// MSAA: samples per pixel, 1 or 4 (blend_px_ms). DEPTH: the quads are tested against the view's depth buffer
// (fp.depth_ptr: [y][x][sample] floats, GreaterEqual, no write; src/render/mod.rs:959-974) — the tile's depths are read
// once (single-sampled: a register per pixel; 4x: 4 KB of the wave's LDS, s_depth), a record is compared with the
// tile's [min, max] first: behind everything -> dropped at staging, in front of everything -> the plain path, in between
// -> the per-sample path.
template <int ROWS, int NS>
__device__ __forceinline__ bool all_saturated(const PxMsN<NS> (&T)[ROWS], const float t_eps) {
    bool s = true;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) s = s && T[r].S * T[r].rb < t_eps;
    return s;
}
template <int VARIANT, bool TRACE, int MODE, int ROWS, int MSAA, bool DEPTH, bool BBOX>   // MODE: raster_scan_kernel
__device__ __forceinline__ uint32_t raster_tile(const FrameParams& fp, const float4* __restrict__ records,
                                               const uint32_t* __restrict__ coarse, const uint32_t coarse_cap,
                                               const uint32_t sup_mul, const uint32_t sup_x, Control* ctl,
                                               float4* __restrict__ fb, uint32_t* __restrict__ fb8_default,
                                               const uint32_t want_srgb8, const float t_eps, const float surfel_limit,
                                               float4* const s_rec, uint32_t* const s_queue, float4* const s_depth, const int lane,
                                               const uint32_t tile_v, const int row0, uint32_t& trace_scanned,
                                               uint32_t& trace_blended, uint32_t& trace_staged, uint32_t& work,
                                               [[maybe_unused]] uint32_t (&phase)[4]) {
    BGS_PHASE(0);
    constexpr int REC_V4 = VARIANT == RV_SURFEL ? 6 : 3;   // float4 per record in `records`
    constexpr bool MIDROUND_EXIT = MODE != 0, DENSE = MODE == 1;
    constexpr int ST_V4 = staged_v4(VARIANT, MSAA, DEPTH, DENSE);                   // ... and per staged record in s_rec
    constexpr uint32_t STAGE = 64u;
    constexpr bool ABLATE = BGS_ABLATION != 0;
    // the interior-record loop (blend_interior_ms): whole tiles of OBB quads without overlay or depth buffer, 1 or 4 samples
    // (every sample count; under a depth buffer too: a record the tile's depths leave whole — in front of everything the tile
    // holds — is interior like any other, one they split goes through the per-sample path with the others)
    constexpr bool FAST = BGS_INTERIOR_PATH != 0 && !ABLATE && (VARIANT == RV_OBB || VARIANT == RV_SURFEL) && !BBOX && ROWS == 4;
    // ... and their strips' reach masks (below) where frames are not dense: on a dense frame nearly every strip is reached and the
    // three scalar instructions per strip only cost (dense 1 M -6 % frames/s; scene-like +1.4 %, 5 M scene-like +2 %, dense
    // surfels +4 %: profiles/r6_experiments/strip_reach_ab.txt)
    constexpr bool STRIPS = FAST && !DENSE;
    // half extent of the box the tile's sample positions span around the tile centre (the exact quad-vs-tile test)
    constexpr float HALF = 7.5f + ms_reach(MSAA);

    // (the tile is the wave's: scalar registers for everything derived from it — the tile origin the staging folds into
    // every record used to sit in VGPRs across the blend loop; 92 -> 86 VGPRs for the headline's instantiation)
    const uint32_t tile = __builtin_amdgcn_readfirstlane(tile_v);
    const uint32_t ty = tile / (uint32_t)fp.tiles_x, tx = tile - ty * (uint32_t)fp.tiles_x;
    const int px = (int)tx * TILE_PX + (lane & 15), py0 = (int)ty * TILE_PX + row0 + (lane >> 4);
    // OBB and surfel records are staged tile-local (stage_obb / stage_surfel): pixels are then addressed inside the tile
    const float qx = VARIANT != RV_AABB3D ? (float)(lane & 15) : (float)px + 0.5f;
    const float aspect = fp.viewport_w / fp.viewport_h;
    const float tile_cx = (float)((int)tx * TILE_PX + 8), tile_cy = (float)((int)ty * TILE_PX + 8);
    const float tile_ox = (float)((int)tx * TILE_PX) + 0.5f, tile_oy = (float)((int)ty * TILE_PX) + 0.5f;

    // per-pixel transmittance: one number (single-sampled target) or PxMs (4x)
    typedef typename TransOf<MSAA>::type Trans;
    Trans T[ROWS];
    float* const s_depth_f = reinterpret_cast<float*>(s_depth);   // Sample2 / Sample8: MSAA floats per pixel, pixel = r * 64 + lane
    float cb[ROWS], qy[ROWS], dpx[ROWS];
    v2f crg[ROWS];
    float tile_dmin = 0.0f, tile_dmax = 0.0f;
    {
        float lo = INFINITY, hi = -INFINITY;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int py = py0 + 4 * r;
            qy[r] = VARIANT != RV_AABB3D ? (float)(row0 + (lane >> 4) + 4 * r) : (float)py + 0.5f;
            const bool inside = px < fp.width && py < fp.height;
            T[r] = trans_init<MSAA>(inside);
            crg[r] = (v2f){0.0f, 0.0f};
            cb[r] = 0.0f;
            dpx[r] = 0.0f;
            if constexpr (DEPTH && (MSAA == 1 || MSAA == 4)) {
                float4 d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (inside) {
                    const float* dsrc = reinterpret_cast<const float*>(fp.depth_ptr) + ((size_t)py * (size_t)fp.width + (size_t)px) * MSAA;
                    if constexpr (MSAA == 4) d = *reinterpret_cast<const float4*>(dsrc);
                    else d.x = d.y = d.z = d.w = dsrc[0];
                    lo = fminf(lo, fminf(fminf(d.x, d.y), fminf(d.z, d.w)));
                    hi = fmaxf(hi, fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w)));
                }
                if constexpr (MSAA == 4) s_depth[r * 64 + lane] = d;
                else dpx[r] = d.x;
            } else if constexpr (DEPTH) {
                const float* dsrc = reinterpret_cast<const float*>(fp.depth_ptr) + ((size_t)py * (size_t)fp.width + (size_t)px) * MSAA;
#pragma unroll
                for (int k = 0; k < MSAA; ++k) {
                    const float d = inside ? dsrc[k] : 0.0f;
                    if (inside) { lo = fminf(lo, d); hi = fmaxf(hi, d); }
                    s_depth_f[(r * 64 + lane) * MSAA + k] = d;
                }
            }
        }
        if constexpr (DEPTH) {
            // A ROWS == 1 strip wave takes the range of the WHOLE tile too (three more rows of depths per lane, once per heavy
            // tile): its own strip's range would be a tighter, equally valid pre-test — but which update a record takes
            // (all samples / per sample, which differ in the last bit) follows from it, and a tile must come out the same
            // whichever shape of wave draws it (found in round 6's last exploration: 1 ulp on ~270 pixels of a frame with a
            // depth buffer whose heavy tiles went to strip waves from the second frame on).
            if constexpr (ROWS == 1) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int pyt = (int)ty * TILE_PX + (lane >> 4) + 4 * rr;
                    if (px < fp.width && pyt < fp.height) {
                        const float* dsrc = reinterpret_cast<const float*>(fp.depth_ptr) + ((size_t)pyt * (size_t)fp.width + (size_t)px) * MSAA;
#pragma unroll
                        for (int k = 0; k < MSAA; ++k) { const float d = dsrc[k]; lo = fminf(lo, d); hi = fmaxf(hi, d); }
                    }
                }
            }
            tile_dmin = wave_min(lo);
            tile_dmax = wave_max(hi);
        }
    }

    const uint32_t st = supertile_div(ty, sup_mul) * sup_x + supertile_div(tx, sup_mul);
    // wave-uniform by construction (one tile per wave): say so, so that the scan loop's control flow (list
    // position, queue length, flush decisions, the record loop) is scalar instead of exec-mask loops on VGPRs
    const uint32_t total = __builtin_amdgcn_readfirstlane(min(ctl->coarse_total[st], coarse_cap));
    const uint2* __restrict__ list = reinterpret_cast<const uint2*>(coarse) + (size_t)st * coarse_cap;

    // candidate stream of (rank, tile rectangle) entries, two groups ahead of the scan
    uint32_t rank_cur = 0u, rect_cur = RECT_EMPTY, rank_nxt = 0u, rect_nxt = RECT_EMPTY;
    if ((uint32_t)lane < total) { const uint2 e = list[lane]; rank_cur = e.x; rect_cur = e.y; }
    if ((uint32_t)lane + 64u < total) { const uint2 e = list[lane + 64]; rank_nxt = e.x; rect_nxt = e.y; }

    // Hits are queued across candidate groups and gathered / blended together once FLUSH_AT of them
    // wait (or the next group would not fit, or the list ends): every gather is a dependent global
    // load (~2.5 us under load) that nothing hides, and a sparse list yields only a few hits per
    // group of 64 candidates — 17 gathers per tile on the scene-like workload before, 3 now. A dense
    // list fills the queue with its first group, so nothing changes there.
    constexpr uint32_t FLUSH_AT = 32u;
    bool tile_saturated = false;   // the tile stopped because every pixel was saturated (not because its lists ended)
    uint32_t qn = 0u;  // ranks waiting in s_queue (wave-uniform)
    uint32_t rounds = 0u;  // staging rounds blended so far (a dense frame's heavy tiles: more than one)
    uint32_t base = 0u;
    for (;;) {
        const bool have = base < total;
        // bitwise on purpose: four compares and three scalar ands instead of nested exec-mask regions
        const bool hit = have & (tx >= (rect_cur & 255u)) & (tx <= ((rect_cur >> 8) & 255u)) &
                         (ty >= ((rect_cur >> 16) & 255u)) & (ty <= (rect_cur >> 24));
        const unsigned long long b = __ballot(hit);
        const uint32_t hits = (uint32_t)__popcll(b);
        BGS_PHASE(1);
        const bool fits = qn + hits <= 64u;
        if (have && fits) {  // queue this group's hits and advance the candidate stream
            const uint32_t i2 = base + 128u + (uint32_t)lane;
            uint32_t rank_nn = 0u, rect_nn = RECT_EMPTY;
            if (i2 < total) { const uint2 e = list[i2]; rank_nn = e.x; rect_nn = e.y; }
            // hits of lower lanes: v_mbcnt, no 64-bit lane mask to keep in registers
            if (hit)
                s_queue[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = rank_cur;
            qn += hits;
            if constexpr (TRACE) trace_scanned += min(64u, total - base);
            work += WORK_GROUP;
            rank_cur = rank_nxt; rect_cur = rect_nxt;
            rank_nxt = rank_nn; rect_nxt = rect_nn;
            base += 64u;
        }
        const bool end = base >= total;
        if (qn && (qn >= FLUSH_AT || !fits || end) && !(ABLATE && (fp.debug & 32u))) {  // ablation bit 32: scan only
            const uint32_t cnt = qn;
            qn = 0u;
            bool saturated = false;
            for (uint32_t c0 = 0u; c0 < cnt && !saturated; c0 += STAGE) {
            const uint32_t ccnt = min(STAGE, cnt - c0);
            work += WORK_ROUND + ccnt * WORK_STAGED;
            __builtin_amdgcn_wave_barrier();
            [[maybe_unused]] bool keep_lane = false, interior_lane = false;   // this lane's record: blended at all / INTERIOR (FAST)
            // ... and which of the tile's four 16 x 4 strips it can reach at all (FAST, round 6): a strip none of whose pixels'
            // samples the quad covers costs ~8 vector + 3 scalar instructions to find that out pixel by pixel — on the scene-like
            // and trained-like frames that is one or two of a pair's four strips. The same separating-axis test as `keep`, on the
            // strip's box of pixel centres (x 0..15, y 4r..4r+3) against the reach blend_px_ms / blend_px give a pixel (lim + m)
            [[maybe_unused]] bool strip_lane[4] = {false, false, false, false};
            if ((uint32_t)lane < ccnt) {
                const float4* src = records + (size_t)s_queue[c0 + (uint32_t)lane] * REC_V4;
                float4 r0 = src[0], r1 = src[1];
                // exact test the tile rect cannot do: the quad is the parallelogram |u|,|v| <= 1, so
                // it misses the tile iff the box of the tile's sample positions (its pixel centres for a
                // single-sampled target) lies wholly beyond one of its
                // two axes (the box's own axes are the rect test). The verdict rides in a spare
                // dword of the staged record (keepz_of) and the blend loop skips rejected records.
                bool keep;
                if constexpr (VARIANT == RV_OBB) {
                    stage_obb(r0, r1, tile_ox, tile_oy);
                    // tile centre = first pixel + (7.5, 7.5), half extent HALF in both axes
                    const float uc = fmaf(r0.w, 7.5f, fmaf(r0.z, 7.5f, r0.x));
                    const float vc = fmaf(r1.y, 7.5f, fmaf(r1.x, 7.5f, r0.y));
                    const float eu = HALF * (fabsf(r0.z) + fabsf(r0.w));
                    const float ev = HALF * (fabsf(r1.x) + fabsf(r1.y));
                    keep = !(fabsf(uc) - eu > 1.0001f * OBB_C) && !(fabsf(vc) - ev > 1.0001f * OBB_C);
                } else {  // axis-aligned square: uv = (m00 * dx, m11 * dy)
                    const float dcx = tile_cx - r0.x, dcy = tile_cy - r0.y;
                    const float uc = r0.z * dcx, vc = r0.w * dcy;
                    const float eu = HALF * fabsf(r0.z), ev = HALF * fabsf(r0.w);
                    keep = !(fabsf(uc) - eu > 1.0001f) && !(fabsf(vc) - ev > 1.0001f);
                }
                if constexpr (ABLATE) keep = keep || (fp.debug & 64u);
                if constexpr (VARIANT == RV_OBB) {
                    float4 r2 = src[2];
                    stage_obb_margin<MSAA>(r0, r1);   // p[4] is unused by the OBB record
                    if constexpr (FAST) {
                        // every sample of every pixel of the tile inside the quad (INTERIOR records, above): the pixel centres
                        // reach 7.5 px from the tile centre; ONE margin for both axes, as in blend_px_ms; false for a NaN
                        const float su = 7.5f * (fabsf(r0.z) + fabsf(r0.w)), sv = 7.5f * (fabsf(r1.x) + fabsf(r1.y));
                        const float ucc = fmaf(r0.w, 7.5f, fmaf(r0.z, 7.5f, r0.x)), vcc = fmaf(r1.y, 7.5f, fmaf(r1.x, 7.5f, r0.y));
                        keep_lane = keep && (!DEPTH || r2.w >= tile_dmin);
                        interior_lane = keep && (!DEPTH || r2.w >= tile_dmax) &&
                                        (fabsf(ucc) + su + r1.z <= 0.9999f * OBB_C) && (fabsf(vcc) + sv + r1.z <= 0.9999f * OBB_C);
                        if constexpr (STRIPS) {
                        const float eus = fmaf(7.5f, fabsf(r0.z), 1.5f * fabsf(r0.w)), evs = fmaf(7.5f, fabsf(r1.x), 1.5f * fabsf(r1.y));
                        const float ub = fmaf(r0.z, 7.5f, r0.x), vb = fmaf(r1.x, 7.5f, r0.y);
                        const float reach = 1.0001f * (OBB_C + r1.z);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {   // (a NaN keeps the strip)
                            const float yc = 4.0f * (float)r + 1.5f;
                            strip_lane[r] = keep_lane && !(fabsf(fmaf(r0.w, yc, ub)) - eus > reach) && !(fabsf(fmaf(r1.y, yc, vb)) - evs > reach);
                        }
                        }
                    }
                    r2.w = keepz_of(keep && (!DEPTH || r2.w >= tile_dmin), r2.w);
                    s_rec[lane * ST_V4 + 0] = r0;
                    s_rec[lane * ST_V4 + 1] = r1;
                    s_rec[lane * ST_V4 + 2] = r2;
                    if constexpr (ST_V4 == 4)   // (blend_px_ms's own expressions: the instance-sort rasteriser forms them per pixel)
                        s_rec[lane * ST_V4 + 3] = make_float4(fmaf(r0.w, MS_OY0, r0.z * MS_OX0), fmaf(r0.w, MS_OY1, r0.z * MS_OX1),
                                                              fmaf(r1.y, MS_OY0, r1.x * MS_OX0), fmaf(r1.y, MS_OY1, r1.x * MS_OX1));
                } else if constexpr (VARIANT == RV_AABB3D) {
                    s_rec[lane * ST_V4 + 0] = r0;
                    s_rec[lane * ST_V4 + 1] = r1;
                    float4 r2 = src[2];
                    r2.w = keepz_of(keep && (!DEPTH || r2.w >= tile_dmin), r2.w);
                    s_rec[lane * ST_V4 + 2] = r2;
                } else {
                    float4 st[6];
                    stage_surfel(src, tile_ox, tile_oy, aspect, st);
                    if constexpr (!BBOX) keep = keep && !(surfel_negligible_in_tile(st, surfel_limit) && !(ABLATE && (fp.debug & 64u)));
                    const float z = src[5].x;
                    st[5].y = keepz_of(keep && (!DEPTH || z >= tile_dmin), z);
                    if constexpr (FAST) {
                        // INTERIOR (above): the quad is the square |u|, |v| <= 1 with u = du xl + U0, v = dv yl + V0 over the
                        // tile's pixel centres xl, yl in [0, 15]; blend_px_ms asks for max(|u|, |v|) <= 1 - m, m = 0.375 max(|du|, |dv|)
                        const float um = fmaxf(fabsf(st[0].x), fabsf(fmaf(st[0].y, 15.0f, st[0].x)));
                        const float vm = fmaxf(fabsf(st[0].z), fabsf(fmaf(st[0].w, 15.0f, st[0].z)));
                        const float mg = MSAA == 1 ? 0.0f : ms_reach(MSAA) * fmaxf(fabsf(st[0].y), fabsf(st[0].w));
                        keep_lane = keep && (!DEPTH || z >= tile_dmin);
                        interior_lane = keep && (!DEPTH || z >= tile_dmax) && (fmaxf(um, vm) + mg <= 0.9999f);
                        if constexpr (STRIPS) {
                            const float evs = 1.5f * fabsf(st[0].w), reach = 1.0001f * (1.0f + mg);
#pragma unroll
                            for (int r = 0; r < 4; ++r)   // (u does not depend on the row: `keep` has looked at it)
                                strip_lane[r] = keep_lane && !(fabsf(fmaf(st[0].w, 4.0f * (float)r + 1.5f, st[0].z)) - evs > reach);
                        }
                    }
#pragma unroll
                    for (int v = 0; v < 6; ++v) s_rec[lane * ST_V4 + v] = st[v];
                }
            }
            // make the staged records visible to every lane of this wave before the broadcast reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            BGS_PHASE(2);
            const uint32_t kend = (ABLATE && (fp.debug & 16u)) ? min(ccnt, 1u) : ccnt;  // ablation bit 16: stage, blend 1
            if constexpr (FAST) {
                // the round's records as two bit masks in scalar registers: which to blend, which of those are interior
                unsigned long long todo = __builtin_amdgcn_ballot_w64(keep_lane);
                const unsigned long long inter = __builtin_amdgcn_ballot_w64(interior_lane);
                [[maybe_unused]] unsigned long long reach_mask[4];   // bit k: record k reaches strip r (STRIPS)
                if constexpr (STRIPS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) reach_mask[r] = __builtin_amdgcn_ballot_w64(strip_lane[r]);
                }
                if constexpr (TRACE) trace_staged += ccnt;
                uint32_t since = 0u;   // records since the last look at the tile's saturation
                bool out = false;
                // Runs of a kind, each taken out of `todo` as a bit mask of its own (the records below the first one of the other
                // kind): the loop over a run then turns on ONE scalar compare — asking every record for its kind cost eight
                // scalar instructions per record in the compiler's hands (a shift, a bit test, a select into a lane mask, an
                // and with exec, ...), and scalar instructions count here like vector ones
                while (todo != 0ull) {
                    // a run of interior records
                    const unsigned long long first_other = todo & ~inter;
                    unsigned long long run = first_other != 0ull ? (todo & ((first_other & (0ull - first_other)) - 1ull)) : todo;
                    todo ^= run;
                    unsigned long long run0 = run;   // (the records blended are counted after the run: a scalar addition per record less)
                    while (run != 0ull) {
                        const uint32_t k = (uint32_t)__builtin_ctzll(run);
                        run &= run - 1ull;
                        if constexpr (VARIANT == RV_OBB) {
                            const float4 a0 = s_rec[k * ST_V4 + 0], a1 = s_rec[k * ST_V4 + 1], a2 = s_rec[k * ST_V4 + 2];
                            const float ux = fmaf(a0.z, qx, a0.x), vx = fmaf(a1.x, qx, a0.y);
                            if constexpr (MSAA != 1) {
                                blend_interior_ms(ux, vx, a0.w, a1.y, a2.z, a1.w, a2.x, a2.y, t_eps, qy[0], qy[1], T[0], crg[0], cb[0], T[1], crg[1], cb[1]);
                                blend_interior_ms(ux, vx, a0.w, a1.y, a2.z, a1.w, a2.x, a2.y, t_eps, qy[2], qy[3], T[2], crg[2], cb[2], T[3], crg[3], cb[3]);
                            } else {
                                blend_interior_1(ux, vx, a0.w, a1.y, a2.z, a1.w, a2.x, a2.y, t_eps, qy[0], qy[1], T[0], crg[0], cb[0], T[1], crg[1], cb[1]);
                                blend_interior_1(ux, vx, a0.w, a1.y, a2.z, a1.w, a2.x, a2.y, t_eps, qy[2], qy[3], T[2], crg[2], cb[2], T[3], crg[3], cb[3]);
                            }
                        } else {
                            // a1 = P0x P0y P0z Pxx | a2 = Pxy Pxz Pyx Pyy | a3 = Pyz Dx0 dDx Dy0 | a4 = dDy r g b | a5 = opacity ...
                            const float4 a1 = s_rec[k * ST_V4 + 1], a2 = s_rec[k * ST_V4 + 2], a3 = s_rec[k * ST_V4 + 3],
                                         a4 = s_rec[k * ST_V4 + 4];
                            const float opa = s_rec[k * ST_V4 + 5].x;
                            const float pxq = fmaf(a1.w, qx, a1.x), pyq = fmaf(a2.x, qx, a1.y), pzq = fmaf(a2.y, qx, a1.z);
                            const float ddx = fmaf(a3.z, qx, a3.y);
#pragma unroll
                            for (int r = 0; r < ROWS; ++r) {
                                if constexpr (MSAA != 1) blend_interior_surfel_ms(pxq, pyq, pzq, ddx, a2.z, a2.w, a3.x, a4.x, a3.w, opa, a4.y, a4.z, a4.w, t_eps, qy[r], T[r], crg[r], cb[r]);
                                else blend_interior_surfel_1(pxq, pyq, pzq, ddx, a2.z, a2.w, a3.x, a4.x, a3.w, opa, a4.y, a4.z, a4.w, t_eps, qy[r], T[r], crg[r], cb[r]);
                            }
                        }
                        if constexpr (MIDROUND_EXIT)
                            if (++since == BGS_MIDROUND_PERIOD) { since = 0u; if (__all(all_saturated(T, t_eps))) { out = true; break; } }
                    }
                    work += WORK_BLENDED * (uint32_t)__builtin_popcountll(run0 ^ run);
                    if constexpr (TRACE) trace_blended += (uint32_t)__builtin_popcountll(run0 ^ run);
                    if (out) break;
                    // a run of the others: strip by strip through blend_px_ms / blend_px
                    const unsigned long long first_inter = todo & inter;
                    run = first_inter != 0ull ? (todo & ((first_inter & (0ull - first_inter)) - 1ull)) : todo;
                    todo ^= run;
                    run0 = run;
                    while (run != 0ull) {
                        const uint32_t k = (uint32_t)__builtin_ctzll(run);
                        run &= run - 1ull;
                        StagedRecord<VARIANT> sr;
                        sr.load(s_rec + k * ST_V4);
                        if constexpr (ST_V4 == 4) sr.a3 = s_rec[k * ST_V4 + 3];
                        // the record's depth (a kept record's: > 0), wave-uniform; zmixed: the tile's depths split it
                        [[maybe_unused]] float zr = 0.0f;
                        if constexpr (DEPTH) zr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(VARIANT == RV_SURFEL ? sr.a5.y : sr.a2.w)));
                        const bool zmixed = DEPTH && zr < tile_dmax;
#pragma unroll
                        for (int r = 0; r < ROWS; ++r) {
                            if constexpr (STRIPS) if (((reach_mask[r] >> k) & 1ull) == 0ull) continue;   // (scalar: the record cannot reach this strip)
                            if constexpr (MSAA == 4) {
                                float4 d4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                                if constexpr (DEPTH) if (zmixed) d4 = s_depth[r * 64 + lane];
                                blend_px_ms<VARIANT, DEPTH, false, ST_V4 == 4>(sr, qx, qy[r], aspect, t_eps, T[r], crg[r], cb[r], zr, zmixed, d4);
                            } else if constexpr (MSAA == 1) {
                                blend_px<VARIANT, DEPTH, false>(sr, qx, qy[r], aspect, t_eps, T[r], crg[r], cb[r], zr, dpx[r]);
                            } else {
                                blend_px_msn<VARIANT, DEPTH, MSAA, false>(sr, qx, qy[r], aspect, t_eps, T[r], crg[r], cb[r], zr, zmixed,
                                                                          s_depth_f + (r * 64 + lane) * MSAA);
                            }
                        }
                        if constexpr (MIDROUND_EXIT)
                            if (++since == BGS_MIDROUND_PERIOD) { since = 0u; if (__all(all_saturated(T, t_eps))) { out = true; break; } }
                    }
                    work += WORK_BLENDED * (uint32_t)__builtin_popcountll(run0 ^ run);
                    if constexpr (TRACE) trace_blended += (uint32_t)__builtin_popcountll(run0 ^ run);
                    if (out) break;
                }
            } else
            for (uint32_t k = 0; k < kend; ++k) {
                StagedRecord<VARIANT> sr;
                sr.load(s_rec + k * ST_V4);
                const float keepz = VARIANT == RV_SURFEL ? sr.a5.y : sr.a2.w;
                if constexpr (TRACE) trace_staged += 1u;
                // the record's depth, or the "skip" flag in its sign bit: wave-uniform, a scalar branch
                const float zr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(keepz)));
                if (!keepz_keeps(zr)) continue;
                if constexpr (TRACE) trace_blended += 1u;
                work += WORK_BLENDED;
                const bool zmixed = DEPTH && zr < tile_dmax;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    if constexpr (MSAA == 4) {
                        float4 d4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if constexpr (DEPTH) if (zmixed) d4 = s_depth[r * 64 + lane];
                        blend_px_ms<VARIANT, DEPTH, BBOX>(sr, qx, qy[r], aspect, t_eps, T[r], crg[r], cb[r], zr, zmixed, d4);
                    } else if constexpr (MSAA == 1) {
                        blend_px<VARIANT, DEPTH, BBOX>(sr, qx, qy[r], aspect, t_eps, T[r], crg[r], cb[r], zr, dpx[r]);
                    } else {
                        blend_px_msn<VARIANT, DEPTH, MSAA, BBOX>(sr, qx, qy[r], aspect, t_eps, T[r], crg[r], cb[r], zr, zmixed,
                                                                 s_depth_f + (r * 64 + lane) * MSAA);
                    }
                }
                if constexpr (MIDROUND_EXIT)
                    if ((k & (BGS_MIDROUND_PERIOD - 1u)) == BGS_MIDROUND_PERIOD - 1u && __all(all_saturated(T, t_eps))) break;
            }
            saturated = __all(all_saturated(T, t_eps));
            ++rounds;
            }
            if (saturated) { tile_saturated = true; break; }
            __builtin_amdgcn_wave_barrier();  // blend reads of s_rec / s_queue done before they are rewritten
        }
        if constexpr (ABLATE) if (fp.debug & 32u) qn = 0u;
        if (end && qn == 0u) break;
    }
#if BGS_PHASE_TRACE
    if constexpr (TRACE) phase[3] = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    {
        // pixel coordinates are recomputed from an opaque copy of the lane id: keeping the four row
        // indices alive across the blend loop costs registers the loop needs (they were spilled)
        int lw = lane;
        asm volatile("" : "+v"(lw));
        const int pxw = (int)tx * TILE_PX + (lw & 15), pyw = (int)ty * TILE_PX + row0 + (lw >> 4);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int py = pyw + 4 * r;
            if (pxw < fp.width && py < fp.height) {
                // (multisampled: the resolve — the mean of the samples is C + clear * mean_s(T_s))
                const float Tf = trans_mean(T[r]);
                const float4 c = make_float4(fmaf(Tf, fp.clear[0], crg[r].x), fmaf(Tf, fp.clear[1], crg[r].y),
                                             fmaf(Tf, fp.clear[2], cb[r]), fmaf(Tf, fp.clear[3], 1.0f - Tf));
                const size_t at = (size_t)py * (size_t)fp.width + (size_t)pxw;
                // Streaming stores: a frame's 33 MB target is written once and read by nobody on this chip before
                // the next frames have pushed it out anyway; kept out of the L2 / Infinity Cache allocation the
                // cloud planes that keygen and project+bin read every frame stay resident (+1.5 % dense, +3 %
                // scene-like frames/s, same-box A/B).
                if (!(want_srgb8 & OUT_SKIP_F32)) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((v4f){c.x, c.y, c.z, c.w}, reinterpret_cast<v4f*>(fb + at));
                }
                // the frame in the reference's target format too (Rgba8UnormSrgb, or Rgba16Float for an hdr
                // camera), here instead of in a separate pass over the 33 MB f32 image (the frame's own
                // destination travels in FrameParams)
                if (want_srgb8 & OUT_SRGB8) {
                    __builtin_nontemporal_store(pack_srgb8(c), (fp.srgb8_target ? reinterpret_cast<uint32_t*>(fp.srgb8_target) : fb8_default) + at);
                } else if (want_srgb8 & OUT_RGBA16F) {
                    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
                    const uint2 h = pack_rgba16f(c);
                    __builtin_nontemporal_store((v2u){h.x, h.y}, reinterpret_cast<v2u*>(fp.srgb8_target ? reinterpret_cast<uint2*>(fp.srgb8_target) : reinterpret_cast<uint2*>(fb8_default)) + at);
                }
            }
        }
    }
    return rounds | (tile_saturated ? 0x80000000u : 0u);   // (bit 31: the tile ended saturated)
}

// Waves per SIMD: 8 for the single-sampled ellipse variants (<= 64 VGPRs), 6 for the surfel variant; the multisampled
// instantiations carry six transmittance words per pixel instead of one (5 / 4 waves).
#ifndef BGS_MS_WAVES
// multisampled ellipse variants: 6 since round 6. Round 5's kernel (92-96 VGPRs) spilled 10-25 registers INTO the record loop at
// 80 and lost 10 %; with the tile id in scalar registers and the interior records in their own loop it is 83-86 and the 3-5
// registers that spill at 80 are tile set-up values outside the loops: dense 1 M even, scene-like +1.9 %, trained-like +4.5 %
// frames/s (profiles/r6_experiments/ms_waves_6_ab.txt)
#define BGS_MS_WAVES 6
#endif
constexpr int raster_waves_per_simd(const int variant, const int msaa, const bool depth) {
    return msaa == 8 ? (variant == 2 ? (depth ? 2 : 3) : (depth ? 3 : 4))   // ten transmittance words per pixel; 8 KB of depth samples per wave
         : msaa >= 2 ? (variant == 2 ? (depth ? 3 : 4) : (depth ? 5 : BGS_MS_WAVES)) : (variant == 2 ? (depth ? 5 : 6) : (depth ? 7 : 8));
}
int raster_scan_waves_per_simd(const FrameParams& fp) {
    const int variant = fp.aabb == 0u ? RV_OBB : (fp.gaussian_mode != 0u ? RV_AABB3D : RV_SURFEL);
    return raster_waves_per_simd(variant, (int)fp.sample_count, fp.depth_ptr != 0ull);
}
// MODE (round 6; a bool until then): 0 the plain instantiation; 1 mid-round exit for DENSE frames (supertile level >= 2: heavy-tile
// strips, no strip reach masks, no staged sample offsets — each of them only costs there); 2 mid-round exit for frames of a kind
// whose saturating tiles hold the work at a lower level (trained-like): the sparse frames' machinery plus the exit
template <int VARIANT, bool TRACE = false, int MODE = 0, int MSAA = 1, bool DEPTH = false, bool BBOX = false>
__global__ __launch_bounds__(256, raster_waves_per_simd(VARIANT, MSAA, DEPTH)) void raster_scan_kernel(const FrameParams* __restrict__ fpp, const float4* __restrict__ records,
                                                          const uint32_t* __restrict__ coarse,
                                                          uint32_t coarse_cap, uint32_t sup_mul,
                                                          uint32_t sup_x, Control* ctl,
                                                          float4* __restrict__ fb,
                                                          uint32_t* __restrict__ fb8_default, uint32_t want_srgb8,
                                                          FrameCleanup cl, uint4* __restrict__ trace,
                                                          const uint8_t* __restrict__ heavy_in, uint8_t* __restrict__ heavy_out,
                                                          const uint16_t* __restrict__ order, uint16_t* __restrict__ cost_out) {
    unsigned long long trace_t0 = 0ull;
    uint32_t trace_scanned = 0u, trace_blended = 0u, trace_staged = 0u, work = 0u;
    if constexpr (TRACE) trace_t0 = __builtin_amdgcn_s_memtime();
    const FrameParams fp = *fpp;  // left in device memory by the frame's keygen
    // records staged per round: the whole queue (24 KB of LDS per workgroup for the 96-byte surfel records, five
    // workgroups per CU at its 5 waves/SIMD; rounds of 32 were 1-5 % slower)
    constexpr uint32_t STAGE = 64u;
    constexpr bool MIDROUND_EXIT = MODE != 0;
    __shared__ float4 s_rec_all[4][STAGE * staged_v4(VARIANT, MSAA, DEPTH, MODE == 1)];
    __shared__ uint32_t s_queue_all[4][64];
    __shared__ float4 s_depth_all[4][DEPTH && MSAA > 1 ? 64 * MSAA : 1];   // the tile's depth samples (raster_tile): MSAA floats per pixel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ntiles = (uint32_t)(fp.tiles_x * fp.tiles_y);
    const uint32_t nblocks = (ntiles + 3u) / 4u;
    // HEAVY TILES (MIDROUND_EXIT instantiations, i.e. dense frames; kernels.h HeavyFeedback): the workgroups behind the
    // regular grid each take ONE tile of the list a completed frame left — a tile that needed more than one staging round
    // there — and give each of its four 16 x 4 row strips a wave of its own; the regular wave of such a tile steps aside.
    const bool strip_block = MIDROUND_EXIT && blockIdx.x >= nblocks;
    // Surfel frames: four runs per XCD. A surfel tile is ~10x the work of an ellipse tile and only 5120 of the 8160 tile
    // waves are resident at once, so an XCD whose band is the image's (heavier) top or bottom edge ends the launch:
    // dense 1 M surfel frame 544 -> 492 us (runs 2 / 4 / 8: 520 / 492 / 508). The ellipse variants keep one band per XCD
    // — their tiles share lists and records with their neighbours and the split only costs (scene-like 61.5 -> 63.6 us) —
    // except the multisampled dense (mid-round-exit) frames, two runs: the bottom band of the dense 1 M frame holds 13 %
    // more work than the mean band (tile trace), and with 5 waves per SIMD that XCD ends the launch. Same-box A/B, runs
    // 1 / 2 / 4: rasteriser alone 61.3 / 57.8 / 58.0 us, frames in flight 15.9 / 16.3 / 16.1 k frames/s (5 M dense: 49.8 /
    // 47.2 / 47.6 us, in flight the same); single-sampled 45.2 / 44.4 / 46.4 us and no change in flight: one band kept
    // (profiles/r4_experiments/tile_order.txt).
    constexpr uint32_t RUNS = raster_runs(VARIANT, MSAA, MIDROUND_EXIT);
    // `order` (tile_order_kernel): this frame's workgroups in the order of the work a completed frame found in them,
    // heaviest first inside every XCD's share — the same share xcd_remap / xcd_remap_runs deal out
    uint32_t tile = strip_block ? 0xFFFFFFFFu
                  : (order ? (uint32_t)order[blockIdx.x] : raster_block_item<RUNS>(blockIdx.x, nblocks)) * 4u + (uint32_t)wave;
    if constexpr (MIDROUND_EXIT) {
        if (strip_block) {
            const uint32_t have = heavy_in ? min(*reinterpret_cast<const uint32_t*>(heavy_in), HEAVY_CAP) : 0u;
            const uint32_t sb = blockIdx.x - nblocks;
            if (sb >= have) return;
            tile = reinterpret_cast<const uint16_t*>(heavy_in + HEAVY_LIST_OFFSET)[sb];
        } else if (heavy_in && tile < ntiles && heavy_in[HEAVY_FLAGS_OFFSET + tile]) {
            tile = 0xFFFFFFFFu;   // a strip block draws this tile
        }
    }
    const uint32_t draw_count = ctl->sort_overflow ? 0u : ctl->draw_count;   // as the project kernel read it
    const uint32_t cmax_bits = __builtin_amdgcn_readfirstlane(ctl->color_max_bits);
    const float t_eps = frame_t_eps(cmax_bits);
    const float surfel_limit = frame_surfel_limit(cmax_bits);
    if (cl.other_ctl && !strip_block) {
        // the status words of this frame's chained scans are dead by now: zero the used ones, and
        // the Control block the lane's next frame will use; report this frame's counters to the host
        const uint32_t g = blockIdx.x * 256u + (uint32_t)tid, gn = nblocks * 256u;
        uint32_t* zc = reinterpret_cast<uint32_t*>(cl.other_ctl);
        for (uint32_t i = g; i < (uint32_t)(sizeof(Control) / 4u); i += gn) zc[i] = 0u;
        if (blockIdx.x == nblocks - 1u) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ctl);
            uint32_t* host = reinterpret_cast<uint32_t*>(cl.host_ctl);
            constexpr uint32_t HEADER_WORDS = CONTROL_HEADER_WORDS;  // draw_count .. strip_tiles
            constexpr uint32_t COARSE_OFF = (uint32_t)(offsetof(Control, coarse_total) / 4u);
            constexpr uint32_t SPLIT_OFF = (uint32_t)(offsetof(Control, splitters) / 4u);
            static_assert(offsetof(Control, strip_tiles) / 4u == HEADER_WORDS - 2u && offsetof(Control, saturated_tiles_prev) / 4u == HEADER_WORDS - 1u,
                          "strip_tiles and saturated_tiles_prev are the header's last words");
            if ((uint32_t)tid < HEADER_WORDS - 2u) host[tid] = src[tid];
            if ((uint32_t)tid == HEADER_WORDS - 2u) {   // how many tiles this launch draws with strip workgroups (known at its start)
                uint32_t strips = 0u;
                if constexpr (MIDROUND_EXIT) strips = heavy_in ? min(*reinterpret_cast<const uint32_t*>(heavy_in), HEAVY_CAP) : 0u;
                host[tid] = strips;
            }
            if ((uint32_t)tid == HEADER_WORDS - 1u) {   // what the order's cost plane said about saturation (tile_order_kernel's eight pairs)
                uint32_t work = 0u, sat = 0u;
                if (cl.order_stats)
                    for (uint32_t x = 0u; x < 8u; ++x) { work += cl.order_stats[2u * x]; sat += cl.order_stats[2u * x + 1u]; }
                host[tid] = (cl.order_stats && work) ? (uint32_t)(((uint64_t)sat * 0x7FFFu) / work) : 0xFFFFFFFFu;
            }
            host[COARSE_OFF + (uint32_t)tid] = src[COARSE_OFF + (uint32_t)tid];
            // the 1 / (256 sub)-quantile keys of this frame's sorted list: later frames' bucket splitters
            const uint32_t nbk = BUCKET_COUNT * min(max(cl.split_sub, 1u), BUCKET_SUB_MAX);
            // (only the nbk - 1 keys the host reads, plus one ~0 terminator: 1 KB for the headline's sub = 1, not 16 KB)
            for (uint32_t t = (uint32_t)tid; t < nbk; t += 256u)
                host[SPLIT_OFF + t] = (draw_count != 0u && t < nbk - 1u)
                                          ? (cl.sorted[(uint32_t)(((unsigned long long)(t + 1u) * draw_count) / nbk)].x ^ cl.key_xor)
                                          : 0xFFFFFFFFu;
        }
        const uint32_t part_words = (fp.n + KEYGEN_TILE - 1u) / KEYGEN_TILE;
        for (uint32_t i = g; i < part_words; i += gn) cl.part_status[i] = 0u;
        const uint32_t depth_v4 = ((draw_count + cl.depth_tile - 1u) / cl.depth_tile) * (RADIX_BASE / 4u);
        for (uint32_t p = 0u; p < cl.places; ++p) {
            uint4* dst = reinterpret_cast<uint4*>(cl.depth_status + (size_t)p * cl.pass_stride);
            for (uint32_t i = g; i < depth_v4; i += gn) dst[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        const uint32_t bin_v4 = ((draw_count + BIN_RANKS - 1u) / BIN_RANKS) * (MAX_SUPERTILES / 4u);
        uint4* bdst = reinterpret_cast<uint4*>(cl.bin_status);
        for (uint32_t i = g; i < bin_v4; i += gn) bdst[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    uint32_t tile_done = 0xFFFFFFFFu;
    [[maybe_unused]] uint32_t phase[4] = {0u, 0u, 0u, 0u};
    if (tile < ntiles) {  // whole wave; nothing in here synchronises across waves
        uint32_t rounds;
        bool reports = true;   // which wave speaks for the tile in the feedback
        if (MIDROUND_EXIT && strip_block) {
            rounds = raster_tile<VARIANT, false, MODE, 1, MSAA, DEPTH, BBOX>(fp, records, coarse, coarse_cap, sup_mul, sup_x, ctl, fb, fb8_default, want_srgb8,
                                                                   t_eps, surfel_limit, s_rec_all[wave], s_queue_all[wave], s_depth_all[wave], lane, tile, 4 * wave,
                                                                   trace_scanned, trace_blended, trace_staged, work, phase);
            reports = wave == 0;
        } else {
            rounds = raster_tile<VARIANT, TRACE, MODE, 4, MSAA, DEPTH, BBOX>(fp, records, coarse, coarse_cap, sup_mul, sup_x, ctl, fb, fb8_default, want_srgb8,
                                                                   t_eps, surfel_limit, s_rec_all[wave], s_queue_all[wave], s_depth_all[wave], lane, tile, 0,
                                                                   trace_scanned, trace_blended, trace_staged, work, phase);
            tile_done = tile;
        }
        // what the tile cost, for the order of the frames behind this one (kernels.h TileCost): whoever drew it says so —
        // its regular wave, or the first strip's wave for all four (it scanned and staged what the tile's wave would
        // have; a regular wave that stepped aside writes nothing, so the tile keeps a cost worth its place in the order
        // should the next frame draw it whole again)
        // (bit 15: the tile ended saturated, not at the end of its lists — what the host's choice of the mid-round-exit
        // instantiation for frames of this kind follows; the work count keeps 15 bits)
        if (cost_out && reports && lane == 0) cost_out[tile] = (uint16_t)(min(0x7FFFu, work) | ((rounds >> 31) << 15));
        rounds &= 0x7FFFFFFFu;
        if constexpr (MIDROUND_EXIT) {
            // feedback for the frames behind this one: a tile that did not saturate inside its first staging round is
            // heavy (parameter-free: the median tile of a dense frame saturates after ~35 of the round's <= 64 records)
            if (heavy_out && reports && lane == 0) {
                bool heavy = rounds >= 2u;
                if (heavy) {
                    const uint32_t at = atomicAdd(reinterpret_cast<uint32_t*>(heavy_out), 1u);
                    if (at < HEAVY_CAP) reinterpret_cast<uint16_t*>(heavy_out + HEAVY_LIST_OFFSET)[at] = (uint16_t)tile;
                    else heavy = false;   // a flagged tile must be in the list: somebody has to draw it
                }
                heavy_out[HEAVY_FLAGS_OFFSET + tile] = heavy ? 1u : 0u;
            }
        }
    }

    if constexpr (TRACE) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && trace && tile_done != 0xFFFFFFFFu) {
            const uint32_t hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID, all 32 bits
            const uint32_t xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
            trace[2u * tile_done] = make_uint4((uint32_t)trace_t0, (uint32_t)(trace_t0 >> 32), (uint32_t)t1, (uint32_t)(t1 >> 32));
            trace[2u * tile_done + 1u] = make_uint4(hw_id, xcc_id, trace_scanned, min(trace_blended, 0xFFFFu) | (min(trace_staged, 0xFFFFu) << 16));
#if BGS_PHASE_TRACE
            const uint32_t t0w = (uint32_t)trace_t0;
            trace[2u * ntiles + tile_done] = make_uint4(phase[0] - t0w, phase[1] - t0w, phase[2] - t0w, phase[3] - t0w);
#endif
        }
    }
}

// TileCost -> the order of a frame's raster workgroups (kernels.h). Workgroup b of the raster grid runs on XCD b % 8 and
// draws the b / 8-th group of four tiles of that XCD's share (xcd_remap / xcd_remap_runs). Block x of this kernel sorts
// XCD x's share by what a completed frame's waves measured — the group's longest wave, descending, ties in the share's
// own order — and writes order[8 i + x] = the group the XCD's i-th workgroup draws. The hardware hands workgroups out
// in grid order as wave slots free up, so every XCD starts its longest chains first and fills up behind them with short
// ones (longest-processing-time list scheduling) instead of ending on whatever its share's last rows hold. Whatever the
// cost words hold (zeros, another view's frame), `order` is a permutation of the groups: the costs only decide balance.
// Bitonic sort of (0xFFFF - cost) << 16 | i in LDS, up to 2048 groups per XCD (65 535 tiles).
constexpr uint32_t ORDER_MAX_SHARE = 2048u;
template <uint32_t RUNS>
__global__ __launch_bounds__(256) void tile_order_kernel(const uint16_t* __restrict__ cost, uint16_t* __restrict__ order,
                                                         const uint32_t nblocks, const uint32_t ntiles) {
    __shared__ uint32_t s_key[ORDER_MAX_SHARE];
    const uint32_t x = blockIdx.x, tid = threadIdx.x;
    const uint32_t len = nblocks > x ? (nblocks - x + 7u) / 8u : 0u;
    uint32_t n2 = 2u;
    while (n2 < len) n2 <<= 1;
    // what this XCD share's tiles say about saturation (kernels.h tile_order_stats_offset): how much of the frame's work was in
    // tiles that ended saturated — the host's choice of the mid-round-exit rasteriser for frames of this kind
    __shared__ uint32_t s_stats[2];
    if (tid < 2u) s_stats[tid] = 0u;
    uint32_t my_worked = 0u, my_sat = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < n2; i += 256u) {
        uint32_t key = 0xFFFFFFFFu;
        if (i < len) {
            const uint32_t t0 = raster_block_item<RUNS>(8u * i + x, nblocks) * 4u;
            uint32_t c = 0u;
            for (uint32_t k = 0u; k < 4u; ++k)
                if (t0 + k < ntiles) {
                    const uint32_t w = (uint32_t)cost[t0 + k];
                    c = max(c, w & 0x7FFFu);                       // (bit 15: ended saturated)
                    my_worked += w & 0x7FFFu;
                    my_sat += (w >> 15) ? (w & 0x7FFFu) : 0u;
                }
            key = ((0xFFFFu - c) << 16) | i;
        }
        s_key[i] = key;
    }
    if (my_worked) atomicAdd(&s_stats[0], my_worked);
    if (my_sat) atomicAdd(&s_stats[1], my_sat);
    __syncthreads();
    if (tid < 2u) reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(order) + tile_order_stats_offset(ntiles))[2u * x + tid] = s_stats[tid];
    for (uint32_t k = 2u; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
            for (uint32_t p = tid; p < n2 / 2u; p += 256u) {
                const uint32_t a = 2u * j * (p / j) + (p % j), b = a + j;
                const uint32_t ka = s_key[a], kb = s_key[b];
                const bool up = (a & k) == 0u;
                if ((ka > kb) == up) { s_key[a] = kb; s_key[b] = ka; }
            }
            __syncthreads();
        }
    for (uint32_t i = tid; i < len; i += 256u)
        order[8u * i + x] = (uint16_t)raster_block_item<RUNS>(8u * (s_key[i] & 0xFFFFu) + x, nblocks);
}

void launch_tile_order(hipStream_t stream, const uint16_t* cost, uint16_t* order, uint32_t ntiles, const FrameParams& fp,
                       bool midround_exit) {
    const int variant = fp.aabb == 0u ? RV_OBB : (fp.gaussian_mode != 0u ? RV_AABB3D : RV_SURFEL);
    launch_tile_order_runs(stream, cost, order, ntiles, raster_runs(variant, (int)fp.sample_count, midround_exit));
}

void launch_tile_order_runs(hipStream_t stream, const uint16_t* cost, uint16_t* order, uint32_t ntiles, uint32_t runs) {
    const uint32_t nblocks = (ntiles + 3u) / 4u;
    if (nblocks == 0u || nblocks > 8u * ORDER_MAX_SHARE) return;
    if (runs == 4u) hipLaunchKernelGGL((tile_order_kernel<4u>), dim3(8), dim3(256), 0, stream, cost, order, nblocks, ntiles);
    else if (runs == 2u) hipLaunchKernelGGL((tile_order_kernel<2u>), dim3(8), dim3(256), 0, stream, cost, order, nblocks, ntiles);
    else hipLaunchKernelGGL((tile_order_kernel<1u>), dim3(8), dim3(256), 0, stream, cost, order, nblocks, ntiles);
}

void launch_raster_scan(hipStream_t stream, const FrameParams& fp, const FrameParams* d_fp, const void* records,
                        const uint32_t* coarse, uint32_t coarse_cap,
                        uint32_t sup_edge, Control* ctl, float4* framebuffer, uint32_t* srgb8_default,
                        uint32_t out_format, const FrameCleanup& cleanup, uint4* tile_trace, int mode,
                        const uint8_t* heavy_in, uint8_t* heavy_out, const uint16_t* order, uint16_t* cost_out) {
    const uint32_t ntiles = (uint32_t)(fp.tiles_x * fp.tiles_y);
    if (ntiles == 0) return;
    const float4* rec = (const float4*)records;
    const uint32_t sup = sup_edge, sup_mul = supertile_mul(sup_edge);
    const uint32_t sup_x = ((uint32_t)fp.tiles_x + sup - 1u) / sup;
    // instantiations: variant x mid-round exit x samples per pixel; the per-tile trace exists for the variants without a
    // depth buffer, the depth test for the untraced ones. Sample2 / Sample8 and the bounding-box overlay (debug features:
    // nothing in the reference selects them by default) come untraced and without the mid-round exit only — the caller
    // (enqueue_frame) never asks those frames for either.
    const bool depth = fp.depth_ptr != 0ull, bbox = fp.visualize_bbox != 0u;
    const uint32_t ms = fp.sample_count;
    const bool plain = (ms == 1u || ms == 4u) && !bbox;
    if (!plain) mode = 0;
    if (mode != 1) { heavy_in = nullptr; heavy_out = nullptr; }   // (the strips are the dense frames')
#define BGS_LAUNCH_RS4(V, X, TR, MS, DP, BB)                                                      \
    hipLaunchKernelGGL((raster_scan_kernel<V, TR, X, MS, DP, BB>), dim3(grid), dim3(256), 0, stream, d_fp, rec, coarse,      \
                       coarse_cap, sup_mul, sup_x, ctl, framebuffer, srgb8_default, out_format, cleanup, (TR) ? tile_trace : (uint4*)nullptr, heavy_in, heavy_out, order, cost_out)
#define BGS_LAUNCH_RS(V, X)                                                                       \
    do {                                                                                          \
        const uint32_t grid = (ntiles + 3u) / 4u + ((X) && heavy_in ? HEAVY_CAP : 0u);            \
        const bool msaa4 = ms == 4u;                                                              \
        if (depth) { if (msaa4) BGS_LAUNCH_RS4(V, X, false, 4, true, false); else BGS_LAUNCH_RS4(V, X, false, 1, true, false); } \
        else if (tile_trace) { if (msaa4) BGS_LAUNCH_RS4(V, X, true, 4, false, false); else BGS_LAUNCH_RS4(V, X, true, 1, false, false); } \
        else { if (msaa4) BGS_LAUNCH_RS4(V, X, false, 4, false, false); else BGS_LAUNCH_RS4(V, X, false, 1, false, false); } \
    } while (0)
    // the rarely used instantiations: V x samples {1, 2, 4, 8} x depth x overlay, minus the plain ones above
#define BGS_LAUNCH_RSX(V)                                                                         \
    do {                                                                                          \
        const uint32_t grid = (ntiles + 3u) / 4u;                                                 \
        if (bbox) {                                                                               \
            if (depth) { if (ms == 1u) BGS_LAUNCH_RS4(V, false, false, 1, true, true); else if (ms == 2u) BGS_LAUNCH_RS4(V, false, false, 2, true, true); \
                         else if (ms == 4u) BGS_LAUNCH_RS4(V, false, false, 4, true, true); else BGS_LAUNCH_RS4(V, false, false, 8, true, true); } \
            else { if (ms == 1u) BGS_LAUNCH_RS4(V, false, false, 1, false, true); else if (ms == 2u) BGS_LAUNCH_RS4(V, false, false, 2, false, true); \
                   else if (ms == 4u) BGS_LAUNCH_RS4(V, false, false, 4, false, true); else BGS_LAUNCH_RS4(V, false, false, 8, false, true); } \
        } else {                                                                                  \
            if (depth) { if (ms == 2u) BGS_LAUNCH_RS4(V, false, false, 2, true, false); else BGS_LAUNCH_RS4(V, false, false, 8, true, false); } \
            else { if (ms == 2u) BGS_LAUNCH_RS4(V, false, false, 2, false, false); else BGS_LAUNCH_RS4(V, false, false, 8, false, false); } \
        }                                                                                         \
    } while (0)
    if (!plain) {
        if (fp.aabb == 0u) BGS_LAUNCH_RSX(RV_OBB);
        else if (fp.gaussian_mode != 0u) BGS_LAUNCH_RSX(RV_AABB3D);
        else BGS_LAUNCH_RSX(RV_SURFEL);
    }
    else if (fp.aabb == 0u) { if (mode == 1) BGS_LAUNCH_RS(RV_OBB, 1); else if (mode == 2) BGS_LAUNCH_RS(RV_OBB, 2); else BGS_LAUNCH_RS(RV_OBB, 0); }
    else if (fp.gaussian_mode != 0u) { if (mode != 0) BGS_LAUNCH_RS(RV_AABB3D, 1); else BGS_LAUNCH_RS(RV_AABB3D, 0); }   // (no interior path: one exit instantiation)
    else BGS_LAUNCH_RS(RV_SURFEL, 0);
#undef BGS_LAUNCH_RSX
#undef BGS_LAUNCH_RS
#undef BGS_LAUNCH_RS4
}

void launch_raster(hipStream_t stream, const FrameParams& fp, const void* records,
                   const uint2* instances, const uint2* ranges, float4* framebuffer,
                   const float clear_color[4], const Control* ctl) {
    const uint32_t ntiles = (uint32_t)(fp.tiles_x * fp.tiles_y);
    if (ntiles == 0) return;
    const float4 clear = make_float4(clear_color[0], clear_color[1], clear_color[2], clear_color[3]);
    const float4* rec = (const float4*)records;
    const bool depth = fp.depth_ptr != 0ull, bbox = fp.visualize_bbox != 0u;
    const uint32_t ms = fp.sample_count;
#define BGS_LAUNCH_R(V, MS, DP, BB)                                                                                     \
    hipLaunchKernelGGL((raster_kernel<V, MS, DP, BB>), dim3(ntiles), dim3(256), 0, stream, fp, rec, instances, ranges, \
                       framebuffer, clear, ctl)
#define BGS_LAUNCH_R3(V, MS)                                                                      \
    do {                                                                                          \
        if (bbox) { if (depth) BGS_LAUNCH_R(V, MS, true, true); else BGS_LAUNCH_R(V, MS, false, true); }   \
        else { if (depth) BGS_LAUNCH_R(V, MS, true, false); else BGS_LAUNCH_R(V, MS, false, false); }      \
    } while (0)
#define BGS_LAUNCH_R2(V)                                                                          \
    do {                                                                                          \
        if (ms == 4u) BGS_LAUNCH_R3(V, 4); else if (ms == 1u) BGS_LAUNCH_R3(V, 1);               \
        else if (ms == 2u) BGS_LAUNCH_R3(V, 2); else BGS_LAUNCH_R3(V, 8);                        \
    } while (0)
    if (fp.aabb == 0u) BGS_LAUNCH_R2(RV_OBB);
    else if (fp.gaussian_mode != 0u) BGS_LAUNCH_R2(RV_AABB3D);
    else BGS_LAUNCH_R2(RV_SURFEL);
#undef BGS_LAUNCH_R2
#undef BGS_LAUNCH_R3
#undef BGS_LAUNCH_R
}

// ---------------------------------------------------------------------------------------
// Rgba8UnormSrgb encode of the f32 target: what the reference's colour attachment stores
// (TextureFormat::Rgba8UnormSrgb, src/render/mod.rs:917-921, examples/headless.rs:120-123).
// Linear RGB -> sRGB OETF -> unorm8 (round to nearest); alpha is linear. 33 MB read, 8 MB write.
// Used for the multi-GPU framebuffer gather (8.3 MB per 1080p frame instead of 33 MB).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void encode_srgb8_kernel(const float4* __restrict__ fb,
                                                           uint32_t* __restrict__ default_out, uint32_t n,
                                                           const FrameParams* __restrict__ fpp, uint32_t out_format) {
    // the frame's own destination (bgs_set_srgb8_target) travels in FrameParams
    uint32_t* __restrict__ out = fpp->srgb8_target ? reinterpret_cast<uint32_t*>(fpp->srgb8_target) : default_out;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 c = fb[i];
        if (out_format & OUT_RGBA16F) reinterpret_cast<uint2*>(out)[i] = pack_rgba16f(c);
        else out[i] = pack_srgb8(c);
    }
}

// ---------------------------------------------------------------------------------------
// upload-time re-layout: planes -> one aligned record per splat (CloudPtrs)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_cloud_kernel(const uint4* __restrict__ pos, const uint4* __restrict__ a, uint32_t v4_a,
                                                         const uint4* __restrict__ b, uint32_t v4_b, const uint4* __restrict__ c,
                                                         uint32_t v4_c, uint4* __restrict__ out, uint32_t stride_v4, uint32_t n) {
    const size_t words = (size_t)n * stride_v4;
    for (size_t g = (size_t)blockIdx.x * 256u + threadIdx.x; g < words; g += (size_t)gridDim.x * 256u) {
        const uint32_t i = (uint32_t)(g / stride_v4), w = (uint32_t)(g - (size_t)i * stride_v4);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (w == 0u) v = pos[i];
        else if (w < 1u + v4_a) v = a[(size_t)i * v4_a + (w - 1u)];
        else if (w < 1u + v4_a + v4_b) v = b[(size_t)i * v4_b + (w - 1u - v4_a)];
        else if (w < 1u + v4_a + v4_b + v4_c) v = c[(size_t)i * v4_c + (w - 1u - v4_a - v4_b)];
        out[g] = v;
    }
}

void launch_pack_cloud(hipStream_t stream, const uint4* pos, const uint4* a, uint32_t v4_a, const uint4* b, uint32_t v4_b,
                       const uint4* c, uint32_t v4_c, uint4* out, uint32_t stride_v4, uint32_t n) {
    if (n == 0) return;
    const size_t words = (size_t)n * stride_v4;
    const uint32_t blocks = (uint32_t)std::min<size_t>((words + 255u) / 256u, 65536u);
    hipLaunchKernelGGL(pack_cloud_kernel, dim3(blocks), dim3(256), 0, stream, pos, a, v4_a, b, v4_b, c, v4_c, out, stride_v4, n);
}

void launch_encode_srgb8(hipStream_t stream, const float4* framebuffer, uint32_t* default_out, uint32_t pixels,
                         const FrameParams* d_fp, uint32_t out_format) {
    if (pixels == 0) return;
    uint32_t blocks = (pixels + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(encode_srgb8_kernel, dim3(blocks), dim3(256), 0, stream, framebuffer, default_out, pixels, d_fp, out_format);
}

}  // namespace bgs
