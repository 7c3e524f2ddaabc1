// device_math_shim.cpp — TEST-ONLY host build of the product's device math header.
//
// Compiles bevy_gaussian_splatting_amd/csrc/splat_math.h with g++ (BGS_HD expands to nothing)
// so tests/test_device_math_host.py can check the per-splat arithmetic the HIP kernels run
// against the oracle WITHOUT a GPU. This is a pre-flight check, not a product path: libbgs
// never runs these functions on the host and has no CPU fallback.
#include "../../bevy_gaussian_splatting_amd/csrc/frame_params.h"
#include "../../bevy_gaussian_splatting_amd/csrc/splat_math.h"

using namespace bgs;

struct ShimOut {
    int32_t visible, draw;
    float color[4];
    float cx, cy;
    float p[5];
    float radius;
    int32_t tx0, ty0, tx1, ty1;
    float mean[2];
    float T[9];
    float quad_m[4];
    float bounds[4];  // minx maxx miny maxy
    float ndc_z;      // the quad's depth (position.z / position.w)
};

struct ShFloat {
    const float* base;
    void load_all(float* c) const { memcpy(c, base, 48 * sizeof(float)); }
};

extern "C" {

void shim_fill_params(uint32_t n, const bgs_view* view, const bgs_settings* s, FrameParams* fp) {
    fill_frame_params(n, view, s, *fp);
}

// csrc/exact_log.h: the correctly rounded ln of the adaptive cutoff, host build (the device runs the same operations)
// the rasteriser's work order with S runs per XCD, for every workgroup of a grid of n
void shim_xcd_runs_items(uint32_t n, uint32_t S, uint32_t* out) {
    for (uint32_t b = 0; b < n; ++b) out[b] = xcd_runs_item(b, n, S);
}

void shim_ln_f32(const float* x, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = ln_f32_cr(x[i]);
}

uint64_t shim_ln_f32_checksum(uint32_t first_bits, uint32_t count) {
    uint64_t sum = 0;
#pragma omp parallel for schedule(static) reduction(+ : sum)
    for (int64_t i = 0; i < (int64_t)count; ++i) {
        const uint32_t in_bits = first_bits + (uint32_t)i;
        float x; memcpy(&x, &in_bits, 4);
        sum += ln_selftest_mix(in_bits, f2u(ln_f32_cr(x)));
    }
    return sum;
}

uint32_t shim_frame_params_size(void) { return (uint32_t)sizeof(FrameParams); }

// supertile index of a tile for a supertile edge (bgs_device.h: reciprocal multiply)
uint32_t shim_supertile_div(uint32_t tile, uint32_t edge) { return supertile_div(tile, supertile_mul(edge)); }

// host-side adaptive policies (frame_params.h)
uint32_t shim_next_supertile_level(double ratio, uint32_t lv, const uint32_t* edges, double* longer_out) {
    return next_supertile_level(ratio, lv, edges, longer_out);
}
uint32_t shim_pow2_ceil(uint64_t v) { return pow2_ceil_u32(v); }
int shim_splitters_ascending(const uint32_t* key, uint32_t count) { return splitters_ascending(key, count) ? 1 : 0; }

// keys exactly as keygen_kernel stores them (before any final-pass un-inversion)
void shim_sort_keys(const FrameParams* fp, const float* pos_vis, uint32_t n, uint32_t* keys_out) {
    for (uint32_t i = 0; i < n; ++i)
        keys_out[i] = sort_key(*fp, V3{pos_vis[4 * i], pos_vis[4 * i + 1], pos_vis[4 * i + 2]});
}

// the same keys the way the chainless keygen tiles compute them: the straight-line verdict first (sort_key_fast), the
// reference's divisions only for the splats it is unsure about. `unsure_out` counts those.
void shim_sort_keys_two_step(const FrameParams* fp, const float* pos_vis, uint32_t n, uint32_t* keys_out, uint32_t* unsure_out) {
    uint32_t unsure_count = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const V3 p{pos_vis[4 * i], pos_vis[4 * i + 1], pos_vis[4 * i + 2]};
        bool unsure = false;
        uint32_t k;
        if (fp->sort_mode == SORT_NONE) k = sort_key_fast<0>(*fp, p, unsure);
        else if (fp->sort_mode != SORT_RADIX) k = sort_key_fast<2>(*fp, p, unsure);
        else k = sort_key_fast<1>(*fp, p, unsure);
        if (unsure) { k = sort_key(*fp, p); ++unsure_count; }
        keys_out[i] = k;
    }
    *unsure_out = unsure_count;
}

// pos = position_visibility row (4 floats); depth_range = {min_distance, max_distance}
void shim_project(const FrameParams* fp, uint32_t key, const float* pos, const float* rot,
                  const float* so, const float* sh48, const float* depth_range, ShimOut* out) {
    Projected pr;
    memset(&pr, 0, sizeof pr);
    ColorInputs ci{pos[3], depth_range[0], depth_range[1]};
    // the same dispatch as the launchers: the Color-only instantiation unless another mode is asked for
    if (fp->rasterize_mode == RASTERIZE_COLOR && fp->draw_mode == 0u)
        project_splat<false>(*fp, key, V3{pos[0], pos[1], pos[2]}, rot, so, ShFloat{sh48}, ci, pr);
    else
        project_splat<true>(*fp, key, V3{pos[0], pos[1], pos[2]}, rot, so, ShFloat{sh48}, ci, pr);
    memset(out, 0, sizeof *out);
    out->visible = pr.visible;
    out->draw = pr.draw;
    if (!pr.draw) return;
    memcpy(out->color, pr.color, sizeof out->color);
    out->cx = pr.quad.cx;
    out->cy = pr.quad.cy;
    memcpy(out->p, pr.p, sizeof out->p);
    out->radius = pr.radius;
    out->tx0 = pr.tx0; out->ty0 = pr.ty0; out->tx1 = pr.tx1; out->ty1 = pr.ty1;
    out->mean[0] = pr.surfel.mean_x;
    out->mean[1] = pr.surfel.mean_y;
    memcpy(out->T, pr.surfel.T, sizeof out->T);
    out->quad_m[0] = pr.quad.m00; out->quad_m[1] = pr.quad.m01;
    out->quad_m[2] = pr.quad.m10; out->quad_m[3] = pr.quad.m11;
    out->bounds[0] = pr.quad.minx; out->bounds[1] = pr.quad.maxx;
    out->bounds[2] = pr.quad.miny; out->bounds[3] = pr.quad.maxy;
    out->ndc_z = pr.ndc_z;
}

float shim_distance_to_camera(const FrameParams* fp, const float* pos) {
    return distance_to_camera(*fp, V3{pos[0], pos[1], pos[2]});
}

}  // extern "C"
