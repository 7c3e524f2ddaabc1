// bgs.hpp — C++ host side above the C ABI (include/bgs.h): the reference's plugin surface for the
// sort + rasterize path, in the language class of the reference (compiled code; the reference is
// Rust, for which this image has no toolchain). Header-only, C++17, no HIP headers needed: link
// libbgs.so. The Python package bevy_gaussian_splatting_amd mirrors the same surface for the tests.
//
//   reference (Rust)                                   here
//   GaussianSplattingPlugin   (src/lib.rs:40-84)       bgs::GaussianSplattingPlugin
//   CloudSettings + enums     (src/gaussian/settings.rs) bgs::CloudSettings, SortMode, GaussianMode, ...
//   PlanarGaussian3d          (formats/planar_3d.rs)   bgs::PlanarGaussian3d (+ test_model, random)
//   PlanarGaussian3dHandle    (formats/planar_3d.rs)   bgs::PlanarGaussian3dHandle (device-resident cloud)
//   SortEntry                 (src/sort/mod.rs:324)    bgs_sort_entry
//   Camera3d + GaussianCamera (examples/headless.rs)   bgs::View::perspective / View::headless
//
// Every failure of the C ABI becomes a bgs::Error carrying the status and bgs_last_error().
#ifndef BGS_HPP
#define BGS_HPP

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bgs.h"
#include "bgs_diag.h"  // adaptive_counters()

namespace bgs {

class Error : public std::runtime_error {
  public:
    Error(int status, const std::string& what) : std::runtime_error(what), status_(status) {}
    int status() const { return status_; }

  private:
    int status_;
};

// src/sort/mod.rs:46-58
enum class SortMode : uint32_t { None = BGS_SORT_NONE, Radix = BGS_SORT_RADIX, Rayon = BGS_SORT_RAYON, Std = BGS_SORT_STD };
// src/gaussian/settings.rs:17-22
enum class GaussianMode : uint32_t { Gaussian2d = BGS_GAUSSIAN_2D, Gaussian3d = BGS_GAUSSIAN_3D };
// src/gaussian/settings.rs:38-47
enum class RasterizeMode : uint32_t {
    Classification = BGS_RASTERIZE_CLASSIFICATION, Color = BGS_RASTERIZE_COLOR, Depth = BGS_RASTERIZE_DEPTH,
    Normal = BGS_RASTERIZE_NORMAL, OpticalFlow = BGS_RASTERIZE_OPTICAL_FLOW, Position = BGS_RASTERIZE_POSITION,
    Velocity = BGS_RASTERIZE_VELOCITY
};
// src/gaussian/settings.rs:6-12
enum class DrawMode : uint32_t { All = BGS_DRAW_ALL, Selected = BGS_DRAW_SELECTED, HighlightSelected = BGS_DRAW_HIGHLIGHT_SELECTED };
// src/gaussian/settings.rs:79-84
enum class GaussianColorSpace : uint32_t { SrgbRec709Display = BGS_COLOR_SRGB, LinRec709Display = BGS_COLOR_LINEAR };
// src/gaussian/settings.rs:52-77
enum class RadixSortDepthBits : uint32_t { Bits16 = 16, Bits24 = 24, Bits32 = 32 };

using Mat4 = std::array<float, 16>;  // column-major, m[4 * c + r] (glam)

inline Mat4 identity() { return Mat4{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }
inline Mat4 from_translation(float x, float y, float z) {
    Mat4 m = identity();
    m[12] = x; m[13] = y; m[14] = z;
    return m;
}

// CloudSettings::default() (src/gaussian/settings.rs:110-131) + the entity's GlobalTransform and Aabb,
// which extract_gaussians puts into the CloudUniform (src/render/mod.rs:1056-1072).
struct CloudSettings {
    bool aabb = false;
    float global_opacity = 1.0f;
    float global_scale = 1.0f;
    bool opacity_adaptive_radius = true;
    bool visualize_bounding_box = false;   // the quads' frames (src/render/gaussian.wgsl:486-495)
    SortMode sort_mode = SortMode::Radix;
    RadixSortDepthBits radix_sort_depth_bits = RadixSortDepthBits::Bits32;
    DrawMode draw_mode = DrawMode::All;
    GaussianMode gaussian_mode = GaussianMode::Gaussian3d;
    RasterizeMode rasterize_mode = RasterizeMode::Color;
    GaussianColorSpace color_space = GaussianColorSpace::SrgbRec709Display;
    uint32_t num_classes = 1;
    uint32_t sh_degree = 3;  // the reference's compile-time SH_DEGREE (Cargo feature sh3)
    Mat4 transform = identity();
    std::array<float, 3> position_min{0.0f, 0.0f, 0.0f};
    std::array<float, 3> position_max{1.0f, 1.0f, 1.0f};

    bgs_settings to_native() const {
        bgs_settings s;
        bgs_settings_default(&s);
        std::memcpy(s.transform, transform.data(), sizeof s.transform);
        s.global_opacity = global_opacity;
        s.global_scale = global_scale;
        s.gaussian_mode = static_cast<uint32_t>(gaussian_mode);
        s.aabb = aabb ? 1u : 0u;
        s.opacity_adaptive_radius = opacity_adaptive_radius ? 1u : 0u;
        s.color_space = static_cast<uint32_t>(color_space);
        s.radix_depth_bits = static_cast<uint32_t>(radix_sort_depth_bits);
        s.sh_degree = sh_degree;
        s.sort_mode = static_cast<uint32_t>(sort_mode);
        s.rasterize_mode = static_cast<uint32_t>(rasterize_mode);
        s.num_classes = num_classes;
        s.draw_mode = static_cast<uint32_t>(draw_mode);
        s.visualize_bounding_box = visualize_bounding_box ? 1u : 0u;
        for (int i = 0; i < 3; ++i) {
            s.position_min[i] = position_min[i];
            s.position_max[i] = position_max[i];
        }
        s.position_min[3] = s.position_max[3] = 1.0f;  // aabb.min().extend(1.0), src/render/mod.rs:1070
        return s;
    }
};

constexpr int SH_COEFF_COUNT = 48;

// Planar (SoA) cloud: the four planes of `Gaussian3d` (src/gaussian/formats/planar_3d.rs:28-54).
struct PlanarGaussian3d {
    std::vector<std::array<float, 4>> position_visibility;              // x, y, z, visibility
    std::vector<std::array<float, SH_COEFF_COUNT>> spherical_harmonic;  // index 3 * k + channel
    std::vector<std::array<float, 4>> rotation;                         // w, x, y, z
    std::vector<std::array<float, 4>> scale_opacity;                    // sx, sy, sz, opacity

    size_t size() const { return position_visibility.size(); }
    void resize(size_t n) {
        position_visibility.resize(n);
        spherical_harmonic.resize(n);
        rotation.resize(n);
        scale_opacity.resize(n);
    }

    // `PlanarGaussian3d::test_model()` geometry (src/gaussian/formats/planar_3d.rs:193-251): 8 splats at
    // (+-0.5)^3 and a duplicate of the first; identity rotation, scale 0.125, opacity 0.125, random SH.
    static PlanarGaussian3d test_model(uint64_t seed = 0) {
        std::mt19937_64 rng(seed);
        std::uniform_real_distribution<float> sh(-1.0f, 1.0f);
        PlanarGaussian3d c;
        for (float x : {-0.5f, 0.5f})
            for (float y : {-0.5f, 0.5f})
                for (float z : {-0.5f, 0.5f}) {
                    c.position_visibility.push_back({x, y, z, 1.0f});
                    c.rotation.push_back({1.0f, 0.0f, 0.0f, 0.0f});
                    c.scale_opacity.push_back({0.125f, 0.125f, 0.125f, 0.125f});
                    std::array<float, SH_COEFF_COUNT> coeff;
                    for (auto& v : coeff) v = sh(rng);
                    c.spherical_harmonic.push_back(coeff);
                }
        c.position_visibility.push_back(c.position_visibility[0]);
        c.rotation.push_back(c.rotation[0]);
        c.scale_opacity.push_back(c.scale_opacity[0]);
        c.spherical_harmonic.push_back(c.spherical_harmonic[0]);
        return c;
    }

    // random_gaussians_3d_seeded (src/gaussian/formats/planar_3d.rs:120-168,182-191): the reference's
    // distributions — rotation ~ U(-1,1)^4 (not normalised), position ~ U(-20,20)^3 with visibility 1,
    // scale ~ U(0,1)^3, opacity ~ U(0,0.8), SH ~ U(-1,1)^48. The reference draws from rand::StdRng; this
    // is std::mt19937_64, so clouds have the same statistics, not the same bits.
    static PlanarGaussian3d random(size_t n, uint64_t seed) {
        std::mt19937_64 rng(seed);
        auto uni = [&](float lo, float hi) { return std::uniform_real_distribution<float>(lo, hi)(rng); };
        PlanarGaussian3d c;
        c.resize(n);
        for (size_t i = 0; i < n; ++i) {
            for (auto& v : c.rotation[i]) v = uni(-1.0f, 1.0f);
            c.position_visibility[i] = {uni(-20.0f, 20.0f), uni(-20.0f, 20.0f), uni(-20.0f, 20.0f), 1.0f};
            c.scale_opacity[i] = {uni(0.0f, 1.0f), uni(0.0f, 1.0f), uni(0.0f, 1.0f), uni(0.0f, 0.8f)};
            for (auto& v : c.spherical_harmonic[i]) v = uni(-1.0f, 1.0f);
        }
        return c;
    }

    // A synthetic cloud with the STATISTICS of a trained 3DGS asset (round 6; the Python twin and the reasoning:
    // bevy_gaussian_splatting_amd/gaussian.py trained_like_gaussians_3d_seeded): positions on `patches` rectangular
    // surfaces inside the (-20, 20)^3 box with N(0, 0.02) of noise along the normal, log-normal scales around the splat
    // spacing with a flat normal axis, unit quaternions aligned with the patch, bimodal opacity (60 % Beta(8, 1.2), 40 %
    // Beta(1.2, 6)), DC-dominated SH with colours in [0.05, 0.95]. Same per-field order as the Python generator;
    // std::mt19937_64 here, numpy PCG64 there: the same statistics, not the same bits. For `global_scale = 1`.
    static PlanarGaussian3d trained_like(size_t n, uint64_t seed, size_t patches = 96) {
        std::mt19937_64 rng(seed);
        auto uni = [&](double lo, double hi) { return std::uniform_real_distribution<double>(lo, hi)(rng); };
        auto nrm = [&](double mu, double sd) { return std::normal_distribution<double>(mu, sd)(rng); };
        auto beta = [&](double a, double b) {
            const double x = std::gamma_distribution<double>(a, 1.0)(rng), y = std::gamma_distribution<double>(b, 1.0)(rng);
            return x / (x + y);
        };
        struct Patch { double c[3], n[3], tu[3], tv[3], su, sv, rgb[3]; };
        std::vector<Patch> P(patches);
        std::vector<double> cdf(patches);
        double area_sum = 0.0;
        auto cross = [](const double* a, const double* b, double* o) {
            o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
        };
        auto normalise = [](double* v) { const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); for (int k = 0; k < 3; ++k) v[k] /= l; };
        for (auto& p : P) for (double& v : p.c) v = uni(-18.0, 18.0);
        for (auto& p : P) {
            for (double& v : p.n) v = nrm(0.0, 1.0);
            normalise(p.n);
            const double helper[3] = {std::fabs(p.n[0]) < 0.9 ? 1.0 : 0.0, std::fabs(p.n[0]) < 0.9 ? 0.0 : 1.0, 0.0};
            cross(p.n, helper, p.tu);
            normalise(p.tu);
            cross(p.n, p.tu, p.tv);
        }
        for (auto& p : P) { p.su = uni(2.0, 12.0); p.sv = uni(2.0, 12.0); area_sum += p.su * p.sv; }
        for (auto& p : P) for (double& v : p.rgb) v = uni(0.15, 0.85);
        { double acc = 0.0; for (size_t i = 0; i < patches; ++i) { acc += P[i].su * P[i].sv / area_sum; cdf[i] = acc; } }
        const double spacing = std::sqrt(area_sum / (double)std::max<size_t>(n, 1));
        PlanarGaussian3d c;
        c.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const size_t pid = std::min<size_t>(std::lower_bound(cdf.begin(), cdf.end(), uni(0.0, 1.0)) - cdf.begin(), patches - 1);
            const Patch& p = P[pid];
            const double u = uni(-0.5, 0.5) * p.su, v = uni(-0.5, 0.5) * p.sv, off = nrm(0.0, 0.02);
            for (int k = 0; k < 3; ++k) c.position_visibility[i][k] = (float)(p.c[k] + u * p.tu[k] + v * p.tv[k] + off * p.n[k]);
            c.position_visibility[i][3] = 1.0f;
            const double t0 = spacing * std::exp(nrm(0.0, 0.6)), t1 = spacing * std::exp(nrm(0.0, 0.6));
            const double flat = uni(0.05, 0.25) * std::sqrt(t0 * t1);
            const double op = uni(0.0, 1.0) < 0.6 ? beta(8.0, 1.2) : beta(1.2, 6.0);
            c.scale_opacity[i] = {(float)t0, (float)t1, (float)flat, (float)op};
            // local axes (tu turned about the normal, the normal): the quaternion whose rotation maps e_i onto them
            const double ang = uni(0.0, 6.283185307179586), ca = std::cos(ang), sa = std::sin(ang);
            double m[3][3];   // columns = ax, ay, az
            for (int k = 0; k < 3; ++k) { m[k][0] = ca * p.tu[k] + sa * p.tv[k]; m[k][1] = -sa * p.tu[k] + ca * p.tv[k]; m[k][2] = p.n[k]; }
            const double tr = m[0][0] + m[1][1] + m[2][2];
            double q[4] = {std::sqrt(std::max(1.0 + tr, 1e-12)) / 2.0,
                           std::copysign(std::sqrt(std::max(1.0 + m[0][0] - m[1][1] - m[2][2], 0.0)) / 2.0, m[2][1] - m[1][2]),
                           std::copysign(std::sqrt(std::max(1.0 - m[0][0] + m[1][1] - m[2][2], 0.0)) / 2.0, m[0][2] - m[2][0]),
                           std::copysign(std::sqrt(std::max(1.0 - m[0][0] - m[1][1] + m[2][2], 0.0)) / 2.0, m[1][0] - m[0][1])};
            const double ql = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            c.rotation[i] = {(float)(q[0] / ql), (float)(q[1] / ql), (float)(q[2] / ql), (float)(q[3] / ql)};
            auto& sh = c.spherical_harmonic[i];
            for (int ch = 0; ch < 3; ++ch) {
                const double rgb = std::min(std::max(p.rgb[ch] + nrm(0.0, 0.08), 0.05), 0.95);
                sh[ch] = (float)((rgb - 0.5) / 0.2820947917738781);
            }
            for (int k = 1; k < 16; ++k) {
                const double sd = k < 4 ? 0.015 : (k < 9 ? 0.0075 : 0.004);
                for (int ch = 0; ch < 3; ++ch) sh[3 * k + ch] = (float)nrm(0.0, sd);
            }
        }
        return c;
    }
};

// The View uniform the path reads, built the way Bevy builds it for `Camera3d::default()`.
struct View {
    bgs_view native{};
    uint32_t width = 0, height = 0;

    // Camera world transform -> infinite reverse-Z perspective (Bevy PerspectiveProjection defaults:
    // fov_y = pi/4, near = 0.1), clear colour opaque black.
    static View perspective(const Mat4& world_from_view, uint32_t width, uint32_t height,
                            float fov_y = 0.78539816339744830962f, float near_plane = 0.1f) {
        View v;
        v.width = width;
        v.height = height;
        bgs_view_perspective(world_from_view.data(), fov_y, near_plane, width, height, &v.native);
        return v;
    }
    // examples/headless.rs:178-183: Camera3d::default() at (0, 1.5, 5) looking down -Z
    static View headless(uint32_t width = 1920, uint32_t height = 1080) {
        return perspective(from_translation(0.0f, 1.5f, 5.0f), width, height);
    }
    void set_clear_color(float r, float g, float b, float a) {
        native.clear_color[0] = r; native.clear_color[1] = g; native.clear_color[2] = b; native.clear_color[3] = a;
    }
    // The camera's `Msaa` component (CloudPipelineKey.sample_count = msaa.samples(), src/render/mod.rs:357-424): 1 =
    // Msaa::Off, 2 / 4 / 8 = Msaa::Sample2 / Sample4 / Sample8; 4 is Bevy's default, which bgs_view_perspective has already set.
    void set_msaa_samples(uint32_t samples) { native.sample_count = samples; }
    // The view's depth attachment as device memory ([y][x][sample] floats, reverse-Z; src/render/mod.rs:959-974), 0 = none.
    void set_depth(const void* device_ptr) { native.depth_device_ptr = (uint64_t)(uintptr_t)device_ptr; }
};

class GaussianSplattingPlugin;

// A cloud resident in HBM (the RenderAsset side of PlanarGaussian3dHandle). Move-only.
class PlanarGaussian3dHandle {
  public:
    PlanarGaussian3dHandle() = default;
    PlanarGaussian3dHandle(PlanarGaussian3dHandle&& o) noexcept : ctx_(o.ctx_), cloud_(o.cloud_) { o.cloud_ = nullptr; }
    PlanarGaussian3dHandle& operator=(PlanarGaussian3dHandle&& o) noexcept {
        if (this != &o) {
            reset();
            ctx_ = o.ctx_;
            cloud_ = o.cloud_;
            o.cloud_ = nullptr;
        }
        return *this;
    }
    PlanarGaussian3dHandle(const PlanarGaussian3dHandle&) = delete;
    PlanarGaussian3dHandle& operator=(const PlanarGaussian3dHandle&) = delete;
    ~PlanarGaussian3dHandle() { reset(); }
    void reset() {
        if (cloud_) bgs_cloud_free(ctx_, cloud_);
        cloud_ = nullptr;
    }
    uint32_t size() const { return cloud_ ? bgs_cloud_len(cloud_) : 0; }
    const bgs_cloud* get() const { return cloud_; }

  private:
    friend class GaussianSplattingPlugin;
    PlanarGaussian3dHandle(bgs_ctx* ctx, bgs_cloud* cloud) : ctx_(ctx), cloud_(cloud) {}
    bgs_ctx* ctx_ = nullptr;
    bgs_cloud* cloud_ = nullptr;
};

// One context on one HIP device: upload, sort, render, frame pipelining. Not re-entrant (one render
// thread, like Bevy's); one plugin per GPU for multi-GPU.
class GaussianSplattingPlugin {
  public:
    explicit GaussianSplattingPlugin(int hip_device = 0) {
        // the binding's handshake: this header's structs are the library's (a stale libbgs.so is refused, not misread)
        if (bgs_abi_check(BGS_ABI_VERSION, sizeof(bgs_view), sizeof(bgs_settings), sizeof(bgs_stats)) != BGS_OK)
            throw Error(BGS_EINVAL, std::string("bgs_abi_check: ") + bgs_last_error(nullptr));
        const int rc = bgs_create(hip_device, &ctx_);
        if (rc != BGS_OK) throw Error(rc, std::string("bgs_create: ") + bgs_last_error(nullptr));
    }
    ~GaussianSplattingPlugin() { bgs_destroy(ctx_); }
    GaussianSplattingPlugin(const GaussianSplattingPlugin&) = delete;
    GaussianSplattingPlugin& operator=(const GaussianSplattingPlugin&) = delete;

    PlanarGaussian3dHandle upload(const PlanarGaussian3d& c) {
        if (c.spherical_harmonic.size() != c.size() || c.rotation.size() != c.size() || c.scale_opacity.size() != c.size())
            throw Error(BGS_EINVAL, "PlanarGaussian3d planes differ in length");
        bgs_cloud* cloud = nullptr;
        check(bgs_cloud_upload_f32(ctx_, static_cast<uint32_t>(c.size()),
                                   c.size() ? c.position_visibility[0].data() : nullptr,
                                   c.size() ? c.spherical_harmonic[0].data() : nullptr,
                                   c.size() ? c.rotation[0].data() : nullptr,
                                   c.size() ? c.scale_opacity[0].data() : nullptr, &cloud),
              "bgs_cloud_upload_f32");
        return PlanarGaussian3dHandle(ctx_, cloud);
    }

    // run_radix_sort / rayon_sort for one view: the full sorted entry list (culled entries last)
    std::vector<bgs_sort_entry> sort(const PlanarGaussian3dHandle& h, const View& v, const CloudSettings& s) {
        std::vector<bgs_sort_entry> out(h.size());
        const bgs_settings ns = s.to_native();
        check(bgs_sort(ctx_, h.get(), &v.native, &ns, out.data()), "bgs_sort");
        return out;
    }

    // Sort + project + bin + rasterize one view. `rgba_out` (optional): width * height * 4 floats,
    // premultiplied linear RGBA, row 0 = top. With set_async(true) and rgba_out == nullptr the frame is
    // only enqueued on the next lane.
    void render(const PlanarGaussian3dHandle& h, const View& v, const CloudSettings& s, std::vector<float>* rgba_out = nullptr) {
        const bgs_settings ns = s.to_native();
        render(h, v, ns, rgba_out);
    }
    void render(const PlanarGaussian3dHandle& h, const View& v, const bgs_settings& ns, std::vector<float>* rgba_out = nullptr) {
        if (rgba_out) rgba_out->resize(static_cast<size_t>(v.width) * v.height * 4);
        check(bgs_render(ctx_, h.get(), &v.native, &ns, rgba_out ? rgba_out->data() : nullptr), "bgs_render");
    }

    // Frames in flight: lanes = buffer sets, multiplexed onto `streams` HIP streams.
    void set_async(bool on) { check(bgs_set_async(ctx_, on ? 1 : 0), "bgs_set_async"); }
    void set_pipeline_depth(uint32_t lanes) { check(bgs_set_pipeline_depth(ctx_, lanes), "bgs_set_pipeline_depth"); }
    void set_pipeline_streams(uint32_t streams) { check(bgs_set_pipeline_streams(ctx_, streams), "bgs_set_pipeline_streams"); }
    void set_graphs(bool on) { check(bgs_set_graphs(ctx_, on ? 1 : 0), "bgs_set_graphs"); }
    void set_output_srgb8(bool on) { check(bgs_set_output_srgb8(ctx_, on ? 1 : 0), "bgs_set_output_srgb8"); }
    // hdr cameras: the colour attachment is Rgba16Float (src/render/mod.rs:917-921); exclusive with sRGB8
    void set_output_rgba16f(bool on) { check(bgs_set_output_rgba16f(ctx_, on ? 1 : 0), "bgs_set_output_rgba16f"); }
    // frames that write a packed image skip the f32 target
    void set_packed_only(bool on) { check(bgs_set_packed_only(ctx_, on ? 1 : 0), "bgs_set_packed_only"); }
    // forget what completed frames taught the context (grid sizes, splitters, list capacity, supertile level)
    void reset_adaptive_state() { check(bgs_reset_adaptive_state(ctx_), "bgs_reset_adaptive_state"); }
    // cumulative counters of the adaptive machinery (bgs_adaptive_counters: bucket / onesweep frames, re-runs by
    // cause, supertile level changes, current level, list-capacity hint)
    std::array<uint64_t, 8> adaptive_counters() {
        std::array<uint64_t, 8> out{};
        check(bgs_adaptive_counters(ctx_, out.data()), "bgs_adaptive_counters");
        return out;
    }
    // async frames completed inside their render call (a kind of frame new to the context), kinds settled on
    std::pair<uint64_t, uint64_t> learning_counters() {
        uint64_t early = 0, kinds = 0;
        check(bgs_learning_counters(ctx_, &early, &kinds), "bgs_learning_counters");
        return {early, kinds};
    }
    // ---- the multi-GPU frame gather (bgs_comm_*: RCCL's ncclGather behind the C ABI; one process per GPU) ----
    static std::array<uint8_t, BGS_COMM_ID_BYTES> comm_unique_id() {   // rank 0; ship the 128 bytes to every rank
        std::array<uint8_t, BGS_COMM_ID_BYTES> id{};
        const int rc = bgs_comm_unique_id(id.data());
        if (rc != BGS_OK) throw Error(rc, std::string("bgs_comm_unique_id: ") + bgs_last_error(nullptr));
        return id;
    }
    bgs_comm* comm_create(const std::array<uint8_t, BGS_COMM_ID_BYTES>& id, uint32_t world_size, uint32_t rank) {   // collective
        bgs_comm* c = nullptr;
        check(bgs_comm_create(ctx_, id.data(), world_size, rank, &c), "bgs_comm_create");
        return c;
    }
    // enqueue one gather of `bytes` per rank (device memory of COMPLETED frames; recv on the root only); returns its ticket
    uint64_t comm_gather(bgs_comm* c, uint32_t root, const void* send, uint64_t bytes, void* recv) {
        uint64_t ticket = 0;
        check(bgs_comm_gather(ctx_, c, root, send, bytes, recv, &ticket), "bgs_comm_gather");
        return ticket;
    }
    void comm_wait(bgs_comm* c, uint64_t ticket = 0) { check(bgs_comm_wait(ctx_, c, ticket), "bgs_comm_wait"); }
    void comm_destroy(bgs_comm* c) { bgs_comm_destroy(ctx_, c); }
    void* device_alloc(uint64_t bytes) {
        void* p = nullptr;
        check(bgs_device_alloc(ctx_, bytes, &p), "bgs_device_alloc");
        return p;
    }
    void device_free(void* p) { check(bgs_device_free(ctx_, p), "bgs_device_free"); }
    void set_srgb8_target(void* device_ptr) { check(bgs_set_srgb8_target(ctx_, device_ptr), "bgs_set_srgb8_target"); }
    void set_profiling(int level) { check(bgs_set_profiling(ctx_, level), "bgs_set_profiling"); }
    void synchronize() { check(bgs_synchronize(ctx_), "bgs_synchronize"); }
    uint32_t frames_in_flight() {
        uint32_t n = 0;
        check(bgs_frames_in_flight(ctx_, &n), "bgs_frames_in_flight");
        return n;
    }
    // Completes the oldest frame in flight; device pointers of its f32 and (if enabled) sRGB8 images.
    std::pair<void*, void*> pipeline_pop() {
        void *f32 = nullptr, *rgba8 = nullptr;
        check(bgs_pipeline_pop(ctx_, &f32, &rgba8), "bgs_pipeline_pop");
        return {f32, rgba8};
    }
    // The most recent frame's Rgba8UnormSrgb image (the reference's render-target format) on the host.
    std::vector<uint8_t> download_srgb8(const View& v) {
        void* dptr = nullptr;
        uint64_t bytes = 0;
        check(bgs_synchronize(ctx_), "bgs_synchronize");
        check(bgs_framebuffer_srgb8_device_ptr(ctx_, &dptr, &bytes), "bgs_framebuffer_srgb8_device_ptr");
        std::vector<uint8_t> out(static_cast<size_t>(v.width) * v.height * 4);
        if (bytes < out.size()) throw Error(BGS_EINVAL, "sRGB8 image smaller than the view");
        check(bgs_download(ctx_, dptr, out.data(), out.size()), "bgs_download");
        return out;
    }
    // The most recent frame's Rgba16Float image on the host (IEEE binary16 bit patterns, 4 per pixel).
    std::vector<uint16_t> download_rgba16f(const View& v) {
        void* dptr = nullptr;
        uint64_t bytes = 0;
        check(bgs_synchronize(ctx_), "bgs_synchronize");
        check(bgs_framebuffer_rgba16f_device_ptr(ctx_, &dptr, &bytes), "bgs_framebuffer_rgba16f_device_ptr");
        std::vector<uint16_t> out(static_cast<size_t>(v.width) * v.height * 4);
        if (bytes < out.size() * 2) throw Error(BGS_EINVAL, "Rgba16Float image smaller than the view");
        check(bgs_download(ctx_, dptr, out.data(), out.size() * 2), "bgs_download");
        return out;
    }
    void download(const void* device_ptr, void* host_out, uint64_t bytes) {
        check(bgs_download(ctx_, device_ptr, host_out, bytes), "bgs_download");
    }
    bgs_stats stats() {
        bgs_stats st;
        check(bgs_get_stats(ctx_, &st), "bgs_get_stats");
        return st;
    }
    bgs_ctx* native() { return ctx_; }
    // wrap a cloud uploaded through the C ABI directly (e.g. bgs_cloud_upload_f16) into an owning handle
    PlanarGaussian3dHandle adopt(bgs_cloud* cloud) { return PlanarGaussian3dHandle(ctx_, cloud); }

  private:
    void check(int rc, const char* what) {
        if (rc != BGS_OK) throw Error(rc, std::string(what) + ": " + bgs_last_error(ctx_));
    }
    bgs_ctx* ctx_ = nullptr;
};

}  // namespace bgs
#endif  // BGS_HPP
