// bgs_host.hpp — the callers and data formats either side of the hot path (SURVEY section 8 f), in C++17 above
// include/bgs.hpp: the f16 planar cloud (src/gaussian/f16.rs), the sort trigger / throttle policy
// (src/sort/mod.rs:76-86,143-194, src/sort/rayon.rs:124-129), the multi-camera SortedEntries asset
// (src/sort/mod.rs:331-393), compute_aabb (src/gaussian/interface.rs:22-63) and the INRIA `.ply` loader
// (src/io/ply.rs:23-132, with the reference's quirks), the `.gcloud` container reader and the loader's dispatch on
// the file extension (src/io/gcloud/flexbuffers.rs:9-22, src/io/loader.rs:22-61), the precomputed-covariance
// plane (src/gaussian/covariance.rs:4-41, src/gaussian/f32.rs:218-251). Header-only; tests/test_cpp_host.py checks each
// against the Python mirror, which is pinned by known answers.
#ifndef BGS_HOST_HPP
#define BGS_HOST_HPP

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <istream>
#include <iterator>
#include <map>
#include <optional>
#include <sstream>

#include "bgs.hpp"

namespace bgs {

// ---- f16 storage ------------------------------------------------------------------------------
// IEEE round-to-nearest-even f32 -> f16 (what `half::f16::from_f32` does, src/gaussian/f16.rs:244-252)
inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t mag = x & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return (uint16_t)(sign | (mag > 0x7F800000u ? 0x7E00u | ((mag >> 13) & 0x3FFu) : 0x7C00u));
    if (mag >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // rounds to infinity
    if (mag < 0x33000001u) return (uint16_t)sign;               // rounds to zero (<= 2^-25)
    int32_t exp = (int32_t)(mag >> 23) - 127;
    uint32_t man = (mag & 0x7FFFFFu) | 0x800000u;
    uint32_t shift, half_bits;
    if (exp < -14) {  // subnormal half
        shift = (uint32_t)(13 + (-14 - exp));
        half_bits = 0;
    } else {
        shift = 13;
        half_bits = (uint32_t)(exp + 15) << 10;
        man &= 0x7FFFFFu;
    }
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
    return (uint16_t)(sign | (half_bits + q));  // a mantissa carry moves into the exponent, as it must
}
inline uint32_t pack_f32s_to_u32(float upper, float lower) {
    return ((uint32_t)f32_to_f16(upper) << 16) | (uint32_t)f32_to_f16(lower);
}

// position_visibility stays f32; spherical_harmonic[n][24] u32 with the EVEN coefficient in the low
// half (src/render/planar.wgsl:117-130); rotation_scale_opacity[n][4] u32 = [rot0|rot1], [rot2|rot3],
// [s0|s1], [s2|opacity], first value in the high half (src/gaussian/f16.rs:29-55).
struct PlanarGaussian3dF16 {
    std::vector<std::array<float, 4>> position_visibility;
    std::vector<std::array<uint32_t, SH_COEFF_COUNT / 2>> spherical_harmonic;
    std::vector<std::array<uint32_t, 4>> rotation_scale_opacity;
    size_t size() const { return position_visibility.size(); }

    static PlanarGaussian3dF16 from_f32(const PlanarGaussian3d& c) {
        PlanarGaussian3dF16 o;
        const size_t n = c.size();
        o.position_visibility = c.position_visibility;
        o.spherical_harmonic.resize(n);
        o.rotation_scale_opacity.resize(n);
        for (size_t i = 0; i < n; ++i) {
            for (int k = 0; k < SH_COEFF_COUNT / 2; ++k)
                o.spherical_harmonic[i][k] = pack_f32s_to_u32(c.spherical_harmonic[i][2 * k + 1], c.spherical_harmonic[i][2 * k]);
            const auto& r = c.rotation[i];
            const auto& s = c.scale_opacity[i];
            o.rotation_scale_opacity[i] = {pack_f32s_to_u32(r[0], r[1]), pack_f32s_to_u32(r[2], r[3]),
                                           pack_f32s_to_u32(s[0], s[1]), pack_f32s_to_u32(s[2], s[3])};
        }
        return o;
    }
};

inline PlanarGaussian3dHandle upload(GaussianSplattingPlugin& plugin, const PlanarGaussian3dF16& c);

// ---- when to re-sort ----------------------------------------------------------------------------
struct SortConfig {  // src/sort/mod.rs:76-86
    int64_t period_ms = 1000;
};
struct SortTrigger {  // src/sort/mod.rs:143-150, one per GaussianCamera
    size_t camera_index = 0;
    bool needs_sort = false;
    std::array<float, 3> last_camera_position{0.0f, 0.0f, 0.0f};
    std::optional<double> last_sort_time;  // seconds
};
// src/sort/mod.rs:164-193 for one camera; `now` in seconds (injected so the policy is testable)
inline void update_sort_trigger(SortTrigger& t, const std::array<float, 3>& camera_position, int64_t camera_order,
                                const SortConfig& config, double now) {
    if (!t.last_sort_time) {
        if (camera_order < 0) throw Error(BGS_EINVAL, "camera order must be a non-negative index into gaussian cameras");
        t.camera_index = (size_t)camera_order;
        t.needs_sort = true;
        t.last_sort_time = now;
        return;
    }
    if ((now - *t.last_sort_time) * 1000.0 < (double)config.period_ms) return;
    if (t.last_camera_position != camera_position) {
        t.needs_sort = true;
        t.last_sort_time = now;
        t.last_camera_position = camera_position;
    }
}
// src/sort/rayon.rs:124-129: the CPU sorts stretch the period to at least 4x the measured sort time
inline void after_cpu_sort(SortConfig& config, double sort_duration_s) {
    config.period_ms = std::max({config.period_ms, config.period_ms * 4 / 5, (int64_t)4 * (int64_t)(sort_duration_s * 1000.0)});
}

// ---- multi-camera sorted-entry asset ---------------------------------------------------------------
// src/sort/mod.rs:331-393: camera_count * entry_count (key, index) pairs, created with
// entry_count = len_sqrt_ceil(cloud)^2 (:259-262), every chunk initialised to key 1 / identity order
// (:347-354). Camera c owns sorted[c * gaussians, (c + 1) * gaussians) with gaussians = cloud.len() — the
// stride is the CLOUD length both where the sorts write (src/sort/rayon.rs:82-84) and where the draw binds
// (src/render/mod.rs:1548-1554); the square-padding tail is unused.
struct SortedEntries {
    size_t camera_count = 0, entry_count = 0;
    std::vector<bgs_sort_entry> sorted;

    static SortedEntries create(size_t camera_count, size_t entry_count) {
        SortedEntries s;
        s.camera_count = camera_count;
        s.entry_count = entry_count;
        s.sorted.resize(camera_count * entry_count);
        for (size_t c = 0; c < camera_count; ++c)
            for (size_t i = 0; i < entry_count; ++i) s.sorted[c * entry_count + i] = bgs_sort_entry{1u, (uint32_t)i};
        return s;
    }
    static SortedEntries for_cloud(size_t camera_count, size_t cloud_len) {  // auto_insert_sorted_entries, :218-268
        const size_t side = (size_t)std::ceil(std::sqrt((float)cloud_len));
        return create(camera_count, side * side);
    }
    // [begin, end) of camera `camera_index`'s chunk; throws where the reference's `.nth().unwrap()` panics
    std::pair<bgs_sort_entry*, bgs_sort_entry*> chunk(size_t camera_index, size_t gaussians) {
        if ((camera_index + 1) * gaussians > sorted.size()) throw Error(BGS_EINVAL, "camera chunk out of range");
        return {sorted.data() + camera_index * gaussians, sorted.data() + (camera_index + 1) * gaussians};
    }
    // update_sorted_entries_sizes (:270-296): a camera-count change re-creates the asset
    void resize_cameras(size_t cameras) {
        if (cameras != camera_count) *this = create(cameras, entry_count);
    }
};

// Every camera gets its own order (the reference's radix path only ever fills chunk 0, src/sort/mod.rs:427).
inline void sort_cameras(GaussianSplattingPlugin& plugin, const PlanarGaussian3dHandle& h, const std::vector<View>& cameras,
                         const CloudSettings& s, SortedEntries& out) {
    out.resize_cameras(cameras.size());
    const bgs_settings ns = s.to_native();
    for (size_t c = 0; c < cameras.size(); ++c) {
        auto range = out.chunk(c, h.size());
        const int rc = bgs_sort(plugin.native(), h.get(), &cameras[c].native, &ns, range.first);
        if (rc != BGS_OK) throw Error(rc, std::string("bgs_sort: ") + bgs_last_error(plugin.native()));
    }
}

// (min, max) the reference hands to the shaders for a cloud: compute_aabb (src/gaussian/interface.rs:22-63,
// position -/+ 0.1 per splat) -> Bevy Aabb {center, half_extents} (src/gaussian/cloud.rs:56-59) ->
// aabb.min() / max() = center -/+ half_extents (src/render/mod.rs:1070-1071), all in f32.
inline bool compute_aabb(const PlanarGaussian3d& c, std::array<float, 3>& mn_out, std::array<float, 3>& mx_out) {
    if (c.size() == 0) return false;
    std::array<float, 3> mn{INFINITY, INFINITY, INFINITY}, mx{-INFINITY, -INFINITY, -INFINITY};
    for (const auto& p : c.position_visibility)
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], p[k] - 0.1f);
            mx[k] = std::max(mx[k], p[k] + 0.1f);
        }
    for (int k = 0; k < 3; ++k) {
        const float center = (mn[k] + mx[k]) / 2.0f, half = (mx[k] - mn[k]) / 2.0f;
        mn_out[k] = center - half;
        mx_out[k] = center + half;
    }
    return true;
}

// ---- INRIA .ply -> PlanarGaussian3d ---------------------------------------------------------------
constexpr float MAX_SIZE_VARIANCE = 4.0f;  // src/io/ply.rs:21

namespace ply_detail {
struct Prop { std::string name; char kind; int bytes; bool list; int count_bytes = 0; char count_kind = 'u'; };  // kind: 'i' 'u' 'f'; lists: bytes = item size
inline bool scalar_type(const std::string& t, char& kind, int& bytes) {
    static const std::map<std::string, std::pair<char, int>> types = {
        {"char", {'i', 1}}, {"int8", {'i', 1}}, {"uchar", {'u', 1}}, {"uint8", {'u', 1}}, {"short", {'i', 2}},
        {"int16", {'i', 2}}, {"ushort", {'u', 2}}, {"uint16", {'u', 2}}, {"int", {'i', 4}}, {"int32", {'i', 4}},
        {"uint", {'u', 4}}, {"uint32", {'u', 4}}, {"float", {'f', 4}}, {"float32", {'f', 4}}, {"double", {'f', 8}},
        {"float64", {'f', 8}}};
    const auto it = types.find(t);
    if (it == types.end()) return false;
    kind = it->second.first;
    bytes = it->second.second;
    return true;
}
inline float sigmoid(float x) { return 1.0f / (1.0f + std::exp(-x)); }  // src/io/ply.rs:40-42
}  // namespace ply_detail

// src/io/ply.rs:76-132. Only `float` properties reach the splat; required: x y z f_dc_0..2 scale_0 scale_1
// opacity rot_0..3; opacity is a logit; f_rest_i -> coefficient (i % 15) + 1 of channel i / 16 (the
// reference's mapping, reproduced literally); scale clamped to mean +- 4 in log space, then exp;
// rotation normalised; padded with Gaussian3d::default() to a multiple of 32 (a whole block if aligned).
inline PlanarGaussian3d parse_ply_3d(std::istream& in) {
    using namespace ply_detail;
    std::string line;
    if (!std::getline(in, line) || line.substr(0, 3) != "ply") throw Error(BGS_EINVAL, "not a PLY file");
    std::string format;
    struct Element { std::string name; size_t count; std::vector<Prop> props; };
    std::vector<Element> elements;
    for (;;) {
        if (!std::getline(in, line)) throw Error(BGS_EINVAL, "unexpected end of PLY header");
        std::istringstream ls(line);
        std::vector<std::string> tok;
        for (std::string t; ls >> t;) tok.push_back(t);
        if (tok.empty() || tok[0] == "comment" || tok[0] == "obj_info") continue;
        if (tok[0] == "format" && tok.size() > 1) format = tok[1];
        else if (tok[0] == "element" && tok.size() > 2) elements.push_back({tok[1], (size_t)std::stoull(tok[2]), {}});
        else if (tok[0] == "property" && tok.size() > 2 && !elements.empty()) {
            if (tok[1] == "list" && tok.size() > 4) {  // property list <count type> <item type> <name>
                Prop p{tok[4], 'f', 4, true};
                if (!scalar_type(tok[2], p.count_kind, p.count_bytes) || !scalar_type(tok[3], p.kind, p.bytes))
                    throw Error(BGS_EINVAL, "unknown PLY list types " + tok[2] + " " + tok[3]);
                elements.back().props.push_back(p);
            }
            else {
                Prop p{tok[2], 'f', 4, false};
                if (!scalar_type(tok[1], p.kind, p.bytes)) throw Error(BGS_EINVAL, "unknown PLY property type " + tok[1]);
                elements.back().props.push_back(p);
            }
        } else if (tok[0] == "end_header") break;
    }
    const bool ascii = format == "ascii", le = format == "binary_little_endian", be = format == "binary_big_endian";
    if (!ascii && !le && !be) throw Error(BGS_EINVAL, "unsupported PLY format " + format);

    std::map<std::string, std::vector<float>> cols;  // float properties of the vertex element
    std::vector<std::string> rest_order;             // f_rest_* in file order
    size_t n = 0;
    for (const Element& e : elements) {
        const bool vertex = e.name == "vertex";
        if (vertex) {
            for (const char* r : {"x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "scale_0", "scale_1", "opacity", "rot_0",
                                  "rot_1", "rot_2", "rot_3"})
                if (std::none_of(e.props.begin(), e.props.end(), [&](const Prop& p) { return p.name == r; }))
                    throw Error(BGS_EINVAL, "missing required properties");  // ply.rs:93-98
            cols.clear();
            rest_order.clear();
            n = e.count;
            for (const Prop& p : e.props)
                if (!p.list && p.kind == 'f' && p.bytes == 4) {
                    cols[p.name].assign(n, 0.0f);
                    if (p.name.rfind("f_rest_", 0) == 0) rest_order.push_back(p.name);
                }
        }
        if (ascii) {
            for (size_t r = 0; r < e.count; ++r) {
                if (!std::getline(in, line)) throw Error(BGS_EINVAL, "truncated PLY payload");
                std::istringstream ls(line);
                for (const Prop& p : e.props) {
                    double v = 0.0;
                    ls >> v;
                    if (p.list) { for (long k = (long)v; k > 0; --k) { double skip; ls >> skip; } continue; }
                    if (vertex && p.kind == 'f' && p.bytes == 4) cols[p.name][r] = (float)v;
                }
            }
        } else {
            if (std::any_of(e.props.begin(), e.props.end(), [](const Prop& p) { return p.list; })) {
                // an element with list properties (a trailing `face` element of a mesh export, which ply-rs
                // parses and the reference then ignores) has rows of varying length: walk it row by row
                auto read_uint = [&](int bytes, char kind) -> long long {
                    unsigned char b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    in.read((char*)b, bytes);
                    if (in.gcount() != bytes) throw Error(BGS_EINVAL, "truncated PLY payload");
                    if (be) std::reverse(b, b + bytes);
                    if (kind == 'f') { if (bytes == 4) { float x; std::memcpy(&x, b, 4); return (long long)x; } double x; std::memcpy(&x, b, 8); return (long long)x; }
                    unsigned long long v = 0;
                    for (int k = 0; k < bytes; ++k) v |= (unsigned long long)b[k] << (8 * k);
                    if (kind == 'i' && bytes < 8 && (v >> (8 * bytes - 1))) v |= ~0ull << (8 * bytes);
                    return (long long)v;
                };
                for (size_t r = 0; r < e.count; ++r)
                    for (const Prop& p : e.props) {
                        if (p.list) {
                            const long long k = read_uint(p.count_bytes, p.count_kind);
                            if (k < 0) throw Error(BGS_EINVAL, "truncated PLY payload");
                            in.ignore((std::streamsize)k * p.bytes);
                            if (in.gcount() != (std::streamsize)k * p.bytes) throw Error(BGS_EINVAL, "truncated PLY payload");
                        } else if (vertex && p.kind == 'f' && p.bytes == 4) {
                            unsigned char b[4];
                            in.read((char*)b, 4);
                            if (in.gcount() != 4) throw Error(BGS_EINVAL, "truncated PLY payload");
                            if (be) std::swap(b[0], b[3]), std::swap(b[1], b[2]);
                            std::memcpy(&cols[p.name][r], b, 4);
                        } else {
                            in.ignore(p.bytes);
                            if (in.gcount() != p.bytes) throw Error(BGS_EINVAL, "truncated PLY payload");
                        }
                    }
                continue;
            }
            size_t stride = 0;
            for (const Prop& p : e.props) stride += (size_t)p.bytes;
            std::vector<char> raw(stride * e.count);
            in.read(raw.data(), (std::streamsize)raw.size());
            if ((size_t)in.gcount() != raw.size()) throw Error(BGS_EINVAL, "truncated PLY payload");
            if (!vertex) continue;
            size_t off = 0;
            for (const Prop& p : e.props) {
                if (p.kind == 'f' && p.bytes == 4) {
                    std::vector<float>& col = cols[p.name];
                    for (size_t r = 0; r < e.count; ++r) {
                        unsigned char b[4];
                        std::memcpy(b, raw.data() + r * stride + off, 4);
                        if (be) std::swap(b[0], b[3]), std::swap(b[1], b[2]);
                        std::memcpy(&col[r], b, 4);
                    }
                }
                off += (size_t)p.bytes;
            }
        }
    }

    auto col = [&](const std::string& name) -> const std::vector<float>* {
        const auto it = cols.find(name);
        return it == cols.end() ? nullptr : &it->second;
    };
    const size_t pad = 32 - (n % 32);  // ply.rs:127-129
    PlanarGaussian3d c;
    c.resize(n + pad);
    for (size_t i = 0; i < n + pad; ++i) {
        c.position_visibility[i] = {0.0f, 0.0f, 0.0f, 1.0f};  // PositionVisibility::default
        c.spherical_harmonic[i].fill(0.0f);
        c.rotation[i] = {0.0f, 0.0f, 0.0f, 0.0f};
        c.scale_opacity[i] = {0.0f, 0.0f, 0.0f, 0.0f};
    }
    const char* pos_names[4] = {"x", "y", "z", "visibility"};
    for (int k = 0; k < 4; ++k)
        if (const auto* v = col(pos_names[k])) for (size_t i = 0; i < n; ++i) c.position_visibility[i][k] = (*v)[i];
    for (int k = 0; k < 3; ++k) {
        if (const auto* v = col("f_dc_" + std::to_string(k))) for (size_t i = 0; i < n; ++i) c.spherical_harmonic[i][k] = (*v)[i];
        if (const auto* v = col("scale_" + std::to_string(k))) for (size_t i = 0; i < n; ++i) c.scale_opacity[i][k] = (*v)[i];
    }
    if (const auto* v = col("opacity")) for (size_t i = 0; i < n; ++i) c.scale_opacity[i][3] = sigmoid((*v)[i]);
    for (int k = 0; k < 4; ++k)
        if (const auto* v = col("rot_" + std::to_string(k))) for (size_t i = 0; i < n; ++i) c.rotation[i][k] = (*v)[i];
    constexpr int PER_CHANNEL = SH_COEFF_COUNT / 3;  // 16
    for (const std::string& name : rest_order) {
        const int idx = std::stoi(name.substr(7));
        const int channel = idx / PER_CHANNEL, coefficient = (idx % (PER_CHANNEL - 1)) + 1;
        const int slot = coefficient * 3 + channel;
        if (slot < SH_COEFF_COUNT) {
            const auto& v = *col(name);
            for (size_t i = 0; i < n; ++i) c.spherical_harmonic[i][slot] = v[i];
        }
    }
    for (size_t i = 0; i < n; ++i) {  // ply.rs:103-125
        auto& so = c.scale_opacity[i];
        const float mean = ((so[0] + so[1]) + so[2]) / 3.0f;
        for (int k = 0; k < 3; ++k) so[k] = std::exp(std::min(std::max(so[k], mean - MAX_SIZE_VARIANCE), mean + MAX_SIZE_VARIANCE));
        auto& r = c.rotation[i];
        const float norm = std::sqrt(((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) + r[3] * r[3]);
        for (int k = 0; k < 4; ++k) r[k] = r[k] / norm;
    }
    return c;
}

// ---- .gcloud container (src/io/gcloud/flexbuffers.rs:9-22, src/io/codec.rs:8-17) ---------------------
// CloudCodec::decode for PlanarGaussian3d: the file is the serde / FlexBuffers image of
//     { position_visibility: [{position: [f32; 3], visibility: f32}], spherical_harmonic: [{coefficients: (f32 x 48)}],
//       rotation: [{rotation: [f32; 4]}], scale_opacity: [{scale: [f32; 3], opacity: f32}] }
// (structs = maps with sorted keys, arrays / tuples / Vec = vectors; serde also accepts structs as sequences
// in field order, and missing fields take Default). The `flexbuffers` 25.2 crate is not in the reference
// tree: this is a general reader of the published format (google/flatbuffers flexbuffers.h: every type
// and byte width), so it does not depend on a writer's choices. PARITY UNPINNED: no .gcloud file written by
// the reference exists here; held to the Python reader / writer of bevy_gaussian_splatting_amd/io_gcloud.py.
namespace gcloud_detail {
enum : uint8_t { T_NULL = 0, T_INT, T_UINT, T_FLOAT, T_KEY, T_STRING, T_INDIRECT_INT, T_INDIRECT_UINT, T_INDIRECT_FLOAT,
                 T_MAP, T_VECTOR, T_VECTOR_INT, T_VECTOR_UINT, T_VECTOR_FLOAT, T_VECTOR_KEY, T_VECTOR_STRING,
                 T_VECTOR_INT2 = 16, T_VECTOR_FLOAT4 = 24, T_BLOB = 25, T_BOOL = 26, T_VECTOR_BOOL = 36 };

struct Buf {
    const uint8_t* p;
    size_t n;
    void need(size_t pos, size_t bytes) const {
        if (pos > n || bytes > n - pos) throw std::runtime_error("gcloud: FlexBuffer offset out of range");
    }
    uint64_t u(size_t pos, unsigned width) const {
        need(pos, width);
        uint64_t v = 0;
        for (unsigned i = 0; i < width; ++i) v |= (uint64_t)p[pos + i] << (8 * i);
        return v;
    }
    int64_t i(size_t pos, unsigned width) const {
        const uint64_t v = u(pos, width);
        return width == 8 ? (int64_t)v : (int64_t)(v << (64 - 8 * width)) >> (64 - 8 * width);
    }
    double f(size_t pos, unsigned width) const {
        need(pos, width);
        if (width == 4) { float x; std::memcpy(&x, p + pos, 4); return x; }
        if (width == 8) { double x; std::memcpy(&x, p + pos, 8); return x; }
        if (width == 2) {  // IEEE binary16
            const uint32_t h = (uint32_t)u(pos, 2), sgn = h >> 15, e = (h >> 10) & 31u, m = h & 1023u;
            const double mag = e == 0 ? std::ldexp((double)m, -24) : e == 31 ? (m ? NAN : INFINITY) : std::ldexp((double)(m | 1024u), (int)e - 25);
            return sgn ? -mag : mag;
        }
        throw std::runtime_error("gcloud: bad float width");
    }
};

// A value by reference: where its slot is, how wide the slot is, and the packed (type << 2 | log2 width) byte.
struct Ref {
    const Buf* b = nullptr;
    size_t pos = 0;
    unsigned parent_width = 1;
    uint8_t packed = 0;
    uint8_t type() const { return packed >> 2; }
    unsigned width() const { return 1u << (packed & 3u); }
    size_t target() const {
        const uint64_t off = b->u(pos, parent_width);
        if (off > pos) throw std::runtime_error("gcloud: FlexBuffer offset out of range");
        return pos - (size_t)off;
    }
    bool is_null() const { return b == nullptr || type() == T_NULL; }
    double number() const {
        switch (type()) {
            case T_INT: return (double)b->i(pos, parent_width);
            case T_UINT: case T_BOOL: return (double)b->u(pos, parent_width);
            case T_FLOAT: return b->f(pos, parent_width);
            case T_INDIRECT_INT: return (double)b->i(target(), width());
            case T_INDIRECT_UINT: return (double)b->u(target(), width());
            case T_INDIRECT_FLOAT: return b->f(target(), width());
            default: throw std::runtime_error("gcloud: expected a number");
        }
    }
    bool is_map() const { return type() == T_MAP; }
    bool is_vector() const {
        const uint8_t t = type();
        return t == T_MAP || t == T_VECTOR || (t >= T_VECTOR_INT && t <= T_VECTOR_STRING) || t == T_VECTOR_BOOL ||
               (t >= T_VECTOR_INT2 && t <= T_VECTOR_FLOAT4);
    }
    size_t length() const {
        const uint8_t t = type();
        if (t >= T_VECTOR_INT2 && t <= T_VECTOR_FLOAT4) return (size_t)((t - T_VECTOR_INT2) / 3 + 2);
        if (!is_vector()) throw std::runtime_error("gcloud: expected a vector");
        const size_t tg = target();
        if (tg < width()) throw std::runtime_error("gcloud: FlexBuffer offset out of range");
        return (size_t)b->u(tg - width(), width());
    }
    Ref at(size_t index) const {  // element of a vector, typed vector or map (values)
        const uint8_t t = type();
        const size_t len = length(), tg = target();
        const unsigned bw = width();
        if (index >= len) throw std::runtime_error("gcloud: vector index out of range");
        Ref r;
        r.b = b;
        r.pos = tg + index * bw;
        r.parent_width = bw;
        if (t == T_MAP || t == T_VECTOR) {
            b->need(tg + len * bw + index, 1);
            r.packed = b->p[tg + len * bw + index];
        } else {
            uint8_t et;
            if (t == T_VECTOR_BOOL) et = T_BOOL;
            else if (t >= T_VECTOR_INT2) et = (uint8_t)((t - T_VECTOR_INT2) % 3 + T_INT);
            else et = (uint8_t)(t - T_VECTOR_INT + T_INT);
            const uint8_t lg = bw == 1 ? 0 : bw == 2 ? 1 : bw == 4 ? 2 : 3;
            r.packed = (uint8_t)((et << 2) | ((et == T_KEY || et == T_STRING) ? 0 : lg));
        }
        return r;
    }
    std::string key_at(size_t index) const {  // map only
        const size_t tg = target();
        const unsigned bw = width();
        if (tg < 3 * (size_t)bw) throw std::runtime_error("gcloud: FlexBuffer offset out of range");
        const size_t keys_slot = tg - 3 * bw;
        const uint64_t koff = b->u(keys_slot, bw);
        if (koff > keys_slot) throw std::runtime_error("gcloud: FlexBuffer offset out of range");
        const size_t keys_target = keys_slot - (size_t)koff;
        const unsigned kbw = (unsigned)b->u(tg - 2 * bw, bw);
        if (kbw != 1 && kbw != 2 && kbw != 4 && kbw != 8) throw std::runtime_error("gcloud: bad key vector width");
        const size_t kslot = keys_target + index * kbw;
        const uint64_t off = b->u(kslot, kbw);
        if (off > kslot) throw std::runtime_error("gcloud: FlexBuffer offset out of range");
        size_t s0 = kslot - (size_t)off, e = s0;
        while (true) {
            b->need(e, 1);
            if (b->p[e] == 0) break;
            ++e;
        }
        return std::string((const char*)b->p + s0, e - s0);
    }
    Ref find(const char* name) const {  // map only; a missing key gives a null Ref
        const size_t len = length();
        for (size_t k = 0; k < len; ++k)
            if (key_at(k) == name) return at(k);
        return Ref{};
    }
    // a struct field: by name when the struct was written as a map, by position when as a sequence
    Ref field(const char* name, size_t index) const {
        if (is_map()) return find(name);
        if (is_vector()) return index < length() ? at(index) : Ref{};
        return Ref{};
    }
};

inline Ref root(const Buf& b) {
    if (b.n < 3) throw std::runtime_error("gcloud: not a FlexBuffer (too short)");
    const unsigned root_width = b.p[b.n - 1];
    if (root_width != 1 && root_width != 2 && root_width != 4 && root_width != 8)
        throw std::runtime_error("gcloud: not a FlexBuffer (bad root width)");
    if (b.n < 2 + (size_t)root_width) throw std::runtime_error("gcloud: not a FlexBuffer (too short)");
    Ref r;
    r.b = &b;
    r.pos = b.n - 2 - root_width;
    r.parent_width = root_width;
    r.packed = b.p[b.n - 2];
    return r;
}

template <size_t K>
inline void read_floats(const Ref& v, std::array<float, K>& out, size_t first, size_t count) {
    if (v.is_null()) return;  // Default
    if (!v.is_vector() || v.length() != count) throw std::runtime_error("gcloud: array field of unexpected length");
    for (size_t k = 0; k < count; ++k) out[first + k] = (float)v.at(k).number();
}
}  // namespace gcloud_detail

inline PlanarGaussian3d decode_gcloud(const uint8_t* data, size_t size) {
    using namespace gcloud_detail;
    const Buf buf{data, size};
    const Ref top = root(buf);
    if (!top.is_vector()) throw std::runtime_error("gcloud: root is neither a map nor a sequence");
    const Ref pv = top.field("position_visibility", 0), sh = top.field("spherical_harmonic", 1),
              rot = top.field("rotation", 2), so = top.field("scale_opacity", 3);
    size_t n = 0;
    for (const Ref* plane : {&pv, &sh, &rot, &so})
        if (!plane->is_null()) n = std::max(n, plane->length());
    for (const Ref* plane : {&pv, &sh, &rot, &so})
        if (!plane->is_null() && plane->length() != 0 && plane->length() != n)
            throw std::runtime_error("gcloud: planes have different lengths");
    PlanarGaussian3d c;
    c.resize(n);
    for (size_t i = 0; i < n; ++i) {
        c.position_visibility[i] = {0.0f, 0.0f, 0.0f, 1.0f};  // PositionVisibility::default()
        c.spherical_harmonic[i].fill(0.0f);
        c.rotation[i] = {0.0f, 0.0f, 0.0f, 0.0f};
        c.scale_opacity[i] = {0.0f, 0.0f, 0.0f, 0.0f};
    }
    if (!pv.is_null() && pv.length())
        for (size_t i = 0; i < n; ++i) {
            const Ref it = pv.at(i);
            read_floats(it.field("position", 0), c.position_visibility[i], 0, 3);
            const Ref vis = it.field("visibility", 1);
            if (!vis.is_null()) c.position_visibility[i][3] = (float)vis.number();
        }
    if (!sh.is_null() && sh.length())
        for (size_t i = 0; i < n; ++i) read_floats(sh.at(i).field("coefficients", 0), c.spherical_harmonic[i], 0, SH_COEFF_COUNT);
    if (!rot.is_null() && rot.length())
        for (size_t i = 0; i < n; ++i) read_floats(rot.at(i).field("rotation", 0), c.rotation[i], 0, 4);
    if (!so.is_null() && so.length())
        for (size_t i = 0; i < n; ++i) {
            const Ref it = so.at(i);
            read_floats(it.field("scale", 0), c.scale_opacity[i], 0, 3);
            const Ref op = it.field("opacity", 1);
            if (!op.is_null()) c.scale_opacity[i][3] = (float)op.number();
        }
    return c;
}

inline PlanarGaussian3d read_gcloud(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    const std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    return decode_gcloud(bytes.data(), bytes.size());
}

// Gaussian3dLoader (src/io/loader.rs:22-61): the asset loader's dispatch on the file extension.
inline PlanarGaussian3d load_cloud(const std::string& path) {
    const size_t dot = path.rfind('.');
    const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    if (ext == "ply") {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw std::runtime_error("cannot open " + path);
        return parse_ply_3d(in);
    }
    if (ext == "gcloud") return read_gcloud(path);
    throw std::runtime_error("only .ply and .gcloud supported");
}

// ---- precomputed covariance upload (src/gaussian/covariance.rs:4-41, src/gaussian/f32.rs:218-251) ---------
inline std::array<float, 6> compute_covariance_3d(const std::array<float, 4>& q, const std::array<float, 3>& scale) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    // columns of R (Mat3::from_cols), M = S * R (row i of R scaled by scale_i), Sigma = M^T * M
    const float R[3][3] = {{1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y + r * z), 2.0f * (x * z - r * y)},
                           {2.0f * (x * y - r * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z + r * x)},
                           {2.0f * (x * z + r * y), 2.0f * (y * z - r * x), 1.0f - 2.0f * (x * x + y * y)}};  // R[row][col]
    float M[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = scale[i] * R[i][j];
    auto S = [&](int i, int j) { return (M[0][i] * M[0][j] + M[1][i] * M[1][j]) + M[2][i] * M[2][j]; };
    return {S(0, 0), S(0, 1), S(0, 2), S(1, 1), S(1, 2), S(2, 2)};
}

inline PlanarGaussian3dHandle upload_precomputed_covariance(GaussianSplattingPlugin& plugin, const PlanarGaussian3d& c) {
    std::vector<std::array<float, 8>> cov(c.size());
    for (size_t i = 0; i < c.size(); ++i) {
        const auto s6 = compute_covariance_3d(c.rotation[i], {c.scale_opacity[i][0], c.scale_opacity[i][1], c.scale_opacity[i][2]});
        cov[i] = {s6[0], s6[1], s6[2], s6[3], s6[4], s6[5], c.scale_opacity[i][3], 0.0f};
    }
    bgs_cloud* cloud = nullptr;
    const int rc = bgs_cloud_upload_cov3d_f32(plugin.native(), (uint32_t)c.size(), c.size() ? c.position_visibility[0].data() : nullptr,
                                              c.size() ? c.spherical_harmonic[0].data() : nullptr,
                                              c.size() ? cov[0].data() : nullptr, &cloud);
    if (rc != BGS_OK) throw Error(rc, std::string("bgs_cloud_upload_cov3d_f32: ") + bgs_last_error(plugin.native()));
    return plugin.adopt(cloud);
}

// ---- f16 upload -------------------------------------------------------------------------------
inline PlanarGaussian3dHandle upload(GaussianSplattingPlugin& plugin, const PlanarGaussian3dF16& c) {
    bgs_cloud* cloud = nullptr;
    const int rc = bgs_cloud_upload_f16(plugin.native(), (uint32_t)c.size(), c.size() ? c.position_visibility[0].data() : nullptr,
                                        c.size() ? c.spherical_harmonic[0].data() : nullptr,
                                        c.size() ? c.rotation_scale_opacity[0].data() : nullptr, &cloud);
    if (rc != BGS_OK) throw Error(rc, std::string("bgs_cloud_upload_f16: ") + bgs_last_error(plugin.native()));
    return plugin.adopt(cloud);
}

}  // namespace bgs
#endif  // BGS_HOST_HPP
