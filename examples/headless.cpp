// headless.cpp — the reference's examples/headless.rs on this library: a seeded random cloud, Camera3d at
// (0, 1.5, 5), an Rgba8UnormSrgb target, `frames` frames, then <output-dir>/0.png (examples/headless.rs:
// 52-66 frames_to_wait = 40, :120-137 target + cloud, :178-183 camera, :349-409 save). C++17 over
// include/bgs.hpp; links libbgs.so only (no HIP headers on the host side).
//
//   make -C examples        &&  examples/headless --gaussian-count 1000000 --width 1920 --height 1080
//
// Flags follow the reference's GaussianSplattingViewer args (src/utils.rs): --gaussian-count, --seed,
// --width, --height; plus --frames, --output-dir, --depth (lanes in flight), and two hooks for the parity
// test: --cloud <file> (u32 n, then the four f32 planes) and --dump-f32 <file> (the last frame, RGBA f32).
// --input-cloud <file.ply | file.gcloud> loads an INRIA .ply or a .gcloud container (the reference viewer's --input-cloud), --f16 uploads the cloud
// in the f16 planar format.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <sys/stat.h>

#include "../include/bgs_host.hpp"

namespace {

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}

void png_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    out.insert(out.end(), body.begin(), body.end());
    put_be32(out, crc32_update(0xFFFFFFFFu, body.data(), body.size()) ^ 0xFFFFFFFFu);
}

// RGBA8 PNG with stored (uncompressed) deflate blocks: no zlib needed.
void write_png(const std::string& path, const std::vector<uint8_t>& rgba, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (w * 4 + 1));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), rgba.begin() + (size_t)y * w * 4, rgba.begin() + (size_t)(y + 1) * w * 4);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size() || off == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; ++i) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
        if (n == 0) break;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});  // 8 bit, RGBA
    png_chunk(png, "IHDR", ihdr);
    png_chunk(png, "IDAT", z);
    png_chunk(png, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)png.data(), (std::streamsize)png.size());
    if (!f) throw std::runtime_error("cannot write " + path);
}

bgs::PlanarGaussian3d read_planes(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    uint32_t n = 0;
    f.read((char*)&n, 4);
    bgs::PlanarGaussian3d c;
    c.resize(n);
    f.read((char*)c.position_visibility.data(), (std::streamsize)n * 16);
    f.read((char*)c.spherical_harmonic.data(), (std::streamsize)n * bgs::SH_COEFF_COUNT * 4);
    f.read((char*)c.rotation.data(), (std::streamsize)n * 16);
    f.read((char*)c.scale_opacity.data(), (std::streamsize)n * 16);
    if (!f) throw std::runtime_error("cannot read " + path);
    return c;
}

}  // namespace

int main(int argc, char** argv) {
    uint32_t count = 10000, width = 1920, height = 1080, frames = 40, depth = 6;
    uint32_t msaa_samples = 0;   // 0 = not given: the view keeps Bevy's default Msaa (4 samples), as examples/headless.rs does
    uint64_t seed = 0;
    std::string out_dir = "headless_output", cloud_path, dump_path, ply_path;
    bool f16 = false, trained_like = false;
    bgs::CloudSettings settings;   // CloudSettings::default()
    // value names as clap's ValueEnum derives them from the reference's enums (kebab case; src/gaussian/settings.rs:17-57)
    auto gaussian_mode = [](const std::string& v) {
        if (v == "gaussian2d") return bgs::GaussianMode::Gaussian2d;
        if (v == "gaussian3d") return bgs::GaussianMode::Gaussian3d;
        std::fprintf(stderr, "--gaussian-mode %s: gaussian2d | gaussian3d (gaussian4d clouds are outside this path)\n", v.c_str());
        std::exit(2);
    };
    auto rasterize_mode = [](const std::string& v) {
        static const std::pair<const char*, bgs::RasterizeMode> names[] = {
            {"classification", bgs::RasterizeMode::Classification}, {"color", bgs::RasterizeMode::Color},
            {"depth", bgs::RasterizeMode::Depth}, {"normal", bgs::RasterizeMode::Normal},
            {"optical-flow", bgs::RasterizeMode::OpticalFlow}, {"position", bgs::RasterizeMode::Position},
            {"velocity", bgs::RasterizeMode::Velocity}};
        for (const auto& nv : names) if (v == nv.first) return nv.second;
        std::fprintf(stderr, "--rasterization-mode %s: classification | color | depth | normal | optical-flow | position | velocity\n", v.c_str());
        std::exit(2);
    };
    auto depth_bits = [](const std::string& v) {
        if (v == "bits16") return bgs::RadixSortDepthBits::Bits16;
        if (v == "bits24") return bgs::RadixSortDepthBits::Bits24;
        if (v == "bits32") return bgs::RadixSortDepthBits::Bits32;
        std::fprintf(stderr, "--radix-sort-depth-bits %s: bits16 | bits24 | bits32\n", v.c_str());
        std::exit(2);
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        // the reference's flag names (GaussianSplattingViewer, src/utils.rs:25-74; --seed stays as an alias)
        if (a == "--gaussian-count") count = (uint32_t)std::stoul(next());
        else if (a == "--gaussian-seed" || a == "--seed") seed = std::stoull(next());
        else if (a == "--gaussian-mode") settings.gaussian_mode = gaussian_mode(next());
        else if (a == "--rasterization-mode") settings.rasterize_mode = rasterize_mode(next());
        else if (a == "--radix-sort-depth-bits") settings.radix_sort_depth_bits = depth_bits(next());
        else if (a == "--msaa-samples") msaa_samples = (uint32_t)std::stoul(next());
        else if (a == "--trained-like") trained_like = true;   // (this build's: the trained-asset statistics instead of the reference's random cloud)
        else if (a == "--width") width = (uint32_t)std::stoul(next());
        else if (a == "--height") height = (uint32_t)std::stoul(next());
        else if (a == "--frames") frames = (uint32_t)std::stoul(next());
        else if (a == "--depth") depth = (uint32_t)std::stoul(next());
        else if (a == "--output-dir") out_dir = next();
        else if (a == "--cloud") cloud_path = next();
        else if (a == "--dump-f32") dump_path = next();
        else if (a == "--input-cloud") ply_path = next();
        else if (a == "--f16") f16 = true;
        else { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
    }
    try {
        bgs::GaussianSplattingPlugin plugin(0);
        bgs::PlanarGaussian3d cloud;
        if (!ply_path.empty()) {
            cloud = bgs::load_cloud(ply_path);  // .ply or .gcloud, like the reference's Gaussian3dLoader
        } else {
            cloud = !cloud_path.empty() ? read_planes(cloud_path)
                  : (trained_like ? bgs::PlanarGaussian3d::trained_like(count, seed) : bgs::PlanarGaussian3d::random(count, seed));
        }
        bgs::PlanarGaussian3dHandle handle =
            f16 ? bgs::upload(plugin, bgs::PlanarGaussian3dF16::from_f32(cloud)) : plugin.upload(cloud);
        bgs::View view = bgs::View::headless(width, height);  // Camera3d at (0, 1.5, 5), black clear colour
        if (msaa_samples) view.set_msaa_samples(msaa_samples);
        const bgs_settings native = settings.to_native();

        plugin.set_output_srgb8(true);  // the reference's target is TextureFormat::Rgba8UnormSrgb
        plugin.set_profiling(0);
        plugin.set_pipeline_depth(depth);
        plugin.set_async(true);
        for (uint32_t f = 0; f < 8; ++f) plugin.render(handle, view, native);  // allocations, hints
        plugin.synchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t f = 0; f < frames; ++f) plugin.render(handle, view, native);
        plugin.synchronize();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

        mkdir(out_dir.c_str(), 0755);
        write_png(out_dir + "/0.png", plugin.download_srgb8(view), width, height);
        if (!dump_path.empty()) {
            plugin.set_async(false);
            std::vector<float> rgba;
            plugin.render(handle, view, settings, &rgba);
            std::ofstream f(dump_path, std::ios::binary);
            f.write((const char*)rgba.data(), (std::streamsize)(rgba.size() * sizeof(float)));
        }
        const bgs_stats st = plugin.stats();
        std::printf("{\"splats\": %zu, \"width\": %u, \"height\": %u, \"frames\": %u, \"frames_per_s\": %.1f, "
                    "\"visible_splats\": %u, \"output\": \"%s/0.png\"}\n",
                    cloud.size(), width, height, frames, frames / s, st.visible_count, out_dir.c_str());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "headless: %s\n", e.what());
        return 1;
    }
    return 0;
}
