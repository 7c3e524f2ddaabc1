// Test driver for include/bgs_host.hpp (no GPU needed for these subcommands): tests/test_cpp_host.py
// compares every output with the Python mirror.
//   host_tool ply <in.ply> <out.bin>        parse_ply_3d -> u32 n + the four f32 planes
//   host_tool cloud <in.ply|in.gcloud> <out.bin>   load_cloud (Gaussian3dLoader's dispatch) -> the same
//   host_tool cov3d <planes.bin> <out.bin>  compute_covariance_3d -> Covariance3dOpacity rows (8 f32)
//   host_tool f16 <planes.bin> <out.bin>    PlanarGaussian3dF16::from_f32 -> pv f32, sh u32[n][24], rso u32[n][4]
//   host_tool half <in.f32> <out.u16>       f32_to_f16 of every value
//   host_tool trigger <period_ms>           stdin: "t x y z order" per line -> "camera_index needs_sort" per line
//   host_tool entries <cams> <len> <out>    SortedEntries::for_cloud, chunk, resize_cameras
//   host_tool aabb <planes.bin>             compute_aabb
//   host_tool settings                      sizeof / defaults of CloudSettings::to_native()
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../../include/bgs_host.hpp"

static void write_planes(const std::string& path, const bgs::PlanarGaussian3d& c) {
    std::ofstream f(path, std::ios::binary);
    const uint32_t n = (uint32_t)c.size();
    f.write((const char*)&n, 4);
    f.write((const char*)c.position_visibility.data(), (std::streamsize)n * 16);
    f.write((const char*)c.spherical_harmonic.data(), (std::streamsize)n * bgs::SH_COEFF_COUNT * 4);
    f.write((const char*)c.rotation.data(), (std::streamsize)n * 16);
    f.write((const char*)c.scale_opacity.data(), (std::streamsize)n * 16);
}

static bgs::PlanarGaussian3d read_planes(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    uint32_t n = 0;
    f.read((char*)&n, 4);
    bgs::PlanarGaussian3d c;
    c.resize(n);
    f.read((char*)c.position_visibility.data(), (std::streamsize)n * 16);
    f.read((char*)c.spherical_harmonic.data(), (std::streamsize)n * bgs::SH_COEFF_COUNT * 4);
    f.read((char*)c.rotation.data(), (std::streamsize)n * 16);
    f.read((char*)c.scale_opacity.data(), (std::streamsize)n * 16);
    return c;
}

int main(int argc, char** argv) {
    const std::string cmd = argc > 1 ? argv[1] : "";
    try {
        if (cmd == "cloud" && argc == 4) {  // load_cloud: the loader's dispatch on the extension (.ply / .gcloud)
            write_planes(argv[3], bgs::load_cloud(argv[2]));
        } else if (cmd == "cov3d" && argc == 4) {  // Covariance3dOpacity plane of a cloud
            const auto c = read_planes(argv[2]);
            std::ofstream f(argv[3], std::ios::binary);
            for (size_t i = 0; i < c.size(); ++i) {
                const auto s6 = bgs::compute_covariance_3d(c.rotation[i], {c.scale_opacity[i][0], c.scale_opacity[i][1], c.scale_opacity[i][2]});
                const float row[8] = {s6[0], s6[1], s6[2], s6[3], s6[4], s6[5], c.scale_opacity[i][3], 0.0f};
                f.write((const char*)row, 32);
            }
        } else if (cmd == "trained" && argc == 5) {   // trained <n> <seed> <out.bin>: PlanarGaussian3d::trained_like
            write_planes(argv[4], bgs::PlanarGaussian3d::trained_like(std::stoul(argv[2]), std::stoull(argv[3])));
        } else if (cmd == "ply" && argc == 4) {
            std::ifstream in(argv[2], std::ios::binary);
            write_planes(argv[3], bgs::parse_ply_3d(in));
        } else if (cmd == "f16" && argc == 4) {
            const auto h = bgs::PlanarGaussian3dF16::from_f32(read_planes(argv[2]));
            std::ofstream f(argv[3], std::ios::binary);
            f.write((const char*)h.position_visibility.data(), (std::streamsize)h.size() * 16);
            f.write((const char*)h.spherical_harmonic.data(), (std::streamsize)h.size() * 96);
            f.write((const char*)h.rotation_scale_opacity.data(), (std::streamsize)h.size() * 16);
        } else if (cmd == "half" && argc == 4) {
            std::ifstream in(argv[2], std::ios::binary);
            std::ofstream out(argv[3], std::ios::binary);
            float v;
            while (in.read((char*)&v, 4)) {
                const uint16_t h = bgs::f32_to_f16(v);
                out.write((const char*)&h, 2);
            }
        } else if (cmd == "trigger" && argc == 3) {
            bgs::SortConfig config{std::stoll(argv[2])};
            bgs::SortTrigger t;
            double now;
            float x, y, z;
            long order;
            while (std::cin >> now >> x >> y >> z >> order) {
                t.needs_sort = false;
                bgs::update_sort_trigger(t, {x, y, z}, order, config, now);
                std::printf("%zu %d\n", t.camera_index, t.needs_sort ? 1 : 0);
            }
        } else if (cmd == "entries" && argc == 5) {   // entries <cameras> <cloud_len> <out.bin>: for_cloud + a resize
            bgs::SortedEntries e = bgs::SortedEntries::for_cloud(std::stoul(argv[2]), std::stoul(argv[3]));
            auto r = e.chunk(e.camera_count - 1, std::stoul(argv[3]));
            r.first[0].key = 77u;                      // a write through the last camera's chunk
            std::ofstream f(argv[4], std::ios::binary);
            const uint64_t hdr[2] = {e.camera_count, e.entry_count};
            f.write((const char*)hdr, 16);
            f.write((const char*)e.sorted.data(), (std::streamsize)(e.sorted.size() * 8));
            bool threw = false;
            try { e.chunk(e.camera_count, e.entry_count); } catch (const bgs::Error&) { threw = true; }
            e.resize_cameras(e.camera_count + 1);
            std::printf("%d %zu %u\n", threw ? 1 : 0, e.sorted.size(), e.sorted[0].key);
        } else if (cmd == "aabb" && argc == 3) {
            std::array<float, 3> mn, mx;
            const bool ok = bgs::compute_aabb(read_planes(argv[2]), mn, mx);
            std::printf("%d %.9g %.9g %.9g %.9g %.9g %.9g\n", ok ? 1 : 0, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]);
        } else if (cmd == "settings") {
            const bgs_settings s = bgs::CloudSettings().to_native();
            bgs_settings d;
            bgs_settings_default(&d);
            std::printf("%zu %d\n", sizeof(bgs_settings), std::memcmp(&s, &d, sizeof s) == 0 ? 1 : 0);
        } else {
            std::fprintf(stderr, "usage: host_tool ply|f16|half|trigger|settings ...\n");
            return 2;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "host_tool: %s\n", e.what());
        return 1;
    }
    return 0;
}
