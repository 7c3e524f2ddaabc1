// frame_params.h — bgs_view + bgs_settings -> FrameParams (host, plain C++).
// Shared by bgs_frame.hip and the CPU pre-flight shim (tests/host_shim) so both feed the
// per-splat arithmetic exactly the same constants.
#pragma once
#include <string.h>

#include "../../include/bgs.h"
#include "bgs_device.h"

namespace bgs {

inline uint32_t depth_places(const bgs_settings* s) {
    if (s->sort_mode == BGS_SORT_NONE) return 0;
    if (s->sort_mode == BGS_SORT_RADIX) return s->radix_depth_bits / 8u;  // src/render/mod.rs:718
    return 4;  // SORT_RAYON / SORT_STD: full 32-bit f32 keys
}

inline void fill_frame_params(uint32_t n, const bgs_view* view, const bgs_settings* s, FrameParams& fp) {
    memcpy(fp.transform, s->transform, sizeof fp.transform);
    memcpy(fp.view_from_world, view->view_from_world, sizeof fp.view_from_world);
    memcpy(fp.clip_from_world, view->clip_from_world, sizeof fp.clip_from_world);
    fp.cam[0] = view->world_from_view[12];  // view.world_position
    fp.cam[1] = view->world_from_view[13];
    fp.cam[2] = view->world_from_view[14];
    fp.viewport_w = view->viewport[2];
    fp.viewport_h = view->viewport[3];
    fp.focal_x = view->clip_from_view[0] * view->viewport[2];  // src/render/helpers.wgsl:20-23
    fp.focal_y = view->clip_from_view[5] * view->viewport[3];
    fp.global_opacity = s->global_opacity;
    fp.global_scale = s->global_scale;
    fp.n = n;
    fp.key_shift = s->sort_mode == BGS_SORT_RADIX ? 32u - s->radix_depth_bits : 0u;  // mod.rs:719
    fp.gaussian_mode = s->gaussian_mode;
    fp.aabb = s->aabb ? 1u : 0u;
    fp.adaptive_radius = s->opacity_adaptive_radius ? 1u : 0u;
    fp.color_space = s->color_space;
    fp.sh_degree = s->sh_degree;
    fp.sort_mode = s->sort_mode;
    fp.width = (int32_t)view->viewport[2];
    fp.height = (int32_t)view->viewport[3];
    fp.tiles_x = (fp.width + TILE_PX - 1) / TILE_PX;
    fp.tiles_y = (fp.height + TILE_PX - 1) / TILE_PX;
    fp.debug = 0;
    fp.rasterize_mode = s->rasterize_mode;
    fp.num_classes = s->num_classes;
    fp.draw_mode = s->draw_mode;
    memcpy(fp.prev_clip_from_world, view->previous_clip_from_world, sizeof fp.prev_clip_from_world);
    fp.delta_time = view->delta_time;
    memcpy(fp.clear, view->clear_color, sizeof fp.clear);
    fp.srgb8_target = 0;
    fp.sort_path = 0;  // chosen per frame by the host (bgs_frame.hip)
    fp.sample_count = view->sample_count ? view->sample_count : 4u;   // 0 = not set = Msaa::default() = Sample4
    fp.depth_ptr = view->depth_device_ptr;
    // uniform parts of world_to_local_direction (gaussian.wgsl:166-176: normalize(basis[k]) = v / length(v),
    // length = sqrt(dot)) and of the bounding boxes (1.0 / viewport): IEEE binary32, the order of splat_math.h
    for (int k = 0; k < 3; ++k) {
        const float x = s->transform[4 * k], y = s->transform[4 * k + 1], z = s->transform[4 * k + 2];
        const float len = __builtin_sqrtf((x * x + y * y) + z * z);
        fp.basis[3 * k] = x / len;
        fp.basis[3 * k + 1] = y / len;
        fp.basis[3 * k + 2] = z / len;
    }
    fp.inv_viewport_w = 1.0f / view->viewport[2];
    fp.inv_viewport_h = 1.0f / view->viewport[3];
    fp.visualize_bbox = s->visualize_bounding_box ? 1u : 0u;
    for (int i = 0; i < 3; ++i) {
        fp.pos_min[i] = s->position_min[i];
        fp.pos_max[i] = s->position_max[i];
    }
}

// ---- adaptive policies of the host side (plain functions so that the CPU suite can test them) -----------

inline uint32_t pow2_ceil_u32(uint64_t v) {
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return (uint32_t)(p < (1ull << 31) ? p : (1ull << 31));
}

// Supertile level of the next frames from a completed frame's statistics (bgs_frame.hip, finish_lane).
// `ratio` = list entries per visible splat of a frame that ran at level `lv`; `edges` = supertile edge in
// tiles of levels 0..3. Inside [1.6, 4] the level stays. Outside it moves straight to the level the frame's
// own geometry asks for: a splat of s supertile edges overlaps (s + 1)^2 supertiles on average, so
// sqrt(ratio) - 1 is the typical splat extent in edges of THIS frame's level, and the finest level whose
// edge is at least that extent keeps the ratio under 4 — but always at least one level in the direction the
// band was left. *longer_out (if the level gets coarser): by how much the longest list is expected to grow
// (entries scale with the ratio, lists with the supertile area).
inline uint32_t next_supertile_level(double ratio, uint32_t lv, const uint32_t edges[4], double* longer_out) {
    if (longer_out) *longer_out = 1.0;
    if (!(ratio > 4.0 && lv < 3) && !(ratio < 1.6 && lv > 0)) return lv;
    const double root = ratio > 0.0 ? __builtin_sqrt(ratio) : 0.0;
    const double extent_tiles = (root > 1.0 ? root - 1.0 : 0.0) * (double)edges[lv];
    uint32_t target = 3;
    for (uint32_t k = 0; k < 4; ++k)
        if ((double)edges[k] >= extent_tiles) { target = k; break; }
    if (ratio > 4.0) target = target > lv + 1 ? target : lv + 1;
    else target = target < lv - 1 ? target : lv - 1;
    if (target > lv && longer_out) {
        const double e0 = (double)edges[lv], e1 = (double)edges[target];
        const double r1 = (extent_tiles / e1 + 1.0) * (extent_tiles / e1 + 1.0);
        const double longer = (r1 / ratio) * (e1 / e0) * (e1 / e0);
        *longer_out = longer > 1.0 ? longer : 1.0;
    }
    return target;
}

// A splitter table is usable only if it is ascending: bucket(key) = number of splitters <= key is monotone in
// the key exactly then, and the bucket sort's ORDER (not just its balance) rests on that.
inline bool splitters_ascending(const uint32_t* key, uint32_t count) {
    for (uint32_t i = 1; i < count; ++i)
        if (key[i - 1] > key[i]) return false;
    return true;
}

}  // namespace bgs
