// frame_params.h — bgs_view + bgs_settings -> FrameParams (host, plain C++).
// Shared by bgs_api.hip and the CPU pre-flight shim (tests/host_shim) so both feed the
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
    fp.sort_path = 0;  // chosen per frame by the host (bgs_api.hip)
    fp.pad_sort = 0;
    for (int i = 0; i < 3; ++i) {
        fp.pos_min[i] = s->position_min[i];
        fp.pos_max[i] = s->position_max[i];
    }
}

}  // namespace bgs
