// gs_frame.hip -- one entry point per pass: the stages of a frame issued back to back from C (include/gsplat_hip.h,
// "One entry point per pass").  Nothing is computed here: every stage is the stage entry point a host would otherwise call
// through its FFI, with the same arguments, so results are identical to the stage-by-stage path by construction.
#include "gs_common.h"
#include <stdlib.h>

#include <stddef.h>

extern "C" int gs_frame_layout(int32_t *offsets, int n) {
    const size_t at[GS_FRAME_SENTINELS] = {
        offsetof(GsFrame, n_points), offsetof(GsFrame, blend_flags), offsetof(GsFrame, near_plane),
        offsetof(GsFrame, n_keys_capacity), offsetof(GsFrame, xyz), offsetof(GsFrame, q_camera_pointcloud),
        offsetof(GsFrame, attrs), offsetof(GsFrame, keys), offsetof(GsFrame, bin_ranges), offsetof(GsFrame, n_bins),
        offsetof(GsFrame, image), offsetof(GsFrame, tile_order), offsetof(GsFrame, boundary_states),
        offsetof(GsFrame, route_counts), offsetof(GsFrame, list_start), offsetof(GsFrame, grad_image), offsetof(GsFrame, acc),
        offsetof(GsFrame, grad_xyz), offsetof(GsFrame, aux_stream), offsetof(GsFrame, band_row_bounds)};
    for (int i = 0; i < n && i < GS_FRAME_SENTINELS; ++i) offsets[i] = (int32_t)at[i];
    return GS_FRAME_SENTINELS;
}

#define GS_STAGE(call)            \
    do {                          \
        const int _rc = (call);   \
        if (_rc < 0) return _rc;  \
    } while (0)

// the forward stages; *colours_pending: work forked to f->aux_stream that `stream` has not waited for yet
static int frame_forward_stages(GsFrame *f, uint32_t stages, void *stream, bool *colours_pending) {
    const int filter = f->bin_shift == 0 ? 0 : (GS_FILTER_BOX | (f->exact_tile_cull ? GS_FILTER_CULL : 0));
    if ((stages & GS_FWD_POSE_INVERSE) && (stages & GS_FWD_FILTER_COMPACT)) {   // the filter inverts the poses itself
        GS_STAGE(gs_filter_compact_from_poses(f->xyz, f->invalid_mask, f->object_id, f->intrinsics, f->q_pointcloud_camera,
                                              f->t_pointcloud_camera, f->n_objects, f->q_camera_pointcloud,
                                              f->t_camera_pointcloud, f->n_points, f->near_plane, f->far_plane, f->width,
                                              f->height, f->visible_mask, f->ids, f->counters, f->filter_workspace, stream));
    } else {
        if (stages & GS_FWD_POSE_INVERSE)
            GS_STAGE(gs_pose_inverse(f->q_pointcloud_camera, f->t_pointcloud_camera, f->q_camera_pointcloud,
                                     f->t_camera_pointcloud, f->n_objects, stream));
        if (stages & GS_FWD_FILTER_COMPACT)
            GS_STAGE(gs_filter_compact(f->xyz, f->invalid_mask, f->object_id, f->intrinsics, f->q_camera_pointcloud,
                                       f->t_camera_pointcloud, f->n_points, f->near_plane, f->far_plane, f->width,
                                       f->height, f->visible_mask, f->ids, f->counters, f->filter_workspace, stream));
    }
    // GS_FWD_COLOUR_ASYNC: the colours leave the projection and run on the caller's second stream beside the list stages
    // (nothing before the blend reads them); GS_COLOUR_ASYNC=0 switches it off for A/B measurements
    static const bool colour_async_allowed = !(getenv("GS_COLOUR_ASYNC") && atoi(getenv("GS_COLOUR_ASYNC")) == 0);
    const bool colour_async = colour_async_allowed && (stages & GS_FWD_PREPROCESS) && (stages & GS_FWD_COLOUR_ASYNC) &&
                              f->aux_stream != nullptr && f->aux_event_fork != nullptr && f->aux_event_join != nullptr;
    if (stages & GS_FWD_PREPROCESS) {   // launched for the capacity n_points; M is read from the counters on the device
        GS_STAGE((colour_async ? gs_preprocess_geometry : gs_preprocess)(
            f->xyz, f->features, f->object_id, f->intrinsics, f->q_camera_pointcloud, f->t_camera_pointcloud, f->ids,
            f->n_points, 1, f->width, f->height, f->tile_row_begin, f->tile_row_step, f->tile_row_end, f->bin_shift,
            f->exact_tile_cull, f->always_store_rotation, f->depth_scale, f->counters, f->attrs, f->num_overlap_tiles,
            f->num_keys, f->block_sums, f->block_sums_full, stream));
        if (colour_async) {
            GS_CHECK_HIP(hipEventRecord((hipEvent_t)f->aux_event_fork, (hipStream_t)stream));
            GS_CHECK_HIP(hipStreamWaitEvent((hipStream_t)f->aux_stream, (hipEvent_t)f->aux_event_fork, 0));
            const int rc = gs_view_colours(f->xyz, f->features, f->object_id, f->q_camera_pointcloud, f->t_camera_pointcloud,
                                           f->ids, f->n_points, 1, f->counters, f->num_keys, f->attrs, f->aux_stream);
            GS_CHECK_HIP(hipEventRecord((hipEvent_t)f->aux_event_join, (hipStream_t)f->aux_stream));
            *colours_pending = true;   // (joined by the caller of this function whatever happens below)
            if (rc < 0) return rc;
        }
    }
#define GS_JOIN_COLOURS()                                                                                   \
    do {                                                                                                    \
        if (*colours_pending) {                                                                             \
            *colours_pending = false;                                                                       \
            GS_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)f->aux_event_join, 0));       \
        }                                                                                                   \
    } while (0)
    if (stages & (GS_FWD_ROUTE_COUNT | GS_FWD_ROUTE_SCATTER)) GS_JOIN_COLOURS();   // the routed records carry colours
    if (stages & GS_FWD_ROUTE_COUNT)
        GS_STAGE(gs_route_count(f->attrs, f->num_keys, f->n_points, f->counters, f->width, f->height, f->rows_per_band,
                                f->world, f->band_row_bounds, f->route_counts, f->route_workspace, stream));
    if (stages & GS_FWD_ROUTE_SCATTER)
        GS_STAGE(gs_route_scatter(f->attrs, f->num_keys, f->n_points, f->counters, f->width, f->height, f->rows_per_band,
                                  f->world, f->band_row_bounds, f->chunk_capacity, f->route_counts, f->route_send,
                                  f->route_pos, f->route_workspace, stream));
    // owner-sharded band side: the received records are the attrs array of every later stage
    const bool received = (stages & GS_FWD_COUNT_KEYS) != 0 || f->records != nullptr;
    const float *attrs = received ? f->records : f->attrs;
    const int n_list_points = received ? (int)f->n_records : f->n_points;
    if (stages & GS_FWD_COUNT_KEYS) {
        GS_CHECK_HIP(hipMemsetAsync(f->counters, 0, sizeof(int32_t) * GS_NUM_COUNTERS, (hipStream_t)stream));
        GS_STAGE(gs_count_keys(f->records, (int)f->n_records, f->chunk_capacity + 1, f->width, f->height, f->tile_row_begin,
                               f->tile_row_step, f->tile_row_end, f->bin_shift, f->exact_tile_cull, f->depth_scale,
                               f->counters, f->num_overlap_tiles, f->num_keys, f->block_sums, f->block_sums_full, stream));
    }
    // GS_HOST_MIRROR=0: the sizes travel through a copy launch (gs_read_counters_async) instead of being stored to the
    // host's pinned memory by the scan kernel itself
    static const bool host_mirror = !(getenv("GS_HOST_MIRROR") && atoi(getenv("GS_HOST_MIRROR")) == 0);
    const bool scan_stores_sizes = host_mirror && (stages & GS_FWD_SCAN) && (stages & GS_FWD_READ_SIZES) &&
                                   f->host_counters_pinned != nullptr;
    // size_stamp: the sizes travel as stamped words the host polls -- no event behind the scan (gsplat_hip.h)
    const bool stamped = scan_stores_sizes && f->size_stamp != 0;
    GS_REQUIRE(f->size_stamp == 0 || stamped, "size_stamp needs GS_FWD_SCAN | GS_FWD_READ_SIZES and host_counters_pinned");
    if (stages & GS_FWD_SCAN) {
        if (stamped)
            GS_STAGE(gs_scan_block_sums2_stamped(f->block_sums, f->block_sums_full, gs_div_up(n_list_points, GS_BLOCK),
                                                 f->counters, f->host_counters_pinned, (uint32_t)f->size_stamp, stream));
        else
            GS_STAGE(gs_scan_block_sums2_to_host(f->block_sums, f->block_sums_full, gs_div_up(n_list_points, GS_BLOCK),
                                                 f->counters, scan_stores_sizes ? f->host_counters_pinned : nullptr, stream));
    }
    if ((stages & GS_FWD_READ_SIZES) && !stamped) {
        if (!scan_stores_sizes)
            GS_STAGE(gs_read_counters_async(f->counters, f->host_counters_pinned, GS_NUM_COUNTERS, stream));
        if (f->size_event != nullptr) GS_CHECK_HIP(hipEventRecord((hipEvent_t)f->size_event, (hipStream_t)stream));
    }
    const int32_t *n_keys_device = f->counters + GS_COUNTER_NUM_KEYS;
    if (stages & GS_FWD_MAKE_KEYS)
        GS_STAGE(gs_make_keys(attrs, f->num_keys, f->block_sums, n_list_points, f->counters, f->n_keys_capacity, f->width,
                              f->height, f->tile_row_begin, f->tile_row_step, f->tile_row_end, f->bin_shift,
                              f->exact_tile_cull, f->key_depth_bits, f->depth_scale, f->keys, f->payload,
                              f->need_state ? f->num_overlap_tiles : nullptr, f->need_state ? f->block_sums_full : nullptr,
                              f->need_state ? f->slot_offsets : nullptr, stream));
    const size_t range_bytes = sizeof(int32_t) * 2 * (size_t)f->n_bins;
    bool ranges_written = false;
    if (stages & GS_FWD_SORT) {
        GS_REQUIRE(range_bytes % 16 == 0, "n_bins must be even (the ranges are zeroed with 16-byte stores)");
        // GS_SORT_RANGES=0: the ranges always by their own launch (A/B measurements)
        static const bool sort_writes_ranges = !(getenv("GS_SORT_RANGES") && atoi(getenv("GS_SORT_RANGES")) == 0);
        const bool want = sort_writes_ranges && (stages & GS_FWD_RANGES);
        const int rc = gs_sort_pairs_and_zero(f->keys, f->payload, f->keys_alt, f->payload_alt, f->n_keys_capacity,
                                              n_keys_device, f->key_depth_bits, f->depth_bits, f->tile_bits, 1, 1,
                                              f->sort_workspace, f->bin_ranges, range_bytes,
                                              want ? f->bin_ranges : nullptr, want ? f->bin_ranges + f->n_bins : nullptr,
                                              f->n_bins, stream);
        if (rc < 0) return rc;
        f->sorted_in_alt = rc & 1;
        ranges_written = (rc & 2) != 0;
    }
    const void *keys_sorted = f->sorted_in_alt ? f->keys_alt : f->keys;
    const int32_t *payload_sorted = f->sorted_in_alt ? f->payload_alt : f->payload;
    if ((stages & GS_FWD_RANGES) && !ranges_written)
        GS_STAGE(gs_tile_ranges_prezeroed(keys_sorted, f->n_keys_capacity, n_keys_device, f->key_depth_bits, f->bin_ranges,
                                          f->bin_ranges + f->n_bins, f->n_bins, (stages & GS_FWD_SORT) ? 1 : 0, stream));
    GS_JOIN_COLOURS();
    if (stages & GS_FWD_BLEND)
        GS_STAGE(gs_blend_forward_split(
            f->bin_ranges, f->bin_ranges + f->n_bins, payload_sorted, attrs, f->width, f->height, f->tile_row_begin,
            f->tile_row_step, f->tile_row_end, f->bin_shift, filter, f->image, f->depth, f->acc_alpha, f->last_effective,
            f->valid_count, f->blend_flags, nullptr, f->tile_order, f->tile_work, f->walked_list, f->walked_start,
            f->boundary_states, f->n_keys_capacity, f->forward_split_workspace, stream));
    return 0;
}
#undef GS_JOIN_COLOURS

extern "C" {

size_t gs_frame_struct_bytes(void) { return sizeof(GsFrame); }

int gs_frame_forward(GsFrame *f, uint32_t stages, void *stream) {
    GS_REQUIRE(f != nullptr, "frame");
    bool colours_pending = false;
    const int rc = frame_forward_stages(f, stages, stream, &colours_pending);
    // the caller's stream never runs ahead of work this call forked, error or not
    if (colours_pending) GS_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)f->aux_event_join, 0));
    return rc;
}

int gs_frame_backward(GsFrame *f, uint32_t stages, void *stream) {
    GS_REQUIRE(f != nullptr, "frame");
    const bool received = f->records != nullptr;
    const float *attrs = received ? f->records : f->attrs;
    const int n_list_points = received ? (int)f->n_records : f->n_visible;
    if (stages & GS_BWD_BLEND)
        GS_STAGE(gs_blend_backward_split(
            f->list_start, f->list_payload, attrs, f->grad_image, f->acc_alpha, f->last_effective, f->slot_offsets, f->n_slots,
            f->width, f->height, f->tile_row_begin, f->tile_row_step, f->tile_row_end, f->backward_bin_shift,
            f->backward_filter, f->partials, f->slot_flags, f->magnitude_image, nullptr,
            f->blend_flags & (GS_BLEND_TWO_WAVES | GS_BLEND_FOUR_WAVES | GS_BLEND_ONE_WAVE | GS_BLEND_SKEWED_WALKS), f->tile_work, f->tile_order_backward,
            f->boundary_states != nullptr ? f->image : nullptr, f->boundary_states, f->n_keys_capacity, f->split_workspace,
            stream));
    if (stages & GS_BWD_REDUCE)
        GS_STAGE(gs_reduce_partials(f->slot_offsets, f->num_overlap_tiles, f->slot_flags, f->partials, n_list_points, f->acc,
                                    (received || f->tile_row_begin != 0 || f->tile_row_step != 1 ||
                                     f->tile_row_end < f->height / GS_TILE_HEIGHT) ? f->num_keys : nullptr,
                                    f->n_slots, attrs, f->width, f->height, stream));
    if (stages & GS_BWD_GATHER_RETURNED)
        GS_STAGE(gs_gather_returned_rows(f->returned_rows, f->route_pos, f->n_visible, f->n_points, f->world,
                                         f->chunk_capacity, f->acc, stream));
    if (stages & GS_BWD_POINTS)
        GS_STAGE(gs_point_backward(f->xyz, f->features, f->object_id, f->intrinsics, f->q_camera_pointcloud,
                                   f->t_camera_pointcloud, f->t_pointcloud_camera, f->ids, f->visible_mask, f->n_visible,
                                   f->n_points, f->acc, f->attrs, f->num_keys, f->color_max_sh_band, f->grad_q_factor,
                                   f->grad_s_factor, f->grad_alpha_factor, f->grad_color_factor,
                                   f->grad_high_order_color_factor, f->grad_xyz, f->grad_features, f->grad_xyz_visible,
                                   f->grad_features_visible, f->hook_compact, nullptr, nullptr, nullptr, nullptr, f->width,
                                   f->height, stream));
    return 0;
}

}  // extern "C"
