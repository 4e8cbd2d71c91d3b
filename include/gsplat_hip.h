/*
 * gsplat_hip.h -- C ABI of the MI355X (gfx950) differentiable 3D-Gaussian-splatting rasteriser.
 *
 * This is the drop-in boundary for the hot path of wanmeihuali/taichi_3d_gaussian_splatting:
 * each entry point replaces one stage of the reference operator
 *   RAS = taichi_3d_gaussian_splatting/GaussianPointCloudRasterisation.py
 * (kernel or torch glue, cited per function).  The reference has no FFI of its own (its kernels
 * are Taichi JIT functions called with torch tensors, RAS:848-997,1069-1100); a host binds these
 * symbols with ctypes/cffi and passes raw device pointers -- see INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the parameter name starts with `host_`;
 *  - the caller owns every buffer (the library never allocates device memory and keeps no
 *    global mutable state); scratch buffers are sized with the *_workspace_bytes queries;
 *  - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised, except in
 *    gs_read_counters (the two size read-backs the reference also performs, RAS:870,916);
 *  - return value: 0 on success, negative on error; gs_last_error() gives the (thread-local)
 *    message.  Nothing throws across the boundary;
 *  - fp32 everywhere, quaternions (x,y,z,w), row-major matrices, image memory [v,u,c].
 *
 * Packed per-visible-point record `attrs` (float[M][16], 64 B, 16-B aligned), produced by
 * gs_preprocess and gathered by the blend kernels with 16-B loads (rows 0 and 1 decide whether a list
 * entry belongs to a tile; both blend passes stage all four rows):
 *   [0] u  [1] v  [2] z (camera depth)  [3] qmax = 2 ln(255 opacity rescale) + 0.01 (exact-cull bound; +inf = off)
 *   [4] conic A  [5] conic B  [6] conic C  [7] 3-sigma radius   (UTL:257-272, RAS:311-315)
 *   [8] r  [9] g  [10] b  [11] opacity sigmoid(logit)            (RAS:299-310)
 *   [12] amp = opacity * rescale  [13] stop-bracket weight  [14] e_lo: exponents below it are skips for certain (csrc/gs_common.h)  [15] rescale (UTL:266)
 *        (alpha = amp * exp(e) with e the reference's exponent, evaluated in the reference's own operation order by each
 *        pass -- UTL:281-283 forward, UTL:336-339 backward; opacity and rescale also stay apart because the reference
 *        multiplies exp(e) by rescale (UTL:284) and then by the opacity (RAS:447): where a comparison with 1/255 or 1e-4
 *        falls within the proven distance between the two roundings the blend kernels re-evaluate it exactly as the reference
 *        does, csrc/gs_common.h "threshold decisions")
 * Lists.  Sort keys are emitted per BIN of (1 << bin_shift)^2 tiles (bin_shift = 2: 64 x 64 pixels; 0: the
 * reference's per-tile keys).  A blend workgroup (one 16 x 16 tile) walks its bin's depth-sorted list and keeps,
 * in order, the entries that belong to its tile: `filter` = GS_FILTER_BOX (the tile lies in the Gaussian's tile
 * box, RAS:81-103 -- mandatory for bin_shift > 0) | GS_FILTER_CULL (exact contribution test).  The sequence of
 * Gaussians a tile blends is the reference's per-tile sorted list (minus pairs that cannot contribute).
 * Tile-row ownership (image-space sharding): rows {tile_row_begin + k*tile_row_step} below tile_row_end
 * (pass the number of tile rows, or any larger value, for "no upper bound").
 * Backward accumulators `acc` (float[M][12]):
 *   [0..1] dL/duv  [2..4] dL/dcov(00,01,11)  [5..7] dL/drgb  [8] dL/dlogit
 *   [9] sum of |dL/duv| norms  [10] number of affected pixels (int32 bits)  [11] unused
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_FILTER_BOX 1
#define GS_FILTER_CULL 2
#define GS_TILE_WIDTH 16      /* RAS:27 */
#define GS_TILE_HEIGHT 16     /* RAS:28 */
#define GS_BOUNDARY_TILES 3   /* RAS:26 */
#define GS_ATTR_STRIDE 16
#define GS_ACC_STRIDE 12
#define GS_FEATURE_DIM 56     /* RAS:214-226 */

/* counters[] slots written by the library (int32, device) */
#define GS_COUNTER_NUM_VISIBLE 0   /* M */
#define GS_COUNTER_NUM_KEYS 1      /* K (saturates at INT32_MAX) */
#define GS_COUNTER_NUM_SLOTS 2     /* sum of num_overlap_tiles = number of (Gaussian, tile) slots */
#define GS_COUNTER_MAX_DEPTH_KEY 3 /* max over visible points of int32(z * depth_scale) (gs_preprocess) */
#define GS_NUM_COUNTERS 8

/* ABI version of this header.  A TUNING build of the library (measurement arms compiled in: -DGS_TUNING_BUILD=1,
 * tools/build_variants.sh) reports GS_ABI_VERSION + GS_ABI_TUNING_OFFSET, which the product loader refuses. */
#define GS_ABI_VERSION 36
#define GS_ABI_TUNING_OFFSET 1000

const char *gs_last_error(void);
int gs_abi_version(void);

/* Camera<-pointcloud pose from pointcloud<-camera pose.  Replaces inverse_SE3_qt_torch,
 * UTL:426-432 as called at RAS:845-846.  n_obj rows. */
int gs_pose_inverse(const float *q_pointcloud_camera, const float *t_pointcloud_camera,
                    float *q_camera_pointcloud, float *t_camera_pointcloud, int n_obj, void *stream);

/* Frustum test + order-preserving stream compaction.  Replaces filter_point_in_camera
 * (RAS:31-78) and the mask -> index torch glue (RAS:841-870).
 * out: mask int8[N], ids int32[N] (first M valid, ascending), counters[GS_COUNTER_NUM_VISIBLE]=M.
 * counters: int32[GS_NUM_COUNTERS]; ALL of them are reset to zero here (the first stage of a frame), so the caller
 * need not clear the buffer between frames. */
size_t gs_filter_workspace_bytes(int n_points);
int gs_filter_compact(const float *xyz, const int8_t *invalid_mask, const int32_t *object_id,
                      const float *intrinsics, const float *q_camera_pointcloud,
                      const float *t_camera_pointcloud, int n_points, float near_plane,
                      float far_plane, int width, int height, int8_t *mask, int32_t *ids,
                      int32_t *counters, void *workspace, void *stream);

/* gs_pose_inverse + gs_filter_compact in one launch chain: the filter inverts the poses itself (same arithmetic, same bits)
 * and leaves them in q_camera_pointcloud / t_camera_pointcloud for the later stages. */
int gs_filter_compact_from_poses(const float *xyz, const int8_t *invalid_mask, const int32_t *object_id,
                                 const float *intrinsics, const float *q_pointcloud_camera,
                                 const float *t_pointcloud_camera, int n_objects, float *q_camera_pointcloud,
                                 float *t_camera_pointcloud, int n_points, float near_plane, float far_plane,
                                 int width, int height, int8_t *mask, int32_t *ids, int32_t *counters,
                                 void *workspace, void *stream);

/* Blocking read of the device counters into host memory (host sync, as RAS:870,916). */
int gs_read_counters(const int32_t *counters, int32_t *host_counters, int n, void *stream);

/* Per-visible-point projection.  Replaces generate_point_attributes_in_camera_plane
 * (RAS:239-315, incl. the in-place quaternion normalisation RAS:196-205) and
 * generate_num_overlap_tiles (RAS:106-128).  Also emits per-256-point partial sums for the two scans:
 * block_sums (of num_owned_tiles -> key offsets) and block_sums_full (of num_overlap_tiles -> slot
 * offsets of the backward pass), both int32[ceil(M/256)].  tile_row_begin/tile_row_step
 * restrict the tile box to the tile rows  r = begin + k*step  owned by this GPU (1 GPU: 0,1);
 * num_overlap_tiles (hook output, RAS:1136) is always the full box count of the reference and
 * num_keys the number of sort keys this GPU will emit (one per bin reached in an owned tile row; the one that is
 * scanned).  exact_tile_cull != 0 drops (bin, Gaussian) pairs whose alpha is below the 1/255 skip threshold
 * (RAS:451) on every pixel of the bin's tiles inside the Gaussian's box: such pairs never change a pixel, so every
 * operator output is unchanged while the lists that are sorted get shorter; it also stores the bound in attrs[3] for
 * the per-tile test of the blend kernels.  0 = every bin of the box, attrs[3] = +inf.
 * n_visible_on_device != 0: n_visible is only the CAPACITY of ids/outputs (e.g. N) and the kernel takes
 * the actual count from counters[GS_COUNTER_NUM_VISIBLE] as written by gs_filter_compact on the same
 * stream -- the host then needs a single size read-back (M, K, slots together) instead of two.
 * counters (may be NULL; zeroed by gs_filter_compact, else by the caller): counters[GS_COUNTER_MAX_DEPTH_KEY]
 * receives the largest quantised depth int32(z*depth_scale) on screen, so that the host can size the
 * key's depth field to the bits in use (fewer radix passes than the far_plane*depth_scale bound).
 * always_store_rotation: the normalised quaternion is written back only when it differs from what is stored (the
 * contents of `features` are the same either way); != 0 writes it always -- what every training iteration pays, since
 * the optimiser has just moved q: lets a benchmark on a static scene include that traffic. */
int gs_preprocess(const float *xyz, float *features, const int32_t *object_id,
                  const float *intrinsics, const float *q_camera_pointcloud,
                  const float *t_camera_pointcloud, const int32_t *ids, int n_visible,
                  int n_visible_on_device, int width, int height, int tile_row_begin,
                  int tile_row_step, int tile_row_end, int bin_shift, int exact_tile_cull,
                  int always_store_rotation, float depth_scale, int32_t *counters, float *attrs,
                  int32_t *num_overlap_tiles, int32_t *num_keys, int32_t *block_sums,
                  int32_t *block_sums_full, void *stream);

/* The same projection with the colour LEFT OUT (row 2 of a record = 0, 0, 0, opacity), and the colours on their own:
 * gs_view_colours writes floats 8..10 of the records of the Gaussians with num_keys > 0 (RAS:280-282,302-310) -- the same
 * device code as inside gs_preprocess, bit-identical records from either way.  Nothing between projection and blend reads
 * a colour, so gs_frame_forward runs gs_view_colours on a second stream beside key generation, sort and ranges
 * (GS_FWD_COLOUR_ASYNC): a streaming pass over 192 B per Gaussian next to latency-bound launches. */
int gs_preprocess_geometry(const float *xyz, float *features, const int32_t *object_id,
                  const float *intrinsics, const float *q_camera_pointcloud,
                  const float *t_camera_pointcloud, const int32_t *ids, int n_visible,
                  int n_visible_on_device, int width, int height, int tile_row_begin,
                  int tile_row_step, int tile_row_end, int bin_shift, int exact_tile_cull,
                  int always_store_rotation, float depth_scale, int32_t *counters, float *attrs,
                  int32_t *num_overlap_tiles, int32_t *num_keys, int32_t *block_sums,
                  int32_t *block_sums_full, void *stream);
int gs_view_colours(const float *xyz, const float *features, const int32_t *object_id, const float *q_camera_pointcloud,
                    const float *t_camera_pointcloud, const int32_t *ids, int n_visible, int n_visible_on_device,
                    const int32_t *counters, const int32_t *num_keys, float *attrs, void *stream);

/* Exclusive scan of per-block sums (in place) and total -> counters[counter_slot]
 * (GS_COUNTER_NUM_KEYS for block_sums, GS_COUNTER_NUM_SLOTS for block_sums_full).
 * Replaces torch.cumsum/cat, RAS:913-922. */
int gs_scan_block_sums(int32_t *block_sums, int n_blocks, int32_t *counters, int counter_slot,
                       void *stream);
/* Both scans of a frame in one launch: block_sums -> counters[GS_COUNTER_NUM_KEYS],
 * block_sums_full -> counters[GS_COUNTER_NUM_SLOTS]. */
int gs_scan_block_sums2(int32_t *block_sums, int32_t *block_sums_full, int n_blocks, int32_t *counters,
                        void *stream);

/* The same, and the frame's sizes {M, K, slots, depth range} are stored into host_counters_mapped[GS_COUNTER_*] by the
 * kernel itself: host_counters_mapped must be pinned host memory the device can address (hipHostMalloc'd / a pinned torch
 * tensor: the host pointer is valid on the device under ROCm's unified addressing); the values are visible to the host once
 * the launch has completed (record an event behind it).  Saves the copy launch of gs_read_counters_async. */
int gs_scan_block_sums2_to_host(int32_t *block_sums, int32_t *block_sums_full, int n_blocks, int32_t *counters,
                                int32_t *host_counters_mapped, void *stream);

/* The same without the event: every size travels as ONE 64-bit word {stamp << 32 | value} stored with a system-scope
 * store -- host_words[GS_COUNTER_NUM_VISIBLE .. GS_COUNTER_MAX_DEPTH_KEY], uint64[4] -- so the host needs no completion
 * signal: it reads the four words until each carries the stamp it handed out for this frame (gs_wait_stamped_sizes).
 * Recording an event behind the scan costs the stream ~6 us per frame (the next kernel waits for the signal packet);
 * a word is valid or not on its own, so no ordering between the stores is needed.  host_words: host-coherent pinned
 * memory (gs_host_alloc_coherent), 8-byte aligned; stamp != 0 and different from the previous frame's. */
int gs_scan_block_sums2_stamped(int32_t *block_sums, int32_t *block_sums_full, int n_blocks, int32_t *counters,
                                void *host_words, uint32_t stamp, void *stream);
/* Host side of the above: spins (the calling thread only; no HIP call) until the four words carry `stamp`, then
 * sizes4 = {M, K, slots, max depth key}.  Returns 0, or 1 when timeout_us passed first (nothing written). */
int gs_wait_stamped_sizes(const void *host_words, uint32_t stamp, int64_t timeout_us, int32_t *sizes4);
/* Pinned host memory that is coherent with the device while kernels run (hipHostMallocCoherent | Mapped). */
int gs_host_alloc_coherent(int64_t bytes, void **out);
int gs_host_free(void *p);

/* Sort-key generation.  Replaces generate_point_sort_key_by_num_overlap_tiles (RAS:131-172).
 * payload[k] = offset into the visible list.  Key layout:
 *   key_depth_bits == 0 : uint64 keys[k] = (bin_id << 32) + int32(z * depth_scale)   (reference layout; with
 *                         bin_shift = 0 exactly the reference's keys)
 *   key_depth_bits  > 0 : uint32 keys[k] = (bin_id << key_depth_bits) | int32(z * depth_scale)
 *                         same order, valid when 0 <= z*depth_scale < 2^key_depth_bits and the bin
 *                         field fits the remaining bits (halves the sort traffic).
 * bin_id = bin_u + bin_v * ceil(tiles_per_row / 2^bin_shift).  Ownership, bin_shift and exact_tile_cull must be
 * the values passed to gs_preprocess.
 * Optionally (slot_offsets may be NULL: inference) writes slot_offsets int32[M] = exclusive scan of num_overlap_tiles (block_offsets_full = scanned
 * block_sums_full): slot_offsets[i] + (t1v-t0v)*(tile_u-t0u) + (tile_v-t0v) is the reference's key index
 * of the (Gaussian i, tile) pair (RAS:163-166) and addresses its partial-gradient slot in the backward.
 * Sizes that are still on their way to the host: with counters != NULL (the array of gs_filter_compact /
 * gs_scan_block_sums2, same stream) n_visible is only the CAPACITY of the per-point arrays and the kernel takes the
 * count from counters[GS_COUNTER_NUM_VISIBLE]; at most n_keys_capacity keys are written (the host compares
 * counters[GS_COUNTER_NUM_KEYS] with it once the read-back has arrived and redoes the frame if it did not fit). */
int gs_make_keys(const float *attrs, const int32_t *num_keys, const int32_t *block_offsets,
                 int n_visible, const int32_t *counters, int64_t n_keys_capacity, int width,
                 int height, int tile_row_begin, int tile_row_step,
                 int tile_row_end, int bin_shift, int exact_tile_cull, int key_depth_bits,
                 float depth_scale, void *keys,
                 int32_t *payload, const int32_t *num_overlap_tiles,
                 const int32_t *block_offsets_full, int32_t *slot_offsets, void *stream);

/* Stable radix sort of (key, payload) pairs.  Replaces torch.sort + gather (RAS:947-950) with
 * the stable tie rule.  (64-bit keys and large inputs: LSD passes of eight bits, three launches each; 32-bit keys up to
 * ~4 M pairs: MSD-first -- one such pass on the top eight or nine bits, then every bucket sorted by its remaining bits
 * inside one workgroup's LDS.  Same result.)  key_depth_bits selects the key layout (see gs_make_keys).  64-bit layout:
 * only the bit ranges [0,depth_bits) and [32,32+tile_bits) are sorted; depth_bits = 64 sorts the
 * whole key as a signed int64.  32-bit layout: bits [0, key_depth_bits+tile_bits).
 * keys_alt/payload_alt are ping-pong buffers of the same size.  Returns 0 when the sorted pairs are in
 * keys/payload, or -- only if allow_result_in_alt != 0 -- 1 when they are in keys_alt/payload_alt (odd
 * number of passes; saves the copy back).  Negative on error.
 * n_keys_device (may be NULL): device address of the actual number of pairs (e.g. counters + GS_COUNTER_NUM_KEYS);
 * n_keys is then the capacity that sizes grids and workspace, and min(*n_keys_device, n_keys) pairs are sorted. */
size_t gs_sort_workspace_bytes(int64_t n_keys);
int gs_sort_pairs(void *keys, int32_t *payload, void *keys_alt, int32_t *payload_alt,
                  int64_t n_keys, const int32_t *n_keys_device, int key_depth_bits, int depth_bits,
                  int tile_bits, int allow_result_in_alt, void *workspace, void *stream);
/* The same sort; its first launch also zero-fills
 * `also_zero` (16-byte aligned, also_zero_bytes % 16 == 0; may be NULL / 0) -- the frame's list ranges ride along instead
 * of a fill launch of their own (gs_tile_ranges with ranges_are_zeroed).
 * bins_in_any_order != 0 (32-bit keys): the caller needs what the frame needs -- the pairs of one bin (bits
 * [key_depth_bits, key_depth_bits + tile_bits) of the key) CONTIGUOUS and in ascending, stable order within the bin --
 * but not the bins themselves in ascending order: gs_tile_ranges and the blend kernels look a bin's list up by its range.
 * The MSD-first sort then partitions by the LOWEST bits of the bin field, so that a bucket is every 256th (512th) bin of
 * the frame instead of a run of adjacent ones: even buckets whatever the density of the scene (adjacent bins: 79 of the
 * headline frame's 255 buckets exceeded a workgroup's LDS).  Order of the result: (bin mod 2^m, bin, depth, input).
 * tile_start / tile_end (int32[n_tiles] each, may be NULL; zero-filled by the caller or through also_zero): when the
 * MSD-first path runs and its buckets are made of whole bins, the bucket-local sort writes the bins' [start, end) ranges
 * (gs_tile_ranges' output) on its way out, and the return value has bit 1 set: no ranges launch is needed.
 * Returns (>= 0): bit 0 = the sorted pairs are in keys_alt / payload_alt, bit 1 = the ranges were written. */
int gs_sort_pairs_and_zero(void *keys, int32_t *payload, void *keys_alt, int32_t *payload_alt,
                           int64_t n_keys, const int32_t *n_keys_device, int key_depth_bits, int depth_bits,
                           int tile_bits, int allow_result_in_alt, int bins_in_any_order, void *workspace,
                           void *also_zero, size_t also_zero_bytes, int32_t *tile_start, int32_t *tile_end,
                           int n_tiles, void *stream);

/* Per-bin [start,end) ranges (n_tiles = number of bins; per tile with bin_shift = 0).  Replaces
 * find_tile_start_and_end (RAS:175-193) including the zero-initialisation of RAS:954-957.  Needs every bin's keys
 * contiguous, not the bins in ascending order (gs_sort_pairs_and_zero, bins_in_any_order). */
int gs_tile_ranges(const void *keys_sorted, int64_t n_keys, const int32_t *n_keys_device,
                   int key_depth_bits, int32_t *tile_start, int32_t *tile_end, int n_tiles,
                   void *stream);

/* The same with ranges_are_zeroed != 0: the caller has zero-filled both arrays on this stream already (e.g.
 * gs_sort_pairs_and_zero) -- no fill is issued. */
int gs_tile_ranges_prezeroed(const void *keys_sorted, int64_t n_keys, const int32_t *n_keys_device,
                             int key_depth_bits, int32_t *tile_start, int32_t *tile_end, int n_tiles,
                             int ranges_are_zeroed, void *stream);

/* Enqueues a copy of the device counters into PINNED host memory (no synchronisation): the host keeps launching
 * and waits for its own event when it needs the sizes. */
int gs_read_counters_async(const int32_t *counters, int32_t *host_counters_pinned, int n, void *stream);

/* Front-to-back alpha blending.  Replaces gaussian_point_rasterisation (RAS:318-485).
 * bin_start/bin_end: the ranges of gs_tile_ranges; bin_shift/filter: see "Lists" above.
 * Tiles in tile rows this GPU does not own are skipped (their pixels are left untouched).
 * flags = 0: all five outputs are written for owned tiles (also when n_keys == 0: zeros).
 * GS_BLEND_RGB_ONLY: the reference's rgb_only (RAS:464-469,478-484): depth and valid_count are neither computed nor
 *   written (may be NULL).  GS_BLEND_NO_STATE: acc_alpha and last_effective -- the state only the backward pass
 *   reads -- are neither tracked nor written (may be NULL): the inference path.  The two flags combine.
 * last_effective = 1 + list position of the last Gaussian blended into the pixel (bin_start of its bin if none).
 * debug_pixel_hits (may be NULL; tests): uint32[H][W][2] = per pixel {number of blended Gaussians, wrap-around sum
 *   of (payload + 1) * 2654435761}; gs_blend_backward fills the same record for the pairs IT treats as blended.
 * Dispatch order.  Tiles differ in work by an order of magnitude; handed to the hardware in image order the launch ends
 * in a long tail of half-empty CUs.  tile_order (may be NULL): int32[number of owned tiles] scratch; when given, the
 * library fills it with the owned tiles sorted by list length, longest first (one small launch), and dispatches the
 * tiles in that order.  tile_work (may be NULL; needs the state outputs): int32[number of owned tiles], receives per
 * owned tile (n-th in row-major order over the owned rows) the number of list positions gs_blend_backward will walk
 * for it -- pass it on to gs_blend_backward.  Results never depend on the order.
 * Walked lists (binned layouts with the state outputs; both NULL otherwise).  walked_list: int32[n_keys << 2 bin_shift],
 * walked_start: int32[number of tiles of the image].  A tile of a binned layout keeps, while it stages its bin's list, the
 * entries that belong to it; with these buffers it also writes them out -- tile t (0 .. 4^bin_shift - 1, row-major in the bin)
 * of a bin with range [s, e) owns positions (s << 2 bin_shift) + t (e - s) onwards -- stores that first position in
 * walked_start[tile] and reports last_effective as positions in walked_list.  gs_blend_backward is then called with
 * (bin_start = walked_start, payload = walked_list, bin_shift = 0, filter = 0): it walks a plain per-tile list and never
 * examines the bin's other entries again (it never goes beyond what the forward pass walked). */
#define GS_BLEND_RGB_ONLY 1
#define GS_BLEND_NO_STATE 2
#define GS_BLEND_TWO_WAVES 4    /* both blend passes: always the two-waves-per-tile kernels (two pixels per lane) */
#define GS_BLEND_FOUR_WAVES 8   /* both blend passes: the four-waves-per-tile kernels (one pixel per lane) whenever the lists
                                   are per-tile lists taken as they are (bin_shift 0, filter 0).  Default (neither flag):
                                   four waves when at most 3840 tiles are rendered -- a grid that cannot fill the chip with two.
                                   Image, depth, counts, state and hit sets are bit-identical between the two forms. */
#define GS_BLEND_ONE_WAVE 16    /* backward pass: the one-wave-per-tile kernel (four pixels per lane, cross-lane sums through
                                   LDS) whenever the lists are per-tile lists taken as they are.  Default (no flag): larger grids
                                   than the four-wave form's.  Decisions, the |grad uv| image and debug hashes are bit-identical
                                   to the two-wave kernel's; slot sums add the same per-pixel terms in another order. */
#define GS_BLEND_SKEWED_WALKS 64    /* backward pass, two-wave kernel: the walk lengths of this frame are skewed (a few tiles walk many
                                   times the mean: trained scenes) -- the form of the kernel with the shorter dependent chain per hit
                                   entry (cross-lane sums in registers) instead of the one with the higher throughput (through LDS).
                                   Same decisions, sums equal to rounding; a hint, never needed for correctness */
#define GS_BLEND_SPLIT_FORWARD 32   /* gs_blend_forward_split with a workspace: GS_MAX_FORWARD_SPLIT workgroups per tile whatever the
                                   grid size (tests, measurements; implies the four-wave form on per-tile lists) */
int gs_blend_forward(const int32_t *bin_start, const int32_t *bin_end, const int32_t *payload,
                     const float *attrs, int width, int height, int tile_row_begin,
                     int tile_row_step, int tile_row_end, int bin_shift, int filter, float *image,
                     float *depth, float *acc_alpha, int32_t *last_effective, int32_t *valid_count,
                     int flags, uint32_t *debug_pixel_hits, int32_t *tile_order, int32_t *tile_work,
                     int32_t *walked_list, int32_t *walked_start, void *stream);

/* Backward per-pixel pass.  Replaces the pixel loop of gaussian_point_rasterisation_backward
 * (RAS:531-705) WITHOUT its global atomics (RAS:674-696): the partial sums of a (Gaussian, tile) pair
 * are stored as one 48-B record (layout of `acc`) in partials[slot] and slot_flags[slot] is raised
 * (slot: see gs_make_keys; n_slots = counters[GS_COUNTER_NUM_SLOTS]; slot_flags is zeroed by the library and must be
 * allocated with n_slots rounded UP to a multiple of 16 bytes -- the fill is issued as one aligned memset).
 * The walk starts at the tile-wide maximum of last_effective and runs down to bin_start; bin_shift and filter
 * must be the forward's.  alpha is evaluated by the same device function as in gs_blend_forward and the staging
 * filter is the same function on the same records, so both passes treat exactly the same (pixel, Gaussian)
 * pairs as blended.  debug_pixel_hits: see gs_blend_forward.  flags: GS_BLEND_TWO_WAVES / GS_BLEND_FOUR_WAVES or 0
 * (see above; the slot sums of the two forms add the same per-pixel terms in a different order).
 * tile_work / tile_order (may be NULL): dispatch order, see gs_blend_forward.  tile_work (the forward's record) given:
 * tile_order is scratch of the same size and is filled with the tiles sorted by walk length, longest first.  Only
 * tile_order given: it is taken as the caller's permutation of 0 .. n-1 (n-th owned tile in row-major order). */
int gs_blend_backward(const int32_t *bin_start, const int32_t *payload, const float *attrs,
                      const float *grad_image, const float *acc_alpha, const int32_t *last_effective,
                      const int32_t *slot_offsets, int64_t n_slots, int width, int height,
                      int tile_row_begin, int tile_row_step, int tile_row_end, int bin_shift,
                      int filter, float *partials, uint8_t *slot_flags, float *magnitude_image,
                      uint32_t *debug_pixel_hits, int flags, const int32_t *tile_work, int32_t *tile_order,
                      void *stream);


/* List splitting for grids that cannot fill the chip (at most 1024 rendered tiles with per-tile lists taken as they are:
 * bin_shift 0, filter 0 -- what small frames use): the forward pass leaves every pixel's transmittance and colour (and
 * what the colour sums rounded away) at every 128th position of its tile's list in `boundary_states`
 * (gs_blend_boundary_bytes(list_length, width, height) bytes; list_length = the payload's capacity, the SAME value in
 * both calls), and the backward pass gives a tile up to GS_MAX_BACKWARD_SPLIT
 * workgroups, each starting from such a state (RAS:558-704 cut at list positions; `image` = the forward's output image).
 * split_workspace: gs_blend_split_workspace_bytes(width, height) bytes, ZERO when first used; the library leaves its
 * counters zero again.  Slot records stay bitwise reproducible; they differ from the un-split ones in rounding only.
 * With boundary_states / image / split_workspace NULL these are gs_blend_forward / gs_blend_backward. */
#define GS_MAX_BACKWARD_SPLIT 4
/* The forward pass of a still smaller grid is split too (round 6: gs_blend_forward_split; at most GS_FORWARD_SPLIT_TILES
 * rendered tiles, per-tile lists taken as they are):
 * a probe launch leaves, per pixel and list segment, the product of (1 - alpha) over the segment's hits; every segment then
 * blends its share of the list starting from the product of the segments in front of it, with the stop rule (RAS:458-460)
 * applied against that transmittance -- every 1/255 and 1e-4 decision is still taken as the reference takes it, colours,
 * depth and state equal the un-split ones to rounding -- and a third launch adds the segments' results in order.
 * forward_split_workspace: gs_blend_forward_split_workspace_bytes(width, height) bytes of scratch, dead when the call's
 * work has completed; NULL = un-split.  debug_pixel_hits must be ZERO on entry when the forward is split. */
#define GS_MAX_FORWARD_SPLIT 4
#define GS_FORWARD_SPLIT_TILES 320   /* grids of at most this many rendered tiles are split (four workgroups per tile) */
size_t gs_blend_forward_split_workspace_bytes(int width, int height);
size_t gs_blend_boundary_bytes(int64_t list_length, int width, int height);
/* Path statistics of the two-waves-per-tile blend kernels: counted only in a tuning build (-DGS_STATS=1, tools/blend_stats.py);
 * the product build leaves them zero.  Synchronises `stream`, copies the GS_BLEND_STATS counters out and optionally clears
 * them; returns 1 from a counting build, 0 from the product build, negative on error.
 *   ENTRIES       (wave, list entry) visits whose alpha was evaluated          HIT_ENTRIES  visits that ran the hit path
 *   HIT_PIXELS    pixels blended over those visits (of 128 per visit)          HIT_LANES    lanes with at least one of their two
 *   CAREFUL_ENTRIES / BRACKETED  visits handed to the exact-decision path      EXACT_ALPHA  ... that evaluated the reference's expression
 *   REPLAYS       pixel histories replayed for a T' next to 1e-4               REPLAY_ENTRIES  list positions those replays walked */
#define GS_BLEND_STATS 16
#define GS_STAT_FWD_ENTRIES 0
#define GS_STAT_FWD_HIT_ENTRIES 1
#define GS_STAT_FWD_HIT_PIXELS 2
#define GS_STAT_FWD_HIT_LANES 3
#define GS_STAT_FWD_CAREFUL_ENTRIES 4
#define GS_STAT_FWD_EXACT_ALPHA 5
#define GS_STAT_FWD_REPLAYS 6
#define GS_STAT_FWD_REPLAY_ENTRIES 7
#define GS_STAT_BWD_ENTRIES 8
#define GS_STAT_BWD_HIT_ENTRIES 9
#define GS_STAT_BWD_HIT_PIXELS 10
#define GS_STAT_BWD_HIT_LANES 11
#define GS_STAT_BWD_BRACKETED 12
#define GS_STAT_BWD_EXACT_ALPHA 13
#define GS_STAT_FWD_HIT_BLOCKS 14   /* sum over hit visits of the 8 x 8-pixel blocks (of the wave's two) that hold a hit pixel */
#define GS_STAT_BWD_HIT_BLOCKS 15
int gs_blend_read_stats(uint64_t *counters, int clear, void *stream);
size_t gs_blend_split_workspace_bytes(int width, int height);
int gs_blend_forward_with_boundaries(const int32_t *bin_start, const int32_t *bin_end, const int32_t *payload,
                                     const float *attrs, int width, int height, int tile_row_begin,
                                     int tile_row_step, int tile_row_end, int bin_shift, int filter, float *image,
                                     float *depth, float *acc_alpha, int32_t *last_effective, int32_t *valid_count,
                                     int flags, uint32_t *debug_pixel_hits, int32_t *tile_order, int32_t *tile_work,
                                     int32_t *walked_list, int32_t *walked_start, float *boundary_states,
                                     int64_t list_length, void *stream);
int gs_blend_forward_split(const int32_t *bin_start, const int32_t *bin_end, const int32_t *payload,
                           const float *attrs, int width, int height, int tile_row_begin,
                           int tile_row_step, int tile_row_end, int bin_shift, int filter, float *image,
                           float *depth, float *acc_alpha, int32_t *last_effective, int32_t *valid_count,
                           int flags, uint32_t *debug_pixel_hits, int32_t *tile_order, int32_t *tile_work,
                           int32_t *walked_list, int32_t *walked_start, float *boundary_states,
                           int64_t list_length, void *forward_split_workspace, void *stream);
int gs_blend_backward_split(const int32_t *bin_start, const int32_t *payload, const float *attrs,
                            const float *grad_image, const float *acc_alpha, const int32_t *last_effective,
                            const int32_t *slot_offsets, int64_t n_slots, int width, int height,
                            int tile_row_begin, int tile_row_step, int tile_row_end, int bin_shift, int filter,
                            float *partials, uint8_t *slot_flags, float *magnitude_image,
                            uint32_t *debug_pixel_hits, int flags, const int32_t *tile_work, int32_t *tile_order,
                            const float *image, const float *boundary_states, int64_t list_length,
                            void *split_workspace, void *stream);

/* Per-Gaussian sum of its flagged slots, in slot order (bitwise reproducible), into acc float[M][12].
 * Replaces the accumulation side of the reference's atomics (RAS:674-696).
 * attrs (may be NULL) + width/height: the packed records of gs_preprocess; with them a Gaussian of many slots is only
 * looked at where its alpha >= 1/255 level set can reach (a needle's tile box is mostly empty) -- same sums, fewer
 * flags read.  Records written with exact_tile_cull = 0 (attrs[3] = +inf) switch this off by themselves. */
int gs_reduce_partials(const int32_t *slot_offsets, const int32_t *num_overlap_tiles,
                       const uint8_t *slot_flags, const float *partials, int n_visible, float *acc,
                       const int32_t *num_keys /* may be NULL; num_keys[i] == 0: nothing to sum */,
                       int64_t n_slots_hint /* total number of slots (picks the lanes-per-Gaussian variant; 0 = default) */,
                       const float *attrs, int width, int height, void *stream);

/* Multi-GPU (tile-row sharding): sparse exchange of the accumulators.  The reference has no counterpart (single GPU);
 * the stage sits between gs_reduce_partials and gs_point_backward when the frame is sharded over several GPUs.
 * gs_compact_rows: the rows of acc this GPU produced (num_keys[i] > 0: it emitted a sort key for Gaussian i) as an
 *   ascending list -- ids int32[capacity], rows float[capacity][12]; *count = number of produced rows (entries past
 *   `capacity` are dropped, the caller compares).  workspace: gs_compact_rows_workspace_bytes(n_visible).
 * gs_merge_rows: `world` (<= 128) such lists, gathered from all ranks into one buffer -- list g starts at word g * list_stride_words
 *   and holds `capacity` ids followed by `capacity` rows (capacity % 4 == 0) with counts[g] valid entries -- are summed
 *   into the dense acc float[n_visible][12] (rows no list mentions: zeros): the lists are added in rank order, the same
 *   additions in the same order on every rank, so replicated gradients stay bit-identical across ranks. */
size_t gs_compact_rows_workspace_bytes(int n_visible);
int gs_compact_rows(const float *acc, const int32_t *num_keys, int n_visible, int capacity, int32_t *ids,
                    float *rows, int32_t *count, void *workspace, void *stream);
int gs_merge_rows(const int32_t *lists, int64_t list_stride_words, int capacity, const int32_t *counts, int world,
                  int n_visible, float *acc, void *stream);


/* ---- Multi-GPU with OWNER-SHARDED Gaussians (routed exchange on top of the tile-row bands; no reference counterpart).
 * Rank g owns a contiguous block of point-cloud rows, projects only those (gs_filter_compact / gs_preprocess on its block
 * with full-image ownership), and sends each projected 64-B record to the band(s) whose tile rows its tile box reaches:
 *   gs_route_count   : counts[b] = records this rank sends to band b.  Bands = equal blocks of rows_per_band tile rows, or
 *                      -- band_row_bounds != NULL, a HOST array of world + 1 non-decreasing tile rows from 0 to the number
 *                      of tile rows -- band b = rows [bounds[b], bounds[b + 1]): boundaries that balance the bands' work
 *                      on a scene whose Gaussians crowd into some rows (a band may be empty)
 *   gs_route_scatter : send float[world][capacity + 1][16]: chunk b = header slot {count as int32 bits} + the records for
 *                      band b in visible-list order; pos int32[world][n_visible_capacity] = slot (0-based, without the
 *                      header) of record i in chunk b, -1 = not sent there.  The chunks are exchanged with one all-to-all
 *                      (equal splits).  Only records with num_keys[i] > 0 travel (the others are incomplete).
 *   gs_count_keys    : the receiving band's per-record counts (sort keys on its rows, reference box count, depth range)
 *                      over the received buffer taken as an attrs array of world x (capacity + 1) records, headers and
 *                      unused slots counting as Gaussians without keys -- then gs_scan_block_sums2 / gs_make_keys / ... run
 *                      on that buffer unchanged; counters must be zeroed by the caller; counters[GS_COUNTER_NUM_VISIBLE]
 *                      is set to n_slots.
 *   gs_gather_returned_rows : backward.  The band's accumulators float[world x (capacity + 1)][12] go back through the
 *                      same all-to-all; the owner sums, per record, the rows returned by its bands in band order.
 * workspace: gs_route_workspace_bytes(n_visible_capacity, world), shared by gs_route_count and the gs_route_scatter that
 * follows it (the scatter reads the offsets the count left there). */
size_t gs_route_workspace_bytes(int n_visible_capacity, int world);
int gs_route_count(const float *attrs, const int32_t *num_keys, int n_visible_capacity, const int32_t *counters,
                   int width, int height, int rows_per_band, int world, const int32_t *band_row_bounds, int32_t *counts,
                   void *workspace, void *stream);
int gs_route_scatter(const float *attrs, const int32_t *num_keys, int n_visible_capacity, const int32_t *counters,
                     int width, int height, int rows_per_band, int world, const int32_t *band_row_bounds, int capacity,
                     const int32_t *counts,
                     float *send, int32_t *pos, void *workspace, void *stream);
int gs_count_keys(const float *records, int n_slots, int chunk_slots, int width, int height, int tile_row_begin,
                  int tile_row_step, int tile_row_end, int bin_shift, int exact_tile_cull, float depth_scale,
                  int32_t *counters, int32_t *num_overlap_tiles, int32_t *num_keys, int32_t *block_sums,
                  int32_t *block_sums_full, void *stream);
int gs_gather_returned_rows(const float *returned, const int32_t *pos, int n_visible, int n_visible_capacity, int world,
                            int capacity, float *acc, void *stream);

/* Backward per-point pass + gradient post-processing.  Replaces the per-point loop of
 * gaussian_point_rasterisation_backward (RAS:707-772), the dense zero-initialisation
 * (RAS:1051-1053), _clear_grad_by_color_max_sh_band (RAS:1167-1182) and the factor scaling
 * (RAS:1105-1125; factors are the frozen class attributes RAS:782-786, passed explicitly).
 * grad_xyz float[N][3] and grad_features float[N][56] are fully written (zeros for rows that
 * are not visible; with visible_mask = the int8[N] mask of gs_filter_compact only those rows are
 * zero-filled, with NULL both arrays are memset first).  The optional compact outputs (may be NULL) are the hook gathers of
 * RAS:1130-1134: grad_xyz_visible float[M][3], grad_features_visible float[M][56].
 * attrs = the packed records of gs_preprocess (float[M][16]): the colour chain takes sigmoid(SH.Y) from row 2 instead
 * of re-reading the 192 SH bytes of every feature row; num_owned_tiles (int32[M], may be NULL = all rows complete) marks
 * the records whose rows 1..3 were not written (no key emitted on this GPU): for those the colour is re-evaluated
 * from the features, and only if their accumulated colour gradient is non-zero (tile-row sharding).
 * hook_compact (may be NULL): float[7*M], the remaining M-indexed hook fields as consecutive planes --
 * grad_viewspace [M][2] | magnitude_grad_viewspace [M] | num_affected_pixels [M] (int32 bits) | point_depth [M] |
 * point_uv_in_camera [M][2] (RAS:1130-1139) -- written here instead of five strided column copies of acc / attrs.
 * Fused slot reduction: with acc == NULL the accumulator record of every Gaussian is summed here from the slot records
 * of gs_blend_backward (slot_offsets, num_overlap_tiles, slot_flags, partials, image size -- the arguments of
 * gs_reduce_partials; num_owned_tiles doubles as its num_keys): the same code in the same order as gs_reduce_partials,
 * hence the same bits, without the launch and without writing and re-reading acc.  With acc != NULL (a multi-GPU run
 * all-reduces it first) the six trailing arguments are ignored. */
int gs_point_backward(const float *xyz, const float *features, const int32_t *object_id,
                      const float *intrinsics, const float *q_camera_pointcloud,
                      const float *t_camera_pointcloud, const float *t_pointcloud_camera,
                      const int32_t *ids, const int8_t *visible_mask, int n_visible, int n_points,
                      const float *acc, const float *attrs, const int32_t *num_owned_tiles,
                      int color_max_sh_band, float grad_q_factor, float grad_s_factor,
                      float grad_alpha_factor, float grad_color_factor,
                      float grad_high_order_color_factor, float *grad_xyz, float *grad_features,
                      float *grad_xyz_visible, float *grad_features_visible, float *hook_compact,
                      const int32_t *slot_offsets, const int32_t *num_overlap_tiles, const uint8_t *slot_flags,
                      const float *partials, int width, int height, void *stream);


/* ---- One entry point per pass (round 4).  A frame is ~25 launches forward and ~5 backward; issued one by one through a
 * foreign-function interface the HOST becomes the bound on small frames (the reference's 4x / 2x down-sampled first
 * iterations, TRN:139-148; a rank of a sharded frame).  GsFrame carries every buffer and option of a frame -- the
 * caller still owns all of them -- and gs_frame_forward / gs_frame_backward run the stages named in `stages`, in the
 * reference's order, by calling the stage entry points above: same kernels, same arguments, same results as the
 * stage-by-stage calls.  Sizes that live on the device (M, K) are taken from `counters` exactly as the stage functions do
 * (n_visible_on_device, n_keys_device): the caller sizes the key buffers (n_keys_capacity) and the key layout
 * (key_depth_bits / depth_bits / tile_bits) from the previous frame, reads host_counters_pinned after `size_event`
 * (recorded right behind the asynchronous copy of the counters; or, with size_stamp, once the stamped words have arrived:
 * gs_wait_stamped_sizes) and redoes the list stages when the frame did not fit. */
#define GS_FWD_POSE_INVERSE   (1u << 0)   /* gs_pose_inverse                    UTL:426-432                     */
#define GS_FWD_FILTER_COMPACT (1u << 1)   /* gs_filter_compact                  RAS:31-78, 841-870              */
#define GS_FWD_PREPROCESS     (1u << 2)   /* gs_preprocess                      RAS:239-315, 106-128            */
#define GS_FWD_ROUTE_COUNT    (1u << 3)   /* gs_route_count   (owner-sharded)                                   */
#define GS_FWD_ROUTE_SCATTER  (1u << 4)   /* gs_route_scatter (owner-sharded)                                   */
#define GS_FWD_COUNT_KEYS     (1u << 5)   /* gs_count_keys    (owner-sharded: attrs = the received records)     */
#define GS_FWD_SCAN           (1u << 6)   /* gs_scan_block_sums2                RAS:913-922                     */
#define GS_FWD_READ_SIZES     (1u << 7)   /* gs_read_counters_async + event     (the size reads of RAS:870,916) */
#define GS_FWD_MAKE_KEYS      (1u << 8)   /* gs_make_keys                       RAS:131-172                     */
#define GS_FWD_SORT           (1u << 9)   /* gs_sort_pairs_and_zero             RAS:947-950                     */
#define GS_FWD_RANGES         (1u << 10)  /* gs_tile_ranges                     RAS:175-193                     */
#define GS_FWD_BLEND          (1u << 11)  /* gs_blend_forward                   RAS:318-485                     */
#define GS_FWD_COLOUR_ASYNC   (1u << 12)  /* with GS_FWD_PREPROCESS: gs_preprocess_geometry on the stream, gs_view_colours on
                                           * aux_stream (forked / joined through the two aux events) beside the list
                                           * stages; the stream waits for the colours before the blend (or before
                                           * returning).  Ignored when aux_stream is NULL.                        */
#define GS_BWD_BLEND          (1u << 0)   /* gs_blend_backward                  RAS:531-705                     */
#define GS_BWD_REDUCE         (1u << 1)   /* gs_reduce_partials                 (the sums of RAS:674-696)       */
#define GS_BWD_GATHER_RETURNED (1u << 2)  /* gs_gather_returned_rows (owner-sharded)                            */
#define GS_BWD_POINTS         (1u << 3)   /* gs_point_backward                  RAS:707-772, 1051-1053, 1102-1125 */
typedef struct GsFrame {
    /* sizes and options */
    int32_t n_points, n_objects, width, height;
    int32_t tile_row_begin, tile_row_step, tile_row_end;
    int32_t bin_shift, exact_tile_cull, always_store_rotation;
    int32_t key_depth_bits, depth_bits, tile_bits;
    int32_t blend_flags;              /* GS_BLEND_* */
    int32_t need_state;               /* slot_offsets / acc_alpha / last_effective are produced */
    int32_t color_max_sh_band;
    int32_t n_visible;                /* backward: M (known to the host by then); forward: ignored (capacity = n_points) */
    int32_t backward_bin_shift, backward_filter;   /* the backward's list layout (0, 0 on walked lists) */
    int32_t world, rows_per_band, chunk_capacity;  /* owner-sharded stages */
    int32_t sorted_in_alt;            /* OUT (GS_FWD_SORT): 1 = the sorted pairs are in keys_alt / payload_alt */
    float near_plane, far_plane, depth_scale;
    float grad_q_factor, grad_s_factor, grad_alpha_factor, grad_color_factor, grad_high_order_color_factor;
    int64_t n_keys_capacity, n_slots, n_records;   /* n_records: owner-sharded, world * (chunk_capacity + 1) */
    /* inputs (RAS:788-804) */
    const float *xyz; float *features; const int8_t *invalid_mask; const int32_t *object_id;
    const float *intrinsics; const float *q_pointcloud_camera; const float *t_pointcloud_camera;
    /* per-frame state */
    float *q_camera_pointcloud, *t_camera_pointcloud;
    int8_t *visible_mask; int32_t *ids; int32_t *counters; int32_t *host_counters_pinned; void *size_event;
    float *attrs; int32_t *num_overlap_tiles, *num_keys, *block_sums, *block_sums_full;
    void *keys, *keys_alt; int32_t *payload, *payload_alt, *slot_offsets;
    int32_t *bin_ranges;              /* int32[2][number of bins]: start, end (16-byte aligned, bins padded to a multiple of 2) */
    int32_t n_bins;
    int32_t size_stamp;               /* != 0: the sizes travel as stamped words (gs_scan_block_sums2_stamped, host_counters_pinned =
                                       * the uint64[4] words) and size_event is not recorded; 0: plain int32 counters + the event */
    float *image, *depth, *acc_alpha; int32_t *last_effective, *valid_count;
    int32_t *tile_order, *tile_work, *walked_list, *walked_start;
    float *boundary_states; void *split_workspace;   /* list splitting (may be NULL); list length = n_keys_capacity */
    void *forward_split_workspace;                   /* gs_blend_forward_split (may be NULL: the forward is not split) */
    void *filter_workspace, *sort_workspace, *route_workspace;
    int32_t *route_counts, *route_pos; float *route_send; const float *records;
    /* backward */
    const int32_t *list_start, *list_payload;   /* what the backward walks: bin_ranges + sorted payload, or the walked lists */
    const float *grad_image; float *partials; uint8_t *slot_flags; float *magnitude_image; int32_t *tile_order_backward;
    float *acc; const float *returned_rows;
    float *grad_xyz, *grad_features, *grad_xyz_visible, *grad_features_visible, *hook_compact;
    /* GS_FWD_COLOUR_ASYNC: a second stream of the same device and two events, all created by the caller */
    void *aux_stream, *aux_event_fork, *aux_event_join;
    /* owner-sharded stages: HOST array of world + 1 tile rows (band b = rows [b, b + 1) of it), NULL = equal bands */
    const int32_t *band_row_bounds;
} GsFrame;
size_t gs_frame_struct_bytes(void);   /* sizeof(GsFrame): a binding checks its mirror of the struct against it */
/* ... and the byte offsets of GS_FRAME_SENTINELS members spread over the struct, in this order:
 *   n_points, blend_flags, near_plane, n_keys_capacity, xyz, q_camera_pointcloud, attrs, keys, bin_ranges, n_bins, image,
 *   tile_order, boundary_states, route_counts, list_start, grad_image, acc, grad_xyz, aux_stream, band_row_bounds
 * (a mirror built from another header -- two members swapped, a stale library of the same size -- passes the size check
 * and would make the stages read wrong pointers).  Writes min(n, GS_FRAME_SENTINELS) offsets, returns GS_FRAME_SENTINELS. */
#define GS_FRAME_SENTINELS 20
int gs_frame_layout(int32_t *offsets, int n);
int gs_frame_forward(GsFrame *frame, uint32_t stages, void *stream);
int gs_frame_backward(GsFrame *frame, uint32_t stages, void *stream);

/* ---- adaptive-controller kernels (not on the per-frame hot path) ---------------- */

/* Focal vector of every Gaussian's ellipsoid: sqrt(r_max^2 - r_min^2) along the rotated longest axis.
 * Replaces compute_ellipsoid_offset (GaussianPointAdaptiveController.py:10-25, GaussianPoint3D.py:375-388).
 * features float[n][56], offsets float[n][3]. */
int gs_ellipsoid_offsets(const float *features, int n, float *offsets, void *stream);

/* One sample per Gaussian from N(xyz, R S S^T R^T) by Box-Muller on caller-supplied uniforms
 * (float[n][4], each in (0,1]).  Replaces sample_from_point (GaussianPointAdaptiveController.py:27-42,
 * GaussianPoint3D.py:90-94,390-406), whose uniforms come from ti.random() inside the kernel. */
int gs_sample_from_points(const float *xyz, const float *features, const float *uniforms, int n,
                          float *samples, void *stream);

/* Per-iteration statistics of the controller (GaussianPointAdaptiveController.py:130-146), one pass: for every visible
 * Gaussian i with id = ids[i] (ids unique):  num_in_camera[id] += 1; num_pixels[id] += num_affected_pixels[i];
 * view_space_gradients[id] += magnitude[i]; view_space_gradients_avg[id] += magnitude[i] / pixels[i] (0/0 counts as 0);
 * position_gradients[id][0..2] += grad_point_in_camera[i][0..2]; position_gradients_norm[id] += |grad_point_in_camera[i]|.
 * The accumulators are indexed by point id (N rows), the inputs by visible index (M rows). */
int gs_controller_accumulate(const int32_t *ids, const int32_t *num_affected_pixels,
                             const float *magnitude_grad_viewspace, const float *grad_point_in_camera, int n_visible,
                             int32_t *accumulated_num_in_camera, int32_t *accumulated_num_pixels,
                             float *accumulated_view_space_gradients, float *accumulated_view_space_gradients_avg,
                             float *accumulated_position_gradients, float *accumulated_position_gradients_norm,
                             void *stream);

/* ---- fused photometric loss of the trainer --------------------------------------
 * L = (1-lambda) * mean|x-y| + lambda * (1 - SSIM(x,y)), x = clamp(prediction,0,1) when clamp01_prediction.
 * Replaces clamp + permute (GaussianPointTrainer.py:167-170) + LossFunction.forward (LossFunction.py:20-35,
 * SSIM = pytorch_msssim.ssim(data_range=1, size_average=True): 11-tap sigma-1.5 Gaussian, 'valid' window) and
 * the autograd backward of that chain.  prediction is the rasteriser output float[H][W][3]
 * (prediction_is_hwc = 1) or float[3][H][W]; target is float[3][H][W]; H, W >= 11.
 *   ssim_grad_maps float[9][H][W] (written by forward, read by backward; may be null for forward-only use)
 *   workspace      float[gs_loss_workspace_floats(H, W)]
 *   losses         float[3] = {L, L1, 1 - SSIM}
 * backward: grad_total / grad_l1 / grad_dssim are device pointers to the upstream scalar gradients of the
 * three outputs (null = 0); grad_prediction has the layout of prediction and is fully overwritten. */
long long gs_loss_workspace_floats(int height, int width);
int gs_loss_forward(const float *prediction, int prediction_is_hwc, int clamp01_prediction, const float *target,
                    int height, int width, float lambda, float *ssim_grad_maps, float *workspace, float *losses,
                    void *stream);
int gs_loss_backward(const float *prediction, int prediction_is_hwc, int clamp01_prediction, const float *target,
                     const float *ssim_grad_maps, int height, int width, float lambda, const float *grad_total,
                     const float *grad_l1, const float *grad_dssim, float *grad_prediction, void *stream);

/* Scale regulariser R = mean_{live i} ||exp(s_i)||_2 (LossFunction.py:42-54: features[mask == 0, 4:7]) and,
 * when grad_features is not null, its gradient added in place:
 *   grad_features[i][4..6] += weight * (*upstream, 1 if null) * dR/ds_i      for live rows
 * -- the dense [N,56] gradient tensor eager autograd would build and add is never materialised.
 * value_and_count float[2] = {R, number of live Gaussians}; workspace float[gs_scale_regulariser_workspace_floats()]. */
long long gs_scale_regulariser_workspace_floats(void);
int gs_scale_regulariser(const float *features, const int8_t *point_invalid_mask, int n_points, float weight,
                         const float *upstream, float *grad_features, float *workspace, float *value_and_count,
                         void *stream);

/* One Adam step over a flat float buffer, in place (torch.optim.Adam semantics without weight decay / amsgrad;
 * the reference's optimisers, GaussianPointTrainer.py:126-129).  step is the 1-based step count (bias correction);
 * all four buffers hold n floats and are 16-byte aligned. */
int gs_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, double lr,
                 double beta1, double beta2, double eps, int step, void *stream);

/* The same step for the [n_rows][56] feature matrix with the gradient of the scale regulariser
 * scale_regulariser_weight * mean_{live rows} ||exp(s)||_2 (LossFunction.py:42-54) added on the fly to columns 4..6 of
 * the incoming gradient of live rows (point_invalid_mask == 0) -- the regulariser then needs no pass of its own over the
 * feature and gradient matrices.  The gradient buffer itself is not modified.  workspace: int32[256]. */
int gs_adam_step_features(float *features, const float *grad, float *exp_avg, float *exp_avg_sq, long long n_rows,
                          double lr, double beta1, double beta2, double eps, int step,
                          const int8_t *point_invalid_mask, double scale_regulariser_weight, int32_t *workspace,
                          void *stream);

/* Adam over the rows of a fixed-capacity tensor ([N,3] positions or [N,56] features) that skips the rows of invalid
 * points: they are neither read nor written.  The moments of a skipped row decay lazily -- last_step[row] (int32[N],
 * zero-initialised, owned by the caller next to the moments) is the last step that updated the row, and the first update
 * after a gap of d steps starts from exp_avg * beta1^d, exp_avg_sq * beta2^d, what d zero-gradient steps of
 * torch.optim.Adam leave behind (TRN:126-129 runs torch.optim.Adam over the full capacity, 10x the points in the
 * reference's Truck configuration).  scale_regulariser_weight != 0 (56-float rows only): as gs_adam_step_features;
 * workspace: int32[256] then. */
int gs_adam_step_rows(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n_rows, int row_len,
                      double lr, double beta1, double beta2, double eps, int step, const int8_t *point_invalid_mask,
                      int32_t *last_step, double scale_regulariser_weight, int32_t *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HIP_H */
