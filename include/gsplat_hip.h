/* gsplat_hip.h -- C ABI of libgsplat_hip.so: the MI355X (gfx950) rasterization hot path.
 *
 * This is the drop-in boundary.  The first group of entry points replaces, one for one, the
 * functions of the reference's `splat_cuda` extension (joeyan/gaussian_splatting
 * src/bindings.cpp:118-159); the fused per-Gaussian stage, the prefix-mode renderer, the multi-GPU
 * bookkeeping and the training-loop operations further down replace the reference's Python glue
 * around them (cited per function).  All take plain device pointers + sizes + a hipStream_t (as
 * void*); no torch types, no exceptions, no hidden allocation, no device synchronisation.  Work
 * is enqueued on `stream` and the call returns.
 *
 * Conventions
 *   dtype      GS_F32 (0) or GS_F64 (1); all floating tensors of one call share it.
 *   layouts    row-major, contiguous, exactly the reference's tensor shapes (cited per function).
 *   outputs    caller-allocated.  Backward outputs marked "accumulated" are added into
 *              (atomicAdd semantics, render_backward.cu:269-281) and must be zeroed by the caller.
 *   return     0 on success, a negative GS_E* code otherwise; gs_last_error() gives the text.
 *   tile rows  [tile_row0, tile_row1) restricts binning/rendering to those rows of 16-px tiles
 *              (multi-GPU sharding); pass 0 and n_tiles_y for the whole image.
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_F32 0
#define GS_F64 1

#define GS_OK 0
#define GS_EINVAL (-1)   /* bad argument (shape, dtype, n_sh not in {1,4,9,16}, ...) */
#define GS_EHIP (-2)     /* a HIP runtime call or kernel launch failed */

#define GS_TILE 16       /* tile edge in pixels (splat_py/structs.py:4) */
#define GS_PACKED_WIDTH 12 /* scalars per packed splat record, see gs_pack_splats */
#define GS_SORT_PREFIX 1024 /* entries ordered per long tile list in prefix mode, see gs_tile_emit_sort */

const char* gs_last_error(void);
int gs_abi_version(void);

/* ---- per-Gaussian projection ------------------------------------------------------------ */
/* camera_projection_cuda (bindings.cpp:121; projection.cu:9-54).  xyz[N,3], K[3,3] -> uv[N,2] */
int gs_camera_projection(const void* xyz, const void* K, int N, void* uv, int dtype, void* stream);
/* camera_projection_backward_cuda (bindings.cpp:122; projection_backward.cu:9-90).
 * rows with z <= 0 are left untouched (Q10). */
int gs_camera_projection_backward(const void* xyz, const void* K, const void* uv_grad_out, int N,
                                  void* xyz_grad_in, int dtype, void* stream);
/* compute_sigma_world_cuda (bindings.cpp:127; projection.cu:57-152).
 * quaternion[N,4] (w,x,y,z), scale[N,3] (log) -> sigma_world[N,3,3] */
int gs_compute_sigma_world(const void* quaternion, const void* scale, int N, void* sigma_world,
                           int dtype, void* stream);
/* compute_sigma_world_backward_cuda (bindings.cpp:128; projection_backward.cu:174-382) */
int gs_compute_sigma_world_backward(const void* quaternion, const void* scale,
                                    const void* sigma_world_grad_out, int N,
                                    void* quaternion_grad_in, void* scale_grad_in, int dtype,
                                    void* stream);
/* compute_projection_jacobian_cuda (bindings.cpp:133; projection.cu:155-211). -> J[N,2,3] */
int gs_compute_projection_jacobian(const void* xyz, const void* K, int N, void* J, int dtype,
                                   void* stream);
/* compute_projection_jacobian_backward_cuda (bindings.cpp:138; projection_backward.cu:93-166) */
int gs_compute_projection_jacobian_backward(const void* xyz, const void* K,
                                            const void* jac_grad_out, int N, void* xyz_grad_in,
                                            int dtype, void* stream);
/* compute_conic_cuda (bindings.cpp:143; projection.cu:214-311).
 * sigma_world[N,3,3], J[N,2,3], camera_T_world[4,4] -> conic[N,3] = (S00, S01+S10, S11) */
int gs_compute_conic(const void* sigma_world, const void* J, const void* camera_T_world, int N,
                     void* conic, int dtype, void* stream);
/* compute_conic_backward_cuda (bindings.cpp:145; projection_backward.cu:385-550) */
int gs_compute_conic_backward(const void* sigma_world, const void* J, const void* camera_T_world,
                              const void* conic_grad_out, int N, void* sigma_world_grad_in,
                              void* J_grad_in, int dtype, void* stream);

/* ---- spherical harmonics ------------------------------------------------------------------ */
/* precompute_rgb_from_sh_cuda (bindings.cpp:148-152; precompute_sh.cu:8-58,113-250).
 * xyz[N,3], sh_coeff[N,3,n_sh] (or [N,3] when n_sh==1), matrix[4,4] whose translation column is
 * the camera centre (Q8; read on the device, no host sync) -> rgb[N,3] */
int gs_precompute_rgb_from_sh(const void* xyz, const void* sh_coeff, const void* matrix, int N,
                              int n_sh, void* rgb, int dtype, void* stream);
/* precompute_rgb_from_sh_backward_cuda (bindings.cpp:153-157; precompute_sh.cu:61-111,252-389) */
int gs_precompute_rgb_from_sh_backward(const void* xyz, const void* matrix, const void* grad_rgb,
                                       int N, int n_sh, void* grad_sh, int dtype, void* stream);

/* ---- tile binning + per-tile depth sort (fp32 only) ----------------------------------------- */
/* get_sorted_gaussian_list (bindings.cpp:147; tile_culling.cu:124-340) is split in two so that
 * the caller owns every allocation:
 *
 *   1. gs_tile_count   OBB/SAT test of every Gaussian against its candidate tiles
 *                      (tile_culling.cu:8-177), per-tile counts and their exclusive prefix:
 *                      tile_ranges[T+1] == splat_start_end_idx_by_tile_idx.  tile_ranges[T] is the
 *                      instance count S (read it back to size the outputs of step 2).
 *   2. gs_tile_emit_sort  re-runs the test, scatters (z bits, gaussian) keys into each tile's
 *                      segment and sorts every segment front-to-back in LDS
 *                      (tile_culling.cu:179-242,327-329).  Order == the reference's fp64 key
 *                      z + (max_z+1)*tile (tile_culling.cu:236-237) with ties broken by ascending
 *                      Gaussian index.
 *
 * uvs[V,2], xyz_camera_frame[V,3], conic[V,3]; n_tiles = n_tiles_x*n_tiles_y.
 * host_mirror (gs_tile_count, gs_tile_count_cut; may be NULL): a device-accessible pointer to 3 ints of PINNED HOST
 *   memory that receives, when the scan finishes, (instance count, visible count, x) with x = the length of the
 *   LONGEST tile list of the rows (gs_tile_count) / the complete instance count (gs_tile_count_cut) -- the frame's
 *   host read without a copy kernel in the stream (record an event behind the call and wait for it).  A caller that
 *   knows the longest list can tell gs_tile_emit_sort_bounded and gs_render_tiles_prefix_phased what they need
 *   not launch.
 * visible_count: NULL, or a device pointer to the number of valid rows when the inputs are
 *   capacity-V buffers whose fill level only the device knows (gs_preprocess_forward); rows
 *   beyond it are ignored and tile_ranges then has T+2 entries, [T+1] = *visible_count, so that
 *   one 8-byte read returns S and V.  Pass the same V and visible_count to both calls.
 * subset, subset_count: both NULL, or a device list of row indices and a device pointer to its
 *   length: only those rows are tested (multi-GPU band mode: the Gaussians whose candidate window
 *   reaches the rank's rows, gs_halo_plan's send_index).  The list must contain every row that has
 *   a tile in [tile_row0, tile_row1); the resulting lists are then the same as without it.
 * workspace: int32[gs_tile_workspace_ints(n_tiles)] scratch written by step 1 and read by step 2
 *   (per-workgroup tile histograms, or per-tile cursors that step 1 leaves zeroed and step 2 returns to zero;
 *   keep it untouched between the two calls; step 2 may be repeated on it -- a larger capacity);
 * keys: uint64[S] scratch.  S may be an over-estimate (a capacity): instances beyond it are not
 * written and tiles reaching beyond it are left unsorted, so a caller may launch step 2 before it
 * has read the true count and repeat it with the exact S only if the count exceeded the capacity.
 *
 * sort_prefix: 0 = every segment is sorted in full (what get_sorted_gaussian_list returns).
 *   GS_SORT_PREFIX = prefix mode for the fused renderer: the forward pass stops reading a tile's
 *   list once all its pixels are saturated (render.cu:106,162), on dense scenes after a small part
 *   of it, so for a tile with GS_SORT_PREFIX < n <= 8192 entries only the GS_SORT_PREFIX nearest are
 *   selected and ordered into sorted_gaussians[start .. start+GS_SORT_PREFIX); the rest of that
 *   segment is undefined.  Results stay exact: gs_render_tiles_prefix raises tile_flags[t] when a
 *   tile ran out of prefix with an unsaturated pixel, sorts those tiles in full from the untouched
 *   `keys` (gs_tile_sort_flagged) and renders them again -- plain enqueues, no host read. */
size_t gs_tile_workspace_ints(int n_tiles);
int gs_tile_count(const void* uvs, const void* conic, int V, const int32_t* visible_count,
                  const int32_t* subset, const int32_t* subset_count, int n_tiles_x, int n_tiles_y,
                  float mh_dist, int tile_row0, int tile_row1, int32_t* workspace,
                  int32_t* tile_ranges /*[T+1] or [T+2]*/, int32_t* host_mirror, void* stream);
int gs_tile_emit_sort(const void* uvs, const void* xyz_camera_frame, const void* conic, int V,
                      const int32_t* visible_count, const int32_t* subset,
                      const int32_t* subset_count, int n_tiles_x, int n_tiles_y, float mh_dist,
                      int tile_row0, int tile_row1, const int32_t* tile_ranges, int32_t* workspace,
                      uint64_t* keys, int64_t S, int32_t* sorted_gaussians /*[S]*/, int sort_prefix,
                      void* stream);
/* gs_tile_emit_sort with a bound on the list lengths (ABI 6): longest_list >= 0 promises that no tile of the rows has
 * more entries, and the sort launches that only serve longer lists are not enqueued (a sparse frame otherwise pays
 * ~5 us for a kernel that finds nothing to do); < 0 = no promise.  The bound may be a GUESS only if the caller checks
 * it against the count pass's result afterwards and repeats the call when it was too small. */
int gs_tile_emit_sort_bounded(const void* uvs, const void* xyz_camera_frame, const void* conic, int V,
                              const int32_t* visible_count, const int32_t* subset, const int32_t* subset_count,
                              int n_tiles_x, int n_tiles_y, float mh_dist, int tile_row0, int tile_row1,
                              const int32_t* tile_ranges, int32_t* workspace, uint64_t* keys, int64_t S,
                              int32_t* sorted_gaussians /*[S]*/, int sort_prefix, int64_t longest_list, void* stream);
/* Full sort of the tiles t in the row band with tile_flags[t] != 0 (keys as left by
 * gs_tile_emit_sort(..., GS_SORT_PREFIX, ...), same S). */
int gs_tile_sort_flagged(const int32_t* tile_ranges, const uint64_t* keys, int64_t S,
                         const int32_t* tile_flags, int n_tiles_x, int tile_row0, int tile_row1,
                         int32_t* sorted_gaussians, void* stream);

/* ---- depth-bucketed binning ("depth cut", fp32, ABI 6; no reference counterpart) ----------------------------
 * For dense frames the tile lists are several times longer than what the render consumes (workload D: ~2800
 * entries per tile, no pixel composites deeper than the 743rd).  Here the count pass walks the visible Gaussians
 * in NBK = 1024 DEPTH BUCKETS of about equal population (one workgroup per bucket), so that its per-workgroup tile
 * histograms are a cumulative depth histogram per tile and every tile knows, exactly, the last bucket b*(t) up to
 * which it holds at most GS_SORT_PREFIX entries.  Only those are emitted and sorted (a true depth prefix of the
 * complete list, completely ordered).  Exact like the prefix sort: gs_render_tiles_cut flags a truncated tile
 * that reaches the end of its list with an unsaturated pixel, emits + sorts the COMPLETE lists of the flagged tiles
 * into the caller's overflow buffers at full_ranges and renders them again; gs_render_tiles_backward_slab reads a
 * flagged tile's list from there.  No host read in between.
 *
 *   gs_preprocess_forward_cut   gs_preprocess_forward + bin_records float[N,8] (u v conic0 conic1 | conic2 z 0 0 per
 *                               visible Gaussian: what the binning reads, one 32-byte sector), and inside cut_workspace
 *                               the bucket boundaries and every visible Gaussian's bucket.  depth_hist: int32
 *                               [GS_CUT_HIST_BINS] that is ALL ZERO on entry and all zero again on exit (allocate once,
 *                               zero once, hand to every frame on the stream): the depth histogram of every
 *                               sample_stride-th visible Gaussian (sample_stride = gs_cut_sample_stride(N)) whose
 *                               quantiles are the boundaries.  uv, xyz_camera_frame and conic may be NULL here (their
 *                               values are in bin_records: uv = columns 0-1, conic = columns 2-4, z = column 5).
 *   gs_tile_count_cut           buckets, counts, cut: tile_ranges[T+3] (prefix of the KEPT counts; [T] = entries kept
 *                               S', [T+1] = V, [T+2] = all entries S), full_ranges[T+1] (prefix of the complete counts)
 *   gs_tile_emit_sort_cut       keys + sorted lists of the kept entries (capacity S as in gs_tile_emit_sort)
 * workspace: int32[gs_tile_workspace_ints(T)] as for gs_tile_count; cut_workspace: int32[gs_cut_workspace_ints(N, T)],
 * both untouched until the frame's backward has run.  Needs the LDS-histogram regime: gs_cut_supported(...) != 0. */
#define GS_CUT_BUCKETS 1024
#define GS_CUT_HIST_BINS 8192
#define GS_CUT_MAX_SAMPLES 65536
size_t gs_cut_workspace_ints(int n_gaussians, int n_tiles);
int gs_cut_sample_stride(int n_gaussians);
int gs_cut_supported(int n_tiles_x, int tile_row0, int tile_row1, int n_gaussians);
int gs_tile_count_cut(const void* bin_records, int N, const int32_t* visible_count, int n_tiles_x, int n_tiles_y,
                      float mh_dist, int tile_row0, int tile_row1, int32_t* workspace, int32_t* cut_workspace,
                      int32_t* tile_ranges /*[T+3]*/, int32_t* full_ranges /*[T+1]*/, int32_t* host_mirror, void* stream);
int gs_tile_emit_sort_cut(const void* bin_records, int N, int n_tiles_x, int n_tiles_y, float mh_dist, int tile_row0,
                          int tile_row1, const int32_t* tile_ranges, int32_t* workspace, int32_t* cut_workspace,
                          uint64_t* keys, int64_t S, int32_t* sorted_gaussians, void* stream);
/* tests / tools: device pointers into the cut workspace -- b*(t) int32[T], complete counts int32[T], bucket bounds
 * uint32[1024] (sortable depth bits, inclusive upper bound), bucket offsets int32[1025], control int32[2] (deepest
 * bucket any tile wants, flagged tiles of the frame) */
int gs_cut_debug_views(int32_t* cut_workspace, int N, int n_tiles, int32_t** bstar, int32_t** totals, uint32_t** bounds,
                       int32_t** bucket_offsets, int32_t** ctrl);

/* ---- fused per-Gaussian stage (fp32) ----------------------------------------------------------------
 * One pass replacing the PyTorch glue and per-Gaussian kernels of rasterize()
 * (splat_py/rasterize.py:29-99: utils.py:60-72 transform, projection.cu:9-19, the frustum cull
 * :33-49, the boolean-mask gathers :52-75, torch.sigmoid :60, projection.cu:57-257,
 * precompute_sh.cu:8-58 on cat(rgb, sh) :89).  Survivors are compacted in Gaussian order, so
 * the visible index equals the reference's boolean-mask index.
 *   inputs   xyz[N,3] quaternion[N,4] scale[N,3] opacity[N,1] (logits) rgb[N,3]
 *            sh[N,3,n_sh-1] (NULL when n_sh == 1), camera_T_world[4,4], K[3,3]
 *   mh_dist, band_row0, band_row1: a multi-GPU rank passes its tile-row band; rgb_render (and the
 *            colour inside packed) is then evaluated only for Gaussians whose candidate tile window
 *            reaches the band (0 elsewhere).  Pass 0 and n_tile_rows for the whole frame.
 *   workspace int32[gs_preprocess_workspace_ints(N)]
 *   outputs  camera_center[3]; visible_count[1] (= V, on the device); culling_mask uint8[N]
 *            (1 = culled); rank int32[N] (visible index or -1); and, with capacity N rows of which
 *            the first V are written: vis_idx int32, uv[.,2], xyz_camera_frame[.,3], conic[.,3],
 *            opacity_act[.,1] = sigmoid, rgb_render[.,3], packed[.,12] (see gs_pack_splats).  vis_idx and
 *            rgb_render may be NULL (not written: the renderer reads the colour from the packed record). */
size_t gs_preprocess_workspace_ints(int N);
int gs_preprocess_forward(const void* xyz, const void* quaternion, const void* scale,
                          const void* opacity, const void* rgb, const void* sh, int n_sh,
                          const void* camera_T_world, const void* K, int N, int W, int H,
                          float near_thresh, float far_thresh, float cull_mask_padding,
                          float mh_dist, int band_row0, int band_row1,
                          int32_t* workspace, void* camera_center, int32_t* visible_count,
                          uint8_t* culling_mask, int32_t* rank, int32_t* vis_idx, void* uv,
                          void* xyz_camera_frame, void* conic, void* opacity_act, void* rgb_render,
                          void* packed, void* stream);
int gs_preprocess_forward_cut(const void* xyz, const void* quaternion, const void* scale,
                              const void* opacity, const void* rgb, const void* sh, int n_sh,
                              const void* camera_T_world, const void* K, int N, int W, int H,
                              float near_thresh, float far_thresh, float cull_mask_padding,
                              float mh_dist, int band_row0, int band_row1,
                              int32_t* workspace, void* camera_center, int32_t* visible_count,
                              uint8_t* culling_mask, int32_t* rank, int32_t* vis_idx, void* uv,
                              void* xyz_camera_frame, void* conic, void* opacity_act, void* rgb_render,
                              void* packed, void* bin_records, int32_t* cut_workspace, int32_t* depth_hist,
                              int sample_stride, void* stream);

/* Backward of the above (projection_backward.cu:9-471, precompute_sh.cu:61-111 and the autograd of
 * the glue: sigmoid', the cat split, the dense scatter of rasterize.py:52-75, matmul').
 * grad_slab: the render gradients, one row of 9 floats per visible Gaussian in the order
 *   (rgb_render 3 | opacity_act 1 | uv 2 | conic 3) -- the layout gs_render_tiles_backward_slab
 *   accumulates into; the row of visible Gaussian v is grad_slab[(v - v_base) * 9].
 * Writes the dense parameter gradients, every row exactly once (zeros for culled Gaussians):
 * grad_xyz[N,3], grad_quaternion[N,4], grad_scale[N,3], grad_opacity_logit[N,1],
 * grad_rgb_param[N,3], grad_sh[N,3,n_sh-1] (NULL when n_sh == 1).
 * A slice [i0, i1) of the Gaussians (multi-GPU: the slice this rank owns) is processed by passing
 * the parameter / rank / output pointers advanced to row i0, N = i1 - i0, and a slab that holds
 * the rows of the slice's visible Gaussians with v_base = their first visible index;
 * opacity_act stays the full array (it is indexed by v). */
int gs_preprocess_backward(const void* xyz, const void* quaternion, const void* scale, int n_sh,
                           const void* camera_T_world, const void* K, const void* camera_center,
                           const int32_t* rank, const void* opacity_act, const void* grad_slab,
                           int v_base, int N, void* grad_xyz, void* grad_quaternion,
                           void* grad_scale, void* grad_opacity_logit, void* grad_rgb_param,
                           void* grad_sh, void* stream);

/* ---- tile renderer ---------------------------------------------------------------------------- */
/* Packs what the render kernels read per splat into one 48-byte (fp32) record per visible Gaussian:
 *   packed[V][12] = (u, v, r2, opacity | a, b, c, det | 1/det, SH_0*col0, SH_0*col1, SH_0*col2)
 * a/b/c as render.cu:117-128 forms them (+0.25 dilation for fp32, none for fp64); r2 is a
 * conservative squared cutoff radius: a pixel farther than sqrt(r2) from (u, v) provably has
 * alpha < 1/255 and is skipped exactly as render.cu:145-148 would skip it (+inf for fp64, which
 * has no alpha threshold); col = rgb[V,3] when rgb is given with n_sh == 1, stored as the product
 * SH_0 * col that sh_to_rgb forms per pixel (spherical_harmonics.cuh:83) (else unused: the kernels
 * then gather the [3, n_sh] coefficients from `rgb` directly).
 * uvs[V,2], opacity[V,1], conic[V,3], rgb[V,3] or NULL. */
int gs_pack_splats(const void* uvs, const void* opacity, const void* conic, const void* rgb, int V,
                   void* packed, int dtype, void* stream);
/* render_tiles_cuda (bindings.cpp:119; render.cu:8-422), the reference's arguments in the reference's order:
 * uvs[V,2], opacity[V,1], rgb[V,3,n_sh], conic[V,3], view_dir_by_pixel[H,W,3] (ignored when n_sh==1),
 * tile_ranges = splat_start_end_idx_by_tile_idx int32[n_tiles+1], sorted_gaussians =
 * gaussian_idx_by_splat_idx int32[S], background_rgb[3] -> num_splats_per_pixel int32[H,W],
 * final_weight_per_pixel[H,W], image[H,W,3]; then the sizes the tensors carry in the reference.  The
 * per-splat record of the render loops is formed while the tile lists are staged (no packing pass). */
int gs_render_tiles(const void* uvs, const void* opacity, const void* rgb, const void* conic,
                    const void* view_dir_by_pixel, const int32_t* tile_ranges,
                    const int32_t* sorted_gaussians, const void* background_rgb,
                    int32_t* num_splats_per_pixel, void* final_weight_per_pixel, void* image, int W, int H,
                    int n_sh, int tile_row0, int tile_row1, int dtype, void* stream);
/* The same from packed records (gs_pack_splats, or the `packed` output of gs_preprocess_forward): one
 * 48-byte gather per list entry instead of four partial ones -- the form the fused renderer uses. */
int gs_render_tiles_packed(const void* packed, const void* rgb, const void* view_dir_by_pixel,
                           const int32_t* tile_ranges, const int32_t* sorted_gaussians,
                           const void* background_rgb, int W, int H, int n_sh, int tile_row0,
                           int tile_row1, int32_t* num_splats_per_pixel, void* final_weight_per_pixel,
                           void* image, int dtype, void* segment_state, void* stream);
/* The same kernel over lists produced with sort_prefix = GS_SORT_PREFIX (fp32, n_sh == 1).
 * Enqueues (1) a provisional render that reads at most the ordered prefix of a prefix-sorted tile
 * and writes tile_flags[t] = 1 if tile t ran out of it with an unsaturated pixel (0 otherwise),
 * (2) gs_tile_sort_flagged, (3) a second render of the flagged tiles from their full lists.
 * Afterwards every output equals what gs_render_tiles_packed gives on fully sorted lists, and the
 * segments of the flagged tiles in sorted_gaussians are fully sorted (the backward pass reads them).
 * keys, S: as passed to gs_tile_emit_sort.  tile_flags: int32[n_tiles] scratch/out.
 * tile_cost (may be NULL): int32[n_tiles] out, the time each tile of [tile_row0, tile_row1) took to render
 * (16-shader-cycle units) -- a scheduling hint for gs_render_tiles_backward_slab, no effect on any result. */
int gs_render_tiles_prefix(const void* packed, const void* rgb, const int32_t* tile_ranges,
                           int32_t* sorted_gaussians, const uint64_t* keys, int64_t S,
                           const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                           int32_t* tile_flags, int32_t* num_splats_per_pixel,
                           void* final_weight_per_pixel, void* image, int32_t* tile_cost, void* segment_state,
                           void* stream);
/* The same in two separately callable phases (ABI 6): GS_PREFIX_RENDER = step (1), GS_PREFIX_REPAIR = steps (2) + (3)
 * on the flags step (1) left, both = gs_render_tiles_prefix.  A frame none of whose lists exceeds GS_SORT_PREFIX
 * cannot raise a flag: a caller that knows (the count pass's longest list) enqueues the render alone, one that only
 * guessed enqueues the repair when the count says otherwise -- later in the stream, the same result. */
#define GS_PREFIX_RENDER 1
#define GS_PREFIX_REPAIR 2
int gs_render_tiles_prefix_phased(const void* packed, const void* rgb, const int32_t* tile_ranges,
                                  int32_t* sorted_gaussians, const uint64_t* keys, int64_t S,
                                  const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                                  int32_t* tile_flags, int32_t* num_splats_per_pixel, void* final_weight_per_pixel,
                                  void* image, int32_t* tile_cost, void* segment_state, int phases, void* stream);
/* Depth segments of the fused backward (fp32, n_sh == 1; ABI 5).  segment_state (may be NULL): a 16-byte aligned
 * device workspace of gs_render_segment_workspace_bytes(W, H, tile_row0, tile_row1) bytes (ABI 6: sized for the
 * tile rows the three calls are given -- 32 KB per tile of the band, not of the grid) that gs_render_tiles_packed /
 * gs_render_tiles_prefix fill -- per (tile, 128-entry segment of its list, pixel) the transmittance at the
 * segment's far boundary and the colour the segment contributed, per pixel where its walk ends and what the
 * backward's first step does there -- and that gs_render_tiles_backward_slab, given the same pointer, uses to
 * launch one workgroup per (tile, segment) instead of one per tile: 3-4x more, shorter work items, the same
 * gradients up to fp32 rounding (what a multi-GPU rank's band of a few hundred tiles needs to fill the chip;
 * no reference counterpart).  The contents are private to the library. */
size_t gs_render_segment_workspace_bytes(int W, int H, int tile_row0, int tile_row1);
/* render_tiles_backward_cuda (bindings.cpp:120; render_backward.cu:12-595), the reference's arguments in the
 * reference's order.  grad_rgb[V,3,n_sh], grad_opacity[V,1], grad_uv[V,2], grad_conic[V,3] are accumulated.
 * backward_mode: GS_BACKWARD_DEFAULT / _COMPAT / _EXACT (below); COMPAT is bug-compatible with
 * render_backward.cu:185. */
int gs_render_tiles_backward(const void* uvs, const void* opacity, const void* rgb, const void* conic,
                             const void* view_dir_by_pixel, const int32_t* tile_ranges,
                             const int32_t* sorted_gaussians, const void* background_rgb,
                             const int32_t* num_splats_per_pixel, const void* final_weight_per_pixel,
                             const void* grad_image, void* grad_rgb, void* grad_opacity, void* grad_uv,
                             void* grad_conic, int W, int H, int n_sh, int tile_row0, int tile_row1, int dtype,
                             int backward_mode, void* stream);
/* The same from packed records (see gs_render_tiles_packed). */
int gs_render_tiles_backward_packed(const void* packed, const void* rgb, const void* view_dir_by_pixel,
                                    const int32_t* tile_ranges, const int32_t* sorted_gaussians,
                                    const void* background_rgb, const int32_t* num_splats_per_pixel,
                                    const void* final_weight_per_pixel, const void* grad_image, int W,
                                    int H, int n_sh, int tile_row0, int tile_row1, void* grad_rgb,
                                    void* grad_opacity, void* grad_uv, void* grad_conic, int dtype,
                                    int backward_mode, void* stream);
/* The fused renderer's form of the above (fp32, n_sh == 1): the four gradients of a Gaussian are
 * accumulated into one row of grad_slab[V, 9] = (rgb 3 | opacity 1 | uv 2 | conic 3).  zero_slab_rows: 0 = the
 * slab is zero-initialised (or holds values to accumulate onto); n > 0 = the call clears its first n rows itself
 * (16-byte aligned slab), in the launch that also makes the tile order -- one prologue kernel instead of a fill
 * and a single-workgroup kernel one after the other.
 * tile_cost / tile_order: with the costs gs_render_tiles_prefix measured and an int32[n_tiles + 8] workspace, the
 * tiles' workgroups are started longest-first (shorter drain at the end of the kernel; grids below 2048 tiles keep
 * the natural order).  Both NULL: natural order.  tile_order alone: the order gs_render_backward_prologue left in
 * it (the fused frames call the prologue as its own entry so that this one is the render kernel and nothing else).
 * The gradients do not depend on the order.
 * segment_state (may be NULL): the workspace the forward filled -> (tile, depth segment) work items, see
 * gs_render_segment_workspace_bytes; tile_cost / tile_order are then not used.
 * cut_flags / full_ranges / overflow_sorted (all NULL, or all given; ABI 6): a frame rendered by gs_render_tiles_cut --
 * a tile with cut_flags[t] != 0 reads its (complete) list from overflow_sorted at full_ranges[t]. */
/* The backward's prologue as its own call (ABI 6): clears the first zero_slab_rows rows of grad_slab[., 9] (0: none)
 * and, if tile_order is given, leaves the launch order of the rows' tiles there -- longest-first from tile_cost
 * (grids of >= 2048 tiles), the natural order otherwise -- in ONE launch; hand tile_order (and tile_cost = NULL,
 * zero_slab_rows = 0) to gs_render_tiles_backward_slab. */
int gs_render_backward_prologue(void* grad_slab, int64_t zero_slab_rows, const int32_t* tile_cost, int32_t* tile_order,
                                int W, int H, int tile_row0, int tile_row1, void* stream);
int gs_render_tiles_backward_slab(const void* packed, const void* rgb, const int32_t* tile_ranges,
                                  const int32_t* sorted_gaussians, const void* background_rgb,
                                  const int32_t* num_splats_per_pixel,
                                  const void* final_weight_per_pixel, const void* grad_image, int W,
                                  int H, int tile_row0, int tile_row1, void* grad_slab, int64_t zero_slab_rows,
                                  const int32_t* tile_cost, int32_t* tile_order, const void* segment_state,
                                  const int32_t* cut_flags, const int32_t* full_ranges, const int32_t* overflow_sorted,
                                  int backward_mode, void* stream);
/* Forward render of a frame binned by gs_tile_count_cut / gs_tile_emit_sort_cut (fp32, one colour coefficient per
 * channel; ABI 6).  tile_ranges / sorted_gaussians: the kept depth prefixes (capacity S as in gs_render_tiles_prefix);
 * a truncated tile that ends with an unsaturated pixel gets tile_flags[t] = 1, the complete lists of such tiles are
 * emitted into overflow_keys / overflow_sorted (capacity overflow_capacity >= full_ranges[T], the frame's complete
 * instance count; only flagged tiles' segments are written) and they are rendered again -- plain enqueues, no host
 * read; the three repair launches exit at once while nothing is flagged.  bin_records, workspace, cut_workspace: as
 * given to gs_tile_count_cut.  host_flagged (may be NULL): one int of device-accessible pinned host memory that receives
 * the number of flagged tiles (a policy hint for later frames: a frame whose tiles mostly need their complete lists
 * is cheaper without the cut).  Results are those of gs_render_tiles_packed on the complete sorted lists, bit for bit. */
int gs_render_tiles_cut(const void* packed, const void* rgb, const int32_t* tile_ranges, const int32_t* sorted_gaussians,
                        int64_t S, const int32_t* full_ranges, const void* bin_records, int N, float mh_dist,
                        int32_t* workspace, int32_t* cut_workspace, uint64_t* overflow_keys, int32_t* overflow_sorted,
                        int64_t overflow_capacity, const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                        int32_t* tile_flags, int32_t* num_splats_per_pixel, void* final_weight_per_pixel, void* image,
                        int32_t* tile_cost, int32_t* host_flagged, void* stream);
/* Gradient mode of the render backward: the `backward_mode` argument of the three entry points above
 * (ABI 5: per call, so that a trainer switching modes cannot race a backward that the autograd engine's
 * own thread has queued).  GS_BACKWARD_DEFAULT takes the process-wide default, which gs_set_backward_mode
 * sets (initially COMPAT) and which is read once, at the call.
 *   GS_BACKWARD_COMPAT  the reference's arithmetic, including render_backward.cu:185: the transmittance is
 *                       divided back by (1 - alpha) only while the CHUNK-LOCAL splat index is below
 *                       num_splats - 1, so for pixels that composite deeper than the reference's first
 *                       shared-memory chunk (960 splats in fp32 with one colour coefficient) the weights of
 *                       the later chunks are off by a factor 1 / (1 - alpha_last)
 *   GS_BACKWARD_EXACT   the same with the GLOBAL splat index: the mathematically exact gradient of the
 *                       forward pass.  Identical to COMPAT whenever no pixel composites past the first chunk. */
#define GS_BACKWARD_DEFAULT (-1)
#define GS_BACKWARD_COMPAT 0
#define GS_BACKWARD_EXACT 1
int gs_set_backward_mode(int mode);
int gs_get_backward_mode(void);

/* render_depth_cuda (bindings.cpp:158; depth.cu:7-177), fp32 only.  depth_image[H,W] is written
 * only where the accumulated alpha passes alpha_threshold (caller pre-fills with -1). */
int gs_render_depth(const void* packed, const void* xyz_camera_frame, const int32_t* tile_ranges,
                    const int32_t* sorted_gaussians, int W, int H, float alpha_threshold,
                    void* depth_image, void* stream);

/* Cost-balanced bands (multi-GPU; no reference counterpart): the band images of unequal bands are all-gathered as
 * equal chunks of chunk_rows = 16 x (tallest band) + 1 pixel rows, each starting at its rank's band; the last row of
 * a chunk carries the per-tile-row costs of that band (floats; 0 outside the band).
 *   gs_band_row_costs   cost_row[r] = sum over the tiles of row r of min(list length, cap) + tile_cost per tile, for
 *                       r in [tile_row0, tile_row1), 0 elsewhere (n_tile_rows floats)
 *   gs_band_assemble    gathered float[G][chunk_rows][3 W] -> image float[H][3 W] (pixel row y from the chunk of the
 *                       band that holds tile row y / 16, band_rows = the G + 1 boundaries, HOST array) and
 *                       host_costs float[n_tile_rows] (device-accessible PINNED host memory: the summed cost rows,
 *                       what balances the next frame's bands; record an event behind the call and wait for it) */
int gs_band_row_costs(const int32_t* tile_ranges, int n_tiles_x, int n_tile_rows, int tile_row0, int tile_row1,
                      int tile_cost, int cap, void* cost_row, void* stream);
int gs_band_assemble(const void* gathered, const int32_t* band_rows, int G, int chunk_rows, int W, int H, int n_tile_rows,
                     void* image, void* host_costs, void* stream);
/* ---- multi-GPU: tile-row bands with owner-sliced gradients ------------------------------------
 * (no reference counterpart: the reference is single-GPU; BASELINE.json north_star asks for frames
 * sharded by tile rows over up to 8 GPUs.)  Rank b renders tile rows [band_rows[b], band_rows[b+1]);
 * rank r owns the Gaussians [256*owner_blocks[r], 256*owner_blocks[r+1]) and runs the per-Gaussian
 * backward for them, so the partial render gradients of the bands are exchanged sparsely: a row
 * travels only from a rank whose band the Gaussian's candidate tile window reaches to its owner.
 *
 * gs_halo_plan (after gs_preprocess_forward of the same frame; inputs are the capacity-N buffers
 * and the device-side visible_count):
 *   mask[N]        bit s = visible Gaussian v reaches the band of rank s
 *   send_index[N]  the visible indices with bit `rank`, ascending: row k of the send buffer is
 *                  grad_slab[send_index[k]]; consecutive owners' parts, plan[4 + r] rows for owner r
 *   plan[4 + 2G]   device record for the frame's host read: number of rows in send_index, V,
 *                  v_lo, v_hi (visible-index range of the Gaussians this rank owns),
 *                  send counts [G], receive counts [G]
 *   workspace      int32[gs_halo_workspace_ints(N, G)], read again by gs_halo_gather_sum
 * band_rows, owner_blocks: host arrays of G + 1 ints.  G <= GS_MAX_RANKS.
 *
 * gs_halo_gather_sum (after the all_to_all): recv holds, for sender s = 0..G-1 at row offset
 * recv_offsets[s] (host array, the exclusive prefix of the receive counts), the rows of the
 * Gaussians v in [v_lo, v_hi) with bit s, ascending; out[(v - v_lo) * 9 ..] = their sum, every
 * row of the range written once (zeros where no band reaches the Gaussian). */
#define GS_MAX_RANKS 8
size_t gs_halo_workspace_ints(int N, int G);
int gs_halo_plan(const void* uvs, const void* conic, int N, const int32_t* visible_count,
                 const int32_t* preprocess_workspace, int n_tiles_x, int n_tiles_y, float mh_dist,
                 const int32_t* band_rows, const int32_t* owner_blocks, int G, int rank,
                 uint32_t* mask, int32_t* workspace, int32_t* send_index, int32_t* plan,
                 void* stream);
int gs_halo_gather_sum(const uint32_t* mask, const int32_t* workspace, int N, int G, int rank,
                       int v_lo, int v_hi, const void* recv, const int32_t* recv_offsets,
                       void* out, void* stream);

/* ---- multi-GPU: band-compact per-Gaussian stage (ABI 5; no reference counterpart) -------------------------
 * A rank evaluates the per-Gaussian stage in full only for the Gaussians that can reach its band, into arrays
 * compacted to those rows.
 * gs_band_project: for every Gaussian the transform, projection and frustum cull of gs_preprocess_forward (same
 *   culling_mask, rank, vis_idx, uv[V,2], opacity_act[V], visible_count, camera_center, workspace), plus mask[V]:
 *   bit s = the Gaussian can reach the tile rows [band_rows[s], band_rows[s+1]) -- its candidate window
 *   (tile_culling.cu:138-156) bounded from the largest scale, a superset of the exact window that needs no
 *   covariance and is the same on every rank.  halo_workspace: int32[gs_halo_workspace_ints(N, G)].
 * gs_halo_plan_masked: gs_halo_plan's send list, split sizes and gather layout from those masks.
 * gs_preprocess_forward_list: rows l < *list_count of `list` (visible indices, ascending: the send list): Sigma, J,
 *   conic, SH colour and the packed record of Gaussian vis_idx[list[l]], written at row l of uv_l[.,2],
 *   xyz_camera_frame_l[.,3], conic_l[.,3], packed_l[.,12] (capacity rows allocated).  Bit-identical to the rows
 *   gs_preprocess_forward writes.  Binning, sort and render take the compact arrays (visible_count = list_count);
 *   the render-gradient slab [rows, 9] of the band is then the send buffer of the gradient exchange. */
int gs_band_project(const void* xyz, const void* scale, const void* opacity, const void* camera_T_world, const void* K,
                    int N, int W, int H, float near_thresh, float far_thresh, float cull_mask_padding, float mh_dist,
                    const int32_t* band_rows, int G, int32_t* workspace, void* camera_center, int32_t* visible_count,
                    uint8_t* culling_mask, int32_t* rank, int32_t* vis_idx, void* uv, void* opacity_act, uint32_t* mask,
                    int32_t* halo_workspace, void* stream);
int gs_halo_plan_masked(const uint32_t* mask, int N, const int32_t* visible_count,
                        const int32_t* preprocess_workspace, const int32_t* owner_blocks, int G, int rank,
                        int32_t* workspace, int32_t* send_index, int32_t* plan,
                        int32_t* plan_host /* NULL, or 4 + 2 G ints of device-accessible pinned host memory: a copy of
                                              plan, written by the same kernel (the host read without a copy) */,
                        void* stream);
int gs_preprocess_forward_list(const void* xyz, const void* quaternion, const void* scale, const void* rgb, const void* sh,
                               int n_sh, const void* camera_T_world, const void* K, const void* camera_center,
                               const int32_t* list, const int32_t* list_count, int capacity, const int32_t* vis_idx,
                               const void* uv, const void* opacity_act, void* uv_l, void* xyz_camera_frame_l, void* conic_l,
                               void* packed_l, void* stream);

/* ---- touch masks handed from the render forward to the render backward (ABI 8; no reference counterpart) -------
 * The fused backward walks the lists its forward walked and needs the same (64 entries x 4 patches) touch masks.  The
 * `_m` forms of the fused frame's three render entries take touch_masks: uint64[n_tiles_of_the_GRID * 16 * 4] (512 bytes
 * per tile; NULL = the forms without `_m`).  The forward (its repair pass included) stores the masks of the first 1024
 * list entries of every tile it renders: touch_masks[(tile * 16 + word) * 4 + patch]; the backward reads them instead
 * of evaluating 256 ellipse-rectangle tests per 64-entry chunk, and builds the words beyond itself.  Results are
 * bit-identical with and without.  The buffer must be the one the SAME frame's forward wrote. */
int gs_render_tiles_prefix_phased_m(const void* packed, const void* rgb, const int32_t* tile_ranges,
                                    int32_t* sorted_gaussians, const uint64_t* keys, int64_t S,
                                    const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                                    int32_t* tile_flags, int32_t* num_splats_per_pixel, void* final_weight_per_pixel,
                                    void* image, int32_t* tile_cost, void* segment_state, int phases, uint64_t* touch_masks,
                                    void* stream);
int gs_render_tiles_cut_m(const void* packed, const void* rgb, const int32_t* tile_ranges, const int32_t* sorted_gaussians,
                          int64_t S, const int32_t* full_ranges, const void* bin_records, int N, float mh_dist,
                          int32_t* workspace, int32_t* cut_workspace, uint64_t* overflow_keys, int32_t* overflow_sorted,
                          int64_t overflow_capacity, const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                          int32_t* tile_flags, int32_t* num_splats_per_pixel, void* final_weight_per_pixel, void* image,
                          int32_t* tile_cost, int32_t* host_flagged, uint64_t* touch_masks, void* stream);
int gs_render_tiles_backward_slab_m(const void* packed, const void* rgb, const int32_t* tile_ranges,
                                    const int32_t* sorted_gaussians, const void* background_rgb,
                                    const int32_t* num_splats_per_pixel, const void* final_weight_per_pixel,
                                    const void* grad_image, int W, int H, int tile_row0, int tile_row1, void* grad_slab,
                                    int64_t zero_slab_rows, const int32_t* tile_cost, int32_t* tile_order,
                                    const void* segment_state, const int32_t* cut_flags, const int32_t* full_ranges,
                                    const int32_t* overflow_sorted, int backward_mode, const uint64_t* touch_masks,
                                    void* stream);

/* ---- multi-GPU: the fused band frontend (ABI 8; no reference counterpart) --------------------------------------
 * gs_band_frontend = gs_band_project + gs_halo_plan_masked + gs_preprocess_forward_list in four launches instead of
 * nine, with every count taken per 256-block of the GAUSSIAN index (owner slices are whole such blocks, so the
 * exchange plan is 2 G differences of scanned offsets).  Same outputs, bit for bit: culling_mask[N], rank[N]
 * (Gaussian -> visible index or -1), uv[V,2] and opacity_act[V] by visible index, camera_center; the send list
 * send_index[L] (visible indices with this rank's band bit, ascending) and, at its rows l, list_g[L] (their Gaussian
 * indices) and the band-compact arrays uv_l[L,2], xyz_camera_frame_l[L,3], conic_l[L,3], packed_l[L,12]; plan[4 + 2G]
 * = (L, V, v_lo, v_hi, send counts[G], receive counts[G]) on the device and -- plan_host != NULL -- the same ints
 * in device-accessible pinned host memory.  All row arrays need capacity N.  workspace:
 * int32[gs_band_frontend_workspace_ints(N, G)] = per-block counts and scanned offsets of the visible Gaussians and of
 * every band's Gaussians, and a 16-bit mask per Gaussian (bit 15 = visible, bit s = can reach band s); it is read
 * again by the two calls below.  band_rows, owner_blocks: host arrays of G + 1 ints.
 * gs_band_gather_sum (after the all_to_all; recv / recv_offsets as for gs_halo_gather_sum): out[(v - v_lo) * 9 ..] =
 *   the sum over the senders, ascending, of the received rows of every visible Gaussian this rank owns.
 * gs_preprocess_backward_gathered: gs_preprocess_backward for the owned slice (pointers of xyz, quaternion, scale,
 *   rank_of_gaussian and of the six gradients start at Gaussian 256 * owner_blocks[rank]; n rows) with the
 *   render-gradient row of each Gaussian summed on the spot from recv instead of read from a slab -- the same bits as
 *   gs_band_gather_sum followed by gs_preprocess_backward, one launch and one [owned, 9] round trip less. */
size_t gs_band_frontend_workspace_ints(int N, int G);
int gs_band_frontend(const void* xyz, const void* quaternion, const void* scale, const void* opacity, const void* rgb,
                     const void* sh, int n_sh, const void* camera_T_world, const void* K, int N, int W, int H,
                     float near_thresh, float far_thresh, float cull_mask_padding, float mh_dist, const int32_t* band_rows,
                     const int32_t* owner_blocks, int G, int rank, int32_t* workspace, void* camera_center,
                     uint8_t* culling_mask, int32_t* rank_out, void* uv, void* opacity_act, int32_t* send_index,
                     int32_t* list_g, void* uv_l, void* xyz_camera_frame_l, void* conic_l, void* packed_l, int32_t* plan,
                     int32_t* plan_host, void* stream);
int gs_band_gather_sum(const int32_t* workspace, int N, int G, int rank, const int32_t* owner_blocks,
                       const int32_t* rank_of_gaussian, int v_lo, const void* recv, const int32_t* recv_offsets, void* out,
                       void* stream);
int gs_preprocess_backward_gathered(const void* xyz, const void* quaternion, const void* scale, int n_sh,
                                    const void* camera_T_world, const void* K, const void* camera_center,
                                    const int32_t* rank_of_gaussian, const void* opacity_act, const int32_t* workspace,
                                    int N_total, int G, int rank, const int32_t* owner_blocks, const void* recv,
                                    const int32_t* recv_offsets, int n, void* grad_xyz, void* grad_quaternion,
                                    void* grad_scale, void* grad_opacity_logit, void* grad_rgb_param, void* grad_sh,
                                    void* stream);

/* ---- training-loop operations behind the rasterizer ------------------------------
 * gs_adam_step: torch.optim.Adam.step() as the reference uses it (splat_py/optimizer_manager.py:15-42
 * builds the optimizer with one group per parameter tensor and a learning rate each; trainer.py:376
 * steps it): defaults amsgrad=False, weight_decay=0, maximize=False.  One launch updates up to 8
 * parameter tensors in place.  params/grads/exp_avg/exp_avg_sq: host arrays of n_groups device
 * pointers (fp32, numel[k] elements each); lr[k], step[k] (the 1-based step count AFTER the
 * increment, per tensor as in torch's state["step"]); beta1, beta2, eps as the Python floats.
 * Update order == torch/optim/adam.py _single_tensor_adam on the ATen CPU kernels. */
int gs_adam_step(int n_groups, void* const* params, const void* const* grads, void* const* exp_avg,
                 void* const* exp_avg_sq, const int64_t* numel, const double* lr,
                 const int64_t* step, double beta1, double beta2, double eps, void* stream);
/* Measurement aid (ABI 7; bench.py `roofline.hbm_copy_gbs_measured`): a float4 grid-stride stream copy of `bytes` bytes
 * (multiple of 16, as are both pointers) with `blocks` workgroups of 256 lanes, non-temporal loads and stores.
 * No reference counterpart. */
int gs_stream_copy(void* dst, const void* src, size_t bytes, int blocks, void* stream);
/* Densification statistics of trainer.py:378-385 in one pass and without the boolean-mask
 * index_put: for every Gaussian i with rank[i] >= 0 (its visible index; -1 = culled)
 *   uv_grad_accum[i] += |uv_grad[rank[i]] * (K[0,0], K[1,1])|,  grad_accum_count[i] += 1
 * (K: the device-resident fp32 3x3 intrinsics, read on the device: no host sync)
 * and for every i  xyz_grad_accum[i] += |xyz_grad[i]|  (skipped when xyz_grad is NULL).
 * uv_grad: rows of 2 floats, uv_row_stride floats apart (9 for the view of the fused path's slab). */
int gs_accumulate_grad_stats(const void* uv_grad, int uv_row_stride, const int32_t* rank,
                             const void* xyz_grad, const void* K, int N, void* uv_grad_accum,
                             void* xyz_grad_accum, int32_t* grad_accum_count, void* stream);

/* The training loss of trainer.py:363-374, value and gradient in one call:
 *   loss = (1 - ssim_frac) * l1_loss(image, target) + ssim_frac * (1 - SSIM(image, target))
 * SSIM = torchmetrics 1.2.1 StructuralSimilarityIndexMeasure(data_range=1.0) (trainer.py:24): 11x11
 * Gaussian window (sigma 1.5), k1 0.01, k2 0.03, mean over the pixels whose window lies inside the
 * image.  image, target: [H, W, 3] fp32 (the rasterizer's layout; H, W > 10).
 * loss_out: float[4] = (loss, l1, ssim, mse) -- mse is the trainer's debug l2_loss (trainer.py:366-367,
 * psnr = -10 log10(mse)).  grad_image: [H, W, 3] = d loss / d image, or NULL.
 * workspace: gs_ssim_l1_workspace_bytes(H, W) bytes (per-tile partial sums, reduced in a fixed order). */
size_t gs_ssim_l1_workspace_bytes(int H, int W);
int gs_ssim_l1_loss(const void* image, const void* target, int H, int W, float ssim_frac,
                    void* workspace, void* loss_out, void* grad_image, void* stream);

/* Adaptive density control, the data movement of splat_py/trainer.py:114-206 (delete / clone / split) and
 * of splat_py/optimizer_manager.py:78-172 (the same surgery on Adam's exp_avg / exp_avg_sq) in one pass:
 * every source row is read once and every row of the new Gaussian set is written once, for up to 8
 * parameter tensors (fp32, `width[k]` floats per row, N0 rows) together with their optimizer state
 * (src_exp_avg[k] / src_exp_avg_sq[k]: NULL when the tensor has no state yet).  The per-row decisions
 * come as destination indices (gaussian_splatting_amd/densify.py forms them from the reference's masks
 * with prefix sums; -1 = does not apply):
 *   dst_self[i]     row of Gaussian i in the new set when it survives unsplit (parameters + state copied)
 *   dst_clone[i]    row of its clone (trainer.py:123-161): parameters copied, xyz - 0.01 * xyz_grad_accum /
 *                   grad_accum_count, optimizer state zero
 *   split_self[i], split_clone[i]   rank among the n_split split rows when the Gaussian (its clone) is
 *                   split (trainer.py:163-206): `samples` new rows at sample_base + s * n_split + rank with
 *                   xyz + R(q / |q|) (random * exp(scale)), scale = log(exp(scale) / split_scale_factor),
 *                   quaternion = q / |q|, the rest copied, optimizer state zero
 * kind[k]: 0 plain, 1 xyz, 2 scale (log), 3 quaternion (w, x, y, z) -- the tensors the split transforms;
 * xyz / scale / quaternion: the SOURCE tensors again (read by the transform whatever tensor a thread moves);
 * random_samples: [n_split * samples, 3] uniform numbers in the order of trainer.py:176.
 * The destination tensors are caller-allocated with the new row count and must not alias the sources. */
int gs_densify_move(int n_tensors, const void* const* src, void* const* dst, const void* const* src_exp_avg,
                    void* const* dst_exp_avg, const void* const* src_exp_avg_sq, void* const* dst_exp_avg_sq,
                    const int32_t* width, const int32_t* kind, int N0, const int32_t* dst_self,
                    const int32_t* dst_clone, const int32_t* split_self, const int32_t* split_clone,
                    const void* xyz, const void* scale, const void* quaternion, const void* xyz_grad_accum,
                    const int32_t* grad_accum_count, const void* random_samples, int n_split, int samples,
                    int sample_base, float split_scale_factor, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HIP_H */
