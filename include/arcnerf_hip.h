/*
 * arcnerf_hip.h — C ABI of libarcnerf_hip.so: the MI355X (gfx950) volumetric-rendering hot path of ArcNerf.
 *
 * Conventions:
 *   - plain pointers + sizes, no torch types.  All pointers are DEVICE pointers unless the name ends in `_host`.
 *   - the CALLER owns and allocates every buffer (zero-initialised where the reference's wrapper does so);
 *     nothing here allocates, frees or synchronises the device.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Kernels are launched on it and the
 *     call returns immediately.
 *   - return value: 0 = ok, ARCN_EINVAL (-1) = bad argument, ARCN_ELAUNCH (-2) = hip launch error
 *     (arcn_last_error() returns the hip error string).
 *   - fp32 everywhere ("dtype": "f32"); contiguous row-major tensors, same shapes as the reference ops.
 *
 * Each entry point cites the reference interface it replaces (file:line relative to the ArcNerf repo).
 */
#ifndef ARCNERF_HIP_H
#define ARCNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARCN_OK 0
#define ARCN_EINVAL (-1)
#define ARCN_ELAUNCH (-2)

#define ARCN_MAX_LEVELS 32

/* activation codes (tcnn names in arcnerf/models/base_modules/geo_rad_model/tcnn_fusedmlp_module.py:195-213) */
#define ARCN_ACT_NONE 0
#define ARCN_ACT_RELU 1
#define ARCN_ACT_SIGMOID 2
#define ARCN_ACT_TRUNCEXP 3 /* fwd exp(x); bwd g*exp(clamp(x,-15,15))  (arcnerf/ops/trunc_exp.py:7-37) */
#define ARCN_ACT_SOFTPLUS 4
/* the remaining two activations of the reference's fused-MLP map (tcnn_fusedmlp_module.py:195-213), tiny-cuda-nn's definitions: */
#define ARCN_ACT_SQUAREPLUS 5 /* y = (X + sqrt(X^2 + 4)) / 2 / 10 with X = 10 x; dy/dx = Y^2 / (Y^2 + 1), Y = 10 y */
#define ARCN_ACT_SINE 6       /* y = sin(x); dy/dx = cos(x) needs the PRE-activation: elementwise ops and forward passes only - the fused
                               * MLP backward works from the post-activations it saved and refuses Sine (tiny-cuda-nn's fused MLP does too) */

const char *arcn_last_error(void);
int arcn_version(void);

/* ------------------------------------------------------------------------------------------------
 * _volume_func (arcnerf/ops/src/volume_func/volume_func.cpp:277-282)
 * ---------------------------------------------------------------------------------------------- */

/* K1 check_pts_in_occ_voxel (volume_func_kernel.cu:16-68). xyz (n,3); bitfield (n_grid^3) bool bytes;
 * aabb (2,3) = xyz_min,xyz_max; out (n) bool bytes. */
int arcn_check_pts_in_occ_voxel(const float *xyz, const uint8_t *bitfield, const float *aabb, int n_grid, uint8_t *out,
                                int64_t n, void *stream);

/* K2 aabb_intersection (volume_func_kernel.cu:74-166). rays (n_rays,3); aabb (n_v,2,3); near/far/mask (n_rays,n_v);
 * pts (n_rays,n_v,2,3). mask = tmin > 0. */
int arcn_aabb_intersection(const float *rays_o, const float *rays_d, const float *aabb, float *near, float *far,
                           float *pts, uint8_t *mask, int64_t n_rays, int64_t n_v, void *stream);

/* torch-path semantics of aabb_ray_intersection (arcnerf/geometry/ray.py:295-339): aabb (n_v,3,2), eps shift, rays
 * starting inside the box are hits with near = eps. */
int arcn_aabb_intersection_torch(const float *rays_o, const float *rays_d, const float *aabb32, float eps, float *near,
                                 float *far, float *pts, uint8_t *mask, int64_t n_rays, int64_t n_v, void *stream);
/* sphere_ray_intersection (arcnerf/geometry/ray.py:180-255; SphereBound, sphere_bound.py:10-37): radius (n_r) device,
 * origin_host[3] HOST; near/far (n_rays,n_r), pts (n_rays,n_r,2,3) optional, mask (n_rays,n_r) bytes.  Misses give 0/0,
 * rays starting inside hit with near 0, |x| < 1e-5 intermediates are flushed to 0 like set_tensor_to_zeros. */
int arcn_sphere_intersection(const float *rays_o, const float *rays_d, const float *radius, const float *origin_host,
                             float *near, float *far, float *pts, uint8_t *mask, int64_t n_rays, int64_t n_r, void *stream);

/* get_rays (arcnerf/render/ray_helper.py:12-119) with pixel_to_world (geometry/projection.py:8-66) and get_ndc_rays
 * (ray_helper.py:122-153): intrinsic (3,3) and c2w (4,4) row-major DEVICE floats.  index == NULL: all W*H pixels, column-major
 * (wh_order = 1: p = i * H + j) or row-major (p = j * W + i); else index (n) int64 holds column-major pixel ids i * H + j.
 * rays_o, rays_d (n,3); rays_r (n) optional, full-image mode only: the mip-nerf radius |d(i,j) - d(i+1,j)| * 2 / sqrt(12) (the
 * last column takes column W-3's value, as the reference's `dx[-2:-1]` does). */
int arcn_get_rays(int W, int H, const float *intrinsic, const float *c2w, int wh_order, const int64_t *index, int64_t n,
                  int center_pixel, int normalize_rays_d, int ndc, float ndc_near, float *rays_o, float *rays_d, float *rays_r,
                  void *stream);

/* The training batch: Pipeline.fetch_step_ray_sample + fetch_step_bkg_color (arcnerf/trainer/pipeline.py:243-300) on a dataset kept
 * as IMAGES + CAMERAS instead of the per-pixel tensors arcnerf_trainer.py:188-219 concat_train_batch collects (img, mask, rays_o,
 * rays_d, rays_r per pixel: 44 B) and step_crop_center_image / step_ray_sample (pipeline.py:95-118,150-168) crop and shuffle.
 *   rgba (n_img,H,W,4) bytes as a Blender PNG holds them (colours = bytes / 255.0, mask = alpha / 255.0, nerf_dataset.py:107-119)
 *   OR img (n_img,H,W,3) floats with mask (n_img,H,W) floats or NULL; intrinsic (n_img,3,3), c2w (n_img,4,4) row-major; all DEVICE.
 *   ids (n) int64 DEVICE: row ids of the reference's (cropped) dataset tensor, id = view * Hc * Wc + (y - y0) * Wc + (x - x0) for the
 *   window [y0, y0 + Hc) x [x0, x0 + Wc) (no crop: 0, 0, H, W); the shuffle is a permutation of them.  An id outside [0, n_img*Hc*Wc)
 *   is counted into *bad_ids (DEVICE int32, optional) and replaced by 0.
 *   bkg_rand (n,3) DEVICE (the draw of torch.rand_like, pipeline.py:286) or bkg_const_host[3] HOST or neither (no blend); the blend
 *   img * mask + (1 - mask) * bkg happens only when the data has a mask (pipeline.py:281-283).
 * Outputs, each optional, (n, .): rays_o, rays_d (get_rays of the pixel, wh_order=False: y * W + x), rays_r (mip-nerf radius as in
 * full-image mode), img_out (blended target), mask_out, bkg_out (the colour blended in), src_out int64 (row of the UNCROPPED
 * (n_img*H*W, .) tensors: for gathering further per-pixel data such as bounds). */
int arcn_fetch_train_batch(const uint8_t *rgba, const float *img, const float *mask, const float *intrinsic, const float *c2w, int n_img,
                           int H, int W, int y0, int x0, int Hc, int Wc, const int64_t *ids, int64_t n, int center_pixel,
                           int normalize_rays_d, const float *bkg_rand, const float *bkg_const_host, float *rays_o, float *rays_d,
                           float *rays_r, float *img_out, float *mask_out, float *bkg_out, int64_t *src_out, int32_t *bad_ids,
                           void *stream);

/* K3 sparse_volume_sampling (volume_func_kernel.cu:174-291). zvals/mask (n_rays,n_pts) zero-initialised by the
 * caller.  (rng_state, rng_inc) is the host pcg32 BEFORE the call (reference: file-static `pcg32 rng{9121}`,
 * include/common.h:22-23, advanced 2^32 after every launch); the caller owns that bookkeeping (arcn_pcg32_*).
 * counts (n_rays) int32 optional: emitted samples per ray. */
int arcn_sparse_volume_sampling(const float *rays_o, const float *rays_d, const float *near, const float *far, int n_pts,
                                float dt, const float *aabb, int n_grid, const uint8_t *bitfield, float near_distance,
                                uint64_t rng_state, uint64_t rng_inc, float *zvals, uint8_t *mask, int32_t *counts,
                                int64_t n_rays, void *stream);

/* K4 tensor_reduce_max (volume_func_kernel.cu:297-337): uni[idx[i]] = max(uni[idx[i]], full[i]) on the uint32 bit
 * pattern (valid for non-negative floats). uni (n_group) zero-initialised by the caller. */
int arcn_tensor_reduce_max(const float *full, const int64_t *idx, int n_group, float *uni, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * `_bitfield_func` (arcnerf/ops/src/bitfield_func/bitfield_func.cpp:279-286; BitfieldBound, obj_bound/bitfield_bound.py):
 * a float density grid of n_grid^3 cells in MORTON order (volume_func.h:141-168) and its packed bitfield of n_grid^3 / 8
 * bytes.  n_grid must be a power of two <= 1024 (10 Morton bits per axis).  The module keeps its own file-static
 * `pcg32 rng{9121}` shared by K5 and K6 and advanced 2^32 after each of them; here the caller passes (state, inc).
 * ---------------------------------------------------------------------------------------------- */
/* K5 sparse_volume_sampling_bit (bitfield_func_kernel.cu:20-136): K3's loop with the occupancy test
 * density_grid_occupied_at_bit (voxel coordinates truncated, clamped into the grid, Morton bit).  Same outputs as K3. */
int arcn_sparse_volume_sampling_bit(const float *rays_o, const float *rays_d, const float *near, const float *far, int n_pts,
                                    float dt, const float *aabb, int n_grid, const uint8_t *bitfield, float near_distance,
                                    uint64_t rng_state, uint64_t rng_inc, float *zvals, uint8_t *mask, int32_t *counts,
                                    int64_t n_rays, void *stream);
/* K6 generate_grid_samples (bitfield_func_kernel.cu:141-212): n_elements cells picked by the (i, ema_step) uint32 LCG, up to
 * 10 probes for one with density > thresh; positions (n,3) jittered in [0,1)^3, indices (n) int32 Morton cell. */
int arcn_generate_grid_samples(const float *density_grid, int ema_step, int n_elements, int n_grid, float thresh,
                               uint64_t rng_state, uint64_t rng_inc, float *positions, int32_t *indices, void *stream);
/* K7 splat_grid_samples (bitfield_func_kernel.cu:215-252): grid_tmp[idx[i]] = max(grid_tmp[idx[i]], density[i]) on the bit
 * pattern (non-negative values). */
int arcn_splat_grid_samples(const float *density, const int32_t *indices, int n_samples, float *density_grid_tmp, void *stream);
/* K8 ema_grid_samples_nerf (bitfield_func_kernel.cu:257-296): grid = grid < 0 ? grid : max(grid * decay, grid_tmp). */
int arcn_ema_grid_samples_nerf(const float *density_grid_tmp, int n_elements, float decay, float *density_grid, void *stream);
/* K9 update_bitfield / grid_to_bitfield (bitfield_func_kernel.cu:301-345): bit = grid > min(opa_thres, mean).  The mean is
 * the host float of the reference signature, or, when density_grid_mean_dev != NULL, read from that device scalar. */
int arcn_update_bitfield(const float *density_grid, float density_grid_mean, const float *density_grid_mean_dev,
                         uint8_t *bitfield, float opa_thres, int n_grid, void *stream);
/* K10 count_bitfield (bitfield_func_kernel.cu:350-389): counter[0] += 8 for every NON-ZERO byte (the reference tests
 * `byte && (1 << j)`, a logical and); counter is a device float the caller zeroes. */
int arcn_count_bitfield(const uint8_t *bitfield, float *counter, int n_grid, void *stream);

/* ------------------------------------------------------------------------------------------------
 * `_multivol_func` (arcnerf/ops/src/multivol_func/multivol_func.cpp; MultiVol background, models/multivol_bkg_model.py):
 * n_cascade nested volumes — volume m = the inner box scaled 2^m about its centre — each an n_grid^3 Morton grid, stored level
 * after level; with inclusive = 0 the inner volume has no grid and level m lives in slot m-1.  Own pcg32 stream like
 * `_bitfield_func` (K11 and K12 advance it 2^32 per launch).
 * ---------------------------------------------------------------------------------------------- */
/* K11 sparse_sampling_in_multivol_bitfield (multivol_func_kernel.cu:14-146): marching with dt = clamp(t * cone_angle,
 * min_step, max_step) through the cascade (level from the point's largest normalised coordinate, volume_func.h:201-226);
 * min_aabb (2,3) = inner volume, aabb (2,3) = outermost.  inclusive = 0: a ray that re-enters the inner volume drops the
 * samples taken so far.  Outputs as K3. */
int arcn_sparse_sampling_in_multivol_bitfield(const float *rays_o, const float *rays_d, const float *near, const float *far,
                                              int n_pts, float cone_angle, float min_step, float max_step,
                                              const float *min_aabb, const float *aabb, int n_grid, int n_cascade,
                                              const uint8_t *bitfield, float near_distance, int inclusive, uint64_t rng_state,
                                              uint64_t rng_inc, float *zvals, uint8_t *mask, int32_t *counts, int64_t n_rays,
                                              void *stream);
/* K12 generate_grid_samples_multivol (multivol_func_kernel.cu:148-240): K6 over the cascade; aabb (2,3) = inner volume;
 * positions in world space of the drawn level, indices = slot * n_grid^3 + Morton cell. */
int arcn_generate_grid_samples_multivol(const float *density_grid, int ema_step, int n_elements, const float *aabb,
                                        int n_cascade, int n_grid, float thresh, int inclusive, uint64_t rng_state,
                                        uint64_t rng_inc, float *positions, int32_t *indices, void *stream);
/* update_bitfield_multivol (multivol_func_kernel.cu:242-300): K9 over all stored levels. */
int arcn_update_bitfield_multivol(const float *density_grid, float density_grid_mean, const float *density_grid_mean_dev,
                                  uint8_t *bitfield, float opa_thres, int n_grid, int n_cascade, int inclusive, void *stream);

/* host pcg32 helpers (include/pcg32.h:50-165), HOST pointers: state_inc_host[2] = {state, inc}. */
void arcn_pcg32_seed(uint64_t initstate, uint64_t initseq, uint64_t *state_inc_host);
void arcn_pcg32_advance(uint64_t *state_inc_host, int64_t delta);

/* ------------------------------------------------------------------------------------------------
 * Compacted sampler: the same marching as K3 fused with K2, emitting the packed form the fast path consumes.
 *   pass 1  arcn_march_count : near/far (K2 or torch semantics) + march, counts (n_rays) int32, optional dense scratch
 *   pass 2  arcn_exclusive_scan_i32 : offsets (n_rays+1) int32 (offsets[n_rays] = total); max_total > 0 clamps every offset to
 *           the capacity of the packed buffers: rays past it keep a truncated (possibly empty) segment, nothing downstream
 *           indexes beyond the buffers when the rays ask for more samples than they hold; max_out (optional, device) receives
 *           max(counts), the dense width P the reference would have used (fg_model.py:251-262)
 *   pass 3  arcn_march_write : t (S) float, ray_id (S) int32 in ray-major order; dense t scratch reused.
 * scratch_t (n_rays,n_pts) float holds the emitted t of pass 1 (no init needed).
 * bitfield_is_packed: 0 = bool per voxel (K3 layout), 1 = 1 bit per voxel in the same x-major order, 2 = 1 bit per voxel
 * in Morton order with clamped coordinates (the K5 / BitfieldBound layout).
 * ---------------------------------------------------------------------------------------------- */
int arcn_march_count(const float *rays_o, const float *rays_d, const float *aabb, int n_grid, const uint8_t *bitfield,
                     int bitfield_is_packed, int n_pts, float dt, float near_distance, int aabb_torch_semantics,
                     uint64_t rng_state, uint64_t rng_inc, float *scratch_t, int32_t *counts, float *near_out,
                     float *far_out, int64_t n_rays, void *stream);
/* Ray culling for pass 1.  arcn_march_cull_grid: coarse ((n_grid / 4)^3 bytes) = 1 where any voxel of a 4^3 block or of one of its 26
 * neighbour blocks is occupied (tmp: same size; n_grid a multiple of 4, >= 16; rebuild whenever the bitfield changes).
 * arcn_march_count_culled = arcn_march_count with that grid: before marching, a wave tests 64 points spread over its ray's [near, far]
 * (at most one block apart per axis, else the ray is marched as usual) and leaves with count 0 when none of them sees an occupied
 * block - what the reference's loop (volume_func_kernel.cu:174-236) produces after stepping through its empty voxels.  Outputs are
 * bit-identical to arcn_march_count. */
int arcn_march_cull_grid(const uint8_t *bitfield, int bitfield_is_packed, int n_grid, uint8_t *coarse, uint8_t *tmp, void *stream);
int arcn_march_count_culled(const float *rays_o, const float *rays_d, const float *aabb, int n_grid, const uint8_t *bitfield,
                            int bitfield_is_packed, const uint8_t *coarse, int n_pts, float dt, float near_distance,
                            int aabb_torch_semantics, uint64_t rng_state, uint64_t rng_inc, float *scratch_t, int32_t *counts,
                            float *near_out, float *far_out, int64_t n_rays, void *stream);
/* arcn_march_count_culled (coarse may be NULL: arcn_march_count) as n_waves PERSISTENT wavefronts: wave w marches the rays w, w + n_waves, ...
 * instead of one wavefront per ray (n_waves <= 0 or >= n_rays: one wavefront per ray).  Same samples, bit for bit (a ray's jitter comes
 * from the launch's generator advanced by the ray's index).  For a launch that runs BESIDE other work with time to spare - the marching of a
 * batch two training steps ahead on a second stream: 8 320 one-ray wavefronts hold ~2.3 long-lived waves on every SIMD for 60 us, 4 096
 * persistent ones half of that for twice as long, and the step's kernels lose 2 % less to them.  Alone, or with one step
 * to finish in, the one-wavefront-per-ray form is the faster one. */
int arcn_march_count_waves(const float *rays_o, const float *rays_d, const float *aabb, int n_grid, const uint8_t *bitfield,
                            int bitfield_is_packed, const uint8_t *coarse, int n_pts, float dt, float near_distance,
                            int aabb_torch_semantics, uint64_t rng_state, uint64_t rng_inc, float *scratch_t, int32_t *counts,
                            float *near_out, float *far_out, int64_t n_rays, int n_waves, void *stream);
/* The three passes above in ONE launch (no dense scratch): a wave keeps its ray's samples in LDS, the workgroups' counts go through a
 * chained scan (decoupled look-back, ray blocks handed out by ticket), the waves copy their samples to their final offsets.
 * Same outputs as the three-pass form, bit for bit (offsets clamped to `capacity`, p_dense = largest per-ray count).
 * workspace: arcn_march_packed_workspace_bytes(n_rays) bytes of device memory (zeroed by the call).  Should the look-back ever give up
 * on a predecessor (2^24 polls), offsets[n_rays] is set to -1: every consumer of the device-side count sees an empty batch instead of
 * offsets built from an incomplete prefix. */
int64_t arcn_march_packed_workspace_bytes(int64_t n_rays);
int arcn_march_packed(const float *rays_o, const float *rays_d, const float *aabb, int n_grid, const uint8_t *bitfield,
                      int bitfield_is_packed, int n_pts, float dt, float near_distance, int aabb_torch_semantics, uint64_t rng_state,
                      uint64_t rng_inc, int32_t *counts, float *near_out, float *far_out, int32_t *offsets, float *t_packed,
                      int32_t *ray_id, int64_t capacity, int32_t *p_dense, void *workspace, int64_t n_rays, void *stream);
int arcn_exclusive_scan_i32(const int32_t *counts, int32_t *offsets, int64_t n, int64_t max_total, int32_t *max_out,
                            void *stream);
int arcn_march_write(const float *scratch_t, const int32_t *counts, const int32_t *offsets, int n_pts, float *t_packed,
                     int32_t *ray_id, int64_t n_rays, int64_t capacity, void *stream);

/* pts/dirs of packed samples: xyz[s] = o[ray]+d[ray]*t[s] (geometry/ray.py:11-30), dirs[s] = d[ray].
 * n_ptr (device, optional) overrides n with *n_ptr (device-side sample count, no host sync). */
int arcn_packed_points(const float *rays_o, const float *rays_d, const float *t_packed, const int32_t *ray_id,
                       float *xyz, float *dirs, int64_t n, const int32_t *n_ptr, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Encoders
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_levels;
    int32_t n_feat;                              /* F: 1, 2 or 4 */
    int32_t resolutions[ARCN_MAX_LEVELS];        /* hashgrid_encoder.py:144-149 */
    int64_t offsets[ARCN_MAX_LEVELS + 1];        /* cumulative rows, offsets[L] = n_total_embed */
    float min_xyz[3];
    float max_xyz[3];
} arcn_hashgrid_desc;

/* HashGridEmbedder.hashgrid_encode_torch (encoding/hashgrid_encoder.py:191-249 + geometry/volume.py:486-570).
 * xyz (n,3); table (n_total,F); out (n, L*F). desc_host is a HOST pointer (copied into the launch).
 * hash_idx (n,L,8) int32 optional debug output (-1 for out-of-volume). */
int arcn_hashgrid_fwd(const float *xyz, const float *table, const arcn_hashgrid_desc *desc_host, float *out,
                      int32_t *hash_idx, int64_t n, const int32_t *n_ptr, void *stream);
/* The input gradient dxyz = J(xyz; table)^T dout of arcn_hashgrid_bwd differentiated once more, for models that put a loss on
 * d(output)/d(xyz) (NeuS normals on a hash grid: BaseGeoNet.forward_with_grad, base_network.py, with the torch backend of
 * hashgrid_encoder.py:191-230 under double autograd; tcnn's grid encoding has the same second-order terms).  gdx (n,3) = gradient
 * arriving on dxyz.  Outputs, each optional: ddout (n, L*F) = d/d dout; dtable (n_total,F) += d/d table (caller zeroes);
 * d2xyz (n,3) += d/d xyz (caller zeroes; cross terms of the trilinear weights).  workspace (optional, at least
 * arcn_hashgrid_bwd_workspace_floats(desc, 2 * n) floats): the table part runs as a binned scatter (8 single-row records per
 * sample and level, same consumer as arcn_hashgrid_bwd) instead of one float atomic per corner and feature. */
int arcn_hashgrid_bwd_bwd(const float *xyz, const float *gdx, const float *table, const float *dout,
                          const arcn_hashgrid_desc *desc_host, float *ddout, float *dtable, float *d2xyz, float *workspace,
                          int64_t workspace_floats, int64_t n, const int32_t *n_ptr, void *stream);
/* The second-order gathers of NeuS on the hash grid WITHOUT their random reads.  The normal (d enc / d x)^T jac and the gradient of that
 * product with respect to jac (hashgrid_encoder.py through base_network.py:30-44, create_graph = True) read the SAME eight table rows per
 * (sample, level) the forward gathered.  arcn_hashgrid_fwd_corners is arcn_hashgrid_fwd_xcd (below: the XCD-affine gather, same `out`,
 * same level_major / n_cap / n_ptr) that also keeps those rows, LEVEL-major in 16-byte quads - quad j of (sample s, level l) holds rows
 * 2j, 2j + 1 at F = 2 (rows 4j .. 4j + 3 at F = 1) at corners[((l * 2 F + j) * n_cap + s) * 4], zeros outside the grid: 8 F n_cap L floats,
 * every store and every later read one contiguous KiB per wavefront - and the two consumers stream them, one lane per sample:
 *   arcn_hashgrid_dxyz_corners : dxyz (n, 3)    = what arcn_hashgrid_bwd adds to a cleared dxyz        (dout (n, L F) row-major)
 *   arcn_hashgrid_ddout_corners: ddout (n, L F) = the ddout of arcn_hashgrid_bwd_bwd                   (gdx (n, 3))
 * The same arithmetic in the same order on the same values: bit-identical to the table forms.  n_feat 1 or 2. */
int arcn_hashgrid_fwd_corners(const float *xyz, const float *table, const arcn_hashgrid_desc *desc_host, float *out, int level_major,
                              float *corners, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream);
int arcn_hashgrid_dxyz_corners(const float *xyz, const float *corners, const float *dout, const arcn_hashgrid_desc *desc_host, float *dxyz,
                               int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream);
int arcn_hashgrid_ddout_corners(const float *xyz, const float *gdx, const float *corners, const arcn_hashgrid_desc *desc_host, float *ddout,
                                int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream);
/* level-major features lm[(l * n_cap + s) * n_feat + f] (arcn_hashgrid_fwd_xcd / _fwd_corners with level_major = 1) -> row-major
 * rows[s * ld + l * n_feat + f], the operand layout of the dense products (arcn_gemm_*): one LDS-tiled pass, whole cache lines on both
 * sides.  HashGridEmbedder.forward's (n, L F) output (hashgrid_encoder.py:160-189) = the XCD-affine gather + this. */
int arcn_hashgrid_lm_to_rows(const float *lm, int n_levels, int n_feat, int64_t n_cap, float *rows, int64_t ld, int64_t n,
                             const int32_t *n_ptr, void *stream);
/* The table's FIRST- and SECOND-order gradients of one batch in one consumer pass: dtable += d/d table of <dout, enc(xyz)> (what
 * arcn_hashgrid_bwd adds) + d/d table of <gdx, J(xyz; table)^T dout_dx> (what arcn_hashgrid_bwd_bwd adds) - both producers fill the same
 * bins, one accumulation pass over the table instead of two.  workspace: at least arcn_hashgrid_bwd_workspace_floats(desc, 3 * n). */
int arcn_hashgrid_bwd_first_second(const float *xyz, const float *dout, const float *gdx, const float *dout_dx,
                                   const arcn_hashgrid_desc *desc_host, float *dtable, float *workspace, int64_t workspace_floats, int64_t n,
                                   void *stream);
/* arcn_hashgrid_bwd_first_second whose chunk owners ALSO apply the optimiser to the table levels they own alone - the arguments and the
 * contract of arcn_hashgrid_bwd_lm_adam below (table / exp_avg / exp_avg_sq at row 0 of level 0, *fused_levels_host = the levels done;
 * the other levels' gradient lands in dtable).  The NeuS-on-hash-grid step: neither of the table's two gradients goes to HBM. */
int arcn_hashgrid_bwd_first_second_adam(const float *xyz, const float *dout, const float *gdx, const float *dout_dx,
                                        const arcn_hashgrid_desc *desc_host, float *dtable, float *table, float *exp_avg, float *exp_avg_sq,
                                        float lr, float beta1, float beta2, float eps, float weight_decay, float ema_decay, float grad_scale,
                                        int step, int ema_step, float *workspace, int64_t workspace_floats, int64_t n,
                                        uint32_t *fused_levels_host, void *stream);
/* Same result as arcn_hashgrid_fwd (bit-identical), scheduled so that each of the chip's 8 XCDs gathers only its own
 * 2 of 16 levels (a level's table slice then stays in that XCD's L2); n_feat 1 or 2.
 * level_major = 0: out (n, L*F) row-major; level_major = 1: out[(l * n_cap + s) * F + f]. */
int arcn_hashgrid_fwd_xcd(const float *xyz, const float *table, const arcn_hashgrid_desc *desc_host, float *out,
                          int level_major, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream);
/* backward: dtable (n_total,F) accumulated (caller zeroes), dxyz (n,3) optional.
 * workspace (device, workspace_floats >= arcn_hashgrid_bwd_workspace_floats(desc, n) floats, optional; -1 if too small):
 * selects the binned scatter.  A first pass turns every (level, sample) into up to four 16-byte records, one per pair of
 * x-neighbour corners, appended to the bin of the table chunk (<= 16384 rows) that owns their rows; a second pass gives
 * every chunk to one workgroup, which streams its bin, accumulates in LDS and writes the chunk back with plain stores - no
 * global float atomics on the large levels.  Without a workspace, when dxyz is requested, or for n_feat 4, the plain
 * one-atomic-per-corner kernel runs.  Both ADD into dtable. */
int arcn_hashgrid_bwd(const float *xyz, const float *table, const float *dout, const arcn_hashgrid_desc *desc_host,
                      float *dtable, float *dxyz, float *workspace, int64_t workspace_floats, int64_t n,
                      const int32_t *n_ptr, void *stream);
int64_t arcn_hashgrid_bwd_workspace_floats(const arcn_hashgrid_desc *desc_host, int64_t n);
/* ARCN_DETERMINISTIC=1 in the environment (read once): the binned scatter accumulates in 64-bit fixed point with integer LDS atomics
 * (order-independent sums: two runs of a training give bit-identical parameters; the reference's CPU path is deterministic too) instead
 * of float compare-and-swap.  arcn_deterministic() reports the mode.  After arcn_hashgrid_bwd / _lm / _bwd_bwd with a workspace, the two
 * uint32 words at workspace[arcn_hashgrid_bwd_status_offset(desc, n)] hold {bits of the launch's largest |gradient|, 1 if a bin
 * overflowed into the (order-dependent) direct float atomics}: a deterministic run checks the second word is 0. */
int arcn_deterministic(void);
int64_t arcn_hashgrid_bwd_status_offset(const arcn_hashgrid_desc *desc, int64_t n);
/* binned scatter with LEVEL-MAJOR gradients dout_lm[(l * dout_stride + s) * F + f] (the layout arcn_hashgrid_fwd_xcd writes
 * and arcn_mlp_bwd_lm produces); workspace required, dtable only. */
int arcn_hashgrid_bwd_lm(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host,
                         float *dtable, float *workspace, int64_t workspace_floats, int64_t n, const int32_t *n_ptr,
                         void *stream);
/* arcn_hashgrid_bwd_lm restricted to the levels of `level_mask` (bit l = level l): the data-parallel step scatters the table gradient in
 * level GROUPS so that a group's all-reduce (its contiguous slice of the flat gradient buffer) is on the wire while the next group is
 * still being scattered - what DistributedDataParallel's buckets do for the reference (common/trainer/basic_trainer.py:197-198).  The
 * calls of one step must cover every level once, stream-ordered on one workspace.  counters_clear = 0: the call clears the block of bin
 * counters first (the first group of a step); 1: a previous group of the same step did (levels use disjoint counters). */
int arcn_hashgrid_bwd_lm_levels(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host,
                                float *dtable, float *workspace, int64_t workspace_floats, int64_t n, const int32_t *n_ptr,
                                uint32_t level_mask, int counters_clear, void *stream);
/* arcn_hashgrid_bwd_lm with the OPTIMISER fused into the scatter's consumer (single-GPU training step; with several ranks the summed
 * gradient only exists after the all-reduce, so they keep the two calls): the owner workgroup of a table chunk holds the chunk's complete
 * gradient in LDS and applies arcn_adam_ema_step's update (torch.optim.Adam + EMA.ema_step written back, arcnerf/trainer/ema.py:29-43) to
 * its rows of `table` / `exp_avg` / `exp_avg_sq` (each pointing at row 0 of level 0) - the gradient of those levels never goes to HBM.
 * Applies to the levels with one owner per chunk; *fused_levels_host (HOST pointer) receives their bit mask, the other levels' gradient is
 * accumulated into dtable as usual and the caller runs arcn_adam_ema_step on their rows (and on every other parameter).
 * ema_decay < 0: plain Adam; >= 0: the EMA with its shadow aliased onto the parameter (the ema == param form of arcn_adam_ema_step).
 * Not available with ARCN_DETERMINISTIC=1 (mask 0: nothing fused, plain scatter).  counters_clear: see arcn_hashgrid_bwd_counter_words.
 * dout_stride = 0: dout is the sample-major (n, L F) gradient of arcn_hashgrid_bwd instead of the level-major one. */
/* bit mask of the levels arcn_hashgrid_bwd_lm_adam would apply the optimiser to for a workspace plan of n samples (0: none) */
int64_t arcn_hashgrid_bwd_fusable_levels(const arcn_hashgrid_desc *desc_host, int64_t n);
int arcn_hashgrid_bwd_lm_adam(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host, float *dtable,
                              float *table, float *exp_avg, float *exp_avg_sq, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float ema_decay, float grad_scale, int step, int ema_step, float *workspace,
                              int64_t workspace_floats, int counters_clear, int64_t n, const int32_t *n_ptr, uint32_t *fused_levels_host,
                              void *stream);
/* The PLANNED form of the binned scatter.  Everything the scatter's producer derives from the sample positions alone - cells, the runs of
 * samples sharing a cell, the 64-bit modulo of the dense levels, owner bins, ranks, the index half {i0 | i1 << 16, wx} of every record - is
 * computed by arcn_hashgrid_bwd_plan as soon as the samples exist (the training step marches its batches two steps early: on the sampling
 * stream, off the step's critical path) into a plan workspace of arcn_hashgrid_plan_workspace_floats(desc, n) floats (0: no planned form for
 * this table / n).  The step itself then runs arcn_hashgrid_bwd_lm_planned / arcn_hashgrid_bwd_lm_adam_planned on the SAME xyz, n and n_ptr:
 * a streaming fill pass (gradient x weights into the slots the plan reserved: no barriers, no atomics) and the chunk owners of
 * arcn_hashgrid_bwd_lm / _lm_adam.  Same sums as the one-pass form (the order of a row's contributions differs, as it does between two runs
 * of the one-pass form); one plan serves one scatter.  HashGridEmbedder's backward (hashgrid_encoder.py:191-249, torch autograd's
 * index_add of the trilinear weights) is what all of them compute. */
int64_t arcn_hashgrid_plan_workspace_floats(const arcn_hashgrid_desc *desc_host, int64_t n);
int arcn_hashgrid_bwd_plan(const float *xyz, const arcn_hashgrid_desc *desc_host, float *plan_ws, int64_t plan_ws_floats, int64_t n,
                           const int32_t *n_ptr, void *stream);
int arcn_hashgrid_bwd_lm_planned(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host,
                                 float *dtable, float *plan_ws, int64_t plan_ws_floats, float *workspace, int64_t workspace_floats,
                                 int64_t n, const int32_t *n_ptr, void *stream);
int arcn_hashgrid_bwd_lm_adam_planned(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host,
                                      float *dtable, float *table, float *exp_avg, float *exp_avg_sq, float lr, float beta1, float beta2,
                                      float eps, float weight_decay, float ema_decay, float grad_scale, int step, int ema_step,
                                      float *plan_ws, int64_t plan_ws_floats, float *workspace, int64_t workspace_floats, int64_t n,
                                      const int32_t *n_ptr, uint32_t *fused_levels_host, void *stream);
/* The scatter starts from a cleared block of bin counters: the first arcn_hashgrid_bwd_counter_words(desc, n) 32-bit words of its
 * workspace.  counters_clear = 1 above says the caller has cleared them since the previous scatter on this workspace
 * (arcn_ngp_step_tail does, in the launch that ends the step), so the scatter skips its own fill launch. */
int64_t arcn_hashgrid_bwd_counter_words(const arcn_hashgrid_desc *desc_host, int64_t n);

/* FreqEmbedder.forward (encoding/freq_encoder.py:65-88): out (n, D*(include_input + 2*n_freqs)). */
int arcn_freq_fwd(const float *x, int D, int n_freqs, int include_input, float *out, int64_t n, void *stream);
/* the same encoding into n_cols >= od columns of a wider row-major buffer (row stride ld_out floats); the columns behind the od of the
 * encoding are zeroed: the positional block of GeoNet's skip concatenation (linear_network_module.py:174-197) and the view block of the
 * radiance input (encoder_mlp_network.py:62-118) are written where the next layer reads them */
int arcn_freq_fwd_cols(const float *x, int D, int n_freqs, int include_input, float *out, int64_t ld_out, int n_cols, int64_t n, void *stream);
/* out (n, n_cols at row stride ld_out) = (d enc / d x) v for v (n, D) - the adjoint of arcn_freq_bwd (which is (d enc / d x)^T dout) with
 * respect to dout: the second differentiation of a normal taken through the encoding (base_network.py:30-44, create_graph = True) */
int arcn_freq_jvp_cols(const float *x, const float *v, int D, int n_freqs, int include_input, float *out, int64_t ld_out, int n_cols, int64_t n,
                       void *stream);
int arcn_freq_bwd(const float *x, const float *dout, int D, int n_freqs, int include_input, float *dx, int64_t n,
                  void *stream);
/* SHEmbedder torch branch (encoding/sh_encoder.py:101-185): out (n, degree^2 + 3*include_input). */
int arcn_sh_fwd(const float *dirs, int degree, int include_input, float *out, int64_t n, void *stream);
/* fuse_radiance_inputs (arcnerf/models/base_modules/geo_rad_model/encoder_mlp_network.py:93-118) for an identity position block and an SH
 * view block, one pass: out (n, W) = the blocks of mode_host (HOST string over p, v, n, f) in that order - p: pts (3); v: SH of
 * normalize(dirs) = dirs / (|dirs| + 1e-8) (degree^2 columns); n: normals (3); f: n_feat features per row at stride ld_feat (a column slice of
 * the geometry net's padded output).  W = the sum of the block widths. */
int arcn_radiance_inputs(const char *mode_host, const float *pts, const float *dirs, const float *normals, const float *feat, int64_t ld_feat,
                         int n_feat, int sh_degree, float *out, int64_t n, void *stream);

/* geo -> radiance glue of Base3dModel._forward_pts_dir (arcnerf/models/base_3d_model.py:233-254):
 * sigma (n) = sigma_act(geo_out[:,0]) (EncoderMLPGeoNet.handle_output, encoder_mlp_network.py:38-50);
 * rad_in (n, Wf + deg^2) = fuse_radiance_inputs for modes 'fv' (feat_first=1) / 'vf' (encoder_mlp_network.py:93-118):
 * geo_out[:, feat_off:feat_off+Wf] and SH(normalize(dirs)).  geo_out has row stride Wg. */
int arcn_ngp_glue_fwd(const float *geo_out, const float *dirs, int Wg, int feat_off, int Wf, int sh_degree,
                      int feat_first, int sigma_act, float *rad_in, float *sigma, int64_t n, const int32_t *n_ptr,
                      void *stream);
/* the same with the view-direction part evaluated once per RAY: arcn_ngp_ray_sh fills sh_ray (n_rays, deg^2) =
 * SH(normalize(rays_d)) and the glue gathers row ray_id[s] (bit-identical: every sample of a ray carries the ray's
 * direction, fg_model.py:311-316).  Needs feat_off 0 and Wg, Wf, deg^2 multiples of 4. */
int arcn_ngp_ray_sh(const float *rays_d, int sh_degree, float *sh_ray, int64_t n_rays, void *stream);
int arcn_ngp_glue_fwd_rays(const float *geo_out, const float *sh_ray, const int32_t *ray_id, int Wg, int feat_off, int Wf,
                           int sh_degree, int feat_first, int sigma_act, float *rad_in, float *sigma, int64_t n,
                           const int32_t *n_ptr, void *stream);
/* backward of the glue: d_geo_out (n,Wg) = scatter(d_rad_in feature slice) + d_sigma * sigma_act'(geo_out[:,0]). */
int arcn_ngp_glue_bwd(const float *geo_out, const float *d_rad_in, const float *d_sigma, int Wg, int feat_off, int Wf,
                      int sh_degree, int feat_first, int sigma_act, float *d_geo_out, int64_t n, const int32_t *n_ptr,
                      void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fully fused small MLP (replaces tcnn.Network FullyFusedMLP, tcnn_fusedmlp_module.py:66-77,162-173, and
 * GeoNet/RadianceNet stacks without skips, linear_network_module.py:174-197,318-335).
 * Layers i = 0..n_layers-1: y_i = act_i(y_{i-1} W_i^T + b_i); W_i is (dims[i+1], dims[i]) row-major (torch Linear),
 * concatenated in `weights`; `biases` optional (concatenated, dims[i+1] each).  hidden activation = act_hidden for
 * i < n_layers-1, act_out for the last.  n_layers <= 8, dims[i] <= 128.
 * acts (optional): concatenated post-activation outputs of every layer, each (n, dims[i+1]) row-major, layer-major
 * (layer i starts at n_cap*sum_{j<=i}dims[j] ... see arcn_mlp_acts_offset) — saved for the backward.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_layers;
    int32_t dims[10]; /* dims[0] = input, dims[n_layers] = output */
    int32_t act_hidden;
    int32_t act_out;
    int32_t has_bias;
    float softplus_beta;
} arcn_mlp_desc;

int arcn_mlp_fwd(const float *x, const float *weights, const float *biases, const arcn_mlp_desc *desc_host, float *out,
                 float *acts, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream);
/* backward: dout (n, dims[L]); acts as saved by fwd (hidden layers only are read, out is read from `out`).
 * dx (n, dims[0]) optional; dweights/dbiases accumulated (caller zeroes). */
int arcn_mlp_bwd(const float *x, const float *weights, const float *biases, const arcn_mlp_desc *desc_host,
                 const float *out, const float *acts, const float *dout, float *dx, float *dweights, float *dbiases,
                 float *scratch, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream);
/* second half of arcn_mlp_bwd on its own (dW, db from the dpre tensors arcn_mlp_bwd(..., dweights = NULL, ...) left in
 * `scratch`), so a caller can run it on another stream while the dX chain of the next network proceeds. */
int arcn_mlp_bwd_dw(const float *x, const arcn_mlp_desc *desc_host, const float *acts, float *scratch, float *dweights,
                    float *dbiases, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream);
/* Dense layers of the WIDE nets (nn.Linear / DenseLayer stacks of GeoNet / RadianceNet, linear_network_module.py:174-197,318-335, e.g.
 * 8 x 256 with a skip concat, or 32 -> 64 -> 17 on the hash grid) as three f32-MFMA products over a huge row count; the three are
 * each other's gradients, so forward, backward and the double backward of NeuS normals (base_network.py:30-44) all run on them.
 * Row-major operands with explicit leading dimensions (ld_* in floats): a column slice of a wider buffer can be passed directly.
 *   arcn_gemm_nt: y (n_rows, N) = act(x (n_rows, K) . w (N, K)^T + bias)   act = ARCN_ACT_*, bias may be NULL
 *   arcn_gemm_nn: dx (n_rows, K) = dy' (n_rows, N) . w (N, K)
 *   arcn_gemm_tn: dw (N, K) (+)= dy' (n_rows, N)^T . x (n_rows, K), reduced over the rows in a fixed order through `scratch`
 *                 dy' = dy, or dy * (mask > 0) when `mask` (layout of dy; the layer's ReLU output) is given: the activation's
 *                 backward (torch threshold_backward after a DenseLayer, linear.py) folded into the operand load
 *                 (>= arcn_gemm_tn_scratch_floats(n_rows, N, K) floats); accumulate = 1 adds to dw, 0 overwrites */
int arcn_gemm_nt(const float *x, int64_t ld_x, const float *w, const float *bias, float *y, int64_t ld_y, int64_t n_rows,
                 const int32_t *n_ptr, int K, int N, int act, float beta, void *stream);
int arcn_gemm_nn(const float *dy, const float *mask, int64_t ld_dy, const float *w, float *dx, int64_t ld_dx, int64_t n_rows,
                 const int32_t *n_ptr, int N, int K, void *stream);
int64_t arcn_gemm_tn_scratch_floats(int64_t n_rows, int N, int K);
int arcn_gemm_tn(const float *dy, const float *mask, int64_t ld_dy, const float *x, int64_t ld_x, float *dw, float *scratch,
                 int64_t scratch_floats, int64_t n_rows, const int32_t *n_ptr, int N, int K, int accumulate, void *stream);
/* arcn_gemm_tn keeping only the first n_head of the N rows: dw (n_head, K) (+)= (dy^T x)[:n_head] - the weight gradient of a layer whose
 * output is padded to a multiple of 4 columns, straight into the layer's own (unpadded) gradient buffer. */
int arcn_gemm_tn_head(const float *dy, const float *mask, int64_t ld_dy, const float *x, int64_t ld_x, float *dw, float *scratch,
                      int64_t scratch_floats, int64_t n_rows, const int32_t *n_ptr, int N, int K, int n_head, int accumulate, void *stream);
/* The same three products on the bf16 matrix rate at f32 accuracy ("split" forms): every f32 operand is EXACTLY hi + mid + lo in three
 * bf16 numbers, and the six partial products down to 2^-16 are accumulated in f32 (what is dropped is < 2^-24 of the product, below an
 * f32 multiply's own rounding); 6 v_mfma_f32_16x16x32_bf16 replace 8 v_mfma_f32_16x16x4_f32 per 32 reduction elements (2.67 x the
 * matrix rate).  Same meaning and layouts as above.  Requirements: row operands (x, dy, mask) with 16-byte aligned rows, reduction
 * length (K for nt, N for nn) and, for tn, N and K multiples of 4; `ws` = device scratch of arcn_gemm_split_bytes(outputs, reduction
 * length) bytes that receives the split weights (nt: (N, K), nn: (K, N)).  Inf / NaN operands give NaN.
 * ws_ready = 0: the product splits w into ws itself (one small launch per call); ws_ready = 1: ws already holds
 * arcn_gemm_split_weights(w, ...) of this very weight - the reference evaluates a layer on 8 chunks of samples per training step
 * (chunk_processing around linear_network_module.py:174-197), the weights are split once for all of them.
 * arcn_gemm_split_weights: transposed = 0 for arcn_gemm_nt_split (w (n_out, k_red), row stride ld_w), 1 for arcn_gemm_nn_split (the layer's
 * (N, K) weight read as (k_red = N, n_out = K)). */
int64_t arcn_gemm_split_bytes(int n_out, int k_red);
int arcn_gemm_split_weights(const float *w, int ld_w, int transposed, int n_out, int k_red, void *ws, int64_t ws_bytes, void *stream);
int arcn_gemm_nt_split(const float *x, int64_t ld_x, const float *w, const float *bias, float *y, uint32_t *relu_bits, int64_t ld_y, int64_t n_rows,
                       const int32_t *n_ptr, int K, int N, int act, float beta, void *ws, int64_t ws_bytes, int ws_ready, void *stream);
int arcn_gemm_nn_split(const float *dy, const float *mask, const uint32_t *mask_bits, int64_t ld_dy, const float *w, float *dx, int64_t ld_dx,
                       int64_t n_rows, const int32_t *n_ptr, int N, int K, void *ws, int64_t ws_bytes, int ws_ready, void *stream);
/* tn: db (N floats, may be NULL) (+)= the column sums of dy' = the layer's bias gradient, summed from the operand as it is staged.
 * ReLU masks as BITS: arcn_gemm_nt_split(act = ReLU, N % 4 == 0) writes relu_bits (ceil(n_rows / 8) x N / 4 uint32: word [s / 8][f / 4],
 * bit 4 (s % 8) + (f % 4) = (y[s][f] > 0)); nn / tn take them as mask_bits instead of the float `mask` (which then is ignored): the
 * masked backward products read 1 bit instead of 32 per element. */
int arcn_gemm_tn_split(const float *dy, const float *mask, const uint32_t *mask_bits, int64_t ld_dy, const float *x, int64_t ld_x, float *dw, float *db,
                       float *scratch, int64_t scratch_floats, int64_t n_rows, const int32_t *n_ptr, int N, int K, int accumulate, void *stream);
/* The same network with a LEVEL-MAJOR input / input gradient: x_lm[(l * x_stride + s) * 2 + f], 2 features per level (what
 * arcn_hashgrid_fwd_xcd(level_major = 1) writes and arcn_hashgrid_bwd_lm consumes).  Wired for the bias-free 2-layer nets fed by
 * the hash grid (input 32 or 64 wide, hidden <= 64, output <= 16); -1 otherwise.  bwd: dx_lm in the layout of x_lm, dweights
 * required (fused dX + dW kernel).
 * `acts` of arcn_mlp_fwd_lm / arcn_mlp_fwd_cat is OPAQUE: the hidden activations are kept in the tile order of the kernels
 * (1 KiB contiguous per wave-wide access instead of sixteen 64-byte pieces of row-major rows), hidden widths must be multiples of
 * 16, and the buffer (arcn_mlp_acts_floats) only makes sense to the matching arcn_mlp_bwd_lm / arcn_mlp_bwd_cat call with the
 * same n_cap.  arcn_mlp_fwd / arcn_mlp_bwd keep the row-major layout documented above. */
int arcn_mlp_fwd_lm(const float *x_lm, int64_t x_stride, const float *weights, const arcn_mlp_desc *desc_host, float *out,
                    float *acts, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream);
int arcn_mlp_bwd_lm(const float *x_lm, int64_t x_stride, const float *weights, const arcn_mlp_desc *desc_host, const float *out,
                    const float *acts, const float *dout, float *dx_lm, float *dweights, float *scratch, int defer_reduce,
                    int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream);
/* arcn_mlp_fwd_lm (the geometry net) followed by arcn_mlp_fwd_cat (the radiance net on [geo_out | b_table[b_index]] / [b_table[b_index] |
 * geo_out]) in ONE kernel for the NGP nets of nerf_ngp.yaml - bias-free 32 -> 64 (ReLU) -> 16 (linear) and 32 -> 64 -> 64 (ReLU) -> <= 16
 * (sigmoid): the geometry net's output tile is the radiance net's first operand without leaving the registers.  The same arguments, the same
 * outputs BIT FOR BIT as the two calls (geo_out (n, 16), head_out = act(geo_out[:, 0]), rgb, rad_acts in the opaque tile order of
 * arcn_mlp_fwd_cat; the geometry net saves nothing - arcn_mlp_bwd_lm recomputes its hidden layer).  -1 for any other pair of nets.
 * Reference: Base3dModel._forward_pts_dir (base_3d_model.py:233-254) on tcnn_fusedmlp_module.py:66-77,162-192. */
int arcn_ngp_nets_fwd(const float *x_lm, int64_t x_stride, const float *geo_w, const arcn_mlp_desc *geo_desc, float *geo_out,
                      const float *b_table, const int32_t *b_index, int a_first, const float *rad_w, const arcn_mlp_desc *rad_desc,
                      float *rgb, float *rad_acts, float *head_out, int head_act, int64_t n_cap, int64_t n, const int32_t *n_ptr,
                      void *stream);
/* The radiance net of Base3dModel._forward_pts_dir (base_3d_model.py:233-254) with fuse_radiance_inputs
 * (encoder_mlp_network.py:93-118) folded into the first layer's operand load: x = [a | b] (a_first) or [b | a], 16 columns each,
 * a (n,16) row-major per sample (the geometry net's output), b = b_table[b_index[s]] (arcn_ngp_ray_sh rows by ray id).
 * head_out (optional) = head_act(a[:,0]) (sigma, EncoderMLPGeoNet.handle_output).  bwd: da (n,16) = gradient of a incl.
 * d_head * head_act'(a[:,0]) in column 0 (b has no gradient); dweights required.  Bias-free nets 32 -> 64 [-> 64] -> <=16. */
int arcn_mlp_fwd_cat(const float *a, const float *b_table, const int32_t *b_index, int a_first, const float *weights,
                     const arcn_mlp_desc *desc_host, float *out, float *acts, float *head_out, int head_act, int64_t n_cap,
                     int64_t n, const int32_t *n_ptr, void *stream);
int arcn_mlp_bwd_cat(const float *a, const float *b_table, const int32_t *b_index, int a_first, const float *weights,
                     const arcn_mlp_desc *desc_host, const float *out, const float *acts, const float *dout, float *da,
                     const float *d_head, int head_act, float *dweights, float *scratch, int defer_reduce, int64_t n_cap,
                     int64_t n, const int32_t *n_ptr, void *stream);
/* defer_reduce = 1 in the two calls above leaves the per-workgroup dW partials in `scratch`; this adds them into dweights
 * (same n_cap and n).  Lets a caller take the two tiny reductions off the backward's critical path. */
int arcn_mlp_bwd_reduce(const arcn_mlp_desc *desc_host, float *scratch, float *dweights, int64_t n_cap, int64_t n,
                        void *stream);
/* float count the caller must provide in `acts` (hidden layers, rows padded to a multiple of 16) and `scratch` (bwd) for capacity n_cap */
int64_t arcn_mlp_acts_floats(const arcn_mlp_desc *desc_host, int64_t n_cap);
int64_t arcn_mlp_scratch_floats(const arcn_mlp_desc *desc_host, int64_t n_cap);

/* elementwise activation (TruncExp F1 etc.) */
int arcn_act_fwd(const float *x, float *y, int64_t n, int act, float beta, void *stream);
/* y (n) = act(x[i * ld]) * scale, rows behind *n_ptr (may be NULL) left alone: the estimated opacity sigma * dt of an occupancy refresh
 * (base_3d_model.py:368-389 get_est_opacity) from the density column of the geometry net's output, one pass */
int arcn_act_col_scale(const float *x, int64_t ld, float *y, int64_t n, const int32_t *n_ptr, int act, float beta, float scale, void *stream);
/* One elementwise pass of ops.autograd.SdfMlpJacFn's backward (the NeuS-on-hash-grid sdf net with its Jacobian as an explicit output):
 * dz = dh s + c_j u s (1 - s), su = s u over an (n, H) hidden layer, c (H) = beta W2[0]; dz / su may alias dh / u. */
int arcn_sdf_jac_dz(const float *dh, const float *u, const float *s, const float *c, float *dz, float *su, int64_t n, int H, void *stream);
/* The two-layer geometry nets of the NeuS-on-hash-grid + MultiVol step, each as ONE forward and ONE backward kernel (csrc/mlp.hip, the
 * transposed-MFMA form of arcn_mlp_fwd_lm): hash features x (32 columns, LEVEL-major as arcn_hashgrid_fwd_xcd / _fwd_corners write them with
 * level_major = 1: x_lm[(l * x_stride + s) * 2 + f]) -> 64 hidden -> n_out <= 32 outputs, bias-free, w1 (64, 32), w2 (n_out, 64) row-major like
 * torch.nn.Linear.weight.  Replaces, per net, the chain lm_to_rows + arcn_gemm_* + arcn_softplus_grad + arcn_sdf_jac_dz2 + arcn_geo_out_grad +
 * arcn_act_col_scale of trainer.FusedNeusNgpStep (16 launches, every (n, 64) intermediate through HBM); nothing is saved between the two calls
 * - the backward recomputes the hidden layer.
 *   jac_mode 1 - the sdf net (GeoNet with a softplus-beta DenseLayer evaluated through forward_with_grad: sdf_model.py:42-101,
 *     base_network.py:30-44): out (n, n_pad) = W2 softplus(W1 x) (columns n_out .. n_pad - 1 zero), head (n) = out[:, 0] (optional),
 *     jac (n, 32) row-major = d out[:, 0] / d x = W1^T (s W2[0]), s = 1 - exp(-beta h).  Backward inputs: d_col0 (n) = d out[:, 0],
 *     d_feat (row stride ld_feat) = d out[:, 1 .. n_out), d_jac (n, 32) = the gradient of jac; outputs dx (n, 32) row-major (dx_stride 0) or
 *     level-major (dx_stride >= n), dw1 += and dw2 += the weight gradients of BOTH paths (ops.autograd.SdfMlpJacFn.backward's arithmetic).
 *     d_col0 is read at row stride ld_col0 (1: a contiguous vector; the width of a (n, n_out) gradient whose column 0 it is).
 *   x_stride 0: the features are (n, 32) ROWS (HashGridEmbedder.forward's output, hashgrid_encoder.py:160-189) instead of level-major.
 *   jac_mode 0 - a density net (ReLU hidden layer, TruncExp on column 0: linear_network_module.py:174-197 with out_act_cfg TruncExp):
 *     head (n) = exp(out[:, 0]); the backward takes d_col0 = d head and multiplies it by exp(clamp(out_col0[s * ld_out], -15, 15)) (the
 *     reference's TruncExp backward, arcnerf/ops/trunc_exp.py), out_col0 = the forward's out.
 * scratch: arcn_geo2_bwd_scratch_floats(n) floats (per-workgroup weight-gradient tiles, summed by the reduction the call ends with in a fixed
 * order).  dw1 and dw2 must be views of one buffer (|dw2 - dw1| < 2^31 floats: a flattened optimiser's gradient buffer). */
int arcn_geo2_fwd(const float *x_lm, int64_t x_stride, const float *w1, const float *w2, int n_out, int n_pad, int jac_mode, float beta,
                  float *out, float *head, float *jac, int64_t n, const int32_t *n_ptr, void *stream);
int64_t arcn_geo2_bwd_scratch_floats(int64_t n);
int arcn_geo2_bwd(const float *x_lm, int64_t x_stride, const float *w1, const float *w2, int n_out, int jac_mode, float beta,
                  const float *d_col0, int64_t ld_col0, const float *out_col0, int64_t ld_out, const float *d_feat, int64_t ld_feat,
                  const float *d_jac, float *dx, int64_t dx_stride, float *dw1, float *dw2, float *scratch, int64_t n, const int32_t *n_ptr,
                  void *stream);
/* The passes BETWEEN the kernels of the NeuS-on-hash-grid + MultiVol training step (trainer.FusedNeusNgpStep; csrc/step_glue.hip), each
 * one group of the reference's elementwise torch expressions in one launch:
 *   arcn_neus_step_prep  the per-step derived weights of the two geometry nets (sdf_model.py:42-101, base_network.py:30-44, neus_model.py:221-228):
 *     w2p (n_pad, H) = the sdf net's last layer l1w (n_out, H) padded with zero rows; w1j (H, E) = w1 * l1w[0][:, None] (the weight the Jacobian
 *     row of the first output is a product with); bw20 (H) = beta * l1w[0]; scale_out[0] = exp(inv_s[0] * speed) (optional);
 *     wb1p (nb_pad, Hb) = the background density net's last layer padded (bkg_l1w NULL: none).
 *   arcn_geo_out_grad    g_out (n, n_pad) = [d_col0 * act'(out[:, 0]) | d_feat (n_feat columns at row stride ld_feat) | 0]: the gradient of a
 *     geometry net's padded output row from the gradients of its first column THROUGH its activation (act 0: as it is; y_col0 optional =
 *     act(out[:, 0])) and of its feature columns (linear_network_module.py GeoNet.forward).
 *   arcn_neus_blend_loss rgb = rgb_f + T rgb_b, depth = depth_f + T depth_b (full_model.py:278-330, `rgb` blending), loss[0] = weight * mean
 *     Huber_delta(rgb - target) (huber_delta <= 0: weight * mean squared error; img_loss.py:60-100), loss[1] = 0 (the accumulator of the pass
 *     that follows, arcn_eikonal_packed with accumulate bit 1), d_rgb, d_tlast = sum_c d_rgb rgb_b, d_rgb_b = T d_rgb.  workspace:
 *     arcn_neus_blend_loss_workspace_words() 32-bit words of device memory, word 0 ZERO before the first call (the call leaves it zero): the
 *     workgroups' partial losses are added by the last of them in index order - no float atomics, the same bits every run.
 *   arcn_sdf_jac_dz2     arcn_sdf_jac_dz with the by-products of the weight gradients: dz as there, sw = s * w (w (H) = l1w[0]; sw^T d_jac is the
 *     Jacobian path's gradient of the first layer), colsum[j] += sum_i s u (the Jacobian path's gradient of l1w[0]); dz / sw may alias dh / u.
 *   arcn_sum_scale_add   dst[0] += factor * scale_dev[0] * sum(src[0..n)) (scale_dev optional): the gradient of inv_s from the per-ray d scale. */
int arcn_neus_step_prep(const float *w1, const float *l1w, int H, int E, int n_out, int n_pad, float beta, const float *inv_s, float speed,
                        float *w2p, float *w1j, float *bw20, float *scale_out, const float *bkg_l1w, int Hb, int nb_out, int nb_pad, float *wb1p,
                        void *stream);
int arcn_geo_out_grad(const float *d_col0, const float *out, int64_t ld_out, const float *y_col0, int act, float beta, const float *d_feat,
                      int64_t ld_feat, int n_feat, int n_pad, float *g_out, int64_t n, void *stream);
int64_t arcn_neus_blend_loss_workspace_words(void);
int arcn_neus_blend_loss(const float *rgb_f, const float *depth_f, const float *t_last, const float *rgb_b, const float *depth_b,
                         const float *target, int64_t n_rays, float huber_delta, float weight, float *rgb, float *depth, float *d_rgb,
                         float *d_tlast, float *d_rgb_b, float *loss, uint32_t *workspace, void *stream);
int arcn_sdf_jac_dz2(const float *dh, const float *u, const float *s, const float *c, const float *w, float *dz, float *sw, float *colsum,
                     int64_t n, int H, void *stream);
int arcn_sum_scale_add(const float *src, int64_t n, const float *scale_dev, float factor, float *dst, void *stream);
/* The tone mappers of HDR-NeRF (arcnerf/models/hdrnerf_model.py:44-75: per colour channel DenseLayer(1, W) + ReLU, DenseLayer(W, 1) +
 * sigmoid on ln(exposure) + log radiance) with the hidden layer in registers.  x / y / dy / dx (n, C) row-major, params / dparams
 * (C, 3 W + 1) = per channel [w1 (W) | b1 (W) | w2 (W) | b2], W <= 128.  bwd: dx may be NULL; dparams is overwritten (workgroup partials
 * in `scratch`, >= arcn_tonemap_scratch_floats floats, added in a fixed order). */
int arcn_tonemap_fwd(const float *x, const float *params, float *y, int64_t n, int C, int W, void *stream);
int64_t arcn_tonemap_scratch_floats(int64_t n, int C, int W);
int arcn_tonemap_bwd(const float *x, const float *y, const float *dy, const float *params, float *dx, float *dparams, float *scratch,
                     int64_t scratch_floats, int64_t n, int C, int W, void *stream);
/* softplus closed under differentiation (nn.Softplus(beta = 100) of the NeuS sdf net, base_modules/activation.py; the normals are
 * d sdf / d x with create_graph = True and the Eikonal loss differentiates them again, base_network.py:30-44), s = sigmoid(beta z):
 *   arcn_softplus_grad : out = g * s                                  first backward (and d out / d g applied to g); g may be NULL: out = s
 *   arcn_softplus_grad2: dg = h * s, dz = h * g * beta s (1 - s)       backward of arcn_softplus_grad for an incoming h (dg / dz may be NULL)
 * from_y = 1: `z` holds y = softplus(z) instead (a layer with the activation in its product's epilogue keeps no z): s = 1 - e^(-beta y), and
 * dz becomes the gradient with respect to y, h * g * beta (1 - s) (the chain through y multiplies by s again). */
int arcn_softplus_grad(const float *z, const float *g, float *out, int64_t n, float beta, int from_y, void *stream);
/* out = (g + g2) * sigmoid(beta z): arcn_softplus_grad for an activation two gradients arrive at - the ordinary chain and the curvature
 * term of a normal's second differentiation (base_network.py:30-44) - without a pass that adds them first */
int arcn_softplus_grad_sum(const float *z, const float *g, const float *g2, float *out, int64_t n, float beta, int from_y, void *stream);
int arcn_softplus_grad2(const float *z, const float *g, const float *h, float *dg, float *dz, int64_t n, float beta, int from_y, void *stream);
/* The two softplus passes for a gradient that is ONE ROW g_row (H) broadcast over the n samples (the last hidden layer of the wide sdf net's
 * normal chain, g = W_D[0]; base_network.py:30-44 through linear_network_module.py:174-197), z (n, H):
 *   arcn_softplus_grad_row : out = g_row[col] * s
 *   arcn_softplus_grad2_row: dz = h * g_row[col] * ds, colsum[col] += sum over the samples of h * s (the adjoint of the row: all a caller
 *                            needs of dg); H = 4 * (a divisor of 256).
 * arcn_concat2_div: out (n, n_cols) = [a[:, :na] / div | b[:, :nb] / div | 0] from two row-major sources at row strides ld_a / ld_b (b may be
 * NULL with nb = 0): the skip concatenation [h | e] / sqrt2 of linear_network_module.py:174-197 and its adjoints; `/ div` = a product with the
 * float reciprocal, the arithmetic of torch's CUDA division by a python scalar (bit-identical to the expression it replaces). */
int arcn_softplus_grad_row(const float *z, const float *g_row, float *out, int64_t n, int H, float beta, int from_y, void *stream);
int arcn_softplus_grad2_row(const float *z, const float *g_row, const float *h, float *dz, float *colsum, int64_t n, int H, float beta, int from_y,
                            void *stream);
int arcn_concat2_div(const float *a, int64_t ld_a, int na, const float *b, int64_t ld_b, int nb, float div, float *out, int n_cols, int64_t n,
                     void *stream);
int arcn_act_bwd(const float *x, const float *y, const float *dy, float *dx, int64_t n, int act, float beta,
                 void *stream);
/* second backward of an elementwise activation, given the pre-activation x, the incoming dy of the first backward (dx = dy f'(x)) and
 * g = d loss / d dx:  ddy = g f'(x),  d2x = g dy f''(x)  (either output may be NULL).  Makes ops.autograd.ActFn twice differentiable. */
int arcn_act_bwd_bwd(const float *x, const float *dy, const float *g, float *ddy, float *d2x, int64_t n, int act, float beta, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Compositing (arcnerf/render/ray_helper.py:476-620)
 * dense form: sigma/alpha_in (R,P), radiance (R,P,3), zvals (R,P); Pe = P if (add_inf_z || alpha_in) else P-1.
 * per-sample outputs alpha/trans/weights (R,Pe) optional.  bkg (bkg_rows,3), bkg_rows in {0,1,R}.
 * status (device int32, optional) is set to 1 if some delta < 0 (the reference asserts, :534).
 * bwd: d_tlast (R, optional) = upstream gradient of trans_shift[:, -1], the transmittance FullModel.blend_bkg_rgb scales a
 * background model with (full_model.py:278-330); the bkg_color term of rgb uses the same quantity (:569-571).
 * ---------------------------------------------------------------------------------------------- */
int arcn_ray_marching_fwd(const float *sigma, const float *alpha_in, const float *radiance, const float *zvals,
                          const float *noise, const float *bkg, int64_t bkg_rows, int64_t R, int P, int add_inf_z,
                          int white_bkg, float *rgb, float *depth, float *mask, float *alpha_out, float *trans_out,
                          float *weights_out, int32_t *status, void *stream);
int arcn_ray_marching_bwd(const float *sigma, const float *alpha_in, const float *radiance, const float *zvals,
                          const float *noise, const float *bkg, int64_t bkg_rows, int64_t R, int P, int add_inf_z,
                          int white_bkg, const float *d_rgb, const float *d_depth, const float *d_mask, const float *d_tlast,
                          float *d_geo, float *d_radiance, void *stream);

/* Fused training step of the packed compositor: arcn_composite_packed_fwd + arcn_huber_loss_grad (ImgLoss Huber,
 * arcnerf/loss/img_loss.py:60-100, mean over R*3, times loss_weight) + arcn_composite_packed_bwd in one pass per ray - the Huber
 * gradient of a ray depends on that ray's colour only.  target (R,3); rgb/depth/mask/d_rgb (R,..) optional outputs; d_sigma (S),
 * d_radiance (S,3) as the backward; loss_partials (optional, ceil(R / 4) floats): entry w receives the summed loss terms of rays
 * 4w .. 4w+3, the loss of the launch is the sum of the entries (a single scalar accumulator would cost 2080 serialised atomics). */
int arcn_composite_packed_train(const float *sigma, const float *radiance, const float *t_packed, const int32_t *offsets,
                                const float *noise, const float *bkg, int64_t bkg_rows, int64_t R, int p_dense,
                                const int32_t *p_dense_ptr, int add_inf_z, int white_bkg, const float *target, float huber_delta,
                                float loss_weight, float *rgb, float *depth, float *mask, float *d_rgb, float *loss_partials,
                                float *d_sigma, float *d_radiance, const int32_t *counts, void *stream);
/* packed form over (offsets, t): identical numbers to the dense form applied to the reference's padded (R,P') view
 * (mask rows [T..T F..F], padded z = last z): P_dense = the dense column count the reference would have used
 * (max(2, max count), fg_model.py:251-262) read from *p_dense_ptr (device) or p_dense if the pointer is NULL.
 * counts (R, optional; all three packed entry points): the samples the marcher emitted per ray.  The reference never drops samples
 * (fg_model.py:264-318 sizes its tensors from the mask); packed buffers have a capacity, and offsets are clamped to it.  A ray whose
 * segment is shorter than its count (it lies behind the point where the buffers filled up) is NOT rendered from the partial set: it is
 * a ray without samples (background colour, zero gradient to its leftover samples) - an overflowed training step is the exact step of
 * the rays that fit, never a step on truncated rays. */
int arcn_composite_packed_fwd(const float *sigma, const float *radiance, const float *t_packed, const int32_t *offsets,
                              const float *noise, const float *bkg, int64_t bkg_rows, int64_t R, int p_dense,
                              const int32_t *p_dense_ptr, int add_inf_z, int white_bkg, float *rgb, float *depth,
                              float *mask, float *weights_out, const int32_t *counts, void *stream);
int arcn_composite_packed_bwd(const float *sigma, const float *radiance, const float *t_packed, const int32_t *offsets,
                              const float *noise, const float *bkg, int64_t bkg_rows, int64_t R, int p_dense,
                              const int32_t *p_dense_ptr, int add_inf_z, int white_bkg, const float *d_rgb,
                              const float *d_depth, const float *d_mask, float *d_sigma, float *d_radiance, const int32_t *counts,
                              void *stream);

/* ImgLoss with loss_type Huber (arcnerf/loss/img_loss.py:60-100): loss[0] = weight * mean(huber_delta(x - y)) over n
 * elements, dx = d loss / d x.  dx and loss are optional. */
int arcn_huber_loss_grad(const float *x, const float *y, int64_t n, float delta, float weight, float *dx, float *loss,
                         void *stream);

/* NeuS sdf_to_alpha (arcnerf/models/neus_model.py:242-265, cdf = sigmoid(sdf * s), :221-228): mid_sdf, mid_slope, alpha
 * (R, P-1); zvals (R, P); s_dev = the scale as a DEVICE scalar (exp(10 * inv_s) of a learnable parameter).  bwd writes
 * d_sdf, d_slope and ADDS the scalar d s into d_s (optional; caller zeroes). */
int arcn_sdf_to_alpha_fwd(const float *mid_sdf, const float *zvals, const float *mid_slope, const float *s_dev, int clip,
                          float *alpha, int64_t R, int P, void *stream);
int arcn_sdf_to_alpha_bwd(const float *mid_sdf, const float *zvals, const float *mid_slope, const float *s_dev, int clip,
                          const float *d_alpha, float *d_sdf, float *d_slope, float *d_s, int64_t R, int P, void *stream);

/* ------------------------------------------------------------------------------------------------
 * NeuS on packed samples (arcnerf/models/neus_model.py:63-104 `_forward`, :174-202 `handle_mid_pts` in its masked layout, :242-265
 * `sdf_to_alpha`; models/sdf_model.py:42-101 valid-point gather + padded fill; render/ray_helper.py:476-620 alpha= branch): the numbers of
 * the reference's padded (rays, P) chain, P = max(2, longest ray), without any (rays, P) tensor.
 *   arcn_neus_count     n_eval[r] = counts[r] > 0 ? min(counts[r] + 1, P) : 0  (points the nets see per ray: the section mid points plus
 *                       the point behind the last sample; *kmax_dev = the largest count, device int32)
 *   arcn_neus_sections  zvals_dense (n_rays, n_pts) valid-first depths of the marcher, offsets = exclusive scan of n_eval (offsets[n_rays]
 *                       = total) -> t_mid / sec_lo / sec_hi (total) mid point and ends of every evaluated section, ray_id (total),
 *                       slot_map (n_rays, p_dense) int64 optional: padded slot -> packed row (tails: the ray's last row; rays without
 *                       samples: row `total`) for gathering the dense `normal_pts` output.  n_sample_cfg = rays.n_sample of the config.
 *   arcn_neus_render_*  per ray: slope = d . normal, cos-annealed, alpha = sdf_to_alpha(sdf, section, slope, *s_dev), weights, rgb
 *                       (+ T_last * bkg), depth, mask, normal = sum w normalize(normal), t_last = trans_shift[:, -1]; the padded slots
 *                       (zero-length sections repeating the last point: alpha = 1e-5 / (sigmoid(s sdf) + 1e-5), not 0) are walked too.
 *                       Rays without samples take FgModel's defaults (bkg colour | dflt_rgb, depth_far, 0, dflt_nrm, 1); dflt_* are HOST
 *                       pointers to 3 floats.  bwd: d_sdf (total), d_radiance / d_normal (total, 3), d_s_ray (n_rays, optional: per-ray
 *                       partials of d s).
 * ---------------------------------------------------------------------------------------------- */
int arcn_neus_count(const int32_t *counts, const int32_t *kmax_dev, int64_t n_rays, int32_t *n_eval, void *stream);
int arcn_neus_sections(const float *zvals_dense, const int32_t *counts, const int32_t *offsets, int n_pts, float n_sample_cfg,
                       int64_t n_rays, int p_dense, float *t_mid, float *sec_lo, float *sec_hi, int32_t *ray_id, int64_t *slot_map,
                       void *stream);
/* the dense (n_rays, p_dense, 3) view of a packed (total, 3) per-point quantity the reference returns (`normal_pts`): slot j of a ray =
 * its point min(j, n - 1), rays without points = dflt_host[0..2]; bwd = its transpose (the aliased tail summed by the ray's wave). */
/* EikonalLoss (arcnerf/loss/geo_loss.py:12-70, MSE on |n| over `normal_pts`, plain mean) evaluated on the PACKED normals with the dense
 * layout's weights: loss[0] = weight * mean over the (n_rays, p_dense) slots of (|n| - 1)^2 where slot j of a ray = its point
 * min(j, n - 1) and rays without points hold a unit default normal; d_normal (n_pts,3) = its gradient (accumulate: added).  The dense
 * (n_rays, p_dense, 3) tensor the reference's loss reads is never built.  accumulate bit 0: the gradient is ADDED to d_normal; bit 1: the
 * loss is added to loss[0] as it stands (else loss - DEVICE, optional - is cleared by the call).  add_src (optional, rows of 3 floats at
 * row stride ld_add): a second incoming gradient of the normals joined in the same pass (the radiance net's input gradient). */
int arcn_eikonal_packed(const float *normal, const int32_t *ray_id, const int32_t *offsets, int64_t n_pts, int64_t n_rays, int p_dense,
                        float weight, int accumulate, const float *add_src, int64_t ld_add, float *d_normal, float *loss, void *stream);
int arcn_neus_slots_fwd(const float *packed, const int32_t *offsets, int64_t n_rays, int p_dense, const float *dflt_host, float *dense,
                        void *stream);
int arcn_neus_slots_bwd(const float *d_dense, const int32_t *offsets, int64_t n_rays, int p_dense, float *d_packed, void *stream);
int arcn_neus_render_fwd(const float *sdf, const float *radiance, const float *normal, const float *t_mid, const float *sec_lo,
                         const float *sec_hi, const int32_t *offsets, const float *rays_d, const float *s_dev, float cos_anneal,
                         const float *bkg, int64_t bkg_rows, const int32_t *kmax_dev, float depth_far, const float *dflt_rgb_host,
                         const float *dflt_nrm_host, int64_t n_rays, float *rgb, float *depth, float *mask, float *nrm, float *t_last,
                         void *stream);
int arcn_neus_render_bwd(const float *sdf, const float *radiance, const float *normal, const float *t_mid, const float *sec_lo,
                         const float *sec_hi, const int32_t *offsets, const float *rays_d, const float *s_dev, float cos_anneal,
                         const float *bkg, int64_t bkg_rows, const int32_t *kmax_dev, int64_t n_rays, const float *d_rgb,
                         const float *d_depth, const float *d_mask, const float *d_nrm, const float *d_tlast, float *d_sdf,
                         float *d_radiance, float *d_normal, float *d_s_ray, void *stream);

/* sample_cdf (ray_helper.py:432-473): bins/cdf (R,n_pts), u (R,n_sample) -> samples (R,n_sample) sorted,
 * inds (R,n_sample) int32 optional (searchsorted right=True). */
int arcn_sample_cdf(const float *bins, const float *cdf, const float *u, int64_t R, int n_pts, int n_sample, float eps,
                    int do_sort, float *samples, int32_t *inds, void *stream);

/* sample_pdf (ray_helper.py:410-429) in one launch: weights (R, n_pts-1) -> (+eps) / sum -> cdf (R, n_pts) with a leading 0 -> inverse
 * CDF at u -> sorted samples (R, n_sample).  u has u_rows = 1 (one lattice for every ray: det) or R rows.  The cdf is accumulated in
 * double and rounded to float prefix by prefix, which is what torch.cumsum does on the reference's CPU path; cdf_out (R, n_pts)
 * optional. */
int arcn_sample_pdf(const float *bins, const float *weights, const float *u, int64_t R, int n_pts, int n_sample, int u_rows, float eps,
                    int do_sort, float *samples, float *cdf_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Occupancy update (volume_bound.py:160-212, geometry/volume.py:983-1017)
 * ---------------------------------------------------------------------------------------------- */
/* opafield[idx] = old>=0 ? max(old*ema, opacity) : old ; ema < 0 = no ema */
int arcn_update_opafield(float *opafield, const int64_t *flat_idx, const float *opacity, int64_t n, float ema,
                         void *stream);
/* Sort-free equivalent of  unique(voxel_idx) -> segmented max (K4) -> update_opafield_by_voxel_idx  for possibly
 * repeated flat cell indices (volume_bound.py:199-211).  cell_max (n_cells floats) and touched (n_cells bytes) are
 * scratch buffers (cleared inside).  opacity must be non-negative.  n_ptr (device, optional): number of valid entries. */
int arcn_opafield_scatter_update(float *opafield, const int64_t *cell_idx, const float *opacity, int64_t n,
                                 const int32_t *n_ptr, int64_t n_cells, float ema, float *cell_max, uint8_t *touched,
                                 void *stream);
/* thres = min(mean(clamp(opa,0)), threshold) computed on device; bitfield (bool bytes) = opa >= thres.
 * workspace: 2 floats (device). */
int arcn_update_bitfield_by_opafield(const float *opafield, uint8_t *bitfield, int64_t n, float threshold,
                                     float *workspace, void *stream);
/* Cells and sample points of one occupancy refresh after the warm-up (VolumeBound.optimize, volume_bound.py:178-190, which draws
 * torch.randperm(n)[:n / 4] and takes get_occupied_voxel_idx()[:n / 4]): n_s = n_grid^3 / 4 cells drawn uniformly without repetition - the
 * image of [0, n_s) under the seeded bijection (perm_a[2] odd, perm_c[2]; two rounds of x -> a x + c, xor-shift, odd multiply, xor-shift
 * modulo n_grid^3, a power of two) - followed by the first n_s occupied cells of the boolean bitfield in flat-index order; both halves
 * leave ORDERED - the uniform half along the Z-curve of the grid, the occupied half by flat index - so that spatial neighbours stay
 * neighbours in the hash gather that follows.  cells_out (2 n_s int64), n_valid
 * (device int32) = n_s + min(occupied, n_s), entries behind it are not written; pts_out (2 n_s, 3) = voxel centre + a uniform jitter of
 * one voxel (voxel_size_host[3]: the volume's xyz_len / n_grid) from the pcg32 stream (rng_state, rng_inc).  workspace: >= n_grid^3 + 8 (n_grid^3 / 4096 + 1) bytes, 8-byte aligned like the
 * bitfield.  Four small launches, no host synchronisation (the torch formulation was ~45). */
int arcn_refresh_cells_points(const uint8_t *bitfield_bool, int n_grid, const uint64_t *perm_a, const uint64_t *perm_c,
                              const float *voxel_size_host, const float *min_xyz_host, uint64_t rng_state, uint64_t rng_inc, int64_t *cells_out, float *pts_out,
                              int32_t *n_valid, uint8_t *workspace, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser over one flat fp32 buffer: torch.optim.Adam(lr, betas, eps, weight_decay) semantics
 * (common/trainer/optimizer.py:6-54) + EMA.ema_step (arcnerf/trainer/ema.py:29-43), fused.
 * step / ema_step are 1-based. grad_scale multiplies the gradient first (1/world_size after all-reduce).
 * ema (the reference's `old_avg`) may be NULL; when given the debiased average
 *   new = ((1-d)*p + d*old*(1-d^(ema_step-1))) / (1-d^ema_step)
 * is stored to BOTH ema and param, as the reference does.  zero_grad != 0 clears grad in the same pass.
 * ema == param: the two copies are equal after every step, so a caller that is the only writer of `param` between steps
 *   may pass the parameter buffer itself; the value read from it is the old average and no second copy is written
 *   (same bits, 8 B/param less traffic).
 * ---------------------------------------------------------------------------------------------- */
int arcn_adam_ema_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, float *ema, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float ema_decay, float grad_scale,
                       int step, int ema_step, int zero_grad, void *stream);
/* dst = src (n_words 32-bit words) as a kernel; src / dst: device memory or pinned host memory.  The small per-step copies of an
 * asynchronous training step (the sample total -> pinned host, trainer.FusedNgpStep) - see optim.hip. */
int arcn_copy_words(const void *src, void *dst, int64_t n_words, void *stream);
/* arcn_adam_ema_step on up to four runs [lo, lo + n) of the SAME flat buffers in one launch (runs_host: lo0, n0, lo1, n1, ...; every lo a
 * multiple of 4 floats): what is left of the flat parameter buffer of a single-GPU step once the scatter's chunk owners have applied
 * the optimiser to their table levels (arcn_hashgrid_bwd_lm_adam) - the small levels in front of them and the MLP weights behind. */
int arcn_adam_ema_step_runs(float *param, float *grad, float *exp_avg, float *exp_avg_sq, float *ema, const int64_t *runs_host, int n_runs,
                            float lr, float beta1, float beta2, float eps, float weight_decay, float ema_decay, float grad_scale, int step,
                            int ema_step, int zero_grad, void *stream);

/* The end of a single-GPU training step of the packed NGP pipeline in ONE launch (the four launches it replaces - two dW reductions, the
 * optimiser on the rest of the flat buffer, the scatter's counter fill - were 5 - 7 us each, 4 % of the step), same arithmetic as
 *   arcn_mlp_bwd_reduce(desc_a, scratch_a, grad + w_seg_a, n_cap, n), arcn_mlp_bwd_reduce(desc_b, scratch_b, grad + w_seg_b, n_cap, n)
 *     (the deferred dW sums of arcn_mlp_bwd_cat / arcn_mlp_bwd_lm called with defer_reduce = 1), then
 *   arcn_adam_ema_step_runs(param, grad, ..., [runs_host] + the two nets' weight segments, zero_grad = 1)
 *     (torch.optim.Adam as configured by common/trainer/optimizer.py:6-54, EMA.ema_step written back, arcnerf/trainer/ema.py:29-43;
 *      the step of common/trainer/basic_trainer.py:560-577), the owner of a dW element applying the update to it,
 *   and clear_words 32-bit words cleared at `clear` (the scatter's bin counters: arcn_hashgrid_bwd_counter_words; NULL / 0: none).
 * w_seg_a / w_seg_b: first weight of each net in the flat buffers (the nets' weights must not overlap the runs).  Bias-free nets of
 * 2 - 3 layers up to 64 wide (the fused-backward shapes). */
int arcn_ngp_step_tail(const arcn_mlp_desc *desc_a, float *scratch_a, int64_t w_seg_a, const arcn_mlp_desc *desc_b, float *scratch_b,
                       int64_t w_seg_b, int64_t n_cap, int64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, float *ema,
                       const int64_t *runs_host, int n_runs, float lr, float beta1, float beta2, float eps, float weight_decay,
                       float ema_decay, float grad_scale, int step, int ema_step, uint32_t *clear, int64_t clear_words, void *stream);

#ifdef __cplusplus
}
#endif
#endif
