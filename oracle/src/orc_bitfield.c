/* ORACLE (test infrastructure only, never shipped or measured as the product).
 *
 * CPU restatement of the reference's `_bitfield_func` family (K5-K10):
 *   arcnerf/ops/src/bitfield_func/bitfield_func_kernel.cu   (kernels)
 *   arcnerf/ops/include/volume_func.h:136-194               (Morton index + bit test)
 * These kernels exist only as CUDA in the reference (no torch twin), so there is nothing runnable here to pin them
 * against: PARITY UNPINNED.  The restatement follows the .cu text statement by statement; every function cites it.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "orc_common.h"

/* volume_func.h:141-158 */
static inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static inline uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

/* volume_func.h:160-168 */
static inline uint32_t morton3d_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* volume_func.h:170-182: truncating float->int cast, clamped into the grid, Morton order */
static inline uint32_t morton_idx_at(const float pos[3], const float mn[3], const float mx[3], uint32_t n) {
    int c[3];
    for (int k = 0; k < 3; ++k) {
        float vs = (mx[k] - mn[k]) / (float)n;
        float vi = (pos[k] - mn[k]) / vs;
        c[k] = clampi((int)vi, 0, (int)n - 1);
    }
    return morton3d((uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2]);
}

ORC_API void orc_morton3d(const uint32_t *xyz, uint32_t *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = morton3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}

ORC_API void orc_morton3d_invert(const uint32_t *idx, uint32_t *xyz, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        xyz[3 * i] = morton3d_invert(idx[i] >> 0);
        xyz[3 * i + 1] = morton3d_invert(idx[i] >> 1);
        xyz[3 * i + 2] = morton3d_invert(idx[i] >> 2);
    }
}

static inline int in_aabb(const float p[3], const float mn[3], const float mx[3]) {
    return p[0] >= mn[0] && p[1] >= mn[1] && p[2] >= mn[2] && p[0] <= mx[0] && p[1] <= mx[1] && p[2] <= mx[2];
}

/* volume_func.h:99-134 (shared with K3) */
static inline float dist_to_next_voxel(const float pos[3], const float d[3], const float mn[3], const float mx[3],
                                       uint32_t n) {
    float t_min = 0.f;
    for (int k = 0; k < 3; ++k) {
        float center = (mn[k] + mx[k]) / 2.0f;
        float half = (mx[k] - mn[k]) / 2.0f;
        float inv_d = 1.0f / d[k];
        float p = (float)n * pos[k];
        float hs = half * copysignf(1.0f, d[k]);
        float a = p + center;
        a = a + hs;
        float t = (floorf(a) - p) * inv_d;
        if (k == 0 || t < t_min) t_min = t;
    }
    return fmaxf(t_min / (float)n, 0.0f);
}

/* K5 sparse_volume_sampling_bit (bitfield_func_kernel.cu:20-82).  zvals/mask zero-initialised by the caller
 * (ops/bitfield_func.py:27-28); rng is the host generator before the launch; voxel_trace optional (R,n_pts) init -1. */
ORC_API void orc_sparse_volume_sampling_bit(const float *rays_o, const float *rays_d, const float *near, const float *far,
                                            int n_pts, float dt, const float *aabb, int n_grid, const uint8_t *bitfield,
                                            float near_distance, uint64_t rng_state, uint64_t rng_inc, float *zvals,
                                            uint8_t *mask, int32_t *voxel_trace, int32_t *counts, int64_t n_rays) {
    const float *mn = aabb, *mx = aabb + 3;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n_rays; ++i) {
        orc_pcg32 rng = {rng_state, rng_inc};
        orc_pcg32_advance(&rng, (int64_t)(uint32_t)((uint32_t)i * 8u));
        const float *o = rays_o + 3 * i, *d = rays_d + 3 * i;
        float startt = fmaxf(near[i], near_distance);
        float far_end = far[i];
        float jit = dt * orc_pcg32_next_float(&rng);
        startt += jit;
        uint32_t j = 0;
        float t = startt;
        float pos[3];
        float *zr = zvals + i * (int64_t)n_pts;
        uint8_t *mr = mask + i * (int64_t)n_pts;
        while (t <= far_end && j < (uint32_t)n_pts) {
            for (int k = 0; k < 3; ++k) { float a = d[k] * t; pos[k] = o[k] + a; }
            if (!in_aabb(pos, mn, mx)) break;
            uint32_t idx = morton_idx_at(pos, mn, mx, (uint32_t)n_grid);
            if (bitfield[idx / 8] & (1u << (idx % 8))) {
                zr[j] = t;
                mr[j] = 1;
                if (voxel_trace) voxel_trace[i * (int64_t)n_pts + j] = (int32_t)idx;
                ++j;
                t += dt;
            } else {
                float t_target = t + dist_to_next_voxel(pos, d, mn, mx, (uint32_t)n_grid);
                do { t += dt; } while (t < t_target);
            }
        }
        if (counts) counts[i] = (int32_t)j;
        if (j > 0 && j < (uint32_t)n_pts) {
            float last = zr[j - 1];
            while (j < (uint32_t)n_pts) { zr[j] = last; ++j; }
        }
    }
}

/* K6 generate_grid_samples (bitfield_func_kernel.cu:141-181): all index arithmetic is uint32 and wraps */
ORC_API void orc_generate_grid_samples(const float *grid_in, int ema_step, int n_elements_i, int n_grid_i, float thresh,
                                       uint64_t rng_state, uint64_t rng_inc, float *positions, int32_t *indices) {
    const uint32_t n_elements = (uint32_t)n_elements_i, n_grid = (uint32_t)n_grid_i, step = (uint32_t)ema_step;
    const uint32_t n_per_level = n_grid * n_grid * n_grid;
    for (uint32_t i = 0; i < n_elements; ++i) {
        orc_pcg32 rng = {rng_state, rng_inc};
        orc_pcg32_advance(&rng, (int64_t)(uint32_t)(i * 4u));
        uint32_t idx = 0;
        for (uint32_t j = 0; j < 10; ++j) {
            idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % n_per_level;
            if (grid_in[idx] > thresh) break;
        }
        uint32_t pos_idx = idx % n_per_level;
        uint32_t x = morton3d_invert(pos_idx >> 0), y = morton3d_invert(pos_idx >> 1), z = morton3d_invert(pos_idx >> 2);
        float r0 = orc_pcg32_next_float(&rng), r1 = orc_pcg32_next_float(&rng), r2 = orc_pcg32_next_float(&rng);
        positions[3 * (int64_t)i + 0] = ((float)x + r0) / (float)n_grid;
        positions[3 * (int64_t)i + 1] = ((float)y + r1) / (float)n_grid;
        positions[3 * (int64_t)i + 2] = ((float)z + r2) / (float)n_grid;
        indices[i] = (int32_t)idx;
    }
}

/* K7 splat_grid_samples (bitfield_func_kernel.cu:215-229): max on the uint32 bit pattern */
ORC_API void orc_splat_grid_samples(const float *density, const int32_t *indices, int n, float *grid_tmp) {
    for (int i = 0; i < n; ++i) {
        uint32_t a, b;
        memcpy(&a, &grid_tmp[(uint32_t)indices[i]], 4);
        memcpy(&b, &density[i], 4);
        if (b > a) memcpy(&grid_tmp[(uint32_t)indices[i]], &b, 4);
    }
}

/* K8 ema_grid_samples_nerf (bitfield_func_kernel.cu:257-272) */
ORC_API void orc_ema_grid_samples_nerf(const float *grid_tmp, int n_elements, float decay, float *grid) {
    for (int i = 0; i < n_elements; ++i) {
        float importance = grid_tmp[i];
        float prev = grid[i];
        float dec = prev * decay;
        grid[i] = (prev < 0.f) ? prev : fmaxf(dec, importance);
    }
}

/* K9 grid_to_bitfield (bitfield_func_kernel.cu:301-321): threshold = min(opa_thres, mean) */
ORC_API void orc_update_bitfield(const float *grid, float mean, uint8_t *bitfield, float opa_thres, int n_grid) {
    const uint32_t n = (uint32_t)n_grid * (uint32_t)n_grid * (uint32_t)n_grid / 8u;
    const float thresh = opa_thres < mean ? opa_thres : mean;
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t bits = 0;
        for (int j = 0; j < 8; ++j) bits |= grid[(int64_t)i * 8 + j] > thresh ? (uint8_t)(1u << j) : 0;
        bitfield[i] = bits;
    }
}

/* K10 count_bitfield (bitfield_func_kernel.cu:350-366).  The reference tests `byte && (1 << j)` (LOGICAL and), so every
 * non-zero byte contributes 8 to the float counter whatever its population; restated as written. */
ORC_API void orc_count_bitfield(const uint8_t *bitfield, float *counter, int n_grid) {
    const uint32_t n = (uint32_t)n_grid * (uint32_t)n_grid * (uint32_t)n_grid / 8u;
    for (uint32_t i = 0; i < n; ++i)
        for (int j = 0; j < 8; ++j)
            if ((bitfield[i] && (uint8_t)(1u << j)) > 0) counter[0] += 1.0f;
}

/* =======================================================================================
 * `_multivol_func` (arcnerf/ops/src/multivol_func/multivol_func_kernel.cu, volume_func.h:196-298): n_cascade nested
 * volumes, volume m = the inner ("basic") one scaled by 2^m about its centre, each an n_grid^3 Morton bitfield; with
 * inclusive = 0 the inner volume itself has no grid and level m lives in slot m-1.  CUDA only: PARITY UNPINNED.
 * ===================================================================================== */

/* include/common.h:64-67 */
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }

/* volume_func.h:277-279 */
static inline float calc_dt(float t, float cone_angle, float min_step, float max_step) {
    return clampf(t * cone_angle, min_step, max_step);
}

/* volume_func.h:201-226: level = exponent of the largest |coordinate| in units of the inner half side */
static inline int mip_from_pos(const float pos[3], const float mn[3], const float mx[3], uint32_t n_cascades) {
    int e_max = 0;
    for (int k = 0; k < 3; ++k) {
        float center = (mn[k] + mx[k]) / 2.0f;
        float half = (mx[k] - mn[k]) / 2.0f;
        float inv = 1.0f / half;
        float a = fabsf(pos[k] - center) * inv;
        int e;
        frexpf(a, &e);
        if (k == 0 || e > e_max) e_max = e;
    }
    int m = e_max > 0 ? e_max : 0;
    return m < (int)n_cascades - 1 ? m : (int)n_cascades - 1;
}

/* volume_func.h:229-254: shrink the point by 2^-mip about the centre, then the Morton cell of the inner grid */
static inline uint32_t morton_idx_at_multivol(const float pos_in[3], uint32_t mip, const float mn[3], const float mx[3],
                                              uint32_t n) {
    float scale = scalbnf(1.0f, -(int)mip);
    int c[3];
    for (int k = 0; k < 3; ++k) {
        float center = (mn[k] + mx[k]) / 2.0f;
        float p = pos_in[k] - center;
        p = p * scale;
        p = p + center;
        float vs = (mx[k] - mn[k]) / (float)n;
        float vi = (p - mn[k]) / vs;
        c[k] = clampi((int)vi, 0, (int)n - 1);
    }
    return morton3d((uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2]);
}

/* K11 sparse_sampling_in_multivol_bitfield (multivol_func_kernel.cu:14-96).  min_aabb = inner volume, aabb = outermost. */
ORC_API void orc_sparse_sampling_in_multivol_bitfield(const float *rays_o, const float *rays_d, const float *near,
                                                      const float *far, int n_pts, float cone_angle, float min_step,
                                                      float max_step, const float *min_aabb, const float *aabb, int n_grid,
                                                      int n_cascade, const uint8_t *bitfield, float near_distance,
                                                      int inclusive, uint64_t rng_state, uint64_t rng_inc, float *zvals,
                                                      uint8_t *mask, int32_t *counts, int64_t n_rays) {
    const float *mn = aabb, *mx = aabb + 3, *imn = min_aabb, *imx = min_aabb + 3;
    const uint32_t n = (uint32_t)n_grid, level_cells = n * n * n;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n_rays; ++i) {
        orc_pcg32 rng = {rng_state, rng_inc};
        orc_pcg32_advance(&rng, (int64_t)(uint32_t)((uint32_t)i * 8u));
        const float *o = rays_o + 3 * i, *d = rays_d + 3 * i;
        float startt = fmaxf(near[i], near_distance);
        float far_end = far[i];
        float jit = calc_dt(startt, cone_angle, min_step, max_step) * orc_pcg32_next_float(&rng);
        startt += jit;
        uint32_t j = 0;
        float t = startt;
        float pos[3];
        float *zr = zvals + i * (int64_t)n_pts;
        uint8_t *mr = mask + i * (int64_t)n_pts;
        while (t <= far_end && j < (uint32_t)n_pts) {
            for (int k = 0; k < 3; ++k) { float a = d[k] * t; pos[k] = o[k] + a; }
            if (!in_aabb(pos, mn, mx)) break;
            float dt = calc_dt(t, cone_angle, min_step, max_step);
            uint32_t mip = (uint32_t)mip_from_pos(pos, imn, imx, (uint32_t)n_cascade);
            if (mip == 0 && !inclusive) {
                /* the ray re-entered the inner volume: drop everything sampled so far (no bkg -> fg -> bkg sequences) */
                while (j > 0) { zr[j] = 0.0f; mr[j] = 0; j--; }
                zr[j] = 0.0f;
                mr[j] = 0;
                float t_target = t + dist_to_next_voxel(pos, d, imn, imx, n);
                do { t += dt; } while (t < t_target);
            } else {
                uint32_t idx = morton_idx_at_multivol(pos, mip, imn, imx, n);
                uint32_t slot = inclusive ? mip : mip - 1;
                if (bitfield[idx / 8 + (level_cells * slot) / 8] & (1u << (idx % 8))) {
                    zr[j] = t;
                    mr[j] = 1;
                    ++j;
                    t += dt;
                } else {
                    float t_target = t + dist_to_next_voxel(pos, d, imn, imx, n);
                    do { t += calc_dt(t, cone_angle, min_step, max_step); } while (t < t_target);
                }
            }
        }
        if (counts) counts[i] = (int32_t)j;
        if (j > 0 && j < (uint32_t)n_pts) {
            float last = zr[j - 1];
            while (j < (uint32_t)n_pts) { zr[j] = last; ++j; }
        }
    }
}

/* K12 generate_grid_samples_multivol (multivol_func_kernel.cu:148-206).  aabb = inner volume (2,3). */
ORC_API void orc_generate_grid_samples_multivol(const float *grid_in, int ema_step, int n_elements_i, const float *aabb,
                                                int n_cascade, int n_grid_i, float thresh, int inclusive, uint64_t rng_state,
                                                uint64_t rng_inc, float *positions, int32_t *indices) {
    const uint32_t n_elements = (uint32_t)n_elements_i, n_grid = (uint32_t)n_grid_i, step = (uint32_t)ema_step;
    const uint32_t n_cascades = (uint32_t)n_cascade, n_per_level = n_grid * n_grid * n_grid;
    for (uint32_t i = 0; i < n_elements; ++i) {
        orc_pcg32 rng = {rng_state, rng_inc};
        orc_pcg32_advance(&rng, (int64_t)(uint32_t)(i * 4u));
        uint32_t level = 0;
        if (inclusive) {
            level = (uint32_t)(orc_pcg32_next_float(&rng) * (float)n_cascades) % n_cascades;
        } else {
            while (level == 0) level = (uint32_t)(orc_pcg32_next_float(&rng) * (float)n_cascades) % n_cascades;
        }
        uint32_t idx = 0;
        for (uint32_t j = 0; j < 10; ++j) {
            idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % n_per_level;
            idx += (inclusive ? level : level - 1) * n_per_level;
            if (grid_in[idx] > thresh) break;
        }
        uint32_t pos_idx = idx % n_per_level;
        uint32_t c[3] = {morton3d_invert(pos_idx >> 0), morton3d_invert(pos_idx >> 1), morton3d_invert(pos_idx >> 2)};
        float r[3];
        for (int k = 0; k < 3; ++k) r[k] = orc_pcg32_next_float(&rng);
        float scale = scalbnf(1.0f, (int)level);
        for (int k = 0; k < 3; ++k) {
            float center = (aabb[k] + aabb[3 + k]) / 2.0f;
            float len = aabb[3 + k] - aabb[k];
            float p = ((float)c[k] + r[k]) / (float)n_grid;
            p = p - 0.5f;
            p = p * len;
            p = p * scale;
            positions[3 * (int64_t)i + k] = p + center;
        }
        indices[i] = (int32_t)idx;
    }
}

/* update_bitfield_multivol (multivol_func_kernel.cu:242-300): K9's rule over every stored level */
ORC_API void orc_update_bitfield_multivol(const float *grid, float mean, uint8_t *bitfield, float opa_thres, int n_grid,
                                          int n_cascade, int inclusive) {
    const uint32_t n = (uint32_t)n_grid * (uint32_t)n_grid * (uint32_t)n_grid / 8u *
                       (uint32_t)(inclusive ? n_cascade : n_cascade - 1);
    const float thresh = opa_thres < mean ? opa_thres : mean;
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t bits = 0;
        for (int j = 0; j < 8; ++j) bits |= grid[(int64_t)i * 8 + j] > thresh ? (uint8_t)(1u << j) : 0;
        bitfield[i] = bits;
    }
}
