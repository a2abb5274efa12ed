/* ORACLE (test infrastructure only) — sampling / bounds part of the path.  See orc_common.h. */
#include "orc_common.h"

/* ---------------------------------------------------------------------------------------
 * pcg32 exports (arcnerf/ops/include/pcg32.h:50-165).  state/inc are passed in and out so
 * the python side can keep the (seed, call counter) explicit.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_pcg32_init(uint64_t initstate, uint64_t initseq, uint64_t *state_inc) {
    orc_pcg32 r;
    orc_pcg32_seed(&r, initstate, initseq);
    state_inc[0] = r.state;
    state_inc[1] = r.inc;
}

ORC_API void orc_pcg32_advance_state(uint64_t *state_inc, int64_t delta) {
    orc_pcg32 r = {state_inc[0], state_inc[1]};
    orc_pcg32_advance(&r, delta);
    state_inc[0] = r.state;
}

ORC_API void orc_pcg32_draw(uint64_t *state_inc, int n, uint32_t *out_uint, float *out_float) {
    /* draws n values; out_uint and out_float both receive the SAME stream positions
     * (float k is built from uint k), whichever pointers are non-null */
    orc_pcg32 r = {state_inc[0], state_inc[1]};
    for (int i = 0; i < n; ++i) {
        uint32_t u = orc_pcg32_next_uint(&r);
        if (out_uint) out_uint[i] = u;
        if (out_float) {
            union { uint32_t u; float f; } x;
            x.u = (u >> 9) | 0x3f800000u;
            out_float[i] = x.f - 1.0f;
        }
    }
    state_inc[0] = r.state;
}

/* ---------------------------------------------------------------------------------------
 * device helpers of arcnerf/ops/include/volume_func.h
 * ------------------------------------------------------------------------------------- */

/* volume_func.h:17-56 slab test; returns 0 and (-1,-1) on miss */
static inline void slab_test(const float o[3], const float d[3], const float mn[3], const float mx[3],
                             float *tmin_out, float *tmax_out) {
    float tmin = (mn[0] - o[0]) / d[0];
    float tmax = (mx[0] - o[0]) / d[0];
    if (tmin > tmax) { float c = tmin; tmin = tmax; tmax = c; }
    float tymin = (mn[1] - o[1]) / d[1];
    float tymax = (mx[1] - o[1]) / d[1];
    if (tymin > tymax) { float c = tymin; tymin = tymax; tymax = c; }
    if (tmin > tymax || tymin > tmax) { *tmin_out = -1.0f; *tmax_out = -1.0f; return; }
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (mn[2] - o[2]) / d[2];
    float tzmax = (mx[2] - o[2]) / d[2];
    if (tzmin > tzmax) { float c = tzmin; tzmin = tzmax; tzmax = c; }
    if (tmin > tzmax || tzmin > tmax) { *tmin_out = -1.0f; *tmax_out = -1.0f; return; }
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    *tmin_out = tmin;
    *tmax_out = tmax;
}

/* volume_func.h:59-88: voxel_idx = (p-min)/((max-min)/n); flat = x*n*n + y*n + z */
static inline int occupied_at(const float p[3], const uint8_t *bitfield, const float mn[3], const float mx[3],
                              uint32_t n, uint32_t *flat_out) {
    float vi[3];
    for (int k = 0; k < 3; ++k) {
        float vs = (mx[k] - mn[k]) / (float)n;
        vi[k] = (p[k] - mn[k]) / vs;
    }
    float lo = vi[0] < vi[1] ? vi[0] : vi[1]; lo = lo < vi[2] ? lo : vi[2];
    float hi = vi[0] > vi[1] ? vi[0] : vi[1]; hi = hi > vi[2] ? hi : vi[2];
    if (lo < 0 || hi >= (float)n) return 0;
    uint32_t x = (uint32_t)floorf(vi[0]);
    uint32_t y = (uint32_t)floorf(vi[1]);
    uint32_t z = (uint32_t)floorf(vi[2]);
    uint32_t flat = x * (n * n) + y * n + z;
    if (flat_out) *flat_out = flat;
    return bitfield[flat] != 0;
}

/* volume_func.h:92-96 */
static inline int in_aabb(const float p[3], const float mn[3], const float mx[3]) {
    return p[0] >= mn[0] && p[1] >= mn[1] && p[2] >= mn[2] && p[0] <= mx[0] && p[1] <= mx[1] && p[2] <= mx[2];
}

/* volume_func.h:99-134 */
static inline float dist_to_next_voxel(const float pos[3], const float d[3], const float mn[3], const float mx[3],
                                       uint32_t n) {
    float t_min = 0.f;
    for (int k = 0; k < 3; ++k) {
        float center = (mn[k] + mx[k]) / 2.0f;
        float half = (mx[k] - mn[k]) / 2.0f;
        float inv_d = 1.0f / d[k];
        float p = (float)n * pos[k];
        float sgn = copysignf(1.0f, d[k]);
        float hs = half * sgn;
        float a = p + center;
        a = a + hs;
        float t = (floorf(a) - p) * inv_d;
        if (k == 0 || t < t_min) t_min = t;
    }
    return fmaxf(t_min / (float)n, 0.0f);
}

static inline float advance_to_next_voxel(float t, float dt, const float pos[3], const float d[3],
                                          const float mn[3], const float mx[3], uint32_t n) {
    float t_target = t + dist_to_next_voxel(pos, d, mn, mx, n);
    do { t += dt; } while (t < t_target);
    return t;
}

/* ---------------------------------------------------------------------------------------
 * K1  check_pts_in_occ_voxel  (volume_func_kernel.cu:16-37)
 * aabb is (2,3): xyz_min then xyz_max.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_check_pts_in_occ_voxel(const float *xyz, const uint8_t *bitfield, const float *aabb, int n_grid,
                                        uint8_t *out, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)occupied_at(xyz + 3 * i, bitfield, aabb, aabb + 3, (uint32_t)n_grid, NULL);
}

/* ---------------------------------------------------------------------------------------
 * K2  aabb_intersection  (volume_func_kernel.cu:74-123).  aabb (V,2,3).
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_aabb_intersection(const float *rays_o, const float *rays_d, const float *aabb, float *near, float *far,
                                   float *pts, uint8_t *mask, int64_t n_rays, int64_t n_v) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_rays * n_v; ++i) {
        int64_t r = i / n_v, v = i % n_v;
        const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        float tmin, tmax;
        slab_test(o, d, aabb + 6 * v, aabb + 6 * v + 3, &tmin, &tmax);
        if (tmin > 0) { near[i] = tmin; far[i] = tmax; mask[i] = 1; }
        else { near[i] = 0.0f; far[i] = 0.0f; mask[i] = 0; }
        for (int k = 0; k < 3; ++k) {
            float a = near[i] * d[k];
            pts[i * 6 + k] = o[k] + a;
            float b = far[i] * d[k];
            pts[i * 6 + 3 + k] = o[k] + b;
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * torch path of aabb_ray_intersection (arcnerf/geometry/ray.py:295-339); aabb (V,3,2).
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_aabb_intersection_torch(const float *rays_o, const float *rays_d, const float *aabb32, float eps,
                                         float *near_out, float *far_out, float *pts, uint8_t *mask_out, int64_t n_rays,
                                         int64_t n_v) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_rays * n_v; ++i) {
        int64_t r = i / n_v, v = i % n_v;
        const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        const float *bb = aabb32 + 6 * v; /* bb[2*dim+0]=min, bb[2*dim+1]=max */
        float near = 0.0f, far = 10000.0f;
        int mask = 1;
        for (int dim = 0; dim < 3; ++dim) {
            float mn = bb[2 * dim], mx = bb[2 * dim + 1];
            int axis = fabsf(d[dim]) < eps;
            int out = (o[dim] < mn) || (o[dim] > mx);
            if (axis && out) mask = 0;
            float t1 = (mn - o[dim]) / d[dim];
            float t2 = (mx - o[dim]) / d[dim];
            /* torch.min/max over a 2-vector propagate NaN */
            float lo, hi;
            if (isnan(t1) || isnan(t2)) { lo = NAN; hi = NAN; }
            else { lo = t1 < t2 ? t1 : t2; hi = t1 < t2 ? t2 : t1; }
            if (mask && lo > near) near = lo;
            if (mask && hi < far) far = hi;
            if (near > far) mask = 0;
        }
        if (near < 0.0f) near = 0.0f;
        if (far < 0.0f) far = 0.0f;
        if (!mask) { near = 0.0f; far = 0.0f; }
        else { near += eps; far -= eps; }
        near_out[i] = near; far_out[i] = far; mask_out[i] = (uint8_t)mask;
        for (int k = 0; k < 3; ++k) {
            /* get_ray_points_by_zvals: rays_o + zvals * rays_d */
            pts[i * 6 + k] = o[k] + near * d[k];
            pts[i * 6 + 3 + k] = o[k] + far * d[k];
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * sphere_ray_intersection (arcnerf/geometry/ray.py:180-255) with set_tensor_to_zeros (|x| < 1e-5 -> 0,
 * common/utils/torch_utils.py:50-54) and batch_dot_product (sum over the last dim).  radius (n_r), one origin.
 * near/far (n_rays, n_r), pts (n_rays, n_r, 2, 3), mask (n_rays, n_r).
 * ------------------------------------------------------------------------------------- */
static inline float orc_zero_small(float v) { return fabsf(v) < 1e-5f ? 0.0f : v; }

ORC_API void orc_sphere_intersection(const float *rays_o, const float *rays_d, const float *radius, const float *origin,
                                     float *near_out, float *far_out, float *pts, uint8_t *mask_out, int64_t n_rays,
                                     int64_t n_r) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_rays * n_r; ++i) {
        int64_t ray = i / n_r, k = i % n_r;
        const float *o = rays_o + 3 * ray, *d = rays_d + 3 * ray;
        const float r = radius[k];
        float oc[3];
        for (int a = 0; a < 3; ++a) oc[a] = origin[a] - o[a];
        float z_half = oc[0] * d[0];
        z_half = z_half + oc[1] * d[1];
        z_half = z_half + oc[2] * d[2];
        z_half = orc_zero_small(z_half);
        float oc2 = oc[0] * oc[0];
        oc2 = oc2 + oc[1] * oc[1];
        oc2 = oc2 + oc[2] * oc[2];
        const int inside = sqrtf(oc2) <= r;
        int mask = (z_half > 0.0f) || inside;
        float d2 = orc_zero_small(oc2 - z_half * z_half);
        mask = mask && (d2 >= 0.0f);
        float z_off = orc_zero_small(r * r - d2);
        mask = mask && (z_off >= 0.0f);
        z_off = sqrtf(z_off); /* NaN where masked out, overwritten below */
        float near = z_half - z_off, far = z_half + z_off;
        if (!(near >= 0.0f)) near = isnan(near) ? near : 0.0f; /* clamp_min keeps NaN */
        if (!(far >= 0.0f)) far = isnan(far) ? far : 0.0f;
        if (!mask) { near = 0.0f; far = 0.0f; }
        near_out[i] = near; far_out[i] = far; mask_out[i] = (uint8_t)mask;
        for (int a = 0; a < 3; ++a) {
            pts[i * 6 + a] = o[a] + near * d[a];
            pts[i * 6 + 3 + a] = o[a] + far * d[a];
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * K3  sparse_volume_sampling (volume_func_kernel.cu:174-236).  zvals/mask must be
 * zero-initialised by the caller (ops/volume_func.py:100-105).  rng (state,inc) is the
 * host generator BEFORE the launch; the kernel copy advances i*8 per ray.
 * voxel_trace (optional, int32 (R,n_pts), caller-initialised to -1) records the flat voxel
 * index of every emitted sample so integer parity can be asserted.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_sparse_volume_sampling(const float *rays_o, const float *rays_d, const float *near, const float *far,
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
            uint32_t flat = 0;
            if (occupied_at(pos, bitfield, mn, mx, (uint32_t)n_grid, &flat)) {
                zr[j] = t;
                mr[j] = 1;
                if (voxel_trace) voxel_trace[i * (int64_t)n_pts + j] = (int32_t)flat;
                ++j;
                t += dt;
            } else {
                t = advance_to_next_voxel(t, dt, pos, d, mn, mx, (uint32_t)n_grid);
            }
        }
        if (counts) counts[i] = (int32_t)j;
        if (j > 0 && j < (uint32_t)n_pts) {
            float last = zr[j - 1];
            while (j < (uint32_t)n_pts) { zr[j] = last; ++j; }
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * K4  tensor_reduce_max (volume_func_kernel.cu:297-309): atomicMax on the uint32 bit
 * pattern — exact for non-negative floats, reproduced literally (bit-pattern max).
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_tensor_reduce_max(const float *full, const int64_t *idx, int n_group, float *uni, int64_t n) {
    (void)n_group;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t a, b;
        memcpy(&a, &uni[idx[i]], 4);
        memcpy(&b, &full[i], 4);
        if (b > a) memcpy(&uni[idx[i]], &b, 4);
    }
}

/* ---------------------------------------------------------------------------------------
 * Occupancy update (arcnerf/geometry/volume.py:983-1017), voxel_idx given as flat
 * x*n*n+y*n+z, assumed unique.  ema<0 means "None".
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_update_opafield(float *opafield, const int64_t *flat_idx, const float *opacity, int64_t n, float ema) {
    for (int64_t i = 0; i < n; ++i) {
        float old = opafield[flat_idx[i]];
        float upd = opacity[i];
        if (ema >= 0.0f) { float a = old * ema; upd = a > opacity[i] ? a : opacity[i]; if (isnan(a) || isnan(opacity[i])) upd = NAN; }
        opafield[flat_idx[i]] = (old >= 0) ? upd : old;
    }
}

/* thres = min(mean(clamp(opa,0)), threshold); bitfield = opa >= thres ('overwrite') */
ORC_API float orc_update_bitfield_by_opafield(const float *opafield, uint8_t *bitfield, int64_t n, float threshold) {
    /* torch .mean() of a float32 tensor accumulates pairwise in fp32; use double then round,
     * the comparison below is done against the fp32-rounded python float like the reference */
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc += (double)(opafield[i] > 0 ? opafield[i] : 0.0f);
    float mean = (float)(acc / (double)n);
    float thres = mean < threshold ? mean : threshold;
    for (int64_t i = 0; i < n; ++i) bitfield[i] = opafield[i] >= thres;
    return thres;
}

/* ---------------------------------------------------------------------------------------
 * Voxel / grid info (arcnerf/geometry/volume.py:486-531) for one resolution.
 * outputs: voxel_idx int64 (B,3) (-1 if invalid), valid (B), corner idx int64 (B,8,3),
 * weights (B,8) — rows of invalid points are left untouched.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_voxel_grid_info(const float *pts, int64_t n, const float *min_xyz, const float *max_xyz, int n_grid,
                                 int64_t *voxel_idx, uint8_t *valid, int64_t *corner_idx, float *weights) {
    static const int PERM[8][3] = {{0,0,0},{0,1,0},{1,0,0},{1,1,0},{0,0,1},{0,1,1},{1,0,1},{1,1,1}};
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float v[3], vs[3];
        int ok = 1;
        for (int k = 0; k < 3; ++k) {
            vs[k] = (max_xyz[k] - min_xyz[k]) / (float)n_grid;
            v[k] = (pts[3 * i + k] - min_xyz[k]) / vs[k];
            if (!(v[k] >= 0) || !(v[k] < (float)n_grid)) ok = 0;
        }
        valid[i] = (uint8_t)ok;
        int64_t c[3];
        for (int k = 0; k < 3; ++k) { c[k] = ok ? (int64_t)floorf(v[k]) : -1; voxel_idx[3 * i + k] = c[k]; }
        if (!ok) continue;
        float w[3];
        for (int k = 0; k < 3; ++k) {
            /* grid_pts = idx * voxel_size + start_point[0]  (x-min used for every axis, volume.py:515) */
            float g0 = (float)c[k] * vs[k] + min_xyz[0];
            float ww = (pts[3 * i + k] - g0) / vs[k];
            w[k] = ww < 0.0f ? 0.0f : (ww > 1.0f ? 1.0f : ww);
        }
        for (int q = 0; q < 8; ++q) {
            float wx = PERM[q][0] ? w[0] : 1.0f - w[0];
            float wy = PERM[q][1] ? w[1] : 1.0f - w[1];
            float wz = PERM[q][2] ? w[2] : 1.0f - w[2];
            if (corner_idx) for (int k = 0; k < 3; ++k) corner_idx[(i * 8 + q) * 3 + k] = c[k] + PERM[q][k];
            if (weights) weights[i * 8 + q] = (wx * wy) * wz;
        }
    }
}

ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------------------------------
 * get_rays (arcnerf/render/ray_helper.py:12-119) = pixel_to_cam (geometry/projection.py:8-34: x_cam = (i - s (j - cy) / fy - cx)
 * / fx * z, y_cam = (j - cy) / fy * z, z = 1) + cam_to_world (einsum R x + t, transformation.py:44-59) - cam_loc, then normalize
 * (v / (|v| + 1e-8)) or the NDC warp (get_ndc_rays, ray_helper.py:122-153).
 * Pixel p of the flat list: index == NULL -> p = i * H + j (wh_order, column-major) or p = j * W + i (row-major); else index[p] is
 * the column-major id i * H + j.  rays_r (optional, full image only, W >= 3): |d(i,j) - d(i+1,j)| * 2 / sqrt(12) (mip-nerf radius);
 * the last column takes the value of column W-3 (the reference appends `dx[-2:-1]`).
 * ------------------------------------------------------------------------------------- */
static void orc_ray_dir(const float *K, const float *c2w, float pi, float pj, int normalize_d, float out[3]) {
    const float fx = K[0], s = K[1], cx = K[2], fy = K[4], cy = K[5];
    float cam[3];
    cam[0] = (pi - (s * (pj - cy) / fy) - cx) / fx * 1.0f;
    cam[1] = (pj - cy) / fy * 1.0f;
    cam[2] = 1.0f;
    for (int k = 0; k < 3; ++k) {
        float w = c2w[4 * k + 0] * cam[0];
        w = w + c2w[4 * k + 1] * cam[1];
        w = w + c2w[4 * k + 2] * cam[2];
        w = w + c2w[4 * k + 3];
        out[k] = w - c2w[4 * k + 3];
    }
    if (normalize_d) {
        float nrm = sqrtf(out[0] * out[0] + out[1] * out[1] + out[2] * out[2]) + 1e-8f;
        for (int k = 0; k < 3; ++k) out[k] = out[k] / nrm;
    }
}

ORC_API void orc_get_rays(int W, int H, const float *K, const float *c2w, int wh_order, const int64_t *index, int64_t n,
                          int center_pixel, int normalize_d, int ndc, float ndc_near, float *rays_o, float *rays_d, float *rays_r) {
    const float off = center_pixel ? 0.5f : 0.0f;
    for (int64_t p = 0; p < n; ++p) {
        int64_t i, j;
        if (index) { i = index[p] / H; j = index[p] % H; }
        else if (wh_order) { i = p / H; j = p % H; }
        else { j = p / W; i = p % W; }
        float d[3], o[3] = {c2w[3], c2w[7], c2w[11]};
        orc_ray_dir(K, c2w, (float)i + off, (float)j + off, normalize_d && !ndc, d);
        if (ndc) {
            const float fx = K[0], fy = K[4];
            float t = -(ndc_near + o[2]) / d[2];
            for (int k = 0; k < 3; ++k) o[k] = o[k] + t * d[k];
            const float ax = -1.0f / ((float)W / (2.0f * fx)), ay = -1.0f / ((float)H / (2.0f * fy));
            float no[3] = {ax * o[0] / o[2], ay * o[1] / o[2], 1.0f + 2.0f * ndc_near / o[2]};
            float nd[3] = {ax * (d[0] / d[2] - o[0] / o[2]), ay * (d[1] / d[2] - o[1] / o[2]), -2.0f * ndc_near / o[2]};
            for (int k = 0; k < 3; ++k) { o[k] = no[k]; d[k] = nd[k]; }
        }
        for (int k = 0; k < 3; ++k) { rays_o[3 * p + k] = o[k]; rays_d[3 * p + k] = d[k]; }
    }
    if (rays_r && !index) {
        for (int64_t p = 0; p < n; ++p) {
            int64_t i, j;
            if (wh_order) { i = p / H; j = p % H; } else { j = p / W; i = p % W; }
            int64_t ia = i < W - 1 ? i : W - 3;   /* `dx[-2:-1]` (ray_helper.py:108,112): the last column takes the SECOND-to-last difference */
            int64_t pa = wh_order ? ia * H + j : j * W + ia, pb = wh_order ? (ia + 1) * H + j : j * W + ia + 1;
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) { float df = rays_d[3 * pa + k] - rays_d[3 * pb + k]; acc += df * df; }
            rays_r[p] = sqrtf(acc) * 2.0f / sqrtf(12.0f);
        }
    }
}
