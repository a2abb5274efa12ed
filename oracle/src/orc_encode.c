/* ORACLE (test infrastructure only) — encoders of the path.  See orc_common.h. */
#include "orc_common.h"
#ifdef _OPENMP
#include <omp.h>
#endif

static const int PERM[8][3] = {{0,0,0},{0,1,0},{1,0,0},{1,1,0},{0,0,1},{0,1,1},{1,0,1},{1,1,1}};

/* fast_hash: arcnerf/models/base_modules/encoding/hashgrid_encoder.py:232-249 (int64 maths) */
static inline int64_t fast_hash3(int64_t cx, int64_t cy, int64_t cz, int64_t size) {
    int64_t h = 0;
    h ^= cx * 1LL;
    h ^= cy * 2654435761LL;
    h ^= cz * 805459861LL;
    return h % size;
}

/* Level geometry shared by fwd/bwd: follows hashgrid_encode_torch (hashgrid_encoder.py:191-230)
 * -> Volume.get_voxel_grid_info_from_xyz (arcnerf/geometry/volume.py:486-531). */
static inline int level_setup(const float *p, const float *mn, const float *mx, int res, int64_t c[3], float w[3],
                              float vs[3], float dw[3]) {
    float v[3];
    int ok = 1;
    for (int k = 0; k < 3; ++k) {
        vs[k] = (mx[k] - mn[k]) / (float)res;
        v[k] = (p[k] - mn[k]) / vs[k];
        if (!(v[k] >= 0) || !(v[k] < (float)res)) ok = 0;
    }
    if (!ok) return 0;
    for (int k = 0; k < 3; ++k) {
        c[k] = (int64_t)floorf(v[k]);
        float g0 = (float)c[k] * vs[k] + mn[0]; /* start_point[0] for every axis (volume.py:515) */
        float ww = (p[k] - g0) / vs[k];
        w[k] = ww < 0.0f ? 0.0f : (ww > 1.0f ? 1.0f : ww);
        if (dw) dw[k] = (ww >= 0.0f && ww <= 1.0f) ? 1.0f / vs[k] : 0.0f; /* torch.clip grad mask */
    }
    return 1;
}

/* ---------------------------------------------------------------------------------------
 * Hash-grid forward.  table (n_total, F) fp32; resolutions[L]; offsets[L+1]; out (S, L*F).
 * hash_idx (optional int64 (S, L, 8)) gets the table row of every corner (-1 if invalid).
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_hashgrid_fwd(const float *xyz, int64_t S, const float *table, int L, int F, const int32_t *resolutions,
                              const int64_t *offsets, const float *min_xyz, const float *max_xyz, float *out,
                              int64_t *hash_idx) {
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < S; ++s) {
        const float *p = xyz + 3 * s;
        for (int l = 0; l < L; ++l) {
            int64_t c[3];
            float w[3], vs[3];
            float *o = out + s * (int64_t)(L * F) + l * F;
            for (int f = 0; f < F; ++f) o[f] = 0.0f;
            if (!level_setup(p, min_xyz, max_xyz, resolutions[l], c, w, vs, NULL)) {
                if (hash_idx) for (int q = 0; q < 8; ++q) hash_idx[(s * L + l) * 8 + q] = -1;
                continue;
            }
            int64_t size = offsets[l + 1] - offsets[l];
            for (int q = 0; q < 8; ++q) {
                int64_t h = fast_hash3(c[0] + PERM[q][0], c[1] + PERM[q][1], c[2] + PERM[q][2], size) + offsets[l];
                if (hash_idx) hash_idx[(s * L + l) * 8 + q] = h;
                float wx = PERM[q][0] ? w[0] : 1.0f - w[0];
                float wy = PERM[q][1] ? w[1] : 1.0f - w[1];
                float wz = PERM[q][2] ? w[2] : 1.0f - w[2];
                float wt = (wx * wy) * wz;
                for (int f = 0; f < F; ++f) {
                    float a = table[h * F + f] * wt;
                    o[f] = o[f] + a;
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * Hash-grid backward: dtable (n_total,F) += scatter(dout * w); dxyz (S,3) optional.
 * dxyz follows autograd through w = clip((p-g0)/vs,0,1): d w/d p = 1/vs inside (0,1), else 0.
 * (torch.clip passes gradient where min <= x <= max, boundaries included.)
 * Deterministic summation order (sample-major per table row) for any thread count.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_hashgrid_bwd(const float *xyz, int64_t S, const float *table, const float *dout, int L, int F,
                              const int32_t *resolutions, const int64_t *offsets, const float *min_xyz,
                              const float *max_xyz, float *dtable, float *dxyz) {
    /* The table scatter runs as L x n_slices independent tasks: task (l, k) walks ALL samples of level l in order and applies the
     * updates whose row falls into its k-th slice of the level's rows.  Rows of different tasks are disjoint and every row still
     * receives its contributions in sample-major order, so the result is bit-identical to the serial loop for any thread count
     * (the price: the cell / hash arithmetic of a level is repeated n_slices times). */
    if (dtable) {
        int nthreads = 1;
#ifdef _OPENMP
        nthreads = omp_get_max_threads();
#endif
        int n_slices = nthreads / (L > 0 ? L : 1);
        if (n_slices < 1) n_slices = 1;
        if (n_slices > 16) n_slices = 16;
#pragma omp parallel for schedule(dynamic, 1)
        for (int task = 0; task < L * n_slices; ++task) {
            const int l = task / n_slices, k = task % n_slices;
            const int64_t size = offsets[l + 1] - offsets[l];
            const int64_t lo = size * k / n_slices, hi = size * (k + 1) / n_slices;
            for (int64_t s = 0; s < S; ++s) {
                const float *p = xyz + 3 * s;
                int64_t c[3];
                float w[3], vs[3], dw[3];
                if (!level_setup(p, min_xyz, max_xyz, resolutions[l], c, w, vs, dw)) continue;
                const float *g = dout + s * (int64_t)(L * F) + l * F;
                for (int q = 0; q < 8; ++q) {
                    const int64_t r = fast_hash3(c[0] + PERM[q][0], c[1] + PERM[q][1], c[2] + PERM[q][2], size);
                    if (r < lo || r >= hi) continue;
                    const int64_t h = r + offsets[l];
                    float wx = PERM[q][0] ? w[0] : 1.0f - w[0];
                    float wy = PERM[q][1] ? w[1] : 1.0f - w[1];
                    float wz = PERM[q][2] ? w[2] : 1.0f - w[2];
                    float wt = (wx * wy) * wz;
                    for (int f = 0; f < F; ++f) dtable[h * F + f] += g[f] * wt;
                }
            }
        }
    }
    if (!dxyz) return;
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < S; ++s) {
        const float *p = xyz + 3 * s;
        float gx[3] = {0.f, 0.f, 0.f};
        for (int l = 0; l < L; ++l) {
            int64_t c[3];
            float w[3], vs[3], dw[3];
            if (!level_setup(p, min_xyz, max_xyz, resolutions[l], c, w, vs, dw)) continue;
            const float *g = dout + s * (int64_t)(L * F) + l * F;
            int64_t size = offsets[l + 1] - offsets[l];
            for (int q = 0; q < 8; ++q) {
                int64_t h = fast_hash3(c[0] + PERM[q][0], c[1] + PERM[q][1], c[2] + PERM[q][2], size) + offsets[l];
                float wx = PERM[q][0] ? w[0] : 1.0f - w[0];
                float wy = PERM[q][1] ? w[1] : 1.0f - w[1];
                float wz = PERM[q][2] ? w[2] : 1.0f - w[2];
                float dot = 0.f;
                for (int f = 0; f < F; ++f) dot += g[f] * table[h * F + f];
                float sx = PERM[q][0] ? 1.0f : -1.0f, sy = PERM[q][1] ? 1.0f : -1.0f, sz = PERM[q][2] ? 1.0f : -1.0f;
                gx[0] += dot * sx * wy * wz * dw[0];
                gx[1] += dot * wx * sy * wz * dw[1];
                gx[2] += dot * wx * wy * sz * dw[2];
            }
        }
        for (int k = 0; k < 3; ++k) dxyz[3 * s + k] = gx[k];
    }
}

/* ---------------------------------------------------------------------------------------
 * Hash-grid backward differentiated once more (autograd of hashgrid_encode_torch, hashgrid_encoder.py:191-230, taken twice:
 * BaseGeoNet.forward_with_grad builds normals with create_graph=True and the loss reaches them).
 * With y_f = sum_q T[r_q,f] W_q and W_q = a_x a_y a_z (a_k = w_k or 1 - w_k), the first backward gives
 *   dx_k = sum_q (sum_f dout_f T[r_q,f]) dW_q/dp_k.
 * For an upstream gradient gdx (S,3) on dx, with D_q = sum_k gdx_k dW_q/dp_k:
 *   ddout_f        = sum_q T[r_q,f] D_q                      (a gather with weights D)
 *   dtable[r_q,f] += dout_f D_q                              (a scatter with weights D)
 *   d2x_j          = sum_q (sum_f dout_f T[r_q,f]) dD_q/dp_j (cross terms only: each a_k is piecewise linear)
 * Any output pointer may be NULL.  Serial, sample-major summation order.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_hashgrid_bwd_bwd(const float *xyz, int64_t S, const float *gdx, const float *table, const float *dout, int L,
                                  int F, const int32_t *resolutions, const int64_t *offsets, const float *min_xyz,
                                  const float *max_xyz, float *ddout, float *dtable, float *d2xyz) {
    for (int64_t s = 0; s < S; ++s) {
        const float *p = xyz + 3 * s, *gd = gdx + 3 * s;
        float hx[3] = {0.f, 0.f, 0.f};
        for (int l = 0; l < L; ++l) {
            int64_t c[3];
            float w[3], vs[3], dw[3];
            float *dd = ddout ? ddout + s * (int64_t)(L * F) + l * F : NULL;
            if (dd) for (int f = 0; f < F; ++f) dd[f] = 0.0f;
            if (!level_setup(p, min_xyz, max_xyz, resolutions[l], c, w, vs, dw)) continue;
            const float *g = dout + s * (int64_t)(L * F) + l * F;
            int64_t size = offsets[l + 1] - offsets[l];
            for (int q = 0; q < 8; ++q) {
                int64_t h = fast_hash3(c[0] + PERM[q][0], c[1] + PERM[q][1], c[2] + PERM[q][2], size) + offsets[l];
                float a[3], sd[3]; /* a_k and d a_k / d p_k */
                for (int k = 0; k < 3; ++k) {
                    a[k] = PERM[q][k] ? w[k] : 1.0f - w[k];
                    sd[k] = (PERM[q][k] ? 1.0f : -1.0f) * dw[k];
                }
                float D = gd[0] * sd[0] * a[1] * a[2];
                D = D + gd[1] * a[0] * sd[1] * a[2];
                D = D + gd[2] * a[0] * a[1] * sd[2];
                float dot = 0.f;
                for (int f = 0; f < F; ++f) {
                    if (dd) dd[f] = dd[f] + table[h * F + f] * D;
                    if (dtable) dtable[h * F + f] += g[f] * D;
                    dot += g[f] * table[h * F + f];
                }
                if (d2xyz) {
                    hx[0] += dot * sd[0] * (gd[1] * sd[1] * a[2] + gd[2] * a[1] * sd[2]);
                    hx[1] += dot * sd[1] * (gd[0] * sd[0] * a[2] + gd[2] * a[0] * sd[2]);
                    hx[2] += dot * sd[2] * (gd[0] * sd[0] * a[1] + gd[1] * a[0] * sd[1]);
                }
            }
        }
        if (d2xyz) for (int k = 0; k < 3; ++k) d2xyz[3 * s + k] = hx[k];
    }
}

/* ---------------------------------------------------------------------------------------
 * FreqEmbedder.forward (encoding/freq_encoder.py:65-88), log_sampling, (sin, cos).
 * out (S, D*(include_input + 2*n_freqs)); order: x, then per freq: sin(all dims), cos(all dims).
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_freq_fwd(const float *x, int64_t S, int D, int n_freqs, int include_input, float *out) {
    int od = D * (include_input ? 1 : 0) + D * 2 * n_freqs;
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < S; ++s) {
        float *o = out + s * od;
        int c = 0;
        if (include_input) for (int k = 0; k < D; ++k) o[c++] = x[s * D + k];
        for (int f = 0; f < n_freqs; ++f) {
            float freq = ldexpf(1.0f, f);
            for (int k = 0; k < D; ++k) o[c++] = sinf(x[s * D + k] * freq);
            for (int k = 0; k < D; ++k) o[c++] = cosf(x[s * D + k] * freq);
        }
    }
}

ORC_API void orc_freq_bwd(const float *x, const float *dout, int64_t S, int D, int n_freqs, int include_input, float *dx) {
    int od = D * (include_input ? 1 : 0) + D * 2 * n_freqs;
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < S; ++s) {
        const float *g = dout + s * od;
        for (int k = 0; k < D; ++k) {
            float acc = 0.f;
            int c = 0;
            if (include_input) { acc += g[k]; c = D; }
            for (int f = 0; f < n_freqs; ++f) {
                float freq = ldexpf(1.0f, f);
                float a = x[s * D + k] * freq;
                acc += g[c + k] * cosf(a) * freq;
                acc -= g[c + D + k] * sinf(a) * freq;
                c += 2 * D;
            }
            dx[s * D + k] = acc;
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * SHEmbedder torch branch (encoding/sh_encoder.py:101-185): input mapped (d+1)/2 and the
 * polynomials evaluated on THAT value (literal restatement).  degree 1..5; out (S, deg^2 [+3]).
 * ------------------------------------------------------------------------------------- */
static const float SH1[1] = {0.28209479177387814f};
static const float SH2[3] = {-0.4886025119029199f, 0.4886025119029199f, -0.4886025119029199f};
static const float SH3[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                             0.5462742152960396f};
static const float SH4[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
static const float SH5[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f, -0.6690465435572892f,
                             0.10578554691520431f, -0.6690465435572892f, 0.47308734787878004f, -1.7701307697799304f,
                             0.6258357354491761f};

ORC_API void orc_sh_fwd(const float *dirs, int64_t S, int degree, int include_input, float *out) {
    int od = degree * degree + (include_input ? 3 : 0);
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < S; ++s) {
        float *o = out + s * od;
        int c = 0;
        if (include_input) { o[0] = dirs[3 * s]; o[1] = dirs[3 * s + 1]; o[2] = dirs[3 * s + 2]; c = 3; }
        float x = (dirs[3 * s] + 1.0f) / 2.0f, y = (dirs[3 * s + 1] + 1.0f) / 2.0f, z = (dirs[3 * s + 2] + 1.0f) / 2.0f;
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        o[c++] = SH1[0];
        if (degree <= 1) continue;
        o[c++] = SH2[0] * y; o[c++] = SH2[1] * z; o[c++] = SH2[2] * x;
        if (degree <= 2) continue;
        o[c++] = SH3[0] * xy; o[c++] = SH3[1] * yz; o[c++] = SH3[2] * (3.0f * zz - 1.0f);
        o[c++] = SH3[3] * xz; o[c++] = SH3[4] * (xx - yy);
        if (degree <= 3) continue;
        o[c++] = SH4[0] * y * (3.0f * xx - yy);
        o[c++] = SH4[1] * xy * z;
        o[c++] = SH4[2] * y * (5.0f * zz - 1.0f);
        o[c++] = SH4[3] * z * (5.0f * zz - 3.0f);
        o[c++] = SH4[4] * x * (5.0f * zz - 1.0f);
        o[c++] = SH4[5] * z * (xx - yy);
        o[c++] = SH4[6] * x * (xx - 3.0f * yy);
        if (degree <= 4) continue;
        o[c++] = SH5[0] * xy * (xx - yy);
        o[c++] = SH5[1] * yz * (3.0f * xx - yy);
        o[c++] = SH5[2] * xy * (7.0f * zz - 1.0f);
        o[c++] = SH5[3] * yz * (7.0f * zz - 3.0f);
        o[c++] = SH5[4] * (zz * (35.0f * zz - 30.0f) + 3.0f);
        o[c++] = SH5[5] * xz * (7.0f * zz - 3.0f);
        o[c++] = SH5[6] * (xx - yy) * (7.0f * zz - 1.0f);
        o[c++] = SH5[7] * xz * (xx - 3.0f * yy);
        o[c++] = SH5[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
    }
}
