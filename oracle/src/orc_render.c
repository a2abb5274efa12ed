/* ORACLE (test infrastructure only) — z sampling helpers and alpha compositing.  See orc_common.h. */
#include "orc_common.h"

/* ---------------------------------------------------------------------------------------
 * ray_marching forward (arcnerf/render/ray_helper.py:476-593) + alpha_to_weights (:596-620).
 *
 * sigma (R,P) or NULL, alpha_in (R,P) or NULL (NeuS branch, :550-556), radiance (R,P,3) or NULL,
 * zvals (R,P), noise (R,Pe) or NULL (pre-drawn randn*noise_std), bkg (Rb,3) with Rb in {0,1,R}.
 * Pe = P if add_inf_z or alpha_in given, else P-1 (last sample dropped, :541-545).
 * Per-sample outputs (R,Pe): alpha, trans, weights (any may be NULL).
 * Returns -1 if some delta < 0 (the reference asserts, :534), else 0.
 * ------------------------------------------------------------------------------------- */
ORC_API int orc_ray_marching_fwd(const float *sigma, const float *alpha_in, const float *radiance, const float *zvals,
                                 const float *noise, const float *bkg, int64_t bkg_rows, int64_t R, int P, int add_inf_z,
                                 int white_bkg, float *rgb, float *depth, float *mask, float *alpha_out, float *trans_out,
                                 float *weights_out) {
    const int Pe = (add_inf_z || alpha_in) ? P : P - 1;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t r = 0; r < R; ++r) {
        const float *z = zvals + r * P;
        float T = 1.0f, acc_d = 0.f, acc_m = 0.f, acc_c[3] = {0.f, 0.f, 0.f};
        float T_last = 1.0f;
        for (int i = 0; i < Pe; ++i) {
            float delta;
            if (i < P - 1) {
                delta = z[i + 1] - z[i];
                if (fabsf(delta) < 1e-5f) delta = 0.0f;
                if (delta < 0) bad |= 1;
            } else {
                delta = 1e10f;
            }
            float a;
            if (alpha_in) {
                a = alpha_in[r * P + i];
            } else {
                float s = sigma[r * P + i];
                if (noise) s = s + noise[r * Pe + i];
                s = s > 0.f ? s : 0.f;
                a = 1.0f - expf(-s * delta);
            }
            float w = a * T;
            T_last = T;
            if (alpha_out) alpha_out[r * Pe + i] = a;
            if (trans_out) trans_out[r * Pe + i] = T;
            if (weights_out) weights_out[r * Pe + i] = w;
            acc_d += w * z[i];
            acc_m += w;
            if (radiance) for (int c = 0; c < 3; ++c) acc_c[c] += w * radiance[(r * P + i) * 3 + c];
            float q = (1.0f - a) + 1e-10f;
            T = T * q;
        }
        if (depth) depth[r] = acc_d;
        if (mask) mask[r] = acc_m;
        if (rgb && radiance) {
            for (int c = 0; c < 3; ++c) {
                float v = acc_c[c];
                if (bkg && bkg_rows > 0) v = v + T_last * bkg[(bkg_rows == 1 ? 0 : r) * 3 + c];
                else if (white_bkg) v = v + (1.0f - acc_m);
                rgb[r * 3 + c] = v;
            }
        }
    }
    return bad ? -1 : 0;
}

/* ---------------------------------------------------------------------------------------
 * ray_marching backward (what torch autograd produces for the graph above).
 * Inputs: forward inputs + d_rgb (R,3), d_depth (R), d_mask (R) (any may be NULL = zeros).
 * Outputs: d_sigma (R,P) (or d_alpha (R,P) when alpha_in), d_radiance (R,P,3).
 * Dropped last sample (Pe = P-1) receives zero gradient.
 * ------------------------------------------------------------------------------------- */
ORC_API void orc_ray_marching_bwd(const float *sigma, const float *alpha_in, const float *radiance, const float *zvals,
                                  const float *noise, const float *bkg, int64_t bkg_rows, int64_t R, int P,
                                  int add_inf_z, int white_bkg, const float *d_rgb, const float *d_depth,
                                  const float *d_mask, float *d_geo, float *d_radiance) {
    const int Pe = (add_inf_z || alpha_in) ? P : P - 1;
#pragma omp parallel
    {
        float *a_ = (float *)malloc(sizeof(float) * P * 4);
        float *T_ = a_ + P, *dl_ = a_ + 2 * P, *q_ = a_ + 3 * P;
#pragma omp for schedule(static)
        for (int64_t r = 0; r < R; ++r) {
            const float *z = zvals + r * P;
            float T = 1.0f;
            for (int i = 0; i < Pe; ++i) {
                float delta;
                if (i < P - 1) { delta = z[i + 1] - z[i]; if (fabsf(delta) < 1e-5f) delta = 0.0f; }
                else delta = 1e10f;
                float a;
                if (alpha_in) a = alpha_in[r * P + i];
                else {
                    float s = sigma[r * P + i];
                    if (noise) s = s + noise[r * Pe + i];
                    s = s > 0.f ? s : 0.f;
                    a = 1.0f - expf(-s * delta);
                }
                a_[i] = a; T_[i] = T; dl_[i] = delta;
                q_[i] = (1.0f - a) + 1e-10f;
                T = T * q_[i];
            }
            float g_rgb[3] = {d_rgb ? d_rgb[3 * r] : 0.f, d_rgb ? d_rgb[3 * r + 1] : 0.f, d_rgb ? d_rgb[3 * r + 2] : 0.f};
            float g_d = d_depth ? d_depth[r] : 0.f;
            float g_m = d_mask ? d_mask[r] : 0.f;
            int use_bkg = (bkg && bkg_rows > 0 && radiance);
            if (!use_bkg && white_bkg && radiance) g_m = g_m - (g_rgb[0] + g_rgb[1] + g_rgb[2]);
            /* B = d L / d T_last from the background term */
            float B = 0.f;
            if (use_bkg) {
                const float *bk = bkg + (bkg_rows == 1 ? 0 : r) * 3;
                B = g_rgb[0] * bk[0] + g_rgb[1] * bk[1] + g_rgb[2] * bk[2];
            }
            /* suffix = sum_{j>i} w_j g_j  + (T_last * B if i < Pe-1) */
            float suffix = 0.f;
            for (int i = Pe - 1; i >= 0; --i) {
                float gi = g_d * z[i] + g_m;
                if (radiance) for (int c = 0; c < 3; ++c) gi += g_rgb[c] * radiance[(r * P + i) * 3 + c];
                float w = a_[i] * T_[i];
                float dalpha = T_[i] * gi - suffix / q_[i];
                if (d_radiance) for (int c = 0; c < 3; ++c) d_radiance[(r * P + i) * 3 + c] = w * g_rgb[c];
                if (alpha_in) d_geo[r * P + i] = dalpha;
                else {
                    float s = sigma[r * P + i];
                    if (noise) s = s + noise[r * Pe + i];
                    /* d alpha / d sigma = delta * exp(-relu(s) delta) for s > 0 */
                    d_geo[r * P + i] = s > 0.f ? dalpha * dl_[i] * expf(-s * dl_[i]) : 0.f;
                }
                suffix += w * gi;
                if (i == Pe - 1) suffix += T_[Pe - 1] * B;
            }
            if (Pe < P) {
                d_geo[r * P + P - 1] = 0.f;
                if (d_radiance) for (int c = 0; c < 3; ++c) d_radiance[(r * P + P - 1) * 3 + c] = 0.f;
            }
        }
        free(a_);
    }
}

/* ---------------------------------------------------------------------------------------
 * sample_cdf (ray_helper.py:432-473): u given explicitly (R,n) (linspace for det).
 * searchsorted(cdf,u,right=True): first index with cdf[idx] > u.  inds_out (R,n) optional,
 * samples sorted ascending per row when do_sort.
 * ------------------------------------------------------------------------------------- */
static int cmp_float(const void *a, const void *b) {
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

ORC_API void orc_sample_cdf(const float *bins, const float *cdf, const float *u, int64_t R, int n_pts, int n_sample,
                            float eps, int do_sort, float *samples, int64_t *inds_out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        const float *b = bins + r * n_pts, *c = cdf + r * n_pts;
        for (int k = 0; k < n_sample; ++k) {
            float uu = u[r * n_sample + k];
            int lo = 0, hi = n_pts; /* upper bound */
            while (lo < hi) { int mid = (lo + hi) >> 1; if (c[mid] <= uu) lo = mid + 1; else hi = mid; }
            int ind = lo;
            if (inds_out) inds_out[r * n_sample + k] = ind;
            int below = ind - 1 < 0 ? 0 : (ind - 1 > n_pts - 1 ? n_pts - 1 : ind - 1);
            int above = ind > n_pts - 1 ? n_pts - 1 : ind;
            float denom = c[above] - c[below];
            if (denom < eps) denom = 1.0f;
            float t = (uu - c[below]) / denom;
            samples[r * n_sample + k] = b[below] + t * (b[above] - b[below]);
        }
        if (do_sort) qsort(samples + r * n_sample, (size_t)n_sample, sizeof(float), cmp_float);
    }
}

/* sample_pdf (ray_helper.py:410-429): weights (R,n_pts-1) -> cdf (R,n_pts).
 * torch.cumsum on the CPU keeps the running sum of a float tensor in double (at::acc_type<float, false>) and rounds every prefix to
 * float: restated exactly (given torch's normaliser this reproduces golden G2's cdf bit for bit).  torch.sum is a vectorised float
 * sum whose order depends on the host ISA; restated as the double sum rounded to float (within an ulp or two of it). */
ORC_API void orc_weights_to_cdf(const float *weights, int64_t R, int n_w, float eps, float *cdf) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        const float *w = weights + r * n_w;
        double tot = 0.0;
        for (int i = 0; i < n_w; ++i) tot += (double)(w[i] + eps);
        const float sum = (float)tot;
        double acc = 0.0;
        cdf[r * (n_w + 1)] = 0.f;
        for (int i = 0; i < n_w; ++i) {
            const float pdf = (w[i] + eps) / sum;
            acc += (double)pdf;
            cdf[r * (n_w + 1) + i + 1] = (float)acc;
        }
    }
}

/* get_zvals_from_near_far (ray_helper.py:231-264), perturb off.
 * t_vals = torch.linspace(0,1,n) (or n+2 with ends removed); torch's linspace kernel is
 * symmetric: idx < n/2 ? start + step*idx : end - step*(n-1-idx), step=(end-start)/(n-1). */
static inline float linspace01(int idx, int n) {
    if (n == 1) return 0.0f;
    float step = (1.0f - 0.0f) / (float)(n - 1);
    return idx < n / 2 ? 0.0f + step * (float)idx : 1.0f - step * (float)(n - 1 - idx);
}

ORC_API void orc_zvals_from_near_far(const float *near, const float *far, int64_t R, int n_pts, int inclusive,
                                     int inverse_linear, float *zvals) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        for (int i = 0; i < n_pts; ++i) {
            float t = inclusive ? linspace01(i, n_pts) : linspace01(i + 1, n_pts + 2);
            float z;
            if (inverse_linear) {
                float a = 1.0f / (near[r] + 1e-8f) * (1.0f - t);
                float b = 1.0f / (far[r] + 1e-8f) * t;
                z = 1.0f / (a + b);
            } else {
                z = near[r] + (far[r] - near[r]) * t;
            }
            zvals[r * n_pts + i] = z;
        }
    }
}


/* ---------------------------------------------------------------------------------------
 * NeuS interval opacity: sdf_to_alpha (arcnerf/models/neus_model.py:242-265) with sdf_to_cdf = sigmoid(sdf * s)
 * (:221-228).  mid_sdf, mid_slope, alpha (R, P-1); zvals (R, P).  Backward: d alpha -> d mid_sdf, d mid_slope and the
 * scalar d s (sum over all intervals); torch.clip passes the gradient on [0, 1] inclusive.
 * ------------------------------------------------------------------------------------- */
static inline float orc_sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

ORC_API void orc_sdf_to_alpha_fwd(const float *mid_sdf, const float *zvals, const float *mid_slope, float s, int clip,
                                  float *alpha, int64_t R, int P) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r)
        for (int k = 0; k < P - 1; ++k) {
            const int64_t i = r * (P - 1) + k;
            const float dist = zvals[r * P + k + 1] - zvals[r * P + k];
            const float h = mid_slope[i] * dist * 0.5f;
            const float pc = orc_sigmoidf((mid_sdf[i] - h) * s), nc = orc_sigmoidf((mid_sdf[i] + h) * s);
            float a = (pc - nc + 1e-5f) / (pc + 1e-5f);
            if (clip) a = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);
            alpha[i] = a;
        }
}

ORC_API void orc_sdf_to_alpha_bwd(const float *mid_sdf, const float *zvals, const float *mid_slope, float s, int clip,
                                  const float *d_alpha, float *d_sdf, float *d_slope, double *d_s, int64_t R, int P) {
    double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
    for (int64_t r = 0; r < R; ++r)
        for (int k = 0; k < P - 1; ++k) {
            const int64_t i = r * (P - 1) + k;
            const float dist = zvals[r * P + k + 1] - zvals[r * P + k];
            const float h = mid_slope[i] * dist * 0.5f;
            const float prev = mid_sdf[i] - h, next = mid_sdf[i] + h;
            const float pc = orc_sigmoidf(prev * s), nc = orc_sigmoidf(next * s);
            const float B = pc + 1e-5f;
            const float a = (pc - nc + 1e-5f) / B;
            float g = d_alpha[i];
            if (clip && (a < 0.0f || a > 1.0f)) g = 0.0f;
            const float gp = g * (nc / (B * B)) * (pc * (1.0f - pc)); /* d / d (prev * s) */
            const float gn = -g * (1.0f / B) * (nc * (1.0f - nc));    /* d / d (next * s) */
            d_sdf[i] = (gp + gn) * s;
            d_slope[i] = (gn - gp) * s * (dist * 0.5f);
            acc += (double)(gp * prev + gn * next);
        }
    *d_s = acc;
}
