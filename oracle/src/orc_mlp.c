/* ORACLE (test infrastructure only) — dense layers of GeoNet / RadianceNet.  See orc_common.h.
 *
 * One layer at a time; python composes layers, skip concats and the geo/feat split exactly as
 * arcnerf/models/base_modules/geo_rad_model/linear_network_module.py:174-197,318-335 and
 * tcnn_fusedmlp_module.py:81-111,177-192 do.  y = act(x W^T + b), W is (N,K) row-major like
 * torch.nn.Linear.weight.  Accumulation is k-sequential in fp32 (one rounding per product and
 * per add, no FMA) — the same chain an f32 MFMA produces.
 */
#include "orc_common.h"

enum { ORC_ACT_NONE = 0, ORC_ACT_RELU = 1, ORC_ACT_SIGMOID = 2, ORC_ACT_TRUNCEXP = 3, ORC_ACT_SOFTPLUS = 4 };

static inline float act_fwd(float v, int act, float beta) {
    switch (act) {
    case ORC_ACT_RELU: return v > 0.f ? v : 0.f;
    case ORC_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case ORC_ACT_TRUNCEXP: return expf(v); /* arcnerf/ops/trunc_exp.py:14-17 */
    case ORC_ACT_SOFTPLUS: { float bv = beta * v; return bv > 20.f ? v : log1pf(expf(bv)) / beta; } /* torch softplus */
    default: return v;
    }
}

/* derivative wrt pre-activation, given pre-activation v and post-activation y */
static inline float act_bwd(float v, float y, int act, float beta) {
    switch (act) {
    case ORC_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case ORC_ACT_SIGMOID: return y * (1.0f - y);
    case ORC_ACT_TRUNCEXP: { float c = v < -15.f ? -15.f : (v > 15.f ? 15.f : v); return expf(c); } /* trunc_exp.py:19-23 */
    case ORC_ACT_SOFTPLUS: { float bv = beta * v; return bv > 20.f ? 1.f : 1.0f / (1.0f + expf(-bv)); }
    default: return 1.f;
    }
}

/* elementwise activation exports (TruncExp F1 etc.) */
ORC_API void orc_act_fwd(const float *x, float *y, int64_t n, int act, float beta) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = act_fwd(x[i], act, beta);
}

ORC_API void orc_act_bwd(const float *x, const float *y, const float *dy, float *dx, int64_t n, int act, float beta) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) dx[i] = dy[i] * act_bwd(x[i], y[i], act, beta);
}

/* y (S,N) = act(x (S,K) @ W(N,K)^T + b); pre (optional) receives the pre-activation */
ORC_API void orc_linear_fwd(const float *x, const float *W, const float *b, int64_t S, int K, int N, int act, float beta,
                            float *y, float *pre) {
    float *Wt = (float *)malloc(sizeof(float) * (size_t)K * N);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) Wt[k * N + n] = W[n * K + k];
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * N);
#pragma omp for schedule(static)
        for (int64_t s = 0; s < S; ++s) {
            const float *xs = x + s * K;
            for (int n = 0; n < N; ++n) acc[n] = 0.f;
            for (int k = 0; k < K; ++k) {
                float xv = xs[k];
                const float *w = Wt + k * N;
                for (int n = 0; n < N; ++n) { float p = xv * w[n]; acc[n] = acc[n] + p; }
            }
            for (int n = 0; n < N; ++n) {
                float v = b ? acc[n] + b[n] : acc[n];
                if (pre) pre[s * N + n] = v;
                y[s * N + n] = act_fwd(v, act, beta);
            }
        }
        free(acc);
    }
    free(Wt);
}

/* backward of one layer: given x, W, pre, y, dy  ->  dx (S,K), dW (N,K) +=, db (N) += */
ORC_API void orc_linear_bwd(const float *x, const float *W, const float *pre, const float *y, const float *dy, int64_t S,
                            int K, int N, int act, float beta, float *dx, float *dW, float *db) {
    float *dpre = (float *)malloc(sizeof(float) * (size_t)S * N);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < S * N; ++i) dpre[i] = dy[i] * act_bwd(pre ? pre[i] : y[i], y[i], act, beta);
    if (dx) {
#pragma omp parallel for schedule(static)
        for (int64_t s = 0; s < S; ++s) {
            float *d = dx + s * K;
            for (int k = 0; k < K; ++k) d[k] = 0.f;
            for (int n = 0; n < N; ++n) {
                float g = dpre[s * N + n];
                const float *w = W + n * K;
                for (int k = 0; k < K; ++k) { float p = g * w[k]; d[k] = d[k] + p; }
            }
        }
    }
    if (dW) {
        /* sample-sequential accumulation per (n,k); parallel over n */
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n) {
            float *w = dW + n * K;
            for (int64_t s = 0; s < S; ++s) {
                float g = dpre[s * N + n];
                if (g == 0.f) continue;
                const float *xs = x + s * K;
                for (int k = 0; k < K; ++k) { float p = g * xs[k]; w[k] = w[k] + p; }
            }
        }
    }
    if (db) {
        for (int n = 0; n < N; ++n) {
            float a = 0.f;
            for (int64_t s = 0; s < S; ++s) a += dpre[s * N + n];
            db[n] += a;
        }
    }
    free(dpre);
}
