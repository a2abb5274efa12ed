// The per-element Adam (+ EMA write-back) update shared by adam_ema_kernel (optim.hip) and the scatter consumer that applies the
// optimiser to the table rows it owns (hashgrid.hip).  torch.optim.Adam as configured by common/trainer/optimizer.py:6-54 (L2 weight
// decay folded into the gradient, bias-corrected moments, eps added to sqrt(v_hat)); EMA.ema_step of arcnerf/trainer/ema.py:29-43:
// new = ((1-d) p + d old (1 - d^(n-1))) / (1 - d^n), written back into the parameter.
#pragma once

namespace arcn {

struct AdamHyper {
    float lr, b1, b2, eps, wd, ema_decay, gscale, bc1, bc2_sqrt, deb_old, deb_new;
};

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float *e, float lr, float b1, float b2, float eps,
                                      float wd, float ema_decay, float gscale, float bc1, float bc2_sqrt, float deb_old,
                                      float deb_new) {
    g = g * gscale;
    if (wd != 0.f) g = g + wd * p;
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - (lr / bc1) * (m / denom);
    if (e) {
        const float avg = ((1.0f - ema_decay) * p + ema_decay * (*e) * deb_old) * deb_new;
        *e = avg;
        p = avg;
    }
}

// bias corrections in double like torch (1 - beta**step), passed as fp32
inline AdamHyper make_adam_hyper(float lr, float beta1, float beta2, float eps, float wd, float ema_decay, float gscale, int step,
                                 int ema_step, bool ema) {
    AdamHyper h;
    h.lr = lr; h.b1 = beta1; h.b2 = beta2; h.eps = eps; h.wd = wd; h.ema_decay = ema_decay; h.gscale = gscale;
    h.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    h.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    const double d = (double)ema_decay;
    h.deb_old = ema ? (float)(1.0 - pow(d, (double)(ema_step - 1))) : 0.f;
    h.deb_new = ema ? (float)(1.0 / (1.0 - pow(d, (double)ema_step))) : 0.f;
    return h;
}


#ifdef __HIPCC__
typedef float adam_f4 __attribute__((ext_vector_type(4)));

struct AdamArgs {
    float lr, b1, b2, eps, wd, ema_decay, gscale, bc1, bc2_sqrt, deb_old, deb_new;
    int zero_grad, ema_in_param;
};

// workgroup `bid` of `nblocks` over one contiguous run of n parameters
__device__ __forceinline__ void adam_ema_run(float *__restrict__ param, float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v,
                                             float *__restrict__ ema, int64_t n, int bid, int nblocks, const AdamArgs &a) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)nblocks * blockDim.x;
    for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < n4; i += stride) {
        adam_f4 p4 = reinterpret_cast<adam_f4 *>(param)[i];
        const adam_f4 g4 = reinterpret_cast<const adam_f4 *>(grad)[i];
        adam_f4 m4 = reinterpret_cast<adam_f4 *>(m)[i];
        adam_f4 v4 = reinterpret_cast<adam_f4 *>(v)[i];
        adam_f4 e4 = {0.f, 0.f, 0.f, 0.f};
        if (a.ema_in_param) e4 = p4;
        else if (ema) e4 = reinterpret_cast<adam_f4 *>(ema)[i];
        const bool avg = ema || a.ema_in_param;
        float p[4] = {p4.x, p4.y, p4.z, p4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
        float vv[4] = {v4.x, v4.y, v4.z, v4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            adam1(p[k], g[k], mm[k], vv[k], avg ? &ee[k] : nullptr, a.lr, a.b1, a.b2, a.eps, a.wd, a.ema_decay, a.gscale, a.bc1, a.bc2_sqrt, a.deb_old,
                  a.deb_new);
        reinterpret_cast<adam_f4 *>(param)[i] = adam_f4{p[0], p[1], p[2], p[3]};
        reinterpret_cast<adam_f4 *>(m)[i] = adam_f4{mm[0], mm[1], mm[2], mm[3]};
        reinterpret_cast<adam_f4 *>(v)[i] = adam_f4{vv[0], vv[1], vv[2], vv[3]};
        if (ema) reinterpret_cast<adam_f4 *>(ema)[i] = adam_f4{ee[0], ee[1], ee[2], ee[3]};
        if (a.zero_grad) reinterpret_cast<adam_f4 *>(grad)[i] = adam_f4{0.f, 0.f, 0.f, 0.f};
    }
    // tail
    const int64_t t = (n4 << 2) + (int64_t)bid * blockDim.x + threadIdx.x;
    if (t < n) {
        float old = param[t];
        adam1(param[t], grad[t], m[t], v[t], a.ema_in_param ? &old : (ema ? &ema[t] : nullptr), a.lr, a.b1, a.b2, a.eps, a.wd, a.ema_decay, a.gscale,
              a.bc1, a.bc2_sqrt, a.deb_old, a.deb_new);
        if (a.zero_grad) grad[t] = 0.f;
    }
}


// up to four runs [lo, lo + n) of the SAME flat buffers in one launch: the first b[0] workgroups take run 0, the next b[1] run 1, ...
struct AdamRuns {
    int64_t lo[4], n[4];
    int b[4];
    int count;
};

#endif

}  // namespace arcn
