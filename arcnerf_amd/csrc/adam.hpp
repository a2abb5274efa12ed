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

}  // namespace arcn
