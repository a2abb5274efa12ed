// Fused Adam + EMA over one flat fp32 parameter buffer (gfx950, HBM-streaming: 28 B/param + 8 B/param for the EMA).
// torch.optim.Adam semantics as configured by common/trainer/optimizer.py:6-54 (L2 weight decay folded into the
// gradient, bias-corrected moments, eps added to sqrt(v_hat)); EMA.ema_step of arcnerf/trainer/ema.py:29-43 (JNeRF
// style): new = ((1-d)*p + d*old*(1-d^(n-1))) / (1-d^n), and the average is WRITTEN BACK into the parameter.
// The gradient buffer can be cleared in the same pass (zero_grad), saving one 4 B/param sweep per step.
// Since the average is written back, the shadow copy equals the parameter after every step (ema.py:41-42 stores the same tensor
// in both): a caller that is the only writer of `param` passes ema == param, the kernel then takes the parameter it has just read
// as the old average and writes no second copy (28 B/param in all, bit-identical to keeping the shadow).
#include "common.hpp"
#include "adam.hpp"

namespace arcn {

typedef float f4 __attribute__((ext_vector_type(4)));

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
        f4 p4 = reinterpret_cast<f4 *>(param)[i];
        const f4 g4 = reinterpret_cast<const f4 *>(grad)[i];
        f4 m4 = reinterpret_cast<f4 *>(m)[i];
        f4 v4 = reinterpret_cast<f4 *>(v)[i];
        f4 e4 = {0.f, 0.f, 0.f, 0.f};
        if (a.ema_in_param) e4 = p4;
        else if (ema) e4 = reinterpret_cast<f4 *>(ema)[i];
        const bool avg = ema || a.ema_in_param;
        float p[4] = {p4.x, p4.y, p4.z, p4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
        float vv[4] = {v4.x, v4.y, v4.z, v4.w}, ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            adam1(p[k], g[k], mm[k], vv[k], avg ? &ee[k] : nullptr, a.lr, a.b1, a.b2, a.eps, a.wd, a.ema_decay, a.gscale, a.bc1, a.bc2_sqrt, a.deb_old,
                  a.deb_new);
        reinterpret_cast<f4 *>(param)[i] = f4{p[0], p[1], p[2], p[3]};
        reinterpret_cast<f4 *>(m)[i] = f4{mm[0], mm[1], mm[2], mm[3]};
        reinterpret_cast<f4 *>(v)[i] = f4{vv[0], vv[1], vv[2], vv[3]};
        if (ema) reinterpret_cast<f4 *>(ema)[i] = f4{ee[0], ee[1], ee[2], ee[3]};
        if (a.zero_grad) reinterpret_cast<f4 *>(grad)[i] = f4{0.f, 0.f, 0.f, 0.f};
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

__global__ void __launch_bounds__(256) adam_ema_kernel(float *__restrict__ param, float *__restrict__ grad, float *__restrict__ m,
                                                       float *__restrict__ v, float *__restrict__ ema, int64_t n, AdamArgs a) {
    adam_ema_run(param, grad, m, v, ema, n, blockIdx.x, gridDim.x, a);
}

// up to four runs [lo, lo + n) of the SAME flat buffers in one launch: the first b[0] workgroups take run 0, the next b[1] run 1, ...
struct AdamRuns {
    int64_t lo[4], n[4];
    int b[4];
    int count;
};

__global__ void __launch_bounds__(256) adam_ema_runs_kernel(float *__restrict__ param, float *__restrict__ grad, float *__restrict__ m,
                                                            float *__restrict__ v, float *__restrict__ ema, AdamRuns r, AdamArgs a) {
    int bid = blockIdx.x, k = 0;
    while (k + 1 < r.count && bid >= r.b[k]) { bid -= r.b[k]; ++k; }
    const int64_t lo = r.lo[k];
    adam_ema_run(param + lo, grad + lo, m + lo, v + lo, ema ? ema + lo : nullptr, r.n[k], bid, r.b[k], a);
}

}  // namespace arcn

using namespace arcn;

ARCN_EXPORT int arcn_adam_ema_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, float *ema, int64_t n,
                                   float lr, float beta1, float beta2, float eps, float weight_decay, float ema_decay,
                                   float grad_scale, int step, int ema_step, int zero_grad, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return einval("adam_ema_step: missing/invalid argument");
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(ema)) & 15)
        return einval("adam_ema_step: buffers must be 16-byte aligned");
    if (ema && ema_step < 1) return einval("adam_ema_step: ema_step is 1-based");
    const AdamHyper h = make_adam_hyper(lr, beta1, beta2, eps, weight_decay, ema_decay, grad_scale, step, ema_step, ema != nullptr);
    int64_t blocks = ceil_div<int64_t>((n >> 2) + 1, 256);
    if (blocks > 2048) blocks = 2048;
    const int ema_in_param = ema == param;  // the running average lives in the parameter itself (see the file header)
    const AdamArgs a{lr, beta1, beta2, eps, weight_decay, ema_decay, grad_scale, h.bc1, h.bc2_sqrt, h.deb_old, h.deb_new, zero_grad, ema_in_param};
    hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                       ema_in_param ? static_cast<float *>(nullptr) : ema, n, a);
    return check_launch("adam_ema_step");
}

/* arcn_adam_ema_step on up to four runs [lo, lo + n) of the same flat buffers in ONE launch (runs_host: lo0, n0, lo1, n1, ...; every lo a
 * multiple of 4 floats so that the runs start 16-byte aligned): what is left of the flat parameter buffer when the scatter's chunk owners
 * have applied the optimiser to their table levels (arcn_hashgrid_bwd_lm_adam) - the small levels in front and the MLP weights behind. */
ARCN_EXPORT int arcn_adam_ema_step_runs(float *param, float *grad, float *exp_avg, float *exp_avg_sq, float *ema, const int64_t *runs_host,
                                        int n_runs, float lr, float beta1, float beta2, float eps, float weight_decay, float ema_decay,
                                        float grad_scale, int step, int ema_step, int zero_grad, void *stream) {
    if (n_runs <= 0) return ARCN_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !runs_host || n_runs > 4 || step < 1) return einval("adam_ema_step_runs: missing/invalid argument");
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(ema)) & 15)
        return einval("adam_ema_step_runs: buffers must be 16-byte aligned");
    if (ema && ema_step < 1) return einval("adam_ema_step_runs: ema_step is 1-based");
    AdamRuns r{};
    int total = 0;
    for (int k = 0; k < n_runs; ++k) {
        const int64_t lo = runs_host[2 * k], n = runs_host[2 * k + 1];
        if (lo < 0 || n < 0 || (lo & 3)) return einval("adam_ema_step_runs: a run must start at a multiple of 4 floats");
        if (n == 0) continue;
        int64_t blocks = ceil_div<int64_t>((n >> 2) + 1, 256);
        if (blocks > 1024) blocks = 1024;
        r.lo[r.count] = lo; r.n[r.count] = n; r.b[r.count] = (int)blocks;
        total += (int)blocks;
        ++r.count;
    }
    if (r.count == 0) return ARCN_OK;
    const AdamHyper h = make_adam_hyper(lr, beta1, beta2, eps, weight_decay, ema_decay, grad_scale, step, ema_step, ema != nullptr);
    const int ema_in_param = ema == param;
    const AdamArgs a{lr, beta1, beta2, eps, weight_decay, ema_decay, grad_scale, h.bc1, h.bc2_sqrt, h.deb_old, h.deb_new, zero_grad, ema_in_param};
    hipLaunchKernelGGL(adam_ema_runs_kernel, dim3((unsigned)total), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                       ema_in_param ? static_cast<float *>(nullptr) : ema, r, a);
    return check_launch("adam_ema_step_runs");
}
