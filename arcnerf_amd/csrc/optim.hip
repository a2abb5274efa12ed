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

__global__ void __launch_bounds__(256) adam_ema_kernel(float *__restrict__ param, float *__restrict__ grad, float *__restrict__ m,
                                                       float *__restrict__ v, float *__restrict__ ema, int64_t n, AdamArgs a) {
    adam_ema_run(param, grad, m, v, ema, n, blockIdx.x, gridDim.x, a);
}

__global__ void __launch_bounds__(256) adam_ema_runs_kernel(float *__restrict__ param, float *__restrict__ grad, float *__restrict__ m,
                                                            float *__restrict__ v, float *__restrict__ ema, AdamRuns r, AdamArgs a) {
    int bid = blockIdx.x, k = 0;
    while (k + 1 < r.count && bid >= r.b[k]) { bid -= r.b[k]; ++k; }
    const int64_t lo = r.lo[k];
    adam_ema_run(param + lo, grad + lo, m + lo, v + lo, ema ? ema + lo : nullptr, r.n[k], bid, r.b[k], a);
}

__global__ void __launch_bounds__(256) copy_words_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace arcn

using namespace arcn;

/* dst[0..n_words) = src[0..n_words) as a KERNEL (32-bit words; src / dst: device memory or pinned host memory, which the device addresses
 * directly).  For the small per-step copies of an asynchronous training step (the sample total device -> pinned host, trainer.FusedNgpStep):
 * issued as hipMemcpyAsync between kernel launches they made a step take 4.9 ms instead of 0.75 once the host ran a few dozen operations
 * ahead (the runtime's copy path needs the host's attention; measured in round 4), as kernels they cost 2 us each. */
ARCN_EXPORT int arcn_copy_words(const void *src, void *dst, int64_t n_words, void *stream) {
    if (n_words <= 0) return ARCN_OK;
    if (!src || !dst || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3)) return einval("copy_words: missing or misaligned argument");
    int64_t blocks = ceil_div<int64_t>(n_words, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), static_cast<const uint32_t *>(src),
                       static_cast<uint32_t *>(dst), n_words);
    return check_launch("copy_words");
}

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
