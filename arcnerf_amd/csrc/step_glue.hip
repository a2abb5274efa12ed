// The small passes BETWEEN the kernels of the NeuS-on-hash-grid + MultiVol training step (trainer.FusedNeusNgpStep; the model block of the
// reference's capture_qqtiger_neusngp_multivol.yaml).  As torch expressions they were ~55 launches per step - pads, column copies,
// concatenations, products with a weight row, blends, sums - 12 % of the step's kernel time for arithmetic on a few floats per sample
// (profiles/r5_neus_ngp_multivol_kernel_stats.csv).  Each kernel here is one of those groups in one pass:
//   neus_step_prep     per-step derived weights of both geometry nets (padded last layers, the Jacobian row's folded weight, beta W2[0],
//                      the NeuS scale exp(inv_s * speed))                                           (sdf_model.py:42-101, neus_model.py:221-228)
//   geo_out_grad       the gradient of a geometry net's padded output [col 0 through its activation | features | 0]   (linear_network_module.py)
//   neus_blend_loss    foreground + T_last * background, the image loss and the gradients of the three operands       (full_model.py:278-330,
//                      loss/img_loss.py:60-100)
//   sdf_jac_dz2        arcn_sdf_jac_dz with the two by-products the weight gradients need instead of `s u`            (base_network.py:30-44)
//   sum_scale_add      dst[0] += factor * scale[0] * sum(src): the gradient of inv_s from the per-ray gradients of the scale
#include "common.hpp"

namespace arcn {

__global__ void __launch_bounds__(256)
neus_step_prep_kernel(const float *__restrict__ w1, const float *__restrict__ l1w, int H, int E, int n_out, int n_pad, float beta,
                      const float *__restrict__ inv_s, float speed, float *__restrict__ w2p, float *__restrict__ w1j, float *__restrict__ bw20,
                      float *__restrict__ scale_out, const float *__restrict__ bl1w, int Hb, int nb_out, int nb_pad, float *__restrict__ wb1p) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int i = tid; i < n_pad * H; i += stride) w2p[i] = i < n_out * H ? l1w[i] : 0.f;
    for (int i = tid; i < H * E; i += stride) w1j[i] = w1[i] * l1w[i / E];          // row 0 of the last layer: l1w[j], j = i / E
    for (int i = tid; i < H; i += stride) bw20[i] = beta * l1w[i];
    if (tid == 0 && scale_out) scale_out[0] = expf(inv_s[0] * speed);
    if (bl1w)
        for (int i = tid; i < nb_pad * Hb; i += stride) wb1p[i] = i < nb_out * Hb ? bl1w[i] : 0.f;
}

// g_out (n, n_pad): column 0 = d_col0 * act'(x = out[i * ld_out]) (act 0: d_col0 itself), columns 1 .. n_feat = d_feat[i * ld_feat + c],
// the padding columns 0.  One thread per element: consecutive lanes on consecutive words of g_out.
__global__ void __launch_bounds__(256)
geo_out_grad_kernel(const float *__restrict__ d_col0, const float *__restrict__ out, int64_t ld_out, const float *__restrict__ y_col0, int act,
                    float beta, const float *__restrict__ d_feat, int64_t ld_feat, int n_feat, int n_pad, float *__restrict__ g_out, int64_t n) {
    const int64_t total = n * n_pad;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / n_pad;
        const int c = (int)(e - i * n_pad);
        float v = 0.f;
        if (c == 0) {
            v = d_col0[i];
            if (act != 0) {
                const float x = out[i * ld_out];
                v *= act_grad(x, y_col0 ? y_col0[i] : act_fwd(x, act, beta), act, beta);
            }
        } else if (c <= n_feat) {
            v = d_feat[i * ld_feat + (c - 1)];
        }
        g_out[e] = v;
    }
}

// rgb = rgb_f + T rgb_b, depth = depth_f + T depth_b, the image loss (Huber: delta > 0, the arithmetic of huber_kernel; MSE: delta <= 0) as a
// plain mean over the 3 R elements times weight, d rgb, d T = sum_c d rgb_c rgb_b_c, d rgb_b = T d rgb.  One thread per ray; the loss: every
// workgroup leaves its partial in ws, the LAST one to arrive (ticket in ws[0], which it resets for the next launch) adds the partials in index
// order - no float atomics, no cleared buffer, the same bits every run.  (One workgroup for the whole batch was 42 us for 4096 rays: four
// dependent trips of loads on one CU.)
constexpr int kBlendThreads = 256;
constexpr int kBlendMaxBlocks = 256;

__global__ void __launch_bounds__(kBlendThreads)
neus_blend_loss_kernel(const float *__restrict__ rgb_f, const float *__restrict__ depth_f, const float *__restrict__ t_last,
                       const float *__restrict__ rgb_b, const float *__restrict__ depth_b, const float *__restrict__ target, int64_t R, float delta,
                       float weight, float *__restrict__ rgb, float *__restrict__ depth, float *__restrict__ d_rgb, float *__restrict__ d_tlast,
                       float *__restrict__ d_rgb_b, float *__restrict__ loss, uint32_t *__restrict__ ws) {
    __shared__ float s_part[kBlendThreads / 64];
    __shared__ bool s_last;
    const float scale = weight / (float)(R * 3);
    float acc = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        const float T = t_last[r];
        const float dfv = depth_f[r], dbv = depth_b[r];
        float dt = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float b = rgb_b[3 * r + c];
            const float v = rgb_f[3 * r + c] + T * b;
            rgb[3 * r + c] = v;
            const float d = v - target[3 * r + c];
            float g;
            if (delta > 0.f) {
                const float ad = fabsf(d);
                const bool quad = ad < delta;
                acc += quad ? (0.5f / delta) * ad * ad : ad - 0.5f * delta;
                g = (quad ? d / delta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f))) * scale;
            } else {
                acc += d * d;
                g = 2.0f * d * scale;
            }
            d_rgb[3 * r + c] = g;
            d_rgb_b[3 * r + c] = g * T;
            dt += g * b;
        }
        d_tlast[r] = dt;
        depth[r] = dfv + T * dbv;
    }
    acc = wave_sum(acc);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    float *partials = reinterpret_cast<float *>(ws + 1);
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < kBlendThreads / 64; ++k) t += s_part[k];
        __hip_atomic_store(partials + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(ws, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        float t = 0.f;
        for (unsigned k = 0; k < gridDim.x; ++k) t += __hip_atomic_load(partials + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        loss[0] = t * scale;
        loss[1] = 0.f;      // the accumulator of the loss pass that follows (arcn_eikonal_packed, accumulate bit 1)
        ws[0] = 0u;         // the next launch's ticket
    }
}

typedef float g4v __attribute__((ext_vector_type(4)));

// dz = dh s + c u s (1 - s) (over dh), sw = s * w (over u: the operand of the first layer's Jacobian-path weight gradient, (s w)^T d_jac),
// colsum[j] += sum_i s u: the Jacobian-path gradient of the last layer's first row.  A workgroup of 256 threads walks whole rows (H / 4
// threads per row, 1024 / H rows per trip), so every thread stays in ITS four columns: the column sums are four registers per thread,
// folded through LDS once at the end - one float atomic per column and workgroup.
__global__ void __launch_bounds__(256)
sdf_jac_dz2_kernel(const float *dh, const float *u, const float *__restrict__ s, const float *__restrict__ c,
                   const float *__restrict__ w, float *dz, float *sw, float *__restrict__ colsum, int64_t n, int H) {
    // (dh / dz and u / sw carry no __restrict__: the in-place form passes dz = dh and sw = u - every thread loads its own elements of a trip
    // before it stores them)
    __shared__ float s_sum[256 * 4];
    const int tpr = H >> 2;                       // threads per row (H <= 1024, a multiple of 4; 256 % tpr == 0 checked by the launcher)
    const int rows_per_trip = 256 / tpr;
    const int col = (threadIdx.x % tpr) * 4, sub = threadIdx.x / tpr;
    const g4v cv = *reinterpret_cast<const g4v *>(c + col), wv = *reinterpret_cast<const g4v *>(w + col);
    g4v sum = {0.f, 0.f, 0.f, 0.f};
    // four rows per thread and trip, their twelve loads issued before the first use (one row per trip ran at memory latency: 42 us for
    // 125 K rows of 64, against 25 us for the plain elementwise pass)
    const int64_t step = (int64_t)gridDim.x * rows_per_trip;
    for (int64_t row0 = (int64_t)blockIdx.x * rows_per_trip + sub; row0 < n; row0 += 4 * step) {
        g4v a[4], b[4], sv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = row0 + q * step;
            if (row < n) {
                const int64_t i = row * H + col;
                a[q] = *reinterpret_cast<const g4v *>(dh + i);
                b[q] = *reinterpret_cast<const g4v *>(u + i);
                sv[q] = *reinterpret_cast<const g4v *>(s + i);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = row0 + q * step;
            if (row < n) {
                const int64_t i = row * H + col;
                g4v o1, o2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = sv[q][k] * b[q][k];
                    sum[k] += t;
                    o1[k] = a[q][k] * sv[q][k] + cv[k] * t * (1.0f - sv[q][k]);
                    o2[k] = sv[q][k] * wv[k];
                }
                *reinterpret_cast<g4v *>(dz + i) = o1;
                *reinterpret_cast<g4v *>(sw + i) = o2;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s_sum[threadIdx.x * 4 + k] = sum[k];
    __syncthreads();
    if ((int)threadIdx.x < H) {
        // column j lives in thread (j / 4) + q * tpr, slot j % 4, q = 0 .. rows_per_trip - 1
        const int j = threadIdx.x;
        float t = 0.f;
        for (int q = 0; q < rows_per_trip; ++q) t += s_sum[((j >> 2) + q * tpr) * 4 + (j & 3)];
        if (t != 0.f) atomicAdd(colsum + j, t);
    }
    for (int j = 256 + threadIdx.x; j < H; j += 256) {   // H > 256: tpr > 64, rows_per_trip <= 3
        float t = 0.f;
        for (int q = 0; q < rows_per_trip; ++q) t += s_sum[((j >> 2) + q * tpr) * 4 + (j & 3)];
        if (t != 0.f) atomicAdd(colsum + j, t);
    }
}

__global__ void __launch_bounds__(1024) sum_scale_add_kernel(const float *__restrict__ src, int64_t n, const float *__restrict__ scale, float factor,
                                                             float *__restrict__ dst) {
    __shared__ float s_part[16];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += src[i];
    acc = wave_sum(acc);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += s_part[k];
        dst[0] += t * (scale ? scale[0] : 1.0f) * factor;
    }
}

static inline unsigned glue_grid(int64_t n) {
    int64_t b = ceil_div<int64_t>(n, 256);
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace arcn

using namespace arcn;

ARCN_EXPORT int arcn_neus_step_prep(const float *w1, const float *l1w, int H, int E, int n_out, int n_pad, float beta, const float *inv_s,
                                    float speed, float *w2p, float *w1j, float *bw20, float *scale_out, const float *bkg_l1w, int Hb, int nb_out,
                                    int nb_pad, float *wb1p, void *stream) {
    if (!w1 || !l1w || !w2p || !w1j || !bw20 || H < 1 || E < 1 || n_out < 1 || n_pad < n_out) return einval("neus_step_prep: missing argument");
    if (scale_out && !inv_s) return einval("neus_step_prep: the scale needs inv_s");
    if (bkg_l1w && (!wb1p || Hb < 1 || nb_out < 1 || nb_pad < nb_out)) return einval("neus_step_prep: background last layer incomplete");
    hipLaunchKernelGGL(neus_step_prep_kernel, dim3(8), dim3(256), 0, as_stream(stream), w1, l1w, H, E, n_out, n_pad, beta, inv_s, speed, w2p, w1j,
                       bw20, scale_out, bkg_l1w, Hb, nb_out, nb_pad, wb1p);
    return check_launch("neus_step_prep");
}

ARCN_EXPORT int arcn_geo_out_grad(const float *d_col0, const float *out, int64_t ld_out, const float *y_col0, int act, float beta,
                                  const float *d_feat, int64_t ld_feat, int n_feat, int n_pad, float *g_out, int64_t n, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!d_col0 || !g_out || n_feat < 0 || n_pad < 1 + n_feat || (n_feat > 0 && (!d_feat || ld_feat < n_feat)) || (act != 0 && (!out || ld_out < 1)))
        return einval("geo_out_grad: missing argument");
    hipLaunchKernelGGL(geo_out_grad_kernel, dim3(glue_grid(n * n_pad)), dim3(256), 0, as_stream(stream), d_col0, out, ld_out, y_col0, act, beta, d_feat,
                       ld_feat, n_feat, n_pad, g_out, n);
    return check_launch("geo_out_grad");
}

ARCN_EXPORT int64_t arcn_neus_blend_loss_workspace_words(void) { return 1 + kBlendMaxBlocks; }

ARCN_EXPORT int arcn_neus_blend_loss(const float *rgb_f, const float *depth_f, const float *t_last, const float *rgb_b, const float *depth_b,
                                     const float *target, int64_t n_rays, float huber_delta, float weight, float *rgb, float *depth, float *d_rgb,
                                     float *d_tlast, float *d_rgb_b, float *loss, uint32_t *workspace, void *stream) {
    if (!loss) return einval("neus_blend_loss: loss missing");
    if (n_rays <= 0) {
        if (hipMemsetAsync(loss, 0, 2 * sizeof(float), as_stream(stream)) != hipSuccess) return check_launch("memset");
        return ARCN_OK;
    }
    if (!rgb_f || !depth_f || !t_last || !rgb_b || !depth_b || !target || !rgb || !depth || !d_rgb || !d_tlast || !d_rgb_b || !workspace)
        return einval("neus_blend_loss: missing argument");
    int64_t blocks = ceil_div<int64_t>(n_rays, kBlendThreads);
    if (blocks > kBlendMaxBlocks) blocks = kBlendMaxBlocks;
    hipLaunchKernelGGL(neus_blend_loss_kernel, dim3((unsigned)blocks), dim3(kBlendThreads), 0, as_stream(stream), rgb_f, depth_f, t_last, rgb_b, depth_b,
                       target, n_rays, huber_delta, weight, rgb, depth, d_rgb, d_tlast, d_rgb_b, loss, workspace);
    return check_launch("neus_blend_loss");
}

ARCN_EXPORT int arcn_sdf_jac_dz2(const float *dh, const float *u, const float *s, const float *c, const float *w, float *dz, float *sw, float *colsum,
                                 int64_t n, int H, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!dh || !u || !s || !c || !w || !dz || !sw || !colsum || H < 4 || (H & 3) != 0 || H > 1024 || 256 % (H >> 2) != 0)
        return einval("sdf_jac_dz2: missing argument or hidden width not 4 * (a divisor of 256)");
    if (((uintptr_t)dh | (uintptr_t)u | (uintptr_t)s | (uintptr_t)c | (uintptr_t)w | (uintptr_t)dz | (uintptr_t)sw) & 15u)
        return einval("sdf_jac_dz2: 16-byte aligned tensors");
    const int rows_per_trip = 256 / (H >> 2);
    int64_t blocks = ceil_div<int64_t>(n, rows_per_trip * 8);        // >= 8 rows per thread: the column atomics stay few
    if (blocks > 256) blocks = 256;      // (one same-address float atomic per column and workgroup: 976 workgroups queued ~13 us of them behind a 25 us pass)
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sdf_jac_dz2_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dh, u, s, c, w, dz, sw, colsum, n, H);
    return check_launch("sdf_jac_dz2");
}

ARCN_EXPORT int arcn_sum_scale_add(const float *src, int64_t n, const float *scale_dev, float factor, float *dst, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!src || !dst) return einval("sum_scale_add: missing argument");
    hipLaunchKernelGGL(sum_scale_add_kernel, dim3(1), dim3(1024), 0, as_stream(stream), src, n, scale_dev, factor, dst);
    return check_launch("sum_scale_add");
}
