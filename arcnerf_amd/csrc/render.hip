// Alpha compositing (transmittance scan + weighted sums) and inverse-CDF resampling for gfx950.
//
// Replaces arcnerf/render/ray_helper.py:476-620 (ray_marching, alpha_to_weights) and :432-473 (sample_cdf).
// One 64-lane wavefront owns one ray: samples are consumed 64 at a time, the transmittance
// T_i = prod_{k<i}(1 - alpha_k + 1e-10) is a wave-level prefix product carried across chunks, the per-ray sums are
// wave reductions.  The backward needs TRUE suffix sums (a total-minus-prefix form cancels catastrophically when
// some q_k ~ 1e-10), so it walks the chunks in reverse using per-chunk carries kept in LDS.
//
// HBM traffic (algorithmic): fwd 20 B/sample in (sigma, rgb, z) + 20 B/ray out; bwd +16 B/sample out.
#include "common.hpp"

namespace arcn {

constexpr int kRaysPerBlock = 4;  // 4 waves / 256 threads
constexpr int kMaxChunks = 64;    // <= 4096 visited columns per ray

// ---- sample views ------------------------------------------------------------------------------
// A view exposes, for ray r, ncol(r) visited columns; column k maps to a stored sample and a delta rule.

// The reference's dense (R,P) tensors.
struct DenseView {
    const float *sigma, *alpha_in, *radiance, *zvals, *noise;
    int P, Pe;
    __device__ void patch(const int32_t *) {}
    __device__ int ncol(int64_t) const { return Pe; }
    __device__ bool final_visited(int64_t) const { return true; }  // column Pe-1 is always visited
    __device__ bool real(int64_t, int) const { return true; }
    __device__ int64_t sidx(int64_t r, int k) const { return r * P + k; }
    __device__ int64_t nidx(int64_t r, int k) const { return r * Pe + k; }
    __device__ float delta(int64_t r, int k, bool &neg) const {
        if (k < P - 1) {
            float d = zvals[r * P + k + 1] - zvals[r * P + k];
            if (fabsf(d) < 1e-5f) d = 0.0f;
            if (d < 0.f) neg = true;
            return d;
        }
        return 1e10f;
    }
    __device__ float z(int64_t r, int k) const { return zvals[r * P + k]; }
    __device__ void zero_dropped(int64_t, float *, float *) const {}  // the host wrapper clears the dropped last column
    __device__ void begin_ray(int64_t) {}
    __device__ void zero_truncated(int, float *, float *) const {}
};

// Packed (offsets, t) form reproducing the reference's padded dense (R, P_dense) tensors: valid samples first, the
// tail duplicates the last valid sample (fg_model.py:311-316) so its deltas are 0 and it carries no weight.  Only the
// columns that can carry weight are visited: 0..min(n,Pe)-1 and, with add_inf_z and n < P_dense, the final 1e10
// column (a virtual column aliasing sample n-1).
struct PackedView {
    const float *sigma, *alpha_in, *radiance, *zvals, *noise;
    const int32_t *offsets;
    int P_dense, Pe, add_inf_z;
    __device__ void patch(const int32_t *p_dense_ptr) {
        if (p_dense_ptr) {
            int pd = *p_dense_ptr;
            P_dense = pd < 2 ? 2 : pd;
            Pe = add_inf_z ? P_dense : P_dense - 1;
        }
    }
    // per-ray segment, loaded ONCE by begin_ray (the accessors below are called per sample; re-reading offsets[] there costs a
    // dependent global load each time because the compiler cannot prove the kernel's stores do not alias it)
    int32_t seg_off = 0, seg_n = 0;
    // counts (optional): the samples the marcher emitted per ray.  The packed buffers hold min(total, capacity) samples, the offsets are
    // clamped to the capacity: a ray behind the point where they filled up has a TRUNCATED segment.  Such a ray is never rendered from
    // its partial sample set: it is treated as a ray without samples (background colour, no gradient; the samples it left in the
    // buffers receive zero gradients).  The rays that fit keep their 1 / (3 R) weight of the FULL batch and the step's reported loss still
    // carries the left-out rays' background-vs-target term: an overflowed step is a down-weighted step of the rays that fit, not the step of a
    // smaller batch (the steppers warn and grow the buffers, trainer/fused_step.py).
    const int32_t *counts = nullptr;
    int32_t trunc_n = 0;
    __device__ void begin_ray(int64_t r) {
        seg_off = offsets[r];
        seg_n = offsets[r + 1] - seg_off;
        trunc_n = 0;
        if (counts && seg_n < counts[r]) { trunc_n = seg_n; seg_n = 0; }
    }
    __device__ void zero_truncated(int lane, float *d_geo, float *d_radiance) const {
        for (int k = lane; k < trunc_n; k += 64) {
            const int64_t si = (int64_t)seg_off + k;
            if (d_geo) d_geo[si] = 0.f;
            if (d_radiance) { d_radiance[si * 3] = 0.f; d_radiance[si * 3 + 1] = 0.f; d_radiance[si * 3 + 2] = 0.f; }
        }
    }
    __device__ int count(int64_t) const { return seg_n; }
    __device__ int ncol(int64_t r) const {
        int n = count(r);
        if (n <= 0) return 0;
        int ne = n < Pe ? n : Pe;
        return ne + ((add_inf_z && n < P_dense) ? 1 : 0);
    }
    __device__ bool final_visited(int64_t r) const { return add_inf_z ? true : (count(r) >= Pe); }
    __device__ bool real(int64_t r, int k) const { return k < count(r); }
    __device__ int64_t sidx(int64_t r, int k) const {
        int n = count(r);
        return (int64_t)seg_off + (k < n ? k : n - 1);
    }
    __device__ int64_t nidx(int64_t r, int k) const { return sidx(r, k); }
    __device__ float delta(int64_t r, int k, bool &neg) const {
        int n = count(r);
        if (k >= n) return 1e10f;  // virtual final column
        if (k == n - 1) return (add_inf_z && n == P_dense) ? 1e10f : 0.0f;
        int64_t b = (int64_t)seg_off + k;
        float d = zvals[b + 1] - zvals[b];
        if (fabsf(d) < 1e-5f) d = 0.0f;
        if (d < 0.f) neg = true;
        return d;
    }
    __device__ float z(int64_t r, int k) const { return zvals[sidx(r, k)]; }
    // without the inf column the last of P_dense samples is the dropped column: no visited column covers its gradient
    __device__ void zero_dropped(int64_t r, float *d_geo, float *d_radiance) const {
        const int n = count(r);
        if (!add_inf_z && n >= P_dense && n > 0) {
            const int64_t si = (int64_t)seg_off + n - 1;
            d_geo[si] = 0.f;
            if (d_radiance) { d_radiance[si * 3] = 0.f; d_radiance[si * 3 + 1] = 0.f; d_radiance[si * 3 + 2] = 0.f; }
        }
    }
};

template <typename View>
__device__ __forceinline__ float alpha_at(const View &v, int64_t r, int k, float delta, float *sig_eff) {
    if (v.alpha_in) {
        *sig_eff = 0.f;
        return v.alpha_in[v.sidx(r, k)];
    }
    float s = v.sigma[v.sidx(r, k)];
    if (v.noise) s = s + v.noise[v.nidx(r, k)];
    *sig_eff = s;
    float sr = s > 0.f ? s : 0.f;
    return 1.0f - expf(-sr * delta);
}

// ---- per-ray bodies (one wave = one ray), shared by the forward, backward and fused training kernels --------------------------
struct RayOut {
    float rgb[3], depth, mask, carry, t_last;
};

// forward walk of ray r: transmittance scan, weighted sums.  chunk_carry (optional, LDS row of this wave) receives the
// transmittance at every chunk start - exactly what the backward walk needs, so the fused kernel skips its first pass.
template <typename View>
__device__ __forceinline__ RayOut composite_fwd_ray(View &v, int64_t r, int lane, int nc, const float *__restrict__ bkg,
                                                    int64_t bkg_rows, int white_bkg, float *__restrict__ alpha_out,
                                                    float *__restrict__ trans_out, float *__restrict__ weights_out,
                                                    float *chunk_carry, bool &neg) {
    float carry = 1.0f;
    float acc_d = 0.f, acc_m = 0.f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, t_last = 0.f;
    for (int base = 0; base < nc; base += 64) {
        if (chunk_carry && lane == 0) chunk_carry[base >> 6] = carry;
        const int k = base + lane;
        const bool on = k < nc;
        float a = 0.f, q = 1.f, zz = 0.f, se;
        if (on) {
            float dl = v.delta(r, k, neg);
            a = alpha_at(v, r, k, dl, &se);
            q = (1.0f - a) + 1e-10f;
            zz = v.z(r, k);
        }
        float incl = wave_incl_prod(q);
        float excl = wave_from_below(incl, 1.0f);
        float T = carry * excl;
        float w = a * T;
        if (on) {
            if (v.real(r, k)) {
                if (alpha_out) alpha_out[v.nidx(r, k)] = a;
                if (trans_out) trans_out[v.nidx(r, k)] = T;
                if (weights_out) weights_out[v.nidx(r, k)] = w;
            }
            acc_d += w * zz;
            acc_m += w;
            if (v.radiance) {
                const float *c = v.radiance + v.sidx(r, k) * 3;
                acc_r += w * c[0];
                acc_g += w * c[1];
                acc_b += w * c[2];
            }
            if (k == nc - 1) t_last = T;
        }
        carry = carry * lane_value<63>(incl);
    }
    RayOut o;
    o.depth = wave_sum(acc_d);
    o.mask = wave_sum(acc_m);
    acc_r = wave_sum(acc_r);
    acc_g = wave_sum(acc_g);
    acc_b = wave_sum(acc_b);
    t_last = wave_sum(t_last);
    // trans_shift[:, -1]: T at column Pe-1.  If that column is not among the visited ones every column after the
    // visited ones has q == 1 (in fp32), so it equals the running product.
    if (!(nc > 0 && v.final_visited(r))) t_last = carry;
    o.carry = carry;
    o.t_last = t_last;
    const float sums[3] = {acc_r, acc_g, acc_b};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float x = sums[c];
        if (bkg && bkg_rows > 0) x = x + t_last * bkg[(bkg_rows == 1 ? 0 : r) * 3 + c];
        else if (white_bkg) x = x + (1.0f - o.mask);
        o.rgb[c] = x;
    }
    return o;
}

template <typename View>
__global__ void __launch_bounds__(256)
composite_fwd_kernel(View v, const int32_t *p_dense_ptr, const float *__restrict__ bkg, int64_t bkg_rows, int64_t R,
                     int white_bkg, float *__restrict__ rgb, float *__restrict__ depth, float *__restrict__ mask,
                     float *__restrict__ alpha_out, float *__restrict__ trans_out, float *__restrict__ weights_out,
                     int32_t *status) {
    v.patch(p_dense_ptr);
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    v.begin_ray(r);
    bool neg = false;
    const RayOut o = composite_fwd_ray(v, r, lane, v.ncol(r), bkg, bkg_rows, white_bkg, alpha_out, trans_out, weights_out,
                                       (float *)nullptr, neg);
    if (neg && status && lane == 0) atomicOr(status, 1);
    if (lane == 0) {
        if (depth) depth[r] = o.depth;
        if (mask) mask[r] = o.mask;
        if (rgb && v.radiance) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[r * 3 + c] = o.rgb[c];
        }
    }
}

// reverse walk of ray r given the upstream gradients (g0,g1,g2) on rgb, gd on depth, gm on mask, gt on T_last, the
// transmittance at every chunk start (chunk_carry) and the product over all visited columns (carry).
// suffix_i = sum_{j>i} w_j g_j + (T_last*B if T_last depends on alpha_i); TRUE suffix sums (see the file header).
// d_geo / d_radiance are indexed like sigma / radiance.  Entries no visited column covers are zeroed by the host wrapper
// (dense: dropped last column) or by the ray's own wave (packed: sample n-1 of the longest rays, View::zero_dropped).
template <typename View>
__device__ __forceinline__ void composite_bwd_ray(View &v, int64_t r, int lane, int nc, float carry, const float *chunk_carry,
                                                  float g0, float g1, float g2, float gd, float gm, float gt,
                                                  const float *__restrict__ bkg, int64_t bkg_rows, int white_bkg,
                                                  float *__restrict__ d_geo, float *__restrict__ d_radiance) {
    const int nchunk = (nc + 63) >> 6;
    bool neg = false;
    const bool use_bkg = bkg && bkg_rows > 0 && v.radiance;
    if (!use_bkg && white_bkg && v.radiance) gm = gm - (g0 + g1 + g2);
    float B = 0.f;  // dL/dT_last
    if (use_bkg) {
        const float *bk = bkg + (bkg_rows == 1 ? 0 : r) * 3;
        B = g0 * bk[0] + g1 * bk[1] + g2 * bk[2];
    }
    // upstream gradient of T_last itself (trans_shift[:, -1]): FullModel blends a background model's colour and depth with it
    B += gt;
    const bool final_visited = nc > 0 && v.final_visited(r);
    float suffix_carry = 0.f;
    float virt_dgeo = 0.f, virt_w = 0.f;
    bool has_virt = false;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int k = c * 64 + lane;
        const bool on = k < nc;
        float a = 0.f, q = 1.f, dl = 0.f, se = 0.f, gi = 0.f;
        if (on) {
            dl = v.delta(r, k, neg);
            a = alpha_at(v, r, k, dl, &se);
            q = (1.0f - a) + 1e-10f;
            gi = gd * v.z(r, k) + gm;
            if (v.radiance) {
                const float *cc = v.radiance + v.sidx(r, k) * 3;
                gi += g0 * cc[0] + g1 * cc[1] + g2 * cc[2];
            }
        }
        float incl = wave_incl_prod(q);
        float excl = wave_from_below(incl, 1.0f);
        float T = chunk_carry[c] * excl;
        float w = a * T;
        float term = on ? w * gi : 0.f;
        // when column Pe-1 is visited, T_last = T at that column: it depends on every earlier alpha only
        if (on && k == nc - 1 && final_visited) term += T * B;
        float sfx_incl = wave_incl_suffix_sum(term);
        float sfx_excl = wave_from_above(sfx_incl, 0.f);
        float suffix = sfx_excl + suffix_carry;
        // otherwise T_last is the product over ALL visited q (== carry) and depends on every visited alpha
        if (!final_visited) suffix += carry * B;
        if (on) {
            float dalpha = T * gi - suffix / q;
            float dgeo;
            if (v.alpha_in) dgeo = dalpha;
            else dgeo = se > 0.f ? dalpha * dl * expf(-se * dl) : 0.f;
            if (v.real(r, k)) {
                int64_t si = v.sidx(r, k);
                d_geo[si] = dgeo;
                if (d_radiance) {
                    d_radiance[si * 3 + 0] = w * g0;
                    d_radiance[si * 3 + 1] = w * g1;
                    d_radiance[si * 3 + 2] = w * g2;
                }
            } else {
                has_virt = true;
                virt_dgeo = dgeo;
                virt_w = w;
            }
        }
        suffix_carry += lane_value<0>(sfx_incl);
    }
    // the virtual final column aliases sample n-1, whose own column (delta 0) was stored by another lane of this
    // wave: accumulate after those stores have been issued and completed.
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    if (has_virt) {
        int64_t si = v.sidx(r, nc - 1);
        atomicAdd(&d_geo[si], virt_dgeo);
        if (d_radiance) {
            atomicAdd(&d_radiance[si * 3 + 0], virt_w * g0);
            atomicAdd(&d_radiance[si * 3 + 1], virt_w * g1);
            atomicAdd(&d_radiance[si * 3 + 2], virt_w * g2);
        }
    }
}

template <typename View>
__global__ void __launch_bounds__(256)
composite_bwd_kernel(View v, const int32_t *p_dense_ptr, const float *__restrict__ bkg, int64_t bkg_rows, int64_t R,
                     int white_bkg, const float *__restrict__ d_rgb, const float *__restrict__ d_depth,
                     const float *__restrict__ d_mask, const float *__restrict__ d_tlast, float *__restrict__ d_geo,
                     float *__restrict__ d_radiance) {
    __shared__ float s_carry[kRaysPerBlock][kMaxChunks];
    v.patch(p_dense_ptr);
    const int lane = lane_id();
    const int wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    if (r >= R) return;
    v.begin_ray(r);
    const int nc = v.ncol(r);
    const int nchunk = (nc + 63) >> 6;
    bool neg = false;
    if (lane == 0) v.zero_dropped(r, d_geo, d_radiance);
    v.zero_truncated(lane, d_geo, d_radiance);
    // pass 1: transmittance at every chunk start
    float carry = 1.0f;
    for (int c = 0; c < nchunk; ++c) {
        if (lane == 0) s_carry[wv][c] = carry;
        const int k = c * 64 + lane;
        float q = 1.f, se;
        if (k < nc) {
            float dl = v.delta(r, k, neg);
            float a = alpha_at(v, r, k, dl, &se);
            q = (1.0f - a) + 1e-10f;
        }
        float incl = wave_incl_prod(q);
        carry = carry * lane_value<63>(incl);
    }
    __builtin_amdgcn_wave_barrier();
    const float g0 = d_rgb ? d_rgb[3 * r] : 0.f, g1 = d_rgb ? d_rgb[3 * r + 1] : 0.f, g2 = d_rgb ? d_rgb[3 * r + 2] : 0.f;
    // pass 2: reverse walk
    composite_bwd_ray(v, r, lane, nc, carry, s_carry[wv], g0, g1, g2, d_depth ? d_depth[r] : 0.f, d_mask ? d_mask[r] : 0.f,
                      d_tlast ? d_tlast[r] : 0.f, bkg, bkg_rows, white_bkg, d_geo, d_radiance);
}

// ---- fused training step of the compositor: forward + ImgLoss(Huber) + backward in ONE pass per ray ------------------------
// The Huber gradient of a ray depends on that ray's colour only (d loss / d rgb = huber'(rgb - target) * weight / (3R)), so the
// wave that composited ray r can walk it backwards right away: one launch instead of compositing, loss and backward kernels
// plus a memset, the per-chunk transmittances are kept from the forward walk, and the inputs are read while still in cache.
// loss_partials[workgroup] = sum of the loss terms of that workgroup's 4 rays (plain store; the loss is the sum of the partials).
// An atomic per workgroup onto ONE scalar was measured at +14 us per launch: 2080 same-address float atomics serialise in L2.
template <typename View>
__global__ void __launch_bounds__(256)
composite_train_kernel(View v, const int32_t *p_dense_ptr, const float *__restrict__ bkg, int64_t bkg_rows, int64_t R,
                       int white_bkg, const float *__restrict__ target, float delta, float weight, float *__restrict__ rgb,
                       float *__restrict__ depth, float *__restrict__ mask, float *__restrict__ d_rgb,
                       float *__restrict__ loss_partials, float *__restrict__ d_geo, float *__restrict__ d_radiance) {
    __shared__ float s_carry[kRaysPerBlock][kMaxChunks];
    __shared__ float s_loss[kRaysPerBlock];
    v.patch(p_dense_ptr);
    const int lane = lane_id();
    const int wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlock + wv;
    float my_loss = 0.f;
    if (r < R) {
        v.begin_ray(r);
        const int nc = v.ncol(r);
        bool neg = false;
        if (lane == 0) v.zero_dropped(r, d_geo, d_radiance);
        v.zero_truncated(lane, d_geo, d_radiance);
        const RayOut o = composite_fwd_ray(v, r, lane, nc, bkg, bkg_rows, white_bkg, (float *)nullptr, (float *)nullptr,
                                           (float *)nullptr, s_carry[wv], neg);
        __builtin_amdgcn_wave_barrier();
        // arcnerf/loss/img_loss.py:60-100, mean over all R*3 elements, times weight (same arithmetic as huber_kernel)
        const float scale = weight / (float)(R * 3);
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = o.rgb[c] - target[3 * r + c];
            const float ad = fabsf(d);
            const bool quad = ad < delta;
            my_loss += (quad ? (0.5f / delta) * ad * ad : ad - 0.5f * delta) * scale;
            g[c] = (quad ? d / delta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f))) * scale;
        }
        if (lane == 0) {
            if (depth) depth[r] = o.depth;
            if (mask) mask[r] = o.mask;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (rgb) rgb[r * 3 + c] = o.rgb[c];
                if (d_rgb) d_rgb[r * 3 + c] = g[c];
            }
        }
        composite_bwd_ray(v, r, lane, nc, o.carry, s_carry[wv], g[0], g[1], g[2], 0.f, 0.f, 0.f, bkg, bkg_rows, white_bkg, d_geo,
                          d_radiance);
    }
    if (loss_partials) {
        if (lane == 0) s_loss[wv] = my_loss;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < kRaysPerBlock; ++k) t += s_loss[k];
            loss_partials[blockIdx.x] = t;
        }
    }
}

// ---- inverse CDF resampling ---------------------------------------------------------------------
// one row's searchsorted(right=True) + lerp for the uniforms k = tid, tid + blockDim, ... into s_val, then the row sort and the store
__device__ __forceinline__ void inverse_cdf_row(const float *__restrict__ b, const float *c, const float *__restrict__ u_row, int n_pts,
                                                int n_sample, int n_pad, float eps, int do_sort, float *s_val, float *__restrict__ samples_row,
                                                int32_t *__restrict__ inds_row) {
    for (int k = threadIdx.x; k < n_pad; k += blockDim.x) {
        float out = INFINITY;
        if (k < n_sample) {
            float uu = u_row[k];
            int lo = 0, hi = n_pts;  // searchsorted(right=True): first index with cdf > u
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (c[mid] <= uu) lo = mid + 1; else hi = mid;
            }
            if (inds_row) inds_row[k] = lo;
            int below = lo - 1 < 0 ? 0 : (lo - 1 > n_pts - 1 ? n_pts - 1 : lo - 1);
            int above = lo > n_pts - 1 ? n_pts - 1 : lo;
            float denom = c[above] - c[below];
            if (denom < eps) denom = 1.0f;
            float t = (uu - c[below]) / denom;
            out = b[below] + t * (b[above] - b[below]);
        }
        s_val[k] = out;
    }
    __syncthreads();
    if (do_sort) {  // bitonic sort of the padded row in LDS
        for (int size = 2; size <= n_pad; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = threadIdx.x; i < n_pad; i += blockDim.x) {
                    int j = i ^ stride;
                    if (j > i) {
                        bool up = (i & size) == 0;
                        float x = s_val[i], y = s_val[j];
                        if ((x > y) == up) { s_val[i] = y; s_val[j] = x; }
                    }
                }
                __syncthreads();
            }
        }
    }
    for (int k = threadIdx.x; k < n_sample; k += blockDim.x) samples_row[k] = s_val[k];
}

__global__ void __launch_bounds__(256)
sample_cdf_kernel(const float *__restrict__ bins, const float *__restrict__ cdf, const float *__restrict__ u, int n_pts,
                  int n_sample, int n_pad, float eps, int do_sort, float *__restrict__ samples,
                  int32_t *__restrict__ inds) {
    extern __shared__ __attribute__((aligned(16))) float s_val[];
    const int64_t r = blockIdx.x;
    inverse_cdf_row(bins + r * n_pts, cdf + r * n_pts, u + r * n_sample, n_pts, n_sample, n_pad, eps, do_sort, s_val, samples + r * n_sample,
                    inds ? inds + r * n_sample : nullptr);
}

// sample_pdf (ray_helper.py:410-429) in one launch per call: weights + eps -> / sum -> cumsum -> inverse CDF -> sort.  The cdf is built
// the way torch builds it on the host the reference's CPU path runs on: the running sum is kept in DOUBLE and every prefix is rounded
// to float (at::native cumsum accumulates float tensors in acc_type<float, false> = double), sequentially - a float tree scan differs
// from it by up to ~1e-6 near cdf = 1, and the inverse CDF divides by bin masses down to 1e-5.  The normaliser is the double sum of
// the (weight + eps) floats rounded to float (torch.sum is a vectorised float sum whose order depends on the host's ISA; it is within
// an ulp or two of this).  One workgroup per ray; u has u_rows = 1 (one lattice for all rays) or R rows.
__global__ void __launch_bounds__(256)
sample_pdf_kernel(const float *__restrict__ bins, const float *__restrict__ weights, const float *__restrict__ u, int u_rows, int n_pts,
                  int n_sample, int n_pad, float eps, int do_sort, float *__restrict__ samples, float *__restrict__ cdf_out) {
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float *s_cdf = s_mem;                  // n_pts
    float *s_val = s_mem + n_pts;          // n_pad
    __shared__ double s_part[4];
    __shared__ float s_norm;
    const int64_t r = blockIdx.x;
    const int n_w = n_pts - 1;
    const float *w = weights + r * n_w;
    double part = 0.0;
    for (int k = threadIdx.x; k < n_w; k += blockDim.x) {
        const float v = w[k] + eps;
        s_cdf[k + 1] = v;
        part += (double)v;
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < (int)((blockDim.x + 63) >> 6); ++i) tot += s_part[i];
        s_norm = (float)tot;
    }
    __syncthreads();
    const float norm = s_norm;
    for (int k = threadIdx.x; k < n_w; k += blockDim.x) s_cdf[k + 1] = s_cdf[k + 1] / norm;     // pdf, in place
    __syncthreads();
    if (threadIdx.x == 0) {
        double acc = 0.0;
        s_cdf[0] = 0.0f;
        for (int k = 1; k < n_pts; ++k) {
            acc += (double)s_cdf[k];
            s_cdf[k] = (float)acc;
        }
    }
    __syncthreads();
    if (cdf_out)
        for (int k = threadIdx.x; k < n_pts; k += blockDim.x) cdf_out[r * n_pts + k] = s_cdf[k];
    inverse_cdf_row(bins + r * n_pts, s_cdf, u + (u_rows > 1 ? r * n_sample : 0), n_pts, n_sample, n_pad, eps, do_sort, s_val,
                    samples + r * n_sample, nullptr);
}

// ---- ImgLoss(Huber) value + gradient (arcnerf/loss/img_loss.py:60-100), mean over all R*3 elements, times weight ----
__global__ void __launch_bounds__(256) huber_kernel(const float *__restrict__ x, const float *__restrict__ y, int64_t n, float delta,
                                                    float weight, float *__restrict__ dx, float *__restrict__ loss) {
    float acc = 0.f;
    const float scale = weight / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = x[i] - y[i];
        const float ad = fabsf(d);
        const bool quad = ad < delta;
        acc += quad ? (0.5f / delta) * ad * ad : ad - 0.5f * delta;
        if (dx) dx[i] = (quad ? d / delta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f))) * scale;
    }
    acc = wave_sum(acc);
    if (loss && lane_id() == 0) atomicAdd(loss, acc * scale);
}

// ---- NeuS interval opacity --------------------------------------------------------------------------------------------
// sdf_to_alpha (arcnerf/models/neus_model.py:242-265), cdf = sigmoid(sdf * s) (:221-228).  One lane per interval; s is read
// from device memory (it is exp(10 * inv_s) of a learnable parameter: no host round trip).  Backward also reduces d s.
__device__ __forceinline__ float sigmoidf_dev(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(256)
sdf_to_alpha_fwd_kernel(const float *__restrict__ mid_sdf, const float *__restrict__ zvals, const float *__restrict__ mid_slope,
                        const float *__restrict__ s_ptr, int clip, float *__restrict__ alpha, int64_t R, int P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * (P - 1)) return;
    const int64_t r = i / (P - 1);
    const int k = (int)(i - r * (P - 1));
    const float s = *s_ptr;
    const float dist = zvals[r * P + k + 1] - zvals[r * P + k];
    const float h = mid_slope[i] * dist * 0.5f;
    const float pc = sigmoidf_dev((mid_sdf[i] - h) * s), nc = sigmoidf_dev((mid_sdf[i] + h) * s);
    float a = (pc - nc + 1e-5f) / (pc + 1e-5f);
    if (clip) a = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);
    alpha[i] = a;
}

__global__ void __launch_bounds__(256)
sdf_to_alpha_bwd_kernel(const float *__restrict__ mid_sdf, const float *__restrict__ zvals, const float *__restrict__ mid_slope,
                        const float *__restrict__ s_ptr, int clip, const float *__restrict__ d_alpha, float *__restrict__ d_sdf,
                        float *__restrict__ d_slope, float *__restrict__ d_s, int64_t R, int P) {
    __shared__ float s_part[4];
    const float s = *s_ptr;
    float acc = 0.f;
    const int64_t total = R * (P - 1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (P - 1);
        const int k = (int)(i - r * (P - 1));
        const float dist = zvals[r * P + k + 1] - zvals[r * P + k];
        const float h = mid_slope[i] * dist * 0.5f;
        const float prev = mid_sdf[i] - h, next = mid_sdf[i] + h;
        const float pc = sigmoidf_dev(prev * s), nc = sigmoidf_dev(next * s);
        const float B = pc + 1e-5f;
        const float a = (pc - nc + 1e-5f) / B;
        float g = d_alpha[i];
        if (clip && (a < 0.0f || a > 1.0f)) g = 0.0f;  // torch.clip passes the gradient on [0, 1] inclusive
        const float gp = g * (nc / (B * B)) * (pc * (1.0f - pc));
        const float gn = -g * (1.0f / B) * (nc * (1.0f - nc));
        d_sdf[i] = (gp + gn) * s;
        d_slope[i] = (gn - gp) * s * (dist * 0.5f);
        acc += gp * prev + gn * next;
    }
    if (!d_s) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(d_s, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
}

}  // namespace arcn

using namespace arcn;

ARCN_EXPORT int arcn_huber_loss_grad(const float *x, const float *y, int64_t n, float delta, float weight, float *dx,
                                     float *loss, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !y || !(delta > 0)) return einval("huber_loss_grad: bad argument");
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), as_stream(stream)) != hipSuccess) return check_launch("memset");
    int64_t blocks = ceil_div<int64_t>(n, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(huber_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, y, n, delta, weight, dx, loss);
    return check_launch("huber_loss_grad");
}

ARCN_EXPORT int arcn_ray_marching_fwd(const float *sigma, const float *alpha_in, const float *radiance,
                                      const float *zvals, const float *noise, const float *bkg, int64_t bkg_rows,
                                      int64_t R, int P, int add_inf_z, int white_bkg, float *rgb, float *depth,
                                      float *mask, float *alpha_out, float *trans_out, float *weights_out,
                                      int32_t *status, void *stream) {
    if (R <= 0) return ARCN_OK;
    if ((!sigma && !alpha_in) || !zvals || P < 1) return einval("ray_marching_fwd: sigma/alpha and zvals required");
    if (!(bkg_rows == 0 || bkg_rows == 1 || bkg_rows == R)) return einval("ray_marching_fwd: bkg rows must be 0/1/R");
    DenseView v{sigma, alpha_in, radiance, zvals, noise, P, (add_inf_z || alpha_in) ? P : P - 1};
    dim3 grid((unsigned)ceil_div<int64_t>(R, kRaysPerBlock));
    hipLaunchKernelGGL(composite_fwd_kernel<DenseView>, grid, dim3(256), 0, as_stream(stream), v, nullptr, bkg, bkg_rows,
                       R, white_bkg, rgb, depth, mask, alpha_out, trans_out, weights_out, status);
    return check_launch("ray_marching_fwd");
}

ARCN_EXPORT int arcn_ray_marching_bwd(const float *sigma, const float *alpha_in, const float *radiance,
                                      const float *zvals, const float *noise, const float *bkg, int64_t bkg_rows,
                                      int64_t R, int P, int add_inf_z, int white_bkg, const float *d_rgb,
                                      const float *d_depth, const float *d_mask, const float *d_tlast, float *d_geo,
                                      float *d_radiance, void *stream) {
    if (R <= 0) return ARCN_OK;
    if ((!sigma && !alpha_in) || !zvals || !d_geo || P < 1) return einval("ray_marching_bwd: missing argument");
    const int Pe = (add_inf_z || alpha_in) ? P : P - 1;
    if (Pe > 64 * kMaxChunks) return einval("ray_marching_bwd: at most 4096 samples per ray");
    if (Pe < P) {  // the dropped last column receives zero gradient
        if (hipMemsetAsync(d_geo, 0, sizeof(float) * R * P, as_stream(stream)) != hipSuccess) return check_launch("memset");
        if (d_radiance && hipMemsetAsync(d_radiance, 0, sizeof(float) * R * P * 3, as_stream(stream)) != hipSuccess)
            return check_launch("memset");
    }
    DenseView v{sigma, alpha_in, radiance, zvals, noise, P, Pe};
    dim3 grid((unsigned)ceil_div<int64_t>(R, kRaysPerBlock));
    hipLaunchKernelGGL(composite_bwd_kernel<DenseView>, grid, dim3(256), 0, as_stream(stream), v, nullptr, bkg, bkg_rows,
                       R, white_bkg, d_rgb, d_depth, d_mask, d_tlast, d_geo, d_radiance);
    return check_launch("ray_marching_bwd");
}

ARCN_EXPORT int arcn_composite_packed_fwd(const float *sigma, const float *radiance, const float *t_packed,
                                          const int32_t *offsets, const float *noise, const float *bkg,
                                          int64_t bkg_rows, int64_t R, int p_dense, const int32_t *p_dense_ptr,
                                          int add_inf_z, int white_bkg, float *rgb, float *depth, float *mask,
                                          float *weights_out, const int32_t *counts, void *stream) {
    if (R <= 0) return ARCN_OK;
    if (!sigma || !t_packed || !offsets) return einval("composite_packed_fwd: missing argument");
    if (!(bkg_rows == 0 || bkg_rows == 1 || bkg_rows == R)) return einval("composite_packed_fwd: bkg rows must be 0/1/R");
    if (p_dense < 2) p_dense = 2;
    PackedView v{sigma, nullptr, radiance, t_packed, noise, offsets, p_dense, add_inf_z ? p_dense : p_dense - 1, add_inf_z};
    v.counts = counts;
    dim3 grid((unsigned)ceil_div<int64_t>(R, kRaysPerBlock));
    hipLaunchKernelGGL(composite_fwd_kernel<PackedView>, grid, dim3(256), 0, as_stream(stream), v, p_dense_ptr, bkg,
                       bkg_rows, R, white_bkg, rgb, depth, mask, nullptr, nullptr, weights_out, nullptr);
    return check_launch("composite_packed_fwd");
}

ARCN_EXPORT int arcn_composite_packed_train(const float *sigma, const float *radiance, const float *t_packed, const int32_t *offsets,
                                            const float *noise, const float *bkg, int64_t bkg_rows, int64_t R, int p_dense,
                                            const int32_t *p_dense_ptr, int add_inf_z, int white_bkg, const float *target,
                                            float huber_delta, float loss_weight, float *rgb, float *depth, float *mask,
                                            float *d_rgb, float *loss_partials, float *d_sigma, float *d_radiance, const int32_t *counts,
                                            void *stream) {
    if (R <= 0) return ARCN_OK;
    if (!sigma || !radiance || !t_packed || !offsets || !target || !d_sigma || !d_radiance)
        return einval("composite_packed_train: missing argument");
    if (!(bkg_rows == 0 || bkg_rows == 1 || bkg_rows == R)) return einval("composite_packed_train: bkg rows must be 0/1/R");
    if (!(huber_delta > 0.f)) return einval("composite_packed_train: huber delta must be positive");
    if (p_dense < 2) p_dense = 2;
    if ((add_inf_z ? p_dense : p_dense - 1) > 64 * kMaxChunks) return einval("composite_packed_train: at most 4096 samples per ray");
    PackedView v{sigma, nullptr, radiance, t_packed, noise, offsets, p_dense, add_inf_z ? p_dense : p_dense - 1, add_inf_z};
    v.counts = counts;
    dim3 grid((unsigned)ceil_div<int64_t>(R, kRaysPerBlock));
    hipLaunchKernelGGL(composite_train_kernel<PackedView>, grid, dim3(256), 0, as_stream(stream), v, p_dense_ptr, bkg, bkg_rows, R,
                       white_bkg, target, huber_delta, loss_weight, rgb, depth, mask, d_rgb, loss_partials, d_sigma, d_radiance);
    return check_launch("composite_packed_train");
}

ARCN_EXPORT int arcn_composite_packed_bwd(const float *sigma, const float *radiance, const float *t_packed,
                                          const int32_t *offsets, const float *noise, const float *bkg,
                                          int64_t bkg_rows, int64_t R, int p_dense, const int32_t *p_dense_ptr,
                                          int add_inf_z, int white_bkg, const float *d_rgb, const float *d_depth,
                                          const float *d_mask, float *d_sigma, float *d_radiance, const int32_t *counts,
                                          void *stream) {
    if (R <= 0) return ARCN_OK;
    if (!sigma || !t_packed || !offsets || !d_sigma) return einval("composite_packed_bwd: missing argument");
    if (p_dense < 2) p_dense = 2;
    PackedView v{sigma, nullptr, radiance, t_packed, noise, offsets, p_dense, add_inf_z ? p_dense : p_dense - 1, add_inf_z};
    v.counts = counts;
    dim3 grid((unsigned)ceil_div<int64_t>(R, kRaysPerBlock));
    hipLaunchKernelGGL(composite_bwd_kernel<PackedView>, grid, dim3(256), 0, as_stream(stream), v, p_dense_ptr, bkg,
                       bkg_rows, R, white_bkg, d_rgb, d_depth, d_mask, static_cast<const float *>(nullptr), d_sigma, d_radiance);
    return check_launch("composite_packed_bwd");
}

ARCN_EXPORT int arcn_sample_cdf(const float *bins, const float *cdf, const float *u, int64_t R, int n_pts, int n_sample,
                                float eps, int do_sort, float *samples, int32_t *inds, void *stream) {
    if (R <= 0 || n_sample <= 0) return ARCN_OK;
    if (!bins || !cdf || !u || !samples || n_pts < 1) return einval("sample_cdf: missing argument");
    int n_pad = 1;
    while (n_pad < n_sample) n_pad <<= 1;
    if (n_pad > 16384) return einval("sample_cdf: n_sample too large");
    hipLaunchKernelGGL(sample_cdf_kernel, dim3((unsigned)R), dim3(256), sizeof(float) * n_pad, as_stream(stream), bins, cdf,
                       u, n_pts, n_sample, n_pad, eps, do_sort, samples, inds);
    return check_launch("sample_cdf");
}

ARCN_EXPORT int arcn_sample_pdf(const float *bins, const float *weights, const float *u, int64_t R, int n_pts, int n_sample, int u_rows,
                                float eps, int do_sort, float *samples, float *cdf_out, void *stream) {
    if (R <= 0 || n_sample <= 0) return ARCN_OK;
    if (!bins || !weights || !u || !samples || n_pts < 2) return einval("sample_pdf: missing argument");
    if (u_rows != 1 && u_rows != R) return einval("sample_pdf: u must have 1 or R rows");
    int n_pad = 1;
    while (n_pad < n_sample) n_pad <<= 1;
    if (n_pad + n_pts > 15360) return einval("sample_pdf: n_sample + n_pts too large for one workgroup's LDS");
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)R), dim3(256), sizeof(float) * (size_t)(n_pad + n_pts), as_stream(stream), bins, weights,
                       u, u_rows, n_pts, n_sample, n_pad, eps, do_sort, samples, cdf_out);
    return check_launch("sample_pdf");
}

ARCN_EXPORT int arcn_sdf_to_alpha_fwd(const float *mid_sdf, const float *zvals, const float *mid_slope, const float *s_dev, int clip,
                                      float *alpha, int64_t R, int P, void *stream) {
    if (R <= 0 || P < 2) return ARCN_OK;
    if (!mid_sdf || !zvals || !mid_slope || !s_dev || !alpha) return einval("sdf_to_alpha_fwd: missing argument");
    hipLaunchKernelGGL(sdf_to_alpha_fwd_kernel, dim3((unsigned)ceil_div<int64_t>(R * (P - 1), 256)), dim3(256), 0, as_stream(stream),
                       mid_sdf, zvals, mid_slope, s_dev, clip, alpha, R, P);
    return check_launch("sdf_to_alpha_fwd");
}

ARCN_EXPORT int arcn_sdf_to_alpha_bwd(const float *mid_sdf, const float *zvals, const float *mid_slope, const float *s_dev, int clip,
                                      const float *d_alpha, float *d_sdf, float *d_slope, float *d_s, int64_t R, int P,
                                      void *stream) {
    if (R <= 0 || P < 2) return ARCN_OK;
    if (!mid_sdf || !zvals || !mid_slope || !s_dev || !d_alpha || !d_sdf || !d_slope) return einval("sdf_to_alpha_bwd: missing argument");
    int64_t blocks = ceil_div<int64_t>(R * (P - 1), 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sdf_to_alpha_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), mid_sdf, zvals, mid_slope,
                       s_dev, clip, d_alpha, d_sdf, d_slope, d_s, R, P);
    return check_launch("sdf_to_alpha_bwd");
}
