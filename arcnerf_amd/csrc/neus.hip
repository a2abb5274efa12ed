// NeuS on PACKED samples: section layout + fused (slope -> sdf_to_alpha -> transmittance scan -> weighted sums) render, forward and backward.
//
// Replaces, for a pruned (occupancy-marched) foreground, the dense chain of arcnerf/models/neus_model.py:63-104 (`_forward`), :174-202
// (`handle_mid_pts`, masked layout), :242-265 (`sdf_to_alpha`), arcnerf/models/sdf_model.py:42-101 (valid-point gather / padded scatter)
// and render/ray_helper.py:476-620 (`ray_marching`, alpha= branch) - which the reference runs on padded (rays, P) tensors, P = the
// longest ray of the batch - without materialising any (rays, P) tensor: the numbers are those of the padded form.
//
// The padded form, per ray with k > 0 marched samples z_0 < ... < z_{k-1} (fg_model.py:252-262: P = max(2, max k), tails repeat z_{k-1}):
//   half = (z_{k-1} - z_0) / n_sample * 0.5,  beyond = z_{k-1} + 2 half
//   section ends e_j = z_j (j < k), beyond (j >= k), j = 0..P;   mid points m_j = (e_j + e_{j+1}) / 2, j = 0..P-1
//   the nets see m_0 .. m_{n-1}, n = min(k + 1, P) (the mask of the mid points is [True, mask[:-1]]); slots j >= n repeat slot n-1's
//   sdf / normal / radiance (sdf_model.py:88-99) with a ZERO-length section - whose alpha is NOT zero: (pc - nc + 1e-5) / (pc + 1e-5) =
//   1e-5 / (sigmoid(s sdf) + 1e-5), i.e. ~1 when `beyond` lies inside the surface.  The padded slots therefore carry weight and are
//   walked here too (aliasing the ray's last evaluated point), so that rgb / depth / mask / normal / T_last equal the dense result.
// One 64-lane wavefront owns one ray; slots are consumed 64 at a time with the transmittance as a wave prefix product carried across
// chunks (same scheme as render.hip); the backward walks the chunks in reverse with true suffix sums.
#include "common.hpp"

namespace arcn {

constexpr int kNeusRaysPerBlock = 4;
constexpr int kNeusMaxChunks = 64;   // P <= 4096 slots per ray

// ---- layout -----------------------------------------------------------------------------------------------------------------------
// n_eval[r] = k > 0 ? min(k + 1, P) : 0 with P = max(2, max k): one thread per ray (the scan over n_eval is arcn_exclusive_scan_i32)
__global__ void __launch_bounds__(256) neus_count_kernel(const int32_t *__restrict__ counts, const int32_t *__restrict__ kmax_ptr, int64_t R,
                                                         int32_t *__restrict__ n_eval) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    int P = *kmax_ptr;
    P = P < 2 ? 2 : P;
    const int k = counts[r];
    n_eval[r] = k > 0 ? (k + 1 < P ? k + 1 : P) : 0;
}

// mid points, section ends and ray ids of the evaluated points; optionally the (R, P) map slot -> packed row the dense `normal_pts`
// output is gathered with (padded slots -> the ray's last point, rays without samples -> row `total`, which holds the default normal)
__global__ void __launch_bounds__(256)
neus_sections_kernel(const float *__restrict__ zdense, const int32_t *__restrict__ counts, const int32_t *__restrict__ offsets, int n_pts,
                     float n_sample_cfg, int64_t R, int P, float *__restrict__ t_mid, float *__restrict__ lo, float *__restrict__ hi,
                     int32_t *__restrict__ ray_id, int64_t *__restrict__ slot_map) {
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * kNeusRaysPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    const int k = counts[r];
    const int32_t off = offsets[r];
    const int n = offsets[r + 1] - off;
    const int64_t total = offsets[R];
    if (slot_map) {
        for (int j = lane; j < P; j += 64) slot_map[r * P + j] = n > 0 ? (int64_t)off + (j < n ? j : n - 1) : total;
    }
    if (n <= 0) return;
    const float *z = zdense + r * (int64_t)n_pts;
    const float z0 = z[0], zl = z[k - 1];
    const float half = (zl - z0) / n_sample_cfg * 0.5f;
    const float beyond = zl + half * 2.0f;
    for (int j = lane; j < n; j += 64) {
        const float e0 = j < k ? z[j] : beyond;
        const float e1 = j + 1 < k ? z[j + 1] : beyond;
        t_mid[off + j] = 0.5f * (e1 + e0);
        lo[off + j] = e0;
        hi[off + j] = e1;
        ray_id[off + j] = (int32_t)r;
    }
}

// dense (R, P, C) view of a packed per-point quantity, as the reference's outputs have it (`normal_pts`: sdf_model.py:88-99 padded fill,
// fg_model.py:320-387 defaults): slot j of ray r = point min(j, n - 1), rays without points = dflt.  One wave per ray, coalesced rows.
template <int C>
__global__ void __launch_bounds__(256)
neus_slots_fwd_kernel(const float *__restrict__ packed, const int32_t *__restrict__ offsets, int64_t R, int P, const float dflt0, const float dflt1,
                      const float dflt2, float *__restrict__ dense) {
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * kNeusRaysPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    const int32_t off = offsets[r];
    const int n = offsets[r + 1] - off;
    const float df[3] = {dflt0, dflt1, dflt2};
    float *row = dense + r * (int64_t)P * C;
    for (int e = lane; e < P * C; e += 64) {
        const int j = e / C, c = e - j * C;
        row[e] = n > 0 ? packed[((int64_t)off + (j < n ? j : n - 1)) * C + c] : df[c < 3 ? c : 2];
    }
}

// its transpose: d_packed[point] = sum of the gradients of the slots that alias it (no atomics: the ray's wave sums the tail)
template <int C>
__global__ void __launch_bounds__(256)
neus_slots_bwd_kernel(const float *__restrict__ d_dense, const int32_t *__restrict__ offsets, int64_t R, int P, float *__restrict__ d_packed) {
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * kNeusRaysPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    const int32_t off = offsets[r];
    const int n = offsets[r + 1] - off;
    if (n <= 0) return;
    const float *row = d_dense + r * (int64_t)P * C;
    for (int e = lane; e < (n - 1) * C; e += 64) d_packed[(int64_t)off * C + e] = row[e];
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    for (int j = n - 1 + lane; j < P; j += 64) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += row[j * C + c];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = wave_sum(acc[c]);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) d_packed[((int64_t)off + n - 1) * C + c] = acc[c];
    }
}

// ---- render -----------------------------------------------------------------------------------------------------------------------
struct NeusIn {
    const float *sdf, *radiance, *normal, *t_mid, *lo, *hi, *rays_d, *s_ptr, *bkg;
    const int32_t *offsets, *kmax_ptr;
    int64_t bkg_rows, R;
    float cos_anneal, depth_far;
    float dflt_rgb[3], dflt_nrm[3];
};

struct Slot {
    float a, q, araw, pc, nc, dist, slope, sdf, z, nlen, nh[3], c[3];
    int64_t idx;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// slot k of ray (off, n): aliases point min(k, n-1); zero-length section past the evaluated points
__device__ __forceinline__ Slot neus_slot(const NeusIn &p, int32_t off, int n, int k, const float d[3], float s) {
    Slot o;
    const int j = k < n ? k : n - 1;
    o.idx = (int64_t)off + j;
    o.dist = k < n ? p.hi[o.idx] - p.lo[o.idx] : 0.0f;
    o.sdf = p.sdf[o.idx];
    o.z = p.t_mid[o.idx];
    const float *nv = p.normal + o.idx * 3;
    o.slope = (d[0] * nv[0] + d[1] * nv[1]) + d[2] * nv[2];
    const float u = -o.slope * 0.5f + 0.5f, v = -o.slope;
    const float it = -((u > 0.f ? u : 0.f) * (1.0f - p.cos_anneal) + (v > 0.f ? v : 0.f) * p.cos_anneal);
    const float h = it * o.dist * 0.5f;
    o.pc = sigm((o.sdf - h) * s);
    o.nc = sigm((o.sdf + h) * s);
    o.araw = (o.pc - o.nc + 1e-5f) / (o.pc + 1e-5f);
    o.a = o.araw < 0.0f ? 0.0f : (o.araw > 1.0f ? 1.0f : o.araw);
    o.q = (1.0f - o.a) + 1e-10f;
    o.nlen = sqrtf((nv[0] * nv[0] + nv[1] * nv[1]) + nv[2] * nv[2]);
    const float inv = 1.0f / (o.nlen + 1e-8f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o.nh[c] = nv[c] * inv;
        o.c[c] = p.radiance[o.idx * 3 + c];
    }
    return o;
}

__global__ void __launch_bounds__(256)
neus_render_fwd_kernel(NeusIn p, float *__restrict__ rgb, float *__restrict__ depth, float *__restrict__ mask, float *__restrict__ nrm,
                       float *__restrict__ t_last_out) {
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x * kNeusRaysPerBlock + (threadIdx.x >> 6);
    if (r >= p.R) return;
    const int32_t off = p.offsets[r];
    const int n = p.offsets[r + 1] - off;
    const float *bk = (p.bkg && p.bkg_rows > 0) ? p.bkg + (p.bkg_rows == 1 ? 0 : r) * 3 : nullptr;
    if (n <= 0) {   // FgModel.update_values_for_invalid_rays (fg_model.py:320-387)
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rgb[r * 3 + c] = bk ? bk[c] : p.dflt_rgb[c];
                nrm[r * 3 + c] = p.dflt_nrm[c];
            }
            depth[r] = p.depth_far;
            mask[r] = 0.0f;
            t_last_out[r] = 1.0f;
        }
        return;
    }
    int P = *p.kmax_ptr;
    P = P < 2 ? 2 : P;
    const float d[3] = {p.rays_d[r * 3], p.rays_d[r * 3 + 1], p.rays_d[r * 3 + 2]};
    const float s = *p.s_ptr;
    float carry = 1.0f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, t_last = 0.f;
    for (int base = 0; base < P; base += 64) {
        const int k = base + lane;
        const bool on = k < P;
        Slot sl;
        float a = 0.f, q = 1.f;
        if (on) {
            sl = neus_slot(p, off, n, k, d, s);
            a = sl.a;
            q = sl.q;
        }
        const float incl = wave_incl_prod(q);
        const float T = carry * wave_from_below(incl, 1.0f);
        const float w = a * T;
        if (on) {
            acc[0] += w * sl.c[0]; acc[1] += w * sl.c[1]; acc[2] += w * sl.c[2];
            acc[3] += w * sl.z;
            acc[4] += w;
            acc[5] += w * sl.nh[0]; acc[6] += w * sl.nh[1]; acc[7] += w * sl.nh[2];
            if (k == P - 1) t_last = T;
        }
        carry = carry * lane_value<63>(incl);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = wave_sum(acc[i]);
    t_last = wave_sum(t_last);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rgb[r * 3 + c] = bk ? acc[c] + t_last * bk[c] : acc[c];
            nrm[r * 3 + c] = acc[5 + c];
        }
        depth[r] = acc[3];
        mask[r] = acc[4];
        t_last_out[r] = t_last;
    }
}

// d_sdf (S), d_radiance (S,3), d_normal (S,3), d_s_ray (R): written for every evaluated point / ray (rays without samples: d_s_ray = 0)
__global__ void __launch_bounds__(256)
neus_render_bwd_kernel(NeusIn p, const float *__restrict__ g_rgb, const float *__restrict__ g_depth, const float *__restrict__ g_mask,
                       const float *__restrict__ g_nrm, const float *__restrict__ g_tlast, float *__restrict__ d_sdf,
                       float *__restrict__ d_radiance, float *__restrict__ d_normal, float *__restrict__ d_s_ray) {
    __shared__ float s_carry[kNeusRaysPerBlock][kNeusMaxChunks];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * kNeusRaysPerBlock + wv;
    if (r >= p.R) return;
    const int32_t off = p.offsets[r];
    const int n = p.offsets[r + 1] - off;
    if (n <= 0) {
        if (lane == 0 && d_s_ray) d_s_ray[r] = 0.f;
        return;
    }
    int P = *p.kmax_ptr;
    P = P < 2 ? 2 : P;
    const int nchunk = (P + 63) >> 6;
    const float d[3] = {p.rays_d[r * 3], p.rays_d[r * 3 + 1], p.rays_d[r * 3 + 2]};
    const float s = *p.s_ptr;
    // pass 1: transmittance at every chunk start
    float carry = 1.0f;
    for (int c = 0; c < nchunk; ++c) {
        if (lane == 0) s_carry[wv][c] = carry;
        const int k = c * 64 + lane;
        float q = 1.f;
        if (k < P) q = neus_slot(p, off, n, k, d, s).q;
        carry = carry * lane_value<63>(wave_incl_prod(q));
    }
    __builtin_amdgcn_wave_barrier();
    const float g0 = g_rgb ? g_rgb[3 * r] : 0.f, g1 = g_rgb ? g_rgb[3 * r + 1] : 0.f, g2 = g_rgb ? g_rgb[3 * r + 2] : 0.f;
    const float gd = g_depth ? g_depth[r] : 0.f, gm = g_mask ? g_mask[r] : 0.f;
    const float gn0 = g_nrm ? g_nrm[3 * r] : 0.f, gn1 = g_nrm ? g_nrm[3 * r + 1] : 0.f, gn2 = g_nrm ? g_nrm[3 * r + 2] : 0.f;
    float B = g_tlast ? g_tlast[r] : 0.f;   // dL / dT_last: its own upstream gradient + the background colour term of rgb
    if (p.bkg && p.bkg_rows > 0) {
        const float *bk = p.bkg + (p.bkg_rows == 1 ? 0 : r) * 3;
        B += (g0 * bk[0] + g1 * bk[1]) + g2 * bk[2];
    }
    float suffix_carry = 0.f, ds_acc = 0.f;
    float tail[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // the slots k >= n - 1 all alias point n - 1: d_sdf, d_radiance, d_normal summed
    for (int c = nchunk - 1; c >= 0; --c) {
        const int k = c * 64 + lane;
        const bool on = k < P;
        Slot sl;
        float a = 0.f, q = 1.f, gi = 0.f;
        if (on) {
            sl = neus_slot(p, off, n, k, d, s);
            a = sl.a;
            q = sl.q;
            gi = ((g0 * sl.c[0] + g1 * sl.c[1]) + g2 * sl.c[2]) + gd * sl.z + gm + ((gn0 * sl.nh[0] + gn1 * sl.nh[1]) + gn2 * sl.nh[2]);
        }
        const float incl = wave_incl_prod(q);
        const float T = s_carry[wv][c] * wave_from_below(incl, 1.0f);
        const float w = a * T;
        float term = on ? w * gi : 0.f;
        if (on && k == P - 1) term += T * B;          // T_last = T at the last slot: it depends on every earlier alpha
        const float sfx_incl = wave_incl_suffix_sum(term);
        const float suffix = wave_from_above(sfx_incl, 0.f) + suffix_carry;
        float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (on) {
            float da = T * gi - suffix / q;
            if (sl.araw < 0.0f || sl.araw > 1.0f) da = 0.f;      // torch.clip passes the gradient on [0, 1] inclusive
            const float Bc = sl.pc + 1e-5f;
            const float gp = da * (sl.nc / (Bc * Bc)) * (sl.pc * (1.0f - sl.pc));
            const float gn = -da * (1.0f / Bc) * (sl.nc * (1.0f - sl.nc));
            const float u = -sl.slope * 0.5f + 0.5f, vv = -sl.slope;
            const float it = -((u > 0.f ? u : 0.f) * (1.0f - p.cos_anneal) + (vv > 0.f ? vv : 0.f) * p.cos_anneal);
            const float h = it * sl.dist * 0.5f;
            ds_acc += gp * (sl.sdf - h) + gn * (sl.sdf + h);
            const float d_it = (gn - gp) * s * (sl.dist * 0.5f);
            const float d_slope = d_it * ((u > 0.f ? 0.5f * (1.0f - p.cos_anneal) : 0.f) + (vv > 0.f ? p.cos_anneal : 0.f));
            // normalised normal n / (|n| + 1e-8): J^T g = g / (|n| + eps) - n (n . g) / (|n| (|n| + eps)^2)
            const float gw[3] = {w * gn0, w * gn1, w * gn2};
            const float le = sl.nlen + 1e-8f;
            const float nv[3] = {sl.nh[0] * le, sl.nh[1] * le, sl.nh[2] * le};
            const float ndg = (nv[0] * gw[0] + nv[1] * gw[1]) + nv[2] * gw[2];
            const float kk = sl.nlen > 0.f ? ndg / (sl.nlen * le * le) : 0.f;
            v[0] = (gp + gn) * s;
            v[1] = w * g0; v[2] = w * g1; v[3] = w * g2;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) v[4 + cc] = d_slope * d[cc] + (gw[cc] / le - nv[cc] * kk);
            if (k < n - 1) {
                d_sdf[sl.idx] = v[0];
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    d_radiance[sl.idx * 3 + cc] = v[1 + cc];
                    d_normal[sl.idx * 3 + cc] = v[4 + cc];
                }
#pragma unroll
                for (int i = 0; i < 7; ++i) v[i] = 0.f;
            }
        }
        if ((c + 1) * 64 > n - 1) {   // wave-uniform: this chunk holds slots that alias the last point
#pragma unroll
            for (int i = 0; i < 7; ++i) tail[i] += wave_sum(v[i]);
        }
        suffix_carry += lane_value<0>(sfx_incl);
    }
    ds_acc = wave_sum(ds_acc);
    if (lane == 0) {
        const int64_t li = (int64_t)off + n - 1;
        d_sdf[li] = tail[0];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            d_radiance[li * 3 + cc] = tail[1 + cc];
            d_normal[li * 3 + cc] = tail[4 + cc];
        }
        if (d_s_ray) d_s_ray[r] = ds_acc;
    }
}


// EikonalLoss on `normal_pts` (arcnerf/loss/geo_loss.py:12-70: weight * mean((|n| - 1)^2) over the dense (rays, p_dense) slots) from the
// PACKED normals, value and gradient in one pass: slot j of a ray is its point min(j, n - 1), so a ray's last point counts 1 + p_dense - n
// times; rays without points hold a unit default normal (no loss).  The dense tensor is never built.  d_normal receives (or, accumulate,
// is added) the gradient; loss[0] is added the loss (cleared by the launcher).
__global__ void __launch_bounds__(256)
eikonal_packed_kernel(const float *__restrict__ normal, const int32_t *__restrict__ ray_id, const int32_t *__restrict__ offsets, int64_t n_pts,
                      int p_dense, float scale, int accumulate, const float *__restrict__ add_src, int64_t ld_add, float *__restrict__ d_normal,
                      float *__restrict__ loss) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float term = 0.f;
    if (i < n_pts) {
        const int32_t r = ray_id[i];
        const int32_t off = offsets[r], n = offsets[r + 1] - off;
        const float mult = (i == (int64_t)off + n - 1) ? (float)(1 + (p_dense > n ? p_dense - n : 0)) : 1.0f;
        const float x = normal[3 * i], y = normal[3 * i + 1], z = normal[3 * i + 2];
        const float len = sqrtf(x * x + y * y + z * z);
        const float e = len - 1.0f;
        term = mult * e * e * scale;
        const float g = len > 0.f ? mult * 2.0f * e * scale / len : 0.f;
        float gx = g * x, gy = g * y, gz = g * z;
        if (add_src) { gx += add_src[i * ld_add]; gy += add_src[i * ld_add + 1]; gz += add_src[i * ld_add + 2]; }   // (a second incoming gradient)
        if (accumulate) {
            d_normal[3 * i] += gx; d_normal[3 * i + 1] += gy; d_normal[3 * i + 2] += gz;
        } else {
            d_normal[3 * i] = gx; d_normal[3 * i + 1] = gy; d_normal[3 * i + 2] = gz;
        }
    }
    // one atomic per workgroup (every wave adding to the ONE scalar serialised 2 000 same-address atomics: 27 us for a 2 us kernel)
    __shared__ float s_part[4];
    term = wave_sum(term);
    if (lane_id() == 0) s_part[threadIdx.x >> 6] = term;
    __syncthreads();
    if (loss && threadIdx.x == 0) {
        const float t = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        if (t != 0.f) atomicAdd(loss, t);
    }
}

}  // namespace arcn

using namespace arcn;

ARCN_EXPORT int arcn_neus_count(const int32_t *counts, const int32_t *kmax_dev, int64_t n_rays, int32_t *n_eval, void *stream) {
    if (n_rays <= 0) return ARCN_OK;
    if (!counts || !kmax_dev || !n_eval) return einval("neus_count: missing argument");
    hipLaunchKernelGGL(neus_count_kernel, dim3((unsigned)ceil_div<int64_t>(n_rays, 256)), dim3(256), 0, as_stream(stream), counts, kmax_dev,
                       n_rays, n_eval);
    return check_launch("neus_count");
}

ARCN_EXPORT int arcn_neus_sections(const float *zvals_dense, const int32_t *counts, const int32_t *offsets, int n_pts, float n_sample_cfg,
                                   int64_t n_rays, int p_dense, float *t_mid, float *sec_lo, float *sec_hi, int32_t *ray_id,
                                   int64_t *slot_map, void *stream) {
    if (n_rays <= 0) return ARCN_OK;
    if (!zvals_dense || !counts || !offsets || !t_mid || !sec_lo || !sec_hi || !ray_id || n_pts <= 0 || p_dense < 2)
        return einval("neus_sections: missing argument");
    hipLaunchKernelGGL(neus_sections_kernel, dim3((unsigned)ceil_div<int64_t>(n_rays, kNeusRaysPerBlock)), dim3(256), 0, as_stream(stream),
                       zvals_dense, counts, offsets, n_pts, n_sample_cfg, n_rays, p_dense, t_mid, sec_lo, sec_hi, ray_id, slot_map);
    return check_launch("neus_sections");
}

ARCN_EXPORT int arcn_neus_slots_fwd(const float *packed, const int32_t *offsets, int64_t n_rays, int p_dense, const float *dflt_host,
                                    float *dense, void *stream) {
    if (n_rays <= 0) return ARCN_OK;
    if (!packed || !offsets || !dense || !dflt_host || p_dense < 1) return einval("neus_slots_fwd: missing argument");
    hipLaunchKernelGGL(neus_slots_fwd_kernel<3>, dim3((unsigned)ceil_div<int64_t>(n_rays, kNeusRaysPerBlock)), dim3(256), 0, as_stream(stream),
                       packed, offsets, n_rays, p_dense, dflt_host[0], dflt_host[1], dflt_host[2], dense);
    return check_launch("neus_slots_fwd");
}

ARCN_EXPORT int arcn_neus_slots_bwd(const float *d_dense, const int32_t *offsets, int64_t n_rays, int p_dense, float *d_packed, void *stream) {
    if (n_rays <= 0) return ARCN_OK;
    if (!d_dense || !offsets || !d_packed || p_dense < 1) return einval("neus_slots_bwd: missing argument");
    hipLaunchKernelGGL(neus_slots_bwd_kernel<3>, dim3((unsigned)ceil_div<int64_t>(n_rays, kNeusRaysPerBlock)), dim3(256), 0, as_stream(stream),
                       d_dense, offsets, n_rays, p_dense, d_packed);
    return check_launch("neus_slots_bwd");
}

ARCN_EXPORT int arcn_eikonal_packed(const float *normal, const int32_t *ray_id, const int32_t *offsets, int64_t n_pts, int64_t n_rays, int p_dense,
                                    float weight, int accumulate, const float *add_src, int64_t ld_add, float *d_normal, float *loss,
                                    void *stream) {
    if (loss && !(accumulate & 2) && hipMemsetAsync(loss, 0, sizeof(float), as_stream(stream)) != hipSuccess) return check_launch("memset");
    if (n_pts <= 0 || n_rays <= 0) return ARCN_OK;
    if (!normal || !ray_id || !offsets || !d_normal || p_dense < 1 || (add_src && ld_add < 3)) return einval("eikonal_packed: missing argument");
    const float scale = weight / ((float)n_rays * (float)p_dense);
    hipLaunchKernelGGL(eikonal_packed_kernel, dim3((unsigned)ceil_div<int64_t>(n_pts, 256)), dim3(256), 0, as_stream(stream), normal, ray_id,
                       offsets, n_pts, p_dense, scale, accumulate & 1, add_src, ld_add, d_normal, loss);
    return check_launch("eikonal_packed");
}

static int neus_args(NeusIn &p, const float *sdf, const float *radiance, const float *normal, const float *t_mid, const float *sec_lo,
                     const float *sec_hi, const int32_t *offsets, const float *rays_d, const float *s_dev, float cos_anneal,
                     const float *bkg, int64_t bkg_rows, const int32_t *kmax_dev, float depth_far, const float *dflt_rgb,
                     const float *dflt_nrm, int64_t n_rays) {
    if (!sdf || !radiance || !normal || !t_mid || !sec_lo || !sec_hi || !offsets || !rays_d || !s_dev || !kmax_dev)
        return einval("neus_render: missing argument");
    if (bkg_rows != 0 && bkg_rows != 1 && bkg_rows != n_rays) return einval("neus_render: bkg must have 0, 1 or n_rays rows");
    p.sdf = sdf; p.radiance = radiance; p.normal = normal; p.t_mid = t_mid; p.lo = sec_lo; p.hi = sec_hi; p.rays_d = rays_d;
    p.s_ptr = s_dev; p.bkg = bkg; p.offsets = offsets; p.kmax_ptr = kmax_dev; p.bkg_rows = bkg_rows; p.R = n_rays;
    p.cos_anneal = cos_anneal; p.depth_far = depth_far;
    for (int c = 0; c < 3; ++c) {
        p.dflt_rgb[c] = dflt_rgb ? dflt_rgb[c] : 0.f;
        p.dflt_nrm[c] = dflt_nrm ? dflt_nrm[c] : 0.f;
    }
    return ARCN_OK;
}

ARCN_EXPORT int arcn_neus_render_fwd(const float *sdf, const float *radiance, const float *normal, const float *t_mid, const float *sec_lo,
                                     const float *sec_hi, const int32_t *offsets, const float *rays_d, const float *s_dev,
                                     float cos_anneal, const float *bkg, int64_t bkg_rows, const int32_t *kmax_dev, float depth_far,
                                     const float *dflt_rgb_host, const float *dflt_nrm_host, int64_t n_rays, float *rgb, float *depth,
                                     float *mask, float *nrm, float *t_last, void *stream) {
    if (n_rays <= 0) return ARCN_OK;
    if (!rgb || !depth || !mask || !nrm || !t_last) return einval("neus_render_fwd: missing output");
    NeusIn p;
    int rc = neus_args(p, sdf, radiance, normal, t_mid, sec_lo, sec_hi, offsets, rays_d, s_dev, cos_anneal, bkg, bkg_rows, kmax_dev,
                       depth_far, dflt_rgb_host, dflt_nrm_host, n_rays);
    if (rc) return rc;
    hipLaunchKernelGGL(neus_render_fwd_kernel, dim3((unsigned)ceil_div<int64_t>(n_rays, kNeusRaysPerBlock)), dim3(256), 0, as_stream(stream),
                       p, rgb, depth, mask, nrm, t_last);
    return check_launch("neus_render_fwd");
}

ARCN_EXPORT int arcn_neus_render_bwd(const float *sdf, const float *radiance, const float *normal, const float *t_mid, const float *sec_lo,
                                     const float *sec_hi, const int32_t *offsets, const float *rays_d, const float *s_dev,
                                     float cos_anneal, const float *bkg, int64_t bkg_rows, const int32_t *kmax_dev, int64_t n_rays,
                                     const float *d_rgb, const float *d_depth, const float *d_mask, const float *d_nrm,
                                     const float *d_tlast, float *d_sdf, float *d_radiance, float *d_normal, float *d_s_ray,
                                     void *stream) {
    if (n_rays <= 0) return ARCN_OK;
    if (!d_sdf || !d_radiance || !d_normal) return einval("neus_render_bwd: missing output");
    NeusIn p;
    int rc = neus_args(p, sdf, radiance, normal, t_mid, sec_lo, sec_hi, offsets, rays_d, s_dev, cos_anneal, bkg, bkg_rows, kmax_dev, 0.f,
                       nullptr, nullptr, n_rays);
    if (rc) return rc;
    hipLaunchKernelGGL(neus_render_bwd_kernel, dim3((unsigned)ceil_div<int64_t>(n_rays, kNeusRaysPerBlock)), dim3(256), 0, as_stream(stream),
                       p, d_rgb, d_depth, d_mask, d_nrm, d_tlast, d_sdf, d_radiance, d_normal, d_s_ray);
    return check_launch("neus_render_bwd");
}
