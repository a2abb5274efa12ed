// Frequency and spherical-harmonics encoders + elementwise activations (gfx950).
// FreqEmbedder.forward: arcnerf/models/base_modules/encoding/freq_encoder.py:65-88
// SHEmbedder torch branch: arcnerf/models/base_modules/encoding/sh_encoder.py:101-185 (polynomials evaluated on the
// (d+1)/2 mapped value, literally)
// get_activation / TruncExp: base_modules/activation.py:24-50, arcnerf/ops/trunc_exp.py:7-37
#include "common.hpp"

namespace arcn {

// one lane per output element: coalesced stores; sin/cos arguments x*2^k are exact scalings
__global__ void __launch_bounds__(256) freq_fwd_kernel(const float *__restrict__ x, int D, int n_freqs, int include_input,
                                                       float *__restrict__ out, int64_t n) {
    const int od = D * (include_input ? 1 : 0) + 2 * D * n_freqs;
    const int64_t total = n * od;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / od;
        int c = (int)(i - s * od);
        float v;
        if (include_input) {
            if (c < D) { out[i] = x[s * D + c]; continue; }
            c -= D;
        }
        const int f = c / (2 * D);
        const int rem = c - f * 2 * D;
        const int k = rem % D;
        const float a = x[s * D + k] * ldexpf(1.0f, f);
        v = rem < D ? sinf(a) : cosf(a);
        out[i] = v;
    }
}

// the same encoding written into n_cols >= od columns of a wider row-major buffer (row stride ld_out): columns od .. n_cols - 1 are set to
// zero (the pad of an odd width up to the 16-byte row granule of the products that read the buffer)
__global__ void __launch_bounds__(256) freq_fwd_cols_kernel(const float *__restrict__ x, int D, int n_freqs, int include_input,
                                                            float *__restrict__ out, int64_t ld_out, int n_cols, int64_t n) {
    const int od = D * (include_input ? 1 : 0) + 2 * D * n_freqs;
    const int64_t total = n * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / n_cols;
        int c = (int)(i - s * n_cols);
        float *o = out + s * ld_out + c;
        if (c >= od) { *o = 0.f; continue; }
        if (include_input) {
            if (c < D) { *o = x[s * D + c]; continue; }
            c -= D;
        }
        const int f = c / (2 * D);
        const int rem = c - f * 2 * D;
        const int k = rem % D;
        const float a = x[s * D + k] * ldexpf(1.0f, f);
        *o = rem < D ? sinf(a) : cosf(a);
    }
}

// J v for the encoding's Jacobian J = d enc / d x (od x D, block diagonal per input dimension): out[c] = v[k] for the identity block,
// 2^f cos(2^f x_k) v[k] for sin(2^f x_k), -2^f sin(2^f x_k) v[k] for cos(2^f x_k); columns od .. n_cols - 1 zero.  The adjoint of freq_bwd's
// J^T g: what the second differentiation of a normal n = J^T g (an Eikonal loss on a frequency-encoded sdf net) sends back to g.
__global__ void __launch_bounds__(256) freq_jvp_cols_kernel(const float *__restrict__ x, const float *__restrict__ v, int D, int n_freqs,
                                                            int include_input, float *__restrict__ out, int64_t ld_out, int n_cols, int64_t n) {
    const int od = D * (include_input ? 1 : 0) + 2 * D * n_freqs;
    const int64_t total = n * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / n_cols;
        int c = (int)(i - s * n_cols);
        float *o = out + s * ld_out + c;
        if (c >= od) { *o = 0.f; continue; }
        if (include_input) {
            if (c < D) { *o = v[s * D + c]; continue; }
            c -= D;
        }
        const int f = c / (2 * D);
        const int rem = c - f * 2 * D;
        const int k = rem % D;
        const float sc = ldexpf(1.0f, f);
        const float a = x[s * D + k] * sc;
        *o = (rem < D ? cosf(a) : -sinf(a)) * sc * v[s * D + k];
    }
}

__global__ void __launch_bounds__(256) freq_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dout, int D,
                                                       int n_freqs, int include_input, float *__restrict__ dx, int64_t n) {
    const int od = D * (include_input ? 1 : 0) + 2 * D * n_freqs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * D; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / D;
        const int k = (int)(i - s * D);
        const float *g = dout + s * od;
        float acc = 0.f;
        int c = 0;
        if (include_input) { acc += g[k]; c = D; }
        const float xv = x[i];
        for (int f = 0; f < n_freqs; ++f) {
            const float freq = ldexpf(1.0f, f);
            const float a = xv * freq;
            acc += g[c + k] * cosf(a) * freq;
            acc -= g[c + D + k] * sinf(a) * freq;
            c += 2 * D;
        }
        dx[i] = acc;
    }
}

__device__ __forceinline__ void sh_eval(float dx, float dy, float dz, int degree, float *o) {
    const float x = (dx + 1.0f) / 2.0f, y = (dy + 1.0f) / 2.0f, z = (dz + 1.0f) / 2.0f;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    int c = 0;
    o[c++] = 0.28209479177387814f;
    if (degree <= 1) return;
    o[c++] = -0.4886025119029199f * y;
    o[c++] = 0.4886025119029199f * z;
    o[c++] = -0.4886025119029199f * x;
    if (degree <= 2) return;
    o[c++] = 1.0925484305920792f * xy;
    o[c++] = -1.0925484305920792f * yz;
    o[c++] = 0.31539156525252005f * (3.0f * zz - 1.0f);
    o[c++] = -1.0925484305920792f * xz;
    o[c++] = 0.5462742152960396f * (xx - yy);
    if (degree <= 3) return;
    o[c++] = -0.5900435899266435f * y * (3.0f * xx - yy);
    o[c++] = 2.890611442640554f * xy * z;
    o[c++] = -0.4570457994644658f * y * (5.0f * zz - 1.0f);
    o[c++] = 0.3731763325901154f * z * (5.0f * zz - 3.0f);
    o[c++] = -0.4570457994644658f * x * (5.0f * zz - 1.0f);
    o[c++] = 1.445305721320277f * z * (xx - yy);
    o[c++] = -0.5900435899266435f * x * (xx - 3.0f * yy);
    if (degree <= 4) return;
    o[c++] = 2.5033429417967046f * xy * (xx - yy);
    o[c++] = -1.7701307697799304f * yz * (3.0f * xx - yy);
    o[c++] = 0.9461746957575601f * xy * (7.0f * zz - 1.0f);
    o[c++] = -0.6690465435572892f * yz * (7.0f * zz - 3.0f);
    o[c++] = 0.10578554691520431f * (zz * (35.0f * zz - 30.0f) + 3.0f);
    o[c++] = -0.6690465435572892f * xz * (7.0f * zz - 3.0f);
    o[c++] = 0.47308734787878004f * (xx - yy) * (7.0f * zz - 1.0f);
    o[c++] = -1.7701307697799304f * xz * (xx - 3.0f * yy);
    o[c++] = 0.6258357354491761f * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
}

__global__ void __launch_bounds__(256) sh_fwd_kernel(const float *__restrict__ dirs, int degree, int include_input,
                                                     float *__restrict__ out, int64_t n) {
    const int nsh = degree * degree;
    const int od = nsh + (include_input ? 3 : 0);
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (int64_t)gridDim.x * blockDim.x) {
        float o[25];
        const float dx = dirs[3 * s], dy = dirs[3 * s + 1], dz = dirs[3 * s + 2];
        sh_eval(dx, dy, dz, degree, o);
        float *dst = out + s * od;
        if (include_input) { dst[0] = dx; dst[1] = dy; dst[2] = dz; dst += 3; }
        for (int c = 0; c < nsh; ++c) dst[c] = o[c];
    }
}

// fuse_radiance_inputs (arcnerf/models/base_modules/geo_rad_model/encoder_mlp_network.py:93-118) in one pass for the hash-grid radiance nets:
// out (n, W) = the blocks of `mode` in order - p: the position (3), v: SH_degree(normalize(view dir)) (degree^2; normalize = v / (|v| + 1e-8),
// geometry/transformation.py:11-25), n: the normal (3), f: n_feat geometry features read at row stride ld_feat (a column slice of the
// geometry net's padded output).  Replaces norm + add + div + sh + concat (5 launches, 3 passes over the sample-sized tensors).
struct RadIn {
    int block[4];    // 0 p, 1 v, 2 n, 3 f; -1 unused
    int n_block, width;
};

// one wave = 64 consecutive samples: every lane builds its row in LDS, then the wave stores its 64 x W block - contiguous in memory - with
// consecutive lanes on consecutive words (row-wise stores by one lane each were 38 scattered 4-byte writes per lane: 34 us for 125 K rows)
__global__ void __launch_bounds__(256) radiance_inputs_kernel(RadIn m, const float *__restrict__ pts, const float *__restrict__ dirs,
                                                              const float *__restrict__ normals, const float *__restrict__ feat, int64_t ld_feat,
                                                              int n_feat, int degree, float *__restrict__ out, int64_t n) {
    extern __shared__ float rows[];          // 4 waves x 64 rows x width
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nsh = degree * degree, W = m.width;
    float *mine = rows + (size_t)wave * 64 * W;
    const int64_t n_blocks = (n + 255) / 256;
    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int64_t s0 = blk * 256 + (int64_t)wave * 64;
        const int64_t s = s0 + lane;
        if (s < n) {
            float *dst = mine + lane * W;
            for (int b = 0; b < m.n_block; ++b) {
                const int kind = m.block[b];
                if (kind == 0) { dst[0] = pts[3 * s]; dst[1] = pts[3 * s + 1]; dst[2] = pts[3 * s + 2]; dst += 3; }
                else if (kind == 1) {
                    const float x = dirs[3 * s], y = dirs[3 * s + 1], z = dirs[3 * s + 2];
                    const float len = sqrtf(x * x + y * y + z * z) + 1e-8f;
                    float o[25];
                    sh_eval(x / len, y / len, z / len, degree, o);
                    for (int c = 0; c < nsh; ++c) dst[c] = o[c];
                    dst += nsh;
                } else if (kind == 2) { dst[0] = normals[3 * s]; dst[1] = normals[3 * s + 1]; dst[2] = normals[3 * s + 2]; dst += 3; }
                else {
                    const float *f = feat + s * ld_feat;
                    for (int c = 0; c < n_feat; ++c) dst[c] = f[c];
                    dst += n_feat;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (s0 < n) {
            const int64_t rows_here = (n - s0) < 64 ? (n - s0) : 64;
            const int total = (int)rows_here * W;
            float *g = out + s0 * W;
            for (int i = lane; i < total; i += 64) g[i] = mine[i];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void __launch_bounds__(256) act_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n, int act,
                                                      float beta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = act_fwd(x[i], act, beta);
}

// y[i] = act(x[i * ld]) * scale for the first *n_ptr (or n) rows: the estimated opacity sigma * dt of an occupancy refresh straight from
// the density column of the geometry net's output
__global__ void __launch_bounds__(256) act_col_scale_kernel(const float *__restrict__ x, int64_t ld, float *__restrict__ y, int64_t n,
                                                            const int32_t *__restrict__ n_ptr, int act, float beta, float scale) {
    const int64_t cnt = dev_count(n, n_ptr);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = act_fwd(x[i * ld], act, beta) * scale;
}

__global__ void __launch_bounds__(256) act_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                      const float *__restrict__ dy, float *__restrict__ dx, int64_t n,
                                                      int act, float beta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * act_grad(x[i], y ? y[i] : act_fwd(x[i], act, beta), act, beta);
}

typedef float f4v __attribute__((ext_vector_type(4)));

// ---- softplus closed under differentiation (the sdf nets of NeuS: normals with create_graph, Eikonal loss through them again) -------
// y = softplus_beta(z) (torch threshold 20);  s = dy/dz = sigmoid(beta z);  ds/dz = beta s (1 - s).
//   softplus_grad : out = g * s                                   (first backward; ALSO the gradient of out with respect to g)
//   softplus_grad2: dg = h * s,  dz = h * g * beta s (1 - s)       (backward of softplus_grad for an incoming h: one pass over z, g, h
//                                                                   instead of torch's sigmoid / mul / sigmoid_backward chain)
// one float4 per lane (n a multiple of 4 and 16-byte aligned pointers on the vector path)
// from_y: the argument is y = softplus(z) itself (a layer that applies the activation in its product's epilogue keeps no z):
// s = 1 - e^(-beta y) exactly, and ds is then d s / d y = beta (1 - s) (the chain through y supplies the second factor s).
// (Round 6: on the hardware exponential - v_exp_f32, 1 ulp; the argument's rounding adds <= 1e-7 |beta z| relative, invisible in s and ds -
// instead of libm's expf + expm1f: ~65 VALU instructions per element made these passes, three to five (S, 256) tensors each, run at 4.8 TB/s
// where the same traffic streams at 6.5.  s = -expm1(-x) keeps its RELATIVE accuracy for small x through the series below 1/8.)
__device__ __forceinline__ void softplus_s(float z, float beta, int from_y, float &s, float &ds) {
    const float bv = beta * z;
    if (bv > 20.f) { s = 1.f; ds = 0.f; return; }
    if (from_y) {
        const float q = __builtin_amdgcn_exp2f(-bv * 1.44269504088896341f);       // 1 - s
        const float x = bv;
        const float ser = x * (1.0f - x * 0.5f * (1.0f - x * (1.0f / 3.0f) * (1.0f - x * 0.25f * (1.0f - x * 0.2f * (1.0f - x * (1.0f / 6.0f))))));
        s = (x < 0.125f && x > -0.125f) ? ser : 1.0f - q;
        ds = beta * q;
        return;
    }
    const float e = __builtin_amdgcn_exp2f(bv * 1.44269504088896341f), r = __builtin_amdgcn_rcpf(e + 1.f);
    s = e * r;
    ds = beta * s * r;
}

template <bool VEC>
__global__ void __launch_bounds__(256) softplus_grad_kernel(const float *__restrict__ z, const float *__restrict__ g, const float *__restrict__ g2,
                                                            float *__restrict__ out, int64_t n, float beta, int from_y) {
    constexpr int W = VEC ? 4 : 1;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * W; i < n; i += (int64_t)gridDim.x * blockDim.x * W) {
        float zv[W], gv[W], ov[W];
        if (VEC) {
            *reinterpret_cast<f4v *>(zv) = *reinterpret_cast<const f4v *>(z + i);
            if (g) *reinterpret_cast<f4v *>(gv) = *reinterpret_cast<const f4v *>(g + i);
            else gv[0] = gv[1] = gv[2] = gv[3] = 1.f;      // g == NULL: the derivative s itself
            if (g2) {                                      // a second gradient arriving at the same activation: summed on the way in
                float g2v[W];
                *reinterpret_cast<f4v *>(g2v) = *reinterpret_cast<const f4v *>(g2 + i);
#pragma unroll
                for (int k = 0; k < W; ++k) gv[k] += g2v[k];
            }
        } else { zv[0] = z[i]; gv[0] = (g ? g[i] : 1.f) + (g2 ? g2[i] : 0.f); }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float s, ds;
            softplus_s(zv[k], beta, from_y, s, ds);
            ov[k] = gv[k] * s;
        }
        if (VEC) *reinterpret_cast<f4v *>(out + i) = *reinterpret_cast<const f4v *>(ov);
        else out[i] = ov[0];
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256) softplus_grad2_kernel(const float *__restrict__ z, const float *__restrict__ g, const float *__restrict__ h,
                                                             float *__restrict__ dg, float *__restrict__ dz, int64_t n, float beta, int from_y) {
    constexpr int W = VEC ? 4 : 1;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * W; i < n; i += (int64_t)gridDim.x * blockDim.x * W) {
        float zv[W], gv[W], hv[W], a[W], b[W];
        if (VEC) {
            *reinterpret_cast<f4v *>(zv) = *reinterpret_cast<const f4v *>(z + i);
            *reinterpret_cast<f4v *>(gv) = *reinterpret_cast<const f4v *>(g + i);
            *reinterpret_cast<f4v *>(hv) = *reinterpret_cast<const f4v *>(h + i);
        } else { zv[0] = z[i]; gv[0] = g[i]; hv[0] = h[i]; }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float s, ds;
            softplus_s(zv[k], beta, from_y, s, ds);
            a[k] = hv[k] * s;
            b[k] = hv[k] * gv[k] * ds;
        }
        if (VEC) {
            if (dg) *reinterpret_cast<f4v *>(dg + i) = *reinterpret_cast<const f4v *>(a);
            if (dz) *reinterpret_cast<f4v *>(dz + i) = *reinterpret_cast<const f4v *>(b);
        } else {
            if (dg) dg[i] = a[0];
            if (dz) dz[i] = b[0];
        }
    }
}

// softplus_grad / softplus_grad2 for a gradient that is ONE ROW broadcast over the samples (the last hidden layer of the sdf net's normal
// chain: g = W_D[0], ops/sdf_chain.py).  The plain kernels wanted the row expanded to (n, H) first (a 134 MB copy per pass at 131072 x 256).
//   softplus_grad_row : out = g_row[col] * s
//   softplus_grad2_row: dz = h * g_row[col] * ds, colsum[col] += sum over the rows of h * s - the adjoint of the row itself, which is all the
//                       caller needs of dg (no (n, H) write, no column-sum pass).  Column-stationary threads like sdf_jac_dz2_kernel.
__global__ void __launch_bounds__(256) softplus_grad_row_kernel(const float *__restrict__ z, const float *__restrict__ g_row, float *__restrict__ out,
                                                                int64_t n, int H, float beta, int from_y) {
    const int64_t total = n * H;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const f4v zv = *reinterpret_cast<const f4v *>(z + i), gv = *reinterpret_cast<const f4v *>(g_row + (int)(i % H));
        f4v o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float sv, ds;
            softplus_s(zv[k], beta, from_y, sv, ds);
            o[k] = gv[k] * sv;
        }
        *reinterpret_cast<f4v *>(out + i) = o;
    }
}

__global__ void __launch_bounds__(256) softplus_grad2_row_kernel(const float *__restrict__ z, const float *__restrict__ g_row, const float *__restrict__ h,
                                                                 float *__restrict__ dz, float *__restrict__ colsum, int64_t n, int H, float beta,
                                                                 int from_y) {
    __shared__ float s_sum[256 * 4];
    const int tpr = H >> 2, rows_per_trip = 256 / tpr;
    const int col = (threadIdx.x % tpr) * 4, sub = threadIdx.x / tpr;
    const f4v gv = *reinterpret_cast<const f4v *>(g_row + col);
    f4v sum = {0.f, 0.f, 0.f, 0.f};
    const int64_t step = (int64_t)gridDim.x * rows_per_trip;
    for (int64_t row0 = (int64_t)blockIdx.x * rows_per_trip + sub; row0 < n; row0 += 4 * step) {
        f4v zv[4], hv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = row0 + q * step;
            if (row < n) {
                zv[q] = *reinterpret_cast<const f4v *>(z + row * H + col);
                hv[q] = *reinterpret_cast<const f4v *>(h + row * H + col);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = row0 + q * step;
            if (row < n) {
                f4v o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float sv, ds;
                    softplus_s(zv[q][k], beta, from_y, sv, ds);
                    sum[k] += hv[q][k] * sv;
                    o[k] = hv[q][k] * gv[k] * ds;
                }
                *reinterpret_cast<f4v *>(dz + row * H + col) = o;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s_sum[threadIdx.x * 4 + k] = sum[k];
    __syncthreads();
    for (int j = threadIdx.x; j < H; j += 256) {
        float t = 0.f;
        for (int q = 0; q < rows_per_trip; ++q) t += s_sum[((j >> 2) + q * tpr) * 4 + (j & 3)];
        if (t != 0.f) atomicAdd(colsum + j, t);
    }
}

// out (n, n_cols) = [a[:, :na] / div | b[:, :nb] / div | 0]: the input of the layer behind a skip connection ([h | e] / sqrt2,
// linear_network_module.py:174-197) and its adjoints, from two row-major sources read where they lie (row strides ld_a, ld_b; b may be NULL).
// "/ div" is torch's CUDA division by a python scalar - a product with the float reciprocal (BinaryDivTrueKernel.cu: a * (1 / b) when b is a
// CPU scalar) - so the result equals the module path's `x / math.sqrt(2)` bit for bit (inv = 1: the values themselves).  One thread per element.
__global__ void __launch_bounds__(256) concat2_div_kernel(const float *__restrict__ a, int64_t ld_a, int na, const float *__restrict__ b, int64_t ld_b,
                                                          int nb, float inv, float *__restrict__ out, int n_cols, int64_t n) {
    const int64_t total = n * n_cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / n_cols;
        const int c = (int)(e - i * n_cols);
        float v = 0.f;
        if (c < na) v = a[i * ld_a + c] * inv;
        else if (c < na + nb) v = b[i * ld_b + (c - na)] * inv;
        out[e] = v;
    }
}

// backward of the sdf net with an explicit Jacobian output (ops.autograd.SdfMlpJacFn): per element of the (n, H) hidden layer
//   dz = dh s + c_j u s (1 - s),  su = s u      (c = beta W2[0]; dz and su may alias dh and u)
__global__ void __launch_bounds__(256) sdf_jac_dz_kernel(const float *__restrict__ dh, const float *__restrict__ u, const float *__restrict__ s,
                                                         const float *__restrict__ c, float *__restrict__ dz, float *__restrict__ su, int64_t n, int H) {
    const int64_t total = n * H;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const f4v a = *reinterpret_cast<const f4v *>(dh + i), b = *reinterpret_cast<const f4v *>(u + i), sv = *reinterpret_cast<const f4v *>(s + i);
        const f4v cv = *reinterpret_cast<const f4v *>(c + (int)(i % H));
        f4v o1, o2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = sv[k] * b[k];
            o2[k] = t;
            o1[k] = a[k] * sv[k] + cv[k] * t * (1.0f - sv[k]);
        }
        *reinterpret_cast<f4v *>(dz + i) = o1;
        *reinterpret_cast<f4v *>(su + i) = o2;
    }
}

// ---- the tone mappers of HDR-NeRF (arcnerf/models/hdrnerf_model.py:44-75): per colour channel a 1 -> W -> 1 net, ReLU inside, sigmoid
// out, on ln(exposure) + log radiance of every sample.  As dense layers each channel moves a (samples, W) activation tensor through HBM
// four times per step for 2 W MACs per sample; here the hidden layer lives in registers.  params per channel: [w1 (W) | b1 (W) | w2 (W)
// | b2], x / y / dy / dx are (n, C) row-major, one channel per blockIdx.y.
constexpr int kToneW = 128;

__global__ void __launch_bounds__(256) tonemap_fwd_kernel(const float *__restrict__ x, const float *__restrict__ params, float *__restrict__ y, int64_t n, int C,
                                                          int W) {
    __shared__ float sw[3 * kToneW + 1];
    const int c = blockIdx.y;
    for (int i = threadIdx.x; i < 3 * W + 1; i += 256) sw[i] = params[(int64_t)c * (3 * W + 1) + i];
    __syncthreads();
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < n; s += (int64_t)gridDim.x * 256) {
        const float xv = x[s * C + c];
        float acc = sw[3 * W];
        for (int j = 0; j < W; ++j) acc += sw[2 * W + j] * fmaxf(sw[j] * xv + sw[W + j], 0.f);
        y[s * C + c] = 1.0f / (1.0f + expf(-acc));
    }
}

// one workgroup keeps its parameter-gradient partials in registers over all its 256-sample tiles: per tile, phase 1 with a thread per
// SAMPLE (dz, dx), phase 2 with a thread per HIDDEN UNIT over the tile's samples staged in LDS (the three sums of that unit: no
// cross-lane reduction), then one partial row per workgroup, added in workgroup order by tonemap_reduce_kernel (deterministic)
__global__ void __launch_bounds__(256) tonemap_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                                                          const float *__restrict__ params, float *__restrict__ dx, float *__restrict__ partials, int64_t n,
                                                          int C, int W) {
    __shared__ float sw[3 * kToneW + 1], sx[256], sdz[256], red[3 * kToneW], red2[256];
    const int c = blockIdx.y, tid = threadIdx.x, j = tid & 127, half = tid >> 7;
    for (int i = tid; i < 3 * W + 1; i += 256) sw[i] = params[(int64_t)c * (3 * W + 1) + i];
    __syncthreads();
    const float w1 = j < W ? sw[j] : 0.f, b1 = j < W ? sw[W + j] : 0.f, w2 = j < W ? sw[2 * W + j] : 0.f;
    float a_w1 = 0.f, a_b1 = 0.f, a_w2 = 0.f, a_b2 = 0.f;
    const int64_t tiles = (n + 255) / 256;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t s = t * 256 + tid;
        const bool ok = s < n;
        const float xv = ok ? x[s * C + c] : 0.f, yv = ok ? y[s * C + c] : 0.f, dyv = ok ? dy[s * C + c] : 0.f;
        const float dz = dyv * yv * (1.0f - yv);
        a_b2 += dz;
        if (dx) {
            float dxv = 0.f;
            for (int k = 0; k < W; ++k) dxv += (sw[k] * xv + sw[W + k] > 0.f) ? dz * sw[2 * W + k] * sw[k] : 0.f;
            if (ok) dx[s * C + c] = dxv;
        }
        sx[tid] = xv;
        sdz[tid] = dz;
        __syncthreads();
        if (j < W) {
            for (int i = half; i < 256; i += 2) {
                const float xs = sx[i], dzs = sdz[i];
                const float pre = w1 * xs + b1;
                const float dpre = pre > 0.f ? dzs * w2 : 0.f;
                a_w1 += dpre * xs;
                a_b1 += dpre;
                a_w2 += dzs * fmaxf(pre, 0.f);
            }
        }
        __syncthreads();
    }
    // combine the two sample halves of a hidden unit, and the workgroup's db2
    if (half == 1 && j < W) { red[j] = a_w1; red[kToneW + j] = a_b1; red[2 * kToneW + j] = a_w2; }
    red2[tid] = a_b2;
    __syncthreads();
    float *out = partials + ((int64_t)c * gridDim.x + blockIdx.x) * (3 * W + 1);
    if (half == 0 && j < W) {
        out[j] = a_w1 + red[j];
        out[W + j] = a_b1 + red[kToneW + j];
        out[2 * W + j] = a_w2 + red[2 * kToneW + j];
    }
    if (tid == 0) {
        float sum = 0.f;
        for (int i = 0; i < 256; ++i) sum += red2[i];
        out[3 * W] = sum;
    }
}

__global__ void __launch_bounds__(256) tonemap_reduce_kernel(const float *__restrict__ partials, float *__restrict__ dparams, int n_wg, int C, int W) {
    const int P = 3 * W + 1, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * P) return;
    const int c = i / P, k = i % P;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 3 < n_wg; b += 4) {
        a0 += partials[((int64_t)c * n_wg + b) * P + k];
        a1 += partials[((int64_t)c * n_wg + b + 1) * P + k];
        a2 += partials[((int64_t)c * n_wg + b + 2) * P + k];
        a3 += partials[((int64_t)c * n_wg + b + 3) * P + k];
    }
    for (; b < n_wg; ++b) a0 += partials[((int64_t)c * n_wg + b) * P + k];
    dparams[i] = (a0 + a1) + (a2 + a3);
}

// ---- geo -> radiance glue of Base3dModel._forward_pts_dir (arcnerf/models/base_3d_model.py:233-254) ---------------
// sigma = out_act(geo_out[:,0]) (EncoderMLPGeoNet.handle_output / FusedMLPGeoNet.handle_output_combine);
// rad_in = fuse_radiance_inputs(..) for modes 'fv' / 'vf' (encoder_mlp_network.py:93-118): geo feature slice and
// SH(normalize(view_dir)) concatenated in mode order; normalize = v / (|v| + 1e-8) (geometry/transformation.py:21).
// one lane per float4 of the output rows: loads and stores are fully coalesced (16 B per lane, consecutive lanes consecutive
// addresses).  Requires Wf, Wg, feat_off and deg^2 to be multiples of 4 (NGP: 16/16/0/16); other shapes take the row kernel.

// SH(normalize(d)) once per RAY: every sample of a ray shares its direction, the glue below then gathers the row by ray id
// instead of evaluating the polynomials per sample (4x over, one per output quad)
__global__ void __launch_bounds__(256)
ngp_ray_sh_kernel(const float *__restrict__ rays_d, int degree, float *__restrict__ sh_ray, int64_t n_rays) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz) + 1e-8f;
    float o[25];
    sh_eval(dx / nrm, dy / nrm, dz / nrm, degree, o);
    const int nsh = degree * degree;
    for (int c = 0; c < nsh; ++c) sh_ray[r * nsh + c] = o[c];
}

__global__ void __launch_bounds__(256)
ngp_glue_fwd_vec_kernel(const float *__restrict__ geo_out, const float *__restrict__ dirs, const float *__restrict__ sh_ray,
                        const int32_t *__restrict__ ray_id, int Wg, int feat_off, int Wf,
                        int degree, int feat_first, int sigma_act, float *__restrict__ rad_in, float *__restrict__ sigma,
                        int64_t n, const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int nsh = degree * degree, W = Wf + nsh, Q = W >> 2, Qf = Wf >> 2;
    const int64_t total = cnt * Q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / Q;
        const int q = (int)(i - s * Q);
        const int qf = feat_first ? q : q - (nsh >> 2);     // index inside the feature part
        const int qs = feat_first ? q - Qf : q;              // index inside the SH part
        f4v v;
        if (qf >= 0 && qf < Qf) {
            v = *reinterpret_cast<const f4v *>(geo_out + s * Wg + feat_off + 4 * qf);
            if (qf == 0 && sigma && feat_off == 0) sigma[s] = act_fwd(v.x, sigma_act, 1.0f);
        } else if (sh_ray) {
            v = *reinterpret_cast<const f4v *>(sh_ray + (int64_t)ray_id[s] * nsh + 4 * qs);
        } else {
            float dx = dirs[3 * s], dy = dirs[3 * s + 1], dz = dirs[3 * s + 2];
            float nrm = sqrtf((dx * dx + dy * dy) + dz * dz) + 1e-8f;
            float o[25];
            sh_eval(dx / nrm, dy / nrm, dz / nrm, degree, o);
            v = f4v{o[4 * qs], o[4 * qs + 1], o[4 * qs + 2], o[4 * qs + 3]};
        }
        *reinterpret_cast<f4v *>(rad_in + s * W + 4 * q) = v;
        if (q == 0 && sigma && feat_off != 0) sigma[s] = act_fwd(geo_out[s * Wg], sigma_act, 1.0f);
    }
}

__global__ void __launch_bounds__(256)
ngp_glue_bwd_vec_kernel(const float *__restrict__ geo_out, const float *__restrict__ d_rad_in, const float *__restrict__ d_sigma,
                        int Wg, int feat_off, int Wf, int degree, int feat_first, int sigma_act,
                        float *__restrict__ d_geo_out, int64_t n, const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int nsh = degree * degree, W = Wf + nsh, Q = Wg >> 2;
    const int64_t total = cnt * Q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / Q;
        const int q = (int)(i - s * Q);
        const int c0 = 4 * q;
        f4v v = {0.f, 0.f, 0.f, 0.f};
        if (c0 >= feat_off && c0 + 3 < feat_off + Wf)
            v = *reinterpret_cast<const f4v *>(d_rad_in + s * W + (feat_first ? 0 : nsh) + (c0 - feat_off));
        if (q == 0 && d_sigma) {
            const float x0 = geo_out[s * Wg];
            v.x += d_sigma[s] * act_grad(x0, act_fwd(x0, sigma_act, 1.0f), sigma_act, 1.0f);
        }
        *reinterpret_cast<f4v *>(d_geo_out + s * Wg + c0) = v;
    }
}

__global__ void __launch_bounds__(256)
ngp_glue_fwd_kernel(const float *__restrict__ geo_out, const float *__restrict__ dirs, int Wg, int feat_off, int Wf,
                    int degree, int feat_first, int sigma_act, float *__restrict__ rad_in, float *__restrict__ sigma,
                    int64_t n, const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int nsh = degree * degree, W = Wf + nsh;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cnt; s += (int64_t)gridDim.x * blockDim.x) {
        const float *g = geo_out + s * Wg;
        if (sigma) sigma[s] = act_fwd(g[0], sigma_act, 1.0f);
        float *dst = rad_in + s * W;
        float *fdst = feat_first ? dst : dst + nsh;
        float *sdst = feat_first ? dst + Wf : dst;
        for (int c = 0; c < Wf; ++c) fdst[c] = g[feat_off + c];
        if (nsh > 0) {
            float dx = dirs[3 * s], dy = dirs[3 * s + 1], dz = dirs[3 * s + 2];
            float nrm = sqrtf((dx * dx + dy * dy) + dz * dz) + 1e-8f;
            float o[25];
            sh_eval(dx / nrm, dy / nrm, dz / nrm, degree, o);
            for (int c = 0; c < nsh; ++c) sdst[c] = o[c];
        }
    }
}

__global__ void __launch_bounds__(256)
ngp_glue_bwd_kernel(const float *__restrict__ geo_out, const float *__restrict__ d_rad_in, const float *__restrict__ d_sigma,
                    int Wg, int feat_off, int Wf, int degree, int feat_first, int sigma_act,
                    float *__restrict__ d_geo_out, int64_t n, const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int nsh = degree * degree, W = Wf + nsh;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cnt; s += (int64_t)gridDim.x * blockDim.x) {
        const float *src = d_rad_in + s * W + (feat_first ? 0 : nsh);
        float *dg = d_geo_out + s * Wg;
        for (int c = 0; c < Wg; ++c) {
            float v = (c >= feat_off && c < feat_off + Wf) ? src[c - feat_off] : 0.f;
            if (c == 0 && d_sigma) {
                const float x0 = geo_out[s * Wg];
                v += d_sigma[s] * act_grad(x0, act_fwd(x0, sigma_act, 1.0f), sigma_act, 1.0f);
            }
            dg[c] = v;
        }
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t b = ceil_div<int64_t>(n, 256);
    return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace arcn

using namespace arcn;

ARCN_EXPORT int arcn_freq_fwd(const float *x, int D, int n_freqs, int include_input, float *out, int64_t n, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !out || D < 1 || n_freqs < 0 || (n_freqs == 0 && !include_input)) return einval("freq_fwd: bad argument");
    const int od = D * (include_input ? 1 : 0) + 2 * D * n_freqs;
    hipLaunchKernelGGL(freq_fwd_kernel, dim3(grid_for(n * od)), dim3(256), 0, as_stream(stream), x, D, n_freqs,
                       include_input, out, n);
    return check_launch("freq_fwd");
}

/* arcn_freq_fwd into columns of a wider buffer: row s of the encoding goes to out[s * ld_out + 0 .. od - 1], the columns od .. n_cols - 1
 * behind it are zeroed (freq_encoder.py:10-88; the skip / radiance-input concatenations of linear_network_module.py:174-197 and
 * encoder_mlp_network.py:62-118 then need no copy of the encoding) */
ARCN_EXPORT int arcn_freq_fwd_cols(const float *x, int D, int n_freqs, int include_input, float *out, int64_t ld_out, int n_cols, int64_t n,
                                   void *stream) {
    if (n <= 0) return ARCN_OK;
    const int od = D * (include_input ? 1 : 0) + 2 * D * n_freqs;
    if (!x || !out || D < 1 || n_freqs < 0 || (n_freqs == 0 && !include_input) || n_cols < od || ld_out < n_cols)
        return einval("freq_fwd_cols: bad argument");
    hipLaunchKernelGGL(freq_fwd_cols_kernel, dim3(grid_for(n * n_cols)), dim3(256), 0, as_stream(stream), x, D, n_freqs, include_input, out, ld_out,
                       n_cols, n);
    return check_launch("freq_fwd_cols");
}

/* out (n rows, n_cols >= od columns at row stride ld_out) = (d enc / d x) v for v (n, D): the adjoint of arcn_freq_bwd with respect to its
 * dout (freq_encoder.py:10-88 differentiated twice: base_network.py:30-44 takes normals = d sdf / d x with create_graph = True) */
ARCN_EXPORT int arcn_freq_jvp_cols(const float *x, const float *v, int D, int n_freqs, int include_input, float *out, int64_t ld_out, int n_cols,
                                   int64_t n, void *stream) {
    if (n <= 0) return ARCN_OK;
    const int od = D * (include_input ? 1 : 0) + 2 * D * n_freqs;
    if (!x || !v || !out || D < 1 || n_freqs < 0 || (n_freqs == 0 && !include_input) || n_cols < od || ld_out < n_cols)
        return einval("freq_jvp_cols: bad argument");
    hipLaunchKernelGGL(freq_jvp_cols_kernel, dim3(grid_for(n * n_cols)), dim3(256), 0, as_stream(stream), x, v, D, n_freqs, include_input, out,
                       ld_out, n_cols, n);
    return check_launch("freq_jvp_cols");
}

ARCN_EXPORT int arcn_freq_bwd(const float *x, const float *dout, int D, int n_freqs, int include_input, float *dx,
                              int64_t n, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !dout || !dx || D < 1 || n_freqs < 0) return einval("freq_bwd: bad argument");
    hipLaunchKernelGGL(freq_bwd_kernel, dim3(grid_for(n * D)), dim3(256), 0, as_stream(stream), x, dout, D, n_freqs,
                       include_input, dx, n);
    return check_launch("freq_bwd");
}

ARCN_EXPORT int arcn_sh_fwd(const float *dirs, int degree, int include_input, float *out, int64_t n, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!dirs || !out || degree < 1 || degree > 5) return einval("sh_fwd: degree must be 1..5");
    hipLaunchKernelGGL(sh_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dirs, degree, include_input, out, n);
    return check_launch("sh_fwd");
}

ARCN_EXPORT int arcn_radiance_inputs(const char *mode_host, const float *pts, const float *dirs, const float *normals, const float *feat,
                                     int64_t ld_feat, int n_feat, int sh_degree, float *out, int64_t n, void *stream) {
    if (!mode_host) return einval("radiance_inputs: mode missing");
    RadIn m;
    m.n_block = 0;
    m.width = 0;
    for (const char *c = mode_host; *c; ++c) {
        if (m.n_block == 4) return einval("radiance_inputs: mode is a string of at most 4 of p, v, n, f");
        int kind;
        if (*c == 'p') { kind = 0; if (!pts) return einval("radiance_inputs: p needs pts"); m.width += 3; }
        else if (*c == 'v') { kind = 1; if (!dirs || sh_degree < 1 || sh_degree > 5) return einval("radiance_inputs: v needs dirs and an SH degree 1..5"); m.width += sh_degree * sh_degree; }
        else if (*c == 'n') { kind = 2; if (!normals) return einval("radiance_inputs: n needs normals"); m.width += 3; }
        else if (*c == 'f') { kind = 3; if (!feat || n_feat < 1 || ld_feat < n_feat) return einval("radiance_inputs: f needs features and their row stride"); m.width += n_feat; }
        else return einval("radiance_inputs: mode is a string over p, v, n, f");
        m.block[m.n_block++] = kind;
    }
    if (n <= 0 || m.n_block == 0) return ARCN_OK;
    if (!out) return einval("radiance_inputs: out missing");
    const size_t lds = sizeof(float) * 256 * (size_t)m.width;
    if (lds > 64 * 1024) return einval("radiance_inputs: rows wider than 64 floats");
    hipLaunchKernelGGL(radiance_inputs_kernel, dim3(grid_for(n)), dim3(256), lds, as_stream(stream), m, pts, dirs, normals, feat, ld_feat, n_feat,
                       sh_degree, out, n);
    return check_launch("radiance_inputs");
}

__global__ void __launch_bounds__(256) act_bwd_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ g,
                                                          float *__restrict__ ddy, float *__restrict__ d2x, int64_t n, int act, float beta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i], gi = g[i];
        if (ddy) ddy[i] = gi * act_grad(v, act_fwd(v, act, beta), act, beta);
        if (d2x) d2x[i] = gi * dy[i] * act_grad2(v, act, beta);
    }
}

ARCN_EXPORT int arcn_act_bwd_bwd(const float *x, const float *dy, const float *g, float *ddy, float *d2x, int64_t n, int act, float beta,
                                 void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !dy || !g || (!ddy && !d2x)) return einval("act_bwd_bwd: missing argument");
    hipLaunchKernelGGL(act_bwd_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, dy, g, ddy, d2x, n, act, beta);
    return check_launch("act_bwd_bwd");
}

ARCN_EXPORT int arcn_act_fwd(const float *x, float *y, int64_t n, int act, float beta, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !y) return einval("act_fwd: missing argument");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, y, n, act, beta);
    return check_launch("act_fwd");
}

/* y (n) = act(x[:, 0]) * scale for x (n rows at stride ld): get_est_opacity = sigma * dt of an occupancy refresh (base_3d_model.py:368-389)
 * from the density column of the geometry net's output in one pass; rows behind *n_ptr (may be NULL) are left alone */
ARCN_EXPORT int arcn_act_col_scale(const float *x, int64_t ld, float *y, int64_t n, const int32_t *n_ptr, int act, float beta, float scale,
                                   void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !y || ld < 1) return einval("act_col_scale: missing / invalid argument");
    hipLaunchKernelGGL(act_col_scale_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, ld, y, n, n_ptr, act, beta, scale);
    return check_launch("act_col_scale");
}

ARCN_EXPORT int arcn_act_bwd(const float *x, const float *y, const float *dy, float *dx, int64_t n, int act, float beta,
                             void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !dy || !dx) return einval("act_bwd: missing argument");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, y, dy, dx, n, act, beta);
    return check_launch("act_bwd");
}

static inline bool vec4_ok(int64_t n, const void *a, const void *b, const void *c, const void *d, const void *e) {
    uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e;
    return (n & 3) == 0 && (m & 15u) == 0;
}

static int softplus_grad_launch(const float *z, const float *g, const float *g2, float *out, int64_t n, float beta, int from_y, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!z || !out) return einval("softplus_grad: missing argument");
    if (vec4_ok(n, z, g, out, g2, nullptr))
        hipLaunchKernelGGL(softplus_grad_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, as_stream(stream), z, g, g2, out, n, beta, from_y);
    else
        hipLaunchKernelGGL(softplus_grad_kernel<false>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), z, g, g2, out, n, beta, from_y);
    return check_launch("softplus_grad");
}

ARCN_EXPORT int arcn_softplus_grad(const float *z, const float *g, float *out, int64_t n, float beta, int from_y, void *stream) {
    return softplus_grad_launch(z, g, nullptr, out, n, beta, from_y, stream);
}

/* out = (g + g2) * sigmoid(beta z): arcn_softplus_grad for an activation that two gradients arrive at (the ordinary chain and the
 * curvature term of a normal's second differentiation, base_network.py:30-44), without the pass that adds them first */
ARCN_EXPORT int arcn_softplus_grad_sum(const float *z, const float *g, const float *g2, float *out, int64_t n, float beta, int from_y, void *stream) {
    if (!g || !g2) return einval("softplus_grad_sum: two gradients expected");
    return softplus_grad_launch(z, g, g2, out, n, beta, from_y, stream);
}

ARCN_EXPORT int arcn_softplus_grad2(const float *z, const float *g, const float *h, float *dg, float *dz, int64_t n, float beta, int from_y,
                                    void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!z || !g || !h || (!dg && !dz)) return einval("softplus_grad2: missing argument");
    if (vec4_ok(n, z, g, h, dg, dz))
        hipLaunchKernelGGL(softplus_grad2_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, as_stream(stream), z, g, h, dg, dz, n, beta, from_y);
    else
        hipLaunchKernelGGL(softplus_grad2_kernel<false>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), z, g, h, dg, dz, n, beta, from_y);
    return check_launch("softplus_grad2");
}

ARCN_EXPORT int arcn_softplus_grad_row(const float *z, const float *g_row, float *out, int64_t n, int H, float beta, int from_y, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!z || !g_row || !out || H < 4 || (H & 3) != 0) return einval("softplus_grad_row: missing argument or width not a multiple of 4");
    if (((uintptr_t)z | (uintptr_t)g_row | (uintptr_t)out) & 15u) return einval("softplus_grad_row: 16-byte aligned tensors");
    hipLaunchKernelGGL(softplus_grad_row_kernel, dim3(grid_for(n * H / 4)), dim3(256), 0, as_stream(stream), z, g_row, out, n, H, beta, from_y);
    return check_launch("softplus_grad_row");
}

ARCN_EXPORT int arcn_softplus_grad2_row(const float *z, const float *g_row, const float *h, float *dz, float *colsum, int64_t n, int H, float beta,
                                        int from_y, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!z || !g_row || !h || !dz || !colsum || H < 4 || (H & 3) != 0 || H > 1024 || 256 % (H >> 2) != 0)
        return einval("softplus_grad2_row: missing argument or width not 4 * (a divisor of 256)");
    if (((uintptr_t)z | (uintptr_t)g_row | (uintptr_t)h | (uintptr_t)dz) & 15u) return einval("softplus_grad2_row: 16-byte aligned tensors");
    const int rows_per_trip = 256 / (H >> 2);
    int64_t blocks = ceil_div<int64_t>(n, rows_per_trip * 8);
    if (blocks > 256) blocks = 256;      // (one same-address float atomic per column and workgroup, see arcn_sdf_jac_dz2)
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(softplus_grad2_row_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), z, g_row, h, dz, colsum, n, H, beta, from_y);
    return check_launch("softplus_grad2_row");
}

ARCN_EXPORT int arcn_concat2_div(const float *a, int64_t ld_a, int na, const float *b, int64_t ld_b, int nb, float div, float *out, int n_cols,
                                 int64_t n, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!a || !out || na < 1 || ld_a < na || nb < 0 || (nb > 0 && (!b || ld_b < nb)) || n_cols < na + nb || !(div != 0.f))
        return einval("concat2_div: missing / invalid argument");
    hipLaunchKernelGGL(concat2_div_kernel, dim3(grid_for(n * n_cols)), dim3(256), 0, as_stream(stream), a, ld_a, na, b, ld_b, nb, 1.0f / div, out, n_cols, n);
    return check_launch("concat2_div");
}

ARCN_EXPORT int arcn_sdf_jac_dz(const float *dh, const float *u, const float *s, const float *c, float *dz, float *su, int64_t n, int H, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!dh || !u || !s || !c || !dz || !su || H < 4 || (H & 3) != 0) return einval("sdf_jac_dz: missing argument or hidden width not a multiple of 4");
    if (((uintptr_t)dh | (uintptr_t)u | (uintptr_t)s | (uintptr_t)c | (uintptr_t)dz | (uintptr_t)su) & 15u) return einval("sdf_jac_dz: 16-byte aligned tensors");
    hipLaunchKernelGGL(sdf_jac_dz_kernel, dim3(grid_for(n * H / 4)), dim3(256), 0, as_stream(stream), dh, u, s, c, dz, su, n, H);
    return check_launch("sdf_jac_dz");
}

static int tonemap_wgs(int64_t n) {
    int64_t t = (n + 255) / 256;
    return (int)(t < 1 ? 1 : (t > 512 ? 512 : t));
}

ARCN_EXPORT int arcn_tonemap_fwd(const float *x, const float *params, float *y, int64_t n, int C, int W, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !params || !y || C < 1 || W < 1 || W > kToneW) return einval("tonemap_fwd: missing argument or hidden width outside 1..128");
    int64_t gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(tonemap_fwd_kernel, dim3((unsigned)gx, (unsigned)C), dim3(256), 0, as_stream(stream), x, params, y, n, C, W);
    return check_launch("tonemap_fwd");
}

ARCN_EXPORT int64_t arcn_tonemap_scratch_floats(int64_t n, int C, int W) { return (int64_t)tonemap_wgs(n) * C * (3 * W + 1); }

ARCN_EXPORT int arcn_tonemap_bwd(const float *x, const float *y, const float *dy, const float *params, float *dx, float *dparams, float *scratch,
                                 int64_t scratch_floats, int64_t n, int C, int W, void *stream) {
    if (!x || !y || !dy || !params || !dparams || !scratch || C < 1 || W < 1 || W > kToneW)
        return einval("tonemap_bwd: missing argument or hidden width outside 1..128");
    if (scratch_floats < arcn_tonemap_scratch_floats(n, C, W)) return einval("tonemap_bwd: scratch smaller than arcn_tonemap_scratch_floats");
    const int wgs = n > 0 ? tonemap_wgs(n) : 0;
    if (n > 0)
        hipLaunchKernelGGL(tonemap_bwd_kernel, dim3((unsigned)wgs, (unsigned)C), dim3(256), 0, as_stream(stream), x, y, dy, params, dx, scratch, n, C, W);
    hipLaunchKernelGGL(tonemap_reduce_kernel, dim3((unsigned)((C * (3 * W + 1) + 255) / 256)), dim3(256), 0, as_stream(stream), scratch, dparams, wgs, C, W);
    return check_launch("tonemap_bwd");
}

ARCN_EXPORT int arcn_ngp_glue_fwd(const float *geo_out, const float *dirs, int Wg, int feat_off, int Wf, int sh_degree,
                                  int feat_first, int sigma_act, float *rad_in, float *sigma, int64_t n,
                                  const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!geo_out || !rad_in || Wg < 1 || Wf < 0 || feat_off < 0 || feat_off + Wf > Wg || sh_degree < 0 || sh_degree > 5 ||
        (sh_degree > 0 && !dirs))
        return einval("ngp_glue_fwd: bad argument");
    const int nsh = sh_degree * sh_degree;
    if ((Wg & 3) == 0 && (Wf & 3) == 0 && (nsh & 3) == 0 && feat_off == 0)
        hipLaunchKernelGGL(ngp_glue_fwd_vec_kernel, dim3(grid_for(n * ((Wf + nsh) >> 2))), dim3(256), 0, as_stream(stream), geo_out,
                           dirs, static_cast<const float *>(nullptr), static_cast<const int32_t *>(nullptr), Wg, feat_off, Wf, sh_degree,
                           feat_first, sigma_act, rad_in, sigma, n, n_ptr);
    else
        hipLaunchKernelGGL(ngp_glue_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), geo_out, dirs, Wg, feat_off,
                           Wf, sh_degree, feat_first, sigma_act, rad_in, sigma, n, n_ptr);
    return check_launch("ngp_glue_fwd");
}

ARCN_EXPORT int arcn_ngp_ray_sh(const float *rays_d, int sh_degree, float *sh_ray, int64_t n_rays, void *stream) {
    if (n_rays <= 0) return ARCN_OK;
    if (!rays_d || !sh_ray || sh_degree < 1 || sh_degree > 5) return einval("ngp_ray_sh: degree must be 1..5");
    hipLaunchKernelGGL(ngp_ray_sh_kernel, dim3((unsigned)ceil_div<int64_t>(n_rays, 256)), dim3(256), 0, as_stream(stream), rays_d,
                       sh_degree, sh_ray, n_rays);
    return check_launch("ngp_ray_sh");
}

ARCN_EXPORT int arcn_ngp_glue_fwd_rays(const float *geo_out, const float *sh_ray, const int32_t *ray_id, int Wg, int feat_off,
                                       int Wf, int sh_degree, int feat_first, int sigma_act, float *rad_in, float *sigma,
                                       int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    const int nsh = sh_degree * sh_degree;
    if (!geo_out || !rad_in || !sh_ray || !ray_id || Wg < 1 || Wf < 0 || feat_off != 0 || Wf > Wg || sh_degree < 1 || sh_degree > 5 ||
        (Wg & 3) || (Wf & 3) || (nsh & 3))
        return einval("ngp_glue_fwd_rays: needs feat_off 0 and Wg, Wf, degree^2 multiples of 4");
    hipLaunchKernelGGL(ngp_glue_fwd_vec_kernel, dim3(grid_for(n * ((Wf + nsh) >> 2))), dim3(256), 0, as_stream(stream), geo_out,
                       static_cast<const float *>(nullptr), sh_ray, ray_id, Wg, feat_off, Wf, sh_degree, feat_first, sigma_act, rad_in,
                       sigma, n, n_ptr);
    return check_launch("ngp_glue_fwd_rays");
}

ARCN_EXPORT int arcn_ngp_glue_bwd(const float *geo_out, const float *d_rad_in, const float *d_sigma, int Wg, int feat_off,
                                  int Wf, int sh_degree, int feat_first, int sigma_act, float *d_geo_out, int64_t n,
                                  const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!geo_out || !d_rad_in || !d_geo_out || Wg < 1 || feat_off + Wf > Wg) return einval("ngp_glue_bwd: bad argument");
    const int nsh_b = sh_degree * sh_degree;
    if ((Wg & 3) == 0 && (Wf & 3) == 0 && (nsh_b & 3) == 0 && feat_off == 0)
        hipLaunchKernelGGL(ngp_glue_bwd_vec_kernel, dim3(grid_for(n * (Wg >> 2))), dim3(256), 0, as_stream(stream), geo_out, d_rad_in,
                           d_sigma, Wg, feat_off, Wf, sh_degree, feat_first, sigma_act, d_geo_out, n, n_ptr);
    else
        hipLaunchKernelGGL(ngp_glue_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), geo_out, d_rad_in, d_sigma, Wg,
                           feat_off, Wf, sh_degree, feat_first, sigma_act, d_geo_out, n, n_ptr);
    return check_launch("ngp_glue_bwd");
}
