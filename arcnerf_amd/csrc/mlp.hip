// Fully fused small MLP on gfx950 matrix cores (exact f32 MFMA, v_mfma_f32_16x16x4_f32).
//
// Replaces tcnn.Network(FullyFusedMLP) (arcnerf/models/base_modules/geo_rad_model/tcnn_fusedmlp_module.py:66-77,
// 162-173) and skip-free GeoNet/RadianceNet stacks (linear_network_module.py:174-197,318-335):
//     y_i = act_i(y_{i-1} W_i^T + b_i),   W_i (dims[i+1], dims[i]) row-major like torch.nn.Linear.weight.
//
// Orientation: every layer is computed TRANSPOSED, Y^T = W . X^T, i.e. MFMA rows = output neurons, MFMA columns =
// samples.  With the 16x16x4 f32 MFMA the accumulator of lane (g = lane>>4, j = lane&15) holds neurons 4g..4g+3 of
// sample j, and the B operand of a K-step wants, in lane (g, j), input-neuron k = f(g, step) of sample j.  Choosing
// the K order k = 16t + 4g + ks (t = input tile, ks = 0..3) makes register `ks` of the previous layer's accumulator
// tile t EXACTLY the B operand — activations never leave registers between layers, no LDS round trip, no shuffles.
// The same permutation applied to the A operand means lane (i = lane&15, g) needs W[16mt+i][16t+4g+0..3]: 16 contiguous
// bytes; weights are staged once per workgroup into LDS in that fragment order so each (mt, t) tile is one conflict-
// free ds_read_b128 per lane reused by all NT sample tiles.
// f32 MFMA is bit-for-bit an fmaf-free k-ordered f32 chain (MI355X guide §3), so results match an fp32 torch/CPU MLP to
// summation-order noise (~1e-7 rel) — this is what lets the path meet the 1e-4 RGB parity bar without fp16.
// Roofline: 18.8 kFLOP/sample fwd for the NGP nets; at the 1e8 samples/s target that is 1.9 TFLOP/s of the 157 TFLOP/s
// f32 matrix peak — the MLP is bound by its operand traffic (128 B/sample features in, 16..64 B out), not by MFMA.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "adam.hpp"

namespace arcn {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kMaxLayers = 8;

struct MlpParams {
    int32_t n_layers;
    int32_t dims[kMaxLayers + 1];
    int32_t w_off[kMaxLayers];    // float offset of W_i in `weights`
    int32_t b_off[kMaxLayers];    // float offset of b_i in `biases`
    int32_t lds_off[kMaxLayers];  // float offset of the packed fragments of layer i in LDS
    int32_t act_hidden, act_out, has_bias;
    float beta;
};

__host__ __device__ inline int tiles16(int d) { return (d + 15) >> 4; }

// derivative of the activation expressed through the post-activation value y
__device__ __forceinline__ float act_grad_from_y(float y, int act, float beta) {
    switch (act) {
    case ARCN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case ARCN_ACT_SIGMOID: return y * (1.0f - y);
    case ARCN_ACT_TRUNCEXP: { float c = y < 3.0590232e-7f ? 3.0590232e-7f : (y > 3269017.4f ? 3269017.4f : y); return c; }
    case ARCN_ACT_SOFTPLUS: return 1.0f - expf(-beta * y);
    case ARCN_ACT_SQUAREPLUS: { float Y = 10.f * y, t = Y * Y; return t / (t + 1.f); }
    // (ARCN_ACT_SINE has no derivative in terms of y: build_mlp_params refuses it for the backward)
    default: return 1.f;
    }
}

// Stage W (rows x cols, row-major, leading dimension ld) — or its transpose — into LDS in MFMA A-fragment order:
//   element A[r][c] -> lds[((mt*T + t)*64 + g*16 + i)*4 + ks],  r = 16mt+i, c = 16t+4g+ks, zero padded to 16 multiples.
// TRANSPOSED: A = W^T, i.e. A[r][c] = W[c][r] with W (cols x rows).
// (Round 2 tried the products of these kernels as split-bf16 MFMAs - two planes: 5.6 % faster but the table gradient misses its 1e-3 bar;
// three planes: exact and SLOWER than f32 MFMA here - and the code was removed in round 3; DESIGN.md section 5 keeps the numbers.)
constexpr int kFragLane = 4;                                   // floats per lane of a weight fragment
constexpr int kFragTile = 64 * kFragLane;                      // floats per 16 x 16 fragment

template <bool TRANSPOSED>
__device__ __forceinline__ void stage_fragments(float *lds, const float *__restrict__ W, int rows, int cols) {
    const int MT = tiles16(rows), T = tiles16(cols);
    const int total = MT * T * 256;
    // kBatch independent loads are issued before the first LDS store: a plain load -> store loop runs at L2 latency per
    // element (12..28 trips per thread for the NGP nets, ~20 us at the start of every workgroup)
    constexpr int kBatch = 8;
    for (int e0 = threadIdx.x; e0 < total; e0 += blockDim.x * kBatch) {
        float v[kBatch];
        int dst[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int e = e0 + u * blockDim.x;
            int r, c;
            if (TRANSPOSED) { c = e / (MT * 16); r = e - c * (MT * 16); }  // walk W row-major: W row = c, W col = r
            else { r = e / (T * 16); c = e - r * (T * 16); }
            v[u] = 0.f;
            if (e < total && r < rows && c < cols) v[u] = TRANSPOSED ? W[(int64_t)c * rows + r] : W[(int64_t)r * cols + c];
            const int mt = r >> 4, i = r & 15, t = c >> 4, g = (c & 15) >> 2, ks = c & 3;
            dst[u] = e < total ? ((mt * T + t) * 64 + g * 16 + i) * 4 + ks : -1;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
            if (dst[u] >= 0) {
                lds[dst[u]] = v[u];
            }
    }
}

// Load a (n, width) row-major tensor into the transposed-tile register layout: v[t][nt][r] = src[s0+16nt+j][16t+4g+r]
template <int WT, int NT>
__device__ __forceinline__ void load_tiles(f4 (&v)[WT][NT], const float *__restrict__ src, int width, int64_t s0,
                                           int64_t cnt, int g, int j) {
    const int T = tiles16(width);
    const bool vec = (width & 3) == 0;
#pragma unroll
    for (int t = 0; t < WT; ++t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f4 r = {0.f, 0.f, 0.f, 0.f};
            if (t < T) {
                const int64_t s = s0 + 16 * nt + j;
                const int col = 16 * t + 4 * g;
                if (s < cnt) {
                    const float *p = src + s * width + col;
                    if (vec && col + 3 < width) r = *reinterpret_cast<const f4 *>(p);
                    else {
                        if (col < width) r.x = p[0];
                        if (col + 1 < width) r.y = p[1];
                        if (col + 2 < width) r.z = p[2];
                        if (col + 3 < width) r.w = p[3];
                    }
                }
            }
            v[t][nt] = r;
        }
    }
}

template <int WT, int NT>
__device__ __forceinline__ void store_tiles(const f4 (&v)[WT][NT], float *__restrict__ dst, int width, int64_t s0,
                                            int64_t cnt, int g, int j) {
    const int T = tiles16(width);
    const bool vec = (width & 3) == 0;
#pragma unroll
    for (int t = 0; t < WT; ++t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (t < T) {
                const int64_t s = s0 + 16 * nt + j;
                const int col = 16 * t + 4 * g;
                if (s < cnt) {
                    float *p = dst + s * width + col;
                    const f4 r = v[t][nt];
                    if (vec && col + 3 < width) *reinterpret_cast<f4 *>(p) = r;
                    else {
                        if (col < width) p[0] = r.x;
                        if (col + 1 < width) p[1] = r.y;
                        if (col + 2 < width) p[2] = r.z;
                        if (col + 3 < width) p[3] = r.w;
                    }
                }
            }
        }
    }
}

// out[mt][nt] (+)= sum_t A(mt,t) . in[t][nt]   with A fragments in LDS
template <int WT, int NT>
__device__ __forceinline__ void gemm_tiles(f4 (&out)[WT][NT], const f4 (&in)[WT][NT], const float *lds_frag, int MT, int T,
                                           int lane) {
#pragma unroll
    for (int mt = 0; mt < WT; ++mt) {
        if (mt < MT) {
#pragma unroll
            for (int t = 0; t < WT; ++t) {
                if (t < T) {
                    const f4 a = *reinterpret_cast<const f4 *>(lds_frag + ((mt * T + t) * 64 + lane) * 4);   // (kFragLane == 4 here)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) out[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, in[t][nt].x, out[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) out[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, in[t][nt].y, out[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) out[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, in[t][nt].z, out[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) out[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, in[t][nt].w, out[mt][nt], 0, 0, 0);
                }
            }
        }
    }
}

// ---- forward ------------------------------------------------------------------------------------------
template <int WT, int NT>
__global__ void __launch_bounds__(256)
mlp_fwd_kernel(const float *__restrict__ x, const float *__restrict__ weights, const float *__restrict__ biases, MlpParams P,
               float *__restrict__ out, float *__restrict__ acts, int64_t n_cap, int64_t n, const int32_t *n_ptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int l = 0; l < P.n_layers; ++l) stage_fragments<false>(lds + P.lds_off[l], weights + P.w_off[l], P.dims[l + 1], P.dims[l]);
    __syncthreads();
    const int64_t cnt = dev_count(n, n_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    constexpr int SPW = 16 * NT;
    const int64_t n_tiles = ceil_div_dev(cnt, (int64_t)SPW * 4);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s0 = tile * SPW * 4 + (int64_t)wave * SPW;
        if (s0 >= cnt) continue;
        f4 hin[WT][NT];
        load_tiles<WT, NT>(hin, x, P.dims[0], s0, cnt, g, j);
        int64_t act_off = 0;
        for (int l = 0; l < P.n_layers; ++l) {
            const int N = P.dims[l + 1], K = P.dims[l];
            const int MT = tiles16(N), T = tiles16(K);
            f4 hout[WT][NT];
#pragma unroll
            for (int mt = 0; mt < WT; ++mt) {
                f4 b0 = {0.f, 0.f, 0.f, 0.f};
                if (P.has_bias && mt < MT) {
                    const float *bp = biases + P.b_off[l];
                    const int row = 16 * mt + 4 * g;
                    if (row < N) b0.x = bp[row];
                    if (row + 1 < N) b0.y = bp[row + 1];
                    if (row + 2 < N) b0.z = bp[row + 2];
                    if (row + 3 < N) b0.w = bp[row + 3];
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) hout[mt][nt] = b0;
            }
            gemm_tiles<WT, NT>(hout, hin, lds + P.lds_off[l], MT, T, lane);
            const bool last = (l == P.n_layers - 1);
            const int act = last ? P.act_out : P.act_hidden;
#pragma unroll
            for (int mt = 0; mt < WT; ++mt) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f4 r = hout[mt][nt];
                    if (mt < MT && act != ARCN_ACT_NONE) {
                        r.x = act_fwd(r.x, act, P.beta);
                        r.y = act_fwd(r.y, act, P.beta);
                        r.z = act_fwd(r.z, act, P.beta);
                        r.w = act_fwd(r.w, act, P.beta);
                    }
                    // rows beyond N are exact zeros in W, hence zero pre-activations; keep them zero as next-layer input
                    hin[mt][nt] = r;
                }
            }
            if (last) store_tiles<WT, NT>(hin, out, N, s0, cnt, g, j);
            else if (acts) {
                store_tiles<WT, NT>(hin, acts + act_off, N, s0, cnt, g, j);
                act_off += n_cap * N;
            }
            if (!last && (N & 15)) {
                // padded rows must feed zeros to the next layer even when act(0) != 0 (sigmoid, exp)
                const int row = 16 * (MT - 1) + 4 * g;
#pragma unroll
                for (int mt = 0; mt < WT; ++mt) {
                    if (mt == MT - 1) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if (row >= N) hin[mt][nt].x = 0.f;
                            if (row + 1 >= N) hin[mt][nt].y = 0.f;
                            if (row + 2 >= N) hin[mt][nt].z = 0.f;
                            if (row + 3 >= N) hin[mt][nt].w = 0.f;
                        }
                    }
                }
            }
        }
    }
}

// ---- forward, specialised on the tile shape ---------------------------------------------------------------------------------
// PMC on the generic kernel above: ~10 VALU instructions per MFMA (run-time layer loop, per-tile width guards, a per-element
// activation switch, 64-bit address arithmetic per 16-byte access).  With the number of 16-wide tiles of every layer boundary
// as template parameters the layer loop and every guard resolve at compile time; the widths themselves stay run-time, a
// ragged last tile (the 3-wide RGB output) takes the element-wise path.  Bias-free nets of 2 or 3 layers, widths <= 64.
template <int TT, int NT>
__device__ __forceinline__ void load_tiles_fast(f4 (&v)[4][NT], const float *__restrict__ src, int width, int64_t s0, int64_t cnt,
                                                int g, int j) {
    const bool full = width == 16 * TT;  // wave uniform
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t s = s0 + 16 * nt + j;
        const bool ok = s < cnt;
        const float *p = src + s * width + 4 * g;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f4 r = {0.f, 0.f, 0.f, 0.f};
            if (t < TT && ok) {
                if (full) r = *reinterpret_cast<const f4 *>(p + 16 * t);
                else {
                    const int col = 16 * t + 4 * g;
                    if (col < width) r.x = p[16 * t];
                    if (col + 1 < width) r.y = p[16 * t + 1];
                    if (col + 2 < width) r.z = p[16 * t + 2];
                    if (col + 3 < width) r.w = p[16 * t + 3];
                }
            }
            v[t][nt] = r;
        }
    }
}

template <int TT, int NT>
__device__ __forceinline__ void store_tiles_fast(const f4 (&v)[4][NT], float *__restrict__ dst, int width, int64_t s0, int64_t cnt,
                                                 int g, int j) {
    const bool full = width == 16 * TT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t s = s0 + 16 * nt + j;
        if (s >= cnt) continue;
        float *p = dst + s * width + 4 * g;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const f4 r = v[t][nt];
            if (full) *reinterpret_cast<f4 *>(p + 16 * t) = r;
            else {
                const int col = 16 * t + 4 * g;
                if (col < width) p[16 * t] = r.x;
                if (col + 1 < width) p[16 * t + 1] = r.y;
                if (col + 2 < width) p[16 * t + 2] = r.z;
                if (col + 3 < width) p[16 * t + 3] = r.w;
            }
        }
    }
}

// Saved hidden activations in TILE ORDER (level-major / concat entry points, whose activations only ever go to the matching fused
// backward): the f4 of lane (g, j) - neurons 16t+4g.. of sample 16q+j - sits at ((q * TT + t) * 64 + lane) * 4, so one wave-wide
// store or load is 1 KiB contiguous.  Row-major rows (256 B for 64 neurons) make the same instruction touch 16 separate 64-byte
// pieces: 40 % more HBM write traffic by PMC and a 2 % slower step.  Needs a width of exactly 16 * TT and room for pad16(n_cap) rows.
__host__ __device__ __forceinline__ int64_t pad16(int64_t n) { return (n + 15) & ~(int64_t)15; }

template <int TT, int NT>
__device__ __forceinline__ void store_tiles_frag(const f4 (&v)[4][NT], float *__restrict__ dst, int64_t s0, int64_t cnt, int lane) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (s0 + 16 * nt + (lane & 15) >= cnt) continue;
        float *p = dst + (((s0 >> 4) + nt) * TT * 64 + lane) * 4;
#pragma unroll
        // non-temporal: the saved activations (134 MB per radiance-net pass) are written once and read once by the matching backward -
        // marked as streaming they do not displace the table's lines (step -1.2 % in three alternations, scatter -5 us, gather -1.5 us;
        // profiles/r4_ab_nontemporal.txt.  The same hint on the scatter's 16-byte RECORDS doubled the scatter: partial lines)
        for (int t = 0; t < TT; ++t) __builtin_nontemporal_store(v[t][nt], reinterpret_cast<f4 *>(p + t * 256));
    }
}

template <int TT, int NT>
__device__ __forceinline__ void load_tiles_frag(f4 (&v)[4][NT], const float *__restrict__ src, int64_t s0, int64_t cnt, int lane) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const bool ok = s0 + 16 * nt + (lane & 15) < cnt;
        const float *p = src + (((s0 >> 4) + nt) * TT * 64 + lane) * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f4 r = {0.f, 0.f, 0.f, 0.f};
            if (t < TT && ok) r = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p + t * 256));
            v[t][nt] = r;
        }
    }
}

// Level-major network input (what arcn_hashgrid_fwd_xcd writes, 2 features per level): column c = 2 l + f of sample s lives at
// x[(l * stride + s) * 2 + f].  A lane's 4 columns 16t+4g.. are levels 8t+2g and 8t+2g+1: two 8-byte accesses, and the 16 lanes
// of a level read 128 contiguous bytes.
template <int TT, int NT>
__device__ __forceinline__ void load_tiles_lm2(f4 (&v)[4][NT], const float *__restrict__ src, int64_t stride, int64_t s0, int64_t cnt,
                                               int g, int j) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t s = s0 + 16 * nt + j;
        const bool ok = s < cnt;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f4 r = {0.f, 0.f, 0.f, 0.f};
            if (t < TT && ok) {
                const float *p = src + ((int64_t)(8 * t + 2 * g) * stride + s) * 2;
                const float2 a = *reinterpret_cast<const float2 *>(p);
                const float2 b = *reinterpret_cast<const float2 *>(p + stride * 2);
                r = f4{a.x, a.y, b.x, b.y};
            }
            v[t][nt] = r;
        }
    }
}

template <int TT, int NT>
__device__ __forceinline__ void store_tiles_lm2(const f4 (&v)[4][NT], float *__restrict__ dst, int64_t stride, int64_t s0, int64_t cnt,
                                                int g, int j) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t s = s0 + 16 * nt + j;
        if (s >= cnt) continue;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            float *p = dst + ((int64_t)(8 * t + 2 * g) * stride + s) * 2;
            *reinterpret_cast<float2 *>(p) = make_float2(v[t][nt].x, v[t][nt].y);
            *reinterpret_cast<float2 *>(p + stride * 2) = make_float2(v[t][nt].z, v[t][nt].w);
        }
    }
}

// "Concat" network input, XMODE 2: x = [A | B] (a_first) or [B | A], 16 columns each.  A is row-major (n, 16) per sample;
// B is a small table gathered by an index per sample (the per-ray spherical harmonics, by ray id).  This is the radiance net of
// Base3dModel._forward_pts_dir (base_3d_model.py:233-254) with fuse_radiance_inputs (encoder_mlp_network.py:93-118) folded
// into the first layer's operand load: no (n, 32) input buffer, no glue kernel.  Optional head: head_out[s] =
// act(A[s][0]) (sigma, EncoderMLPGeoNet.handle_output) on the way in; its gradient d_head[s] * act'(A[s][0]) is added to
// column 0 of dA on the way out.
struct MlpCat {
    const float *b_table;
    const int32_t *b_index;
    float *head_out;
    const float *d_head;
    int32_t a_first, head_act;
};

template <int NT>
__device__ __forceinline__ void load_tiles_cat(f4 (&v)[4][NT], const float *__restrict__ a, const MlpCat &c, int64_t s0, int64_t cnt,
                                               int g, int j, bool write_head) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t s = s0 + 16 * nt + j;
        f4 fa = {0.f, 0.f, 0.f, 0.f}, fb = {0.f, 0.f, 0.f, 0.f};
        if (s < cnt) {
            fa = *reinterpret_cast<const f4 *>(a + s * 16 + 4 * g);
            fb = *reinterpret_cast<const f4 *>(c.b_table + (int64_t)c.b_index[s] * 16 + 4 * g);
            if (write_head && c.head_out && g == 0) c.head_out[s] = act_fwd(fa.x, c.head_act, 1.0f);
        }
        v[0][nt] = c.a_first ? fa : fb;
        v[1][nt] = c.a_first ? fb : fa;
        v[2][nt] = f4{0.f, 0.f, 0.f, 0.f};
        v[3][nt] = f4{0.f, 0.f, 0.f, 0.f};
    }
}

// activation of MT tiles, the switch hoisted out of the element loop
template <int MT, int NT>
__device__ __forceinline__ void act_tiles(f4 (&h)[4][NT], int act, float beta) {
    if (act == ARCN_ACT_NONE) return;
    if (act == ARCN_ACT_RELU) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                h[mt][nt].x = fmaxf(h[mt][nt].x, 0.f);
                h[mt][nt].y = fmaxf(h[mt][nt].y, 0.f);
                h[mt][nt].z = fmaxf(h[mt][nt].z, 0.f);
                h[mt][nt].w = fmaxf(h[mt][nt].w, 0.f);
            }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            h[mt][nt].x = act_fwd(h[mt][nt].x, act, beta);
            h[mt][nt].y = act_fwd(h[mt][nt].y, act, beta);
            h[mt][nt].z = act_fwd(h[mt][nt].z, act, beta);
            h[mt][nt].w = act_fwd(h[mt][nt].w, act, beta);
        }
}

// padded rows of a ragged hidden layer must feed zeros to the next layer even when act(0) != 0
template <int MT, int NT>
__device__ __forceinline__ void zero_padded_rows(f4 (&h)[4][NT], int N, int g) {
    if (N == 16 * MT) return;
    const int row = 16 * (MT - 1) + 4 * g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (row >= N) h[MT - 1][nt].x = 0.f;
        if (row + 1 >= N) h[MT - 1][nt].y = 0.f;
        if (row + 2 >= N) h[MT - 1][nt].z = 0.f;
        if (row + 3 >= N) h[MT - 1][nt].w = 0.f;
    }
}

// AH / AO >= 0: the hidden / output activation as a compile-time constant (the level-major and concat entry points with ReLU hidden
// layers and a linear or sigmoid output: the NGP nets).  With the runtime switch the compiler clones the tile loop per activation: the
// radiance forward was 90 KB of code and 160 registers, 11.5 KB and 104 with the constants - room for a third workgroup per CU.
template <int T0, int T1, int T2, int T3, int NT, int XMODE, int AH = -1, int AO = -1, bool RC3 = false>  // XMODE: 0 row-major x, 1 level-major x, 2 concat (MlpCat); RC3: see below
__global__ void __launch_bounds__(256)
mlp_fwd_fixed_kernel(const float *__restrict__ x, int64_t x_stride, MlpCat cat, const float *__restrict__ weights, MlpParams P,
                     float *__restrict__ out, float *__restrict__ acts, int64_t n_cap, int64_t n, const int32_t *n_ptr) {
    constexpr int NL = T3 ? 3 : 2;
    constexpr int TA = T3 ? T3 : 1;
    constexpr bool FRAG = XMODE != 0;  // saved activations in tile order (store_tiles_frag): the level-major / concat entry points
    const int act_h = AH >= 0 ? AH : P.act_hidden, act_o = AO >= 0 ? AO : P.act_out;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int64_t cnt = dev_count(n, n_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    constexpr int SPW = 16 * NT;
    const int64_t n_tiles = ceil_div_dev(cnt, (int64_t)SPW * 4);
    auto load_x = [&](f4 (&dst)[4][NT], int64_t s0) {
        if (XMODE == 2) load_tiles_cat<NT>(dst, x, cat, s0, cnt, g, j, true);
        else if (XMODE == 1) load_tiles_lm2<T0, NT>(dst, x, x_stride, s0, cnt, g, j);
        else load_tiles_fast<T0, NT>(dst, x, P.dims[0], s0, cnt, g, j);
    };
    for (int l = 0; l < NL; ++l) stage_fragments<false>(lds + P.lds_off[l], weights + P.w_off[l], P.dims[l + 1], P.dims[l]);
    __syncthreads();
    const float *w0 = lds + P.lds_off[0], *w1 = lds + P.lds_off[1], *w2 = lds + P.lds_off[NL - 1];
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s0 = tile * SPW * 4 + (int64_t)wave * SPW;
        if (s0 >= cnt) continue;
        f4 h[4][NT], o[4][NT];
        // (requesting a tile's input one tile ahead - the first before the weights are staged - was built and measured: no change of
        // the forward nets or the step in three alternations, +18 VGPRs; round 4)
        load_x(h, s0);
        auto zero = [&](f4 (&a)[4][NT]) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) a[mt][nt] = f4{0.f, 0.f, 0.f, 0.f};
        };
        // layer 0: T0 -> T1 tiles
        zero(o);
        gemm_tiles<4, NT>(o, h, w0, T1, T0, lane);
        act_tiles<T1, NT>(o, act_h, P.beta);
        // the two-layer tile-order nets get no activations at all: their backward recomputes layer 0 from x (32 x 64 MACs per sample
        // against 256 B written here and read there: geometry net forward 40 -> 32 us, backward +1 us; for the three-layer net the
        // backward is MFMA-bound and the same trade loses, 58 -> 55 forward but 86 -> 96 us backward)
        // (RC3: the same trade for the three-layer tile-order net, an A/B instance - ARCN_RAD_RECOMP=1)
        if (acts && !(FRAG && (NL == 2 || RC3))) {
            if (FRAG) store_tiles_frag<T1, NT>(o, acts, s0, cnt, lane);
            else store_tiles_fast<T1, NT>(o, acts, P.dims[1], s0, cnt, g, j);
        }
        zero_padded_rows<T1, NT>(o, P.dims[1], g);
        // layer 1: T1 -> T2 tiles
        zero(h);
        gemm_tiles<4, NT>(h, o, w1, T2, T1, lane);
        if (NL == 2) {
            act_tiles<T2, NT>(h, act_o, P.beta);
            store_tiles_fast<T2, NT>(h, out, P.dims[2], s0, cnt, g, j);
        } else {
            act_tiles<T2, NT>(h, act_h, P.beta);
            if (acts) {
                if (FRAG) store_tiles_frag<T2, NT>(h, acts + pad16(n_cap) * P.dims[1], s0, cnt, lane);
                else store_tiles_fast<T2, NT>(h, acts + n_cap * P.dims[1], P.dims[2], s0, cnt, g, j);
            }
            zero_padded_rows<T2, NT>(h, P.dims[2], g);
            // layer 2: T2 -> T3 tiles
            zero(o);
            gemm_tiles<4, NT>(o, h, w2, TA, T2, lane);
            act_tiles<TA, NT>(o, act_o, P.beta);
            store_tiles_fast<TA, NT>(o, out, P.dims[3], s0, cnt, g, j);
        }
    }
}

// ---- backward, part 1: dpre_l for every layer (stored to scratch) and dx ----------------------------------
template <int WT, int NT>
__global__ void __launch_bounds__(256)
mlp_bwd_dx_kernel(const float *__restrict__ weights, MlpParams P, const float *__restrict__ out,
                  const float *__restrict__ acts, const float *__restrict__ dout, float *__restrict__ dx,
                  float *__restrict__ scratch, int64_t n_cap, int64_t n, const int32_t *n_ptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // A = W_l^T : rows = dims[l] (inputs), cols = dims[l+1] (outputs)
    for (int l = 0; l < P.n_layers; ++l) stage_fragments<true>(lds + P.lds_off[l], weights + P.w_off[l], P.dims[l], P.dims[l + 1]);
    __syncthreads();
    const int64_t cnt = dev_count(n, n_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    constexpr int SPW = 16 * NT;
    const int64_t n_tiles = ceil_div_dev(cnt, (int64_t)SPW * 4);
    // float offsets of layer l inside acts (hidden outputs) and scratch (dpre of every layer)
    int64_t acts_off[kMaxLayers], scr_off[kMaxLayers];
    {
        int64_t a = 0, s = 0;
        for (int l = 0; l < P.n_layers; ++l) {
            acts_off[l] = a;
            scr_off[l] = s;
            if (l < P.n_layers - 1) a += n_cap * P.dims[l + 1];
            s += n_cap * P.dims[l + 1];
        }
    }
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s0 = tile * SPW * 4 + (int64_t)wave * SPW;
        if (s0 >= cnt) continue;
        f4 d[WT][NT];
        load_tiles<WT, NT>(d, dout, P.dims[P.n_layers], s0, cnt, g, j);
        for (int l = P.n_layers - 1; l >= 0; --l) {
            const int N = P.dims[l + 1], K = P.dims[l];
            const bool last = (l == P.n_layers - 1);
            const int act = last ? P.act_out : P.act_hidden;
            if (act != ARCN_ACT_NONE) {
                f4 y[WT][NT];
                load_tiles<WT, NT>(y, last ? out : acts + acts_off[l], N, s0, cnt, g, j);
#pragma unroll
                for (int mt = 0; mt < WT; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        d[mt][nt].x *= act_grad_from_y(y[mt][nt].x, act, P.beta);
                        d[mt][nt].y *= act_grad_from_y(y[mt][nt].y, act, P.beta);
                        d[mt][nt].z *= act_grad_from_y(y[mt][nt].z, act, P.beta);
                        d[mt][nt].w *= act_grad_from_y(y[mt][nt].w, act, P.beta);
                    }
                }
            }
            store_tiles<WT, NT>(d, scratch + scr_off[l], N, s0, cnt, g, j);
            if (l > 0 || dx) {
                f4 dp[WT][NT];
#pragma unroll
                for (int mt = 0; mt < WT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) dp[mt][nt] = f4{0.f, 0.f, 0.f, 0.f};
                gemm_tiles<WT, NT>(dp, d, lds + P.lds_off[l], tiles16(K), tiles16(N), lane);
#pragma unroll
                for (int mt = 0; mt < WT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) d[mt][nt] = dp[mt][nt];
            }
        }
        if (dx) store_tiles<WT, NT>(d, dx, P.dims[0], s0, cnt, g, j);
    }
}

// ---- backward, fused: dx AND the dW partial sums in one pass (bias-free nets of 2 or 3 layers, widths <= 64) -----------------
// The two-kernel path above writes dpre of every layer to scratch and the dW kernel reads x, the hidden activations and
// dpre back: 1.9 KB/sample of HBM traffic for the NGP nets on top of the 0.9 KB the dx pass needs.  Here the dW tiles live in
// registers for the whole kernel: after dpre_l is known (registers, lane (g, j): neurons 4g..4g+3 of sample j) the wave
// transposes it and y_{l-1} through a private 1 KiB LDS tile each (one ds_write_b128 per lane, four ds_read_b32: lane (i, g)
// gets neuron i of sample 4q+g) — exactly the A/B operands of dW += dpre^T . y over 4 samples per MFMA.  At the end the 4
// waves are summed through LDS and the workgroup writes ONE partial per layer for mlp_dw_reduce_kernel.
// T0..T3 = 16-wide tiles per layer boundary (T3 = 0: two layers); the dims themselves stay run-time (ragged widths are zero
// padded by load_tiles / stage_fragments, so e.g. the 3-wide RGB output uses the T3 = 1 instance).
template <int T0, int T1, int T2, int T3, int NT, int XMODE, int AH = -1, int AO = -1, bool RC3 = false>   // AH / AO / RC3: see mlp_fwd_fixed_kernel
__global__ void __launch_bounds__(256, 2)  // 2 workgroups per CU = 2 waves per SIMD: at most 256 VGPR + AGPR per lane
mlp_bwd_fused_kernel(const float *__restrict__ x, int64_t x_stride, MlpCat cat, const float *__restrict__ weights, MlpParams P, const float *__restrict__ out,
                     const float *__restrict__ acts, const float *__restrict__ dout, float *__restrict__ dx,
                     float *__restrict__ partials, int n_slots, int64_t n_cap, int64_t n, const int32_t *n_ptr) {
    constexpr int NL = T3 ? 3 : 2;
    constexpr int WT = 4;
    constexpr int TA = T3 ? T3 : 1;  // tiles of the last boundary when there are three layers
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int lds_w = 0;
    for (int l = 0; l < NL; ++l) {
        stage_fragments<true>(lds + P.lds_off[l], weights + P.w_off[l], P.dims[l], P.dims[l + 1]);
        lds_w = P.lds_off[l] + tiles16(P.dims[l]) * tiles16(P.dims[l + 1]) * kFragTile;
    }
    constexpr bool FRAG = XMODE != 0;        // tile-order activations from the matching forward (store_tiles_frag)
    constexpr bool RECOMP = FRAG && (NL == 2 || RC3);  // ... which saved none for a two-layer net: layer 0 is recomputed from x
    const int act_h = AH >= 0 ? AH : P.act_hidden, act_o = AO >= 0 ? AO : P.act_out;
    // forward fragments of W_0 behind the transposition tiles
    const float *w0_fwd = lds + lds_w + 8192;
    if (RECOMP) stage_fragments<false>(lds + lds_w + 8192, weights + P.w_off[0], P.dims[1], P.dims[0]);
    __syncthreads();
    const int64_t cnt = dev_count(n, n_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    float *trA = lds + lds_w + wave * 2048;  // 4 tiles of 16 samples x 16 neurons: dpre
    float *trB = trA + 1024;                 // 4 tiles: y_{l-1}
    constexpr int SPW = 16 * NT;
    const int64_t n_tiles = ceil_div_dev(cnt, (int64_t)SPW * 4);
    const int64_t a1_off = 0, a2_off = (FRAG ? pad16(n_cap) : n_cap) * P.dims[1];  // hidden activations of layer 0 / layer 1 inside `acts`
    f4 acc0[T1][T0], acc1[T2][T1], acc2[TA][T2];
#pragma unroll
    for (int a = 0; a < T1; ++a)
#pragma unroll
        for (int b = 0; b < T0; ++b) acc0[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < T2; ++a)
#pragma unroll
        for (int b = 0; b < T1; ++b) acc1[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < T2; ++b) acc2[a][b] = f4{0.f, 0.f, 0.f, 0.f};

    // dW (MT x T tiles) += dpre^T . yprev for the NT sample tiles held in registers.  The 16 x 16 transposition tile is row = sample,
    // 16 floats per row, with the four 16-byte chunks of row s XOR-swizzled by (s >> 1) & 3: ds_write_b128 is serviced in contiguous
    // 8-lane groups on 32 banks, and with plain rows the 8 lanes of a group (samples j .. j+7, same chunk) fall on two bank quads -
    // a 4-way conflict, 32 LDS cycles per store instead of 8 (SQ_LDS_BANK_CONFLICT was 52 % of the kernel's LDS cycles).  The
    // ds_read_b32 side (32-lane groups: two consecutive rows, one swizzle value) stays conflict-free.
    const int tr_wr = j * 16 + ((g ^ ((j >> 1) & 3)) << 2);
    int tr_rd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tr_rd[q] = (4 * q + g) * 16 + (((j >> 2) ^ ((2 * q + (g >> 1)) & 3)) << 2) + (j & 3);
    auto accumulate = [&](auto &acc, const f4 (&dpre)[WT][NT], const f4 (&yprev)[WT][NT], auto mt_c, auto t_c) {
        constexpr int MT = decltype(mt_c)::value, T = decltype(t_c)::value;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f4 *>(trA + mt * 256 + tr_wr) = dpre[mt][nt];
#pragma unroll
            for (int t = 0; t < T; ++t) *reinterpret_cast<f4 *>(trB + t * 256 + tr_wr) = yprev[t][nt];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float av[MT], bv[T];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = trA[mt * 256 + tr_rd[q]];
#pragma unroll
                for (int t = 0; t < T; ++t) bv[t] = trB[t * 256 + tr_rd[q]];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[t], acc[mt][t], 0, 0, 0);
            }
        }
    };
    // d *= act'(y) on the first `mt_c` tiles; the activation switch is hoisted out of the element loop
    auto apply_act_grad = [&](f4 (&d)[WT][NT], const f4 (&y)[WT][NT], int act, auto mt_c) {
        constexpr int MT = decltype(mt_c)::value;
        if (act == ARCN_ACT_RELU) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    d[mt][nt].x = y[mt][nt].x > 0.f ? d[mt][nt].x : 0.f;
                    d[mt][nt].y = y[mt][nt].y > 0.f ? d[mt][nt].y : 0.f;
                    d[mt][nt].z = y[mt][nt].z > 0.f ? d[mt][nt].z : 0.f;
                    d[mt][nt].w = y[mt][nt].w > 0.f ? d[mt][nt].w : 0.f;
                }
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                d[mt][nt].x *= act_grad_from_y(y[mt][nt].x, act, P.beta);
                d[mt][nt].y *= act_grad_from_y(y[mt][nt].y, act, P.beta);
                d[mt][nt].z *= act_grad_from_y(y[mt][nt].z, act, P.beta);
                d[mt][nt].w *= act_grad_from_y(y[mt][nt].w, act, P.beta);
            }
    };
    auto back = [&](f4 (&d)[WT][NT], int l, int MT, int T) {  // d <- W_l^T . d
        f4 dp[WT][NT];
#pragma unroll
        for (int mt = 0; mt < WT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dp[mt][nt] = f4{0.f, 0.f, 0.f, 0.f};
        gemm_tiles<WT, NT>(dp, d, lds + P.lds_off[l], MT, T, lane);
#pragma unroll
        for (int mt = 0; mt < WT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) d[mt][nt] = dp[mt][nt];
    };

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s0 = tile * SPW * 4 + (int64_t)wave * SPW;
        if (s0 >= cnt) continue;
        f4 d[WT][NT], y[WT][NT], yp[WT][NT];
        constexpr int TL = T3 ? T3 : T2;  // tiles of the network output
        load_tiles_fast<TL, NT>(d, dout, P.dims[NL], s0, cnt, g, j);
        if (act_o != ARCN_ACT_NONE) {
            load_tiles_fast<TL, NT>(y, out, P.dims[NL], s0, cnt, g, j);
            apply_act_grad(d, y, act_o, std::integral_constant<int, TL>{});
        }
        if (NL == 3) {
            if (FRAG) load_tiles_frag<T2, NT>(yp, acts + a2_off, s0, cnt, lane);
            else load_tiles_fast<T2, NT>(yp, acts + a2_off, P.dims[2], s0, cnt, g, j);
            accumulate(acc2, d, yp, std::integral_constant<int, TA>{}, std::integral_constant<int, T2>{});
            back(d, 2, T2, TA);
            if (act_h != ARCN_ACT_NONE) apply_act_grad(d, yp, act_h, std::integral_constant<int, T2>{});
        }
        f4 xt[WT][NT];
        if (RECOMP) {
            // h_1 = act(W_0 x), the same fragments and MFMA order as the forward: bit-identical to what it would have saved
            if (XMODE == 2) load_tiles_cat<NT>(xt, x, cat, s0, cnt, g, j, false);
            else load_tiles_lm2<T0, NT>(xt, x, x_stride, s0, cnt, g, j);
#pragma unroll
            for (int mt = 0; mt < WT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) yp[mt][nt] = f4{0.f, 0.f, 0.f, 0.f};
            gemm_tiles<WT, NT>(yp, xt, w0_fwd, T1, T0, lane);
            act_tiles<T1, NT>(yp, act_h, P.beta);
        } else if (FRAG) {
            load_tiles_frag<T1, NT>(yp, acts + a1_off, s0, cnt, lane);
        } else {
            load_tiles_fast<T1, NT>(yp, acts + a1_off, P.dims[1], s0, cnt, g, j);
        }
        accumulate(acc1, d, yp, std::integral_constant<int, T2>{}, std::integral_constant<int, T1>{});
        back(d, 1, T1, T2);
        if (act_h != ARCN_ACT_NONE) apply_act_grad(d, yp, act_h, std::integral_constant<int, T1>{});
        if (RECOMP) {
#pragma unroll
            for (int mt = 0; mt < WT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) yp[mt][nt] = xt[mt][nt];
        } else if (XMODE == 2) load_tiles_cat<NT>(yp, x, cat, s0, cnt, g, j, false);
        else if (XMODE == 1) load_tiles_lm2<T0, NT>(yp, x, x_stride, s0, cnt, g, j);
        else load_tiles_fast<T0, NT>(yp, x, P.dims[0], s0, cnt, g, j);
        accumulate(acc0, d, yp, std::integral_constant<int, T1>{}, std::integral_constant<int, T0>{});
        if (dx) {
            if (XMODE == 2) {
                // only the A half of x has a gradient (B is a function of the ray direction): dA (n, 16) = that tile of
                // W_0^T dpre_0, plus the head's gradient in column 0
                back(d, 0, cat.a_first ? 1 : T0, T1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int64_t s = s0 + 16 * nt + j;
                    if (s >= cnt) continue;
                    f4 r = cat.a_first ? d[0][nt] : d[1][nt];
                    if (cat.d_head && g == 0) {
                        const float x0 = cat.a_first ? yp[0][nt].x : yp[1][nt].x;
                        r.x += cat.d_head[s] * act_grad(x0, act_fwd(x0, cat.head_act, 1.0f), cat.head_act, 1.0f);
                    }
                    *reinterpret_cast<f4 *>(dx + s * 16 + 4 * g) = r;
                }
            } else {
                back(d, 0, T0, T1);
                if (XMODE == 1) store_tiles_lm2<T0, NT>(d, dx, x_stride, s0, cnt, g, j);  // dx in the layout of x
                else store_tiles_fast<T0, NT>(d, dx, P.dims[0], s0, cnt, g, j);
            }
        }
    }
    // sum the 4 waves through LDS (the transposition tiles are free now: 4 x 2048 floats >= 16 tiles of 256) and write the
    // workgroup's partial of every layer in the fragment order mlp_dw_reduce_kernel expects: tile (a*4+b), lane, register
    __syncthreads();
    float *red = lds + lds_w;
    auto flush = [&](auto &acc, int l, auto mt_c, auto t_c) {
        constexpr int MT = decltype(mt_c)::value, T = decltype(t_c)::value;
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < T; ++b) {
                        f4 *dst = reinterpret_cast<f4 *>(red + (a * 4 + b) * 256 + lane * 4);
                        *dst = (w == 0) ? acc[a][b] : (*dst + acc[a][b]);
                    }
            }
            __syncthreads();
        }
        f4 *part = reinterpret_cast<f4 *>(partials + ((int64_t)l * n_slots + blockIdx.x) * 4096);
        const f4 *src4 = reinterpret_cast<const f4 *>(red);
        for (int e = threadIdx.x; e < 1024; e += 256) {
            const int tl = e >> 6;  // 64 float4 per tile
            if ((tl >> 2) < MT && (tl & 3) < T) part[e] = src4[e];
        }
        __syncthreads();
    };
    flush(acc0, 0, std::integral_constant<int, T1>{}, std::integral_constant<int, T0>{});
    flush(acc1, 1, std::integral_constant<int, T2>{}, std::integral_constant<int, T1>{});
    if (NL == 3) flush(acc2, 2, std::integral_constant<int, TA>{}, std::integral_constant<int, T2>{});
}

// ---- backward, part 2: dW_l = dpre_l^T . y_{l-1}  (reduction over samples), db_l = sum_s dpre_l ----------------
// grid = (slabs, n_layers * quads): one workgroup owns a slab of samples for one 64x64 quadrant of one layer's dW.
// MFMA rows = output neurons, columns = input neurons, K = samples: both operands come straight from global memory in
// the layout the MFMA wants (lane (i, g): dpre[s+g][16mt+i]; lane (j, g): y[s+g][16nt+j]) — 64 B segments per 16 lanes.
struct DwParams {
    int32_t n_layers;
    int32_t dims[kMaxLayers + 1];
    int32_t w_off[kMaxLayers], b_off[kMaxLayers];
    int32_t quad_first[kMaxLayers + 1];  // prefix of quadrant counts per layer
    int32_t has_bias;
};

__global__ void __launch_bounds__(256)
mlp_bwd_dw_kernel(const float *__restrict__ x, const float *__restrict__ acts, const float *__restrict__ scratch, DwParams P,
                  float *__restrict__ partials, float *__restrict__ bias_partials, int64_t n_cap, int64_t n,
                  const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    // which (layer, quadrant)
    int l = 0;
    while (l + 1 < P.n_layers && (int)blockIdx.y >= P.quad_first[l + 1]) ++l;
    const int q = blockIdx.y - P.quad_first[l];
    const int N = P.dims[l + 1], K = P.dims[l];
    const int qn = (tiles16(K) + 3) / 4;  // quadrants along the input dim
    const int mt0 = (q / qn) * 4, nt0 = (q % qn) * 4;
    const int MT = min(4, tiles16(N) - mt0), NTK = min(4, tiles16(K) - nt0);
    int64_t acts_off = 0, scr_off = 0;
    for (int k = 0; k < l; ++k) { scr_off += n_cap * P.dims[k + 1]; if (k < l - 1) acts_off += n_cap * P.dims[k + 1]; }
    const float *dpre = scratch + scr_off;
    const float *yprev = (l == 0) ? x : acts + acts_off;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    f4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    const int64_t total_waves = (int64_t)gridDim.x * 4;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    // contiguous slab of 4-sample steps per wave
    const int64_t steps = (cnt + 3) >> 2;
    const int64_t per = (steps + total_waves - 1) / total_waves;
    const int64_t st_lo = gw * per, st_hi = (st_lo + per < steps) ? st_lo + per : steps;
    // 4 four-sample steps per iteration: all (up to 32) scalar operand loads are issued before the first MFMA so their
    // L2 latency overlaps; tile rows/cols beyond this layer's size are skipped (wave-uniform).
    constexpr int U = 4;
    for (int64_t st = st_lo; st < st_hi; st += U) {
        float av[U][4], bv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t s = (st + u) * 4 + g;
            const bool ok = (st + u) < st_hi && s < cnt;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int row = 16 * (mt0 + a) + i;
                av[u][a] = (ok && a < MT && row < N) ? dpre[s * N + row] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int col = 16 * (nt0 + b) + i;
                bv[u][b] = (ok && b < NTK && col < K) ? yprev[s * K + col] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a < MT) {
                    bsum[a] += av[u][a];
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (b < NTK) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
                }
            }
        }
    }
    // The 4 waves of the workgroup are summed through ONE 16 KiB LDS tile (wave 0 stores, waves 1..3 add in turn): small
    // enough not to limit occupancy (the kernel is bound by load latency and wants many resident waves).  The workgroup's
    // 64x64 partial then goes to its slot of `partials` with coalesced float4 stores; mlp_dw_reduce_kernel sums the slots.
    // No global atomics here: they run at only ~20 G/s on MI355X.
    __shared__ __attribute__((aligned(16))) float red[16][64][4];
    __shared__ float bred[64];
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    f4 *dst = reinterpret_cast<f4 *>(&red[a * 4 + b][lane][0]);
                    *dst = (w == 0) ? acc[a][b] : (*dst + acc[a][b]);
                }
            if (P.has_bias && nt0 == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    float v = bsum[a];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (g == 0) bred[a * 16 + i] = (w == 0) ? v : bred[a * 16 + i] + v;
                }
            }
        }
        __syncthreads();
    }
    const int64_t slot = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    f4 *part = reinterpret_cast<f4 *>(partials + slot * 4096);
    const f4 *src4 = reinterpret_cast<const f4 *>(&red[0][0][0]);
    for (int e = threadIdx.x; e < 1024; e += 256) part[e] = src4[e];
    if (P.has_bias && bias_partials && nt0 == 0 && threadIdx.x < 64) bias_partials[slot * 64 + threadIdx.x] = bred[threadIdx.x];
}

// sum the per-workgroup partial tiles of one (layer, quadrant) and add them into dW / db.  grid = (4096 / kReduceElems, quads),
// 1024 threads: thread (element e, slice z of kReduceSlices) sums every kReduceSlices-th slot with ALL its loads in flight at once
// (the kernel is pure load latency: 512 slots x 16 KiB per layer), the slices meet in LDS and slice 0 adds the total into dW with
// a plain read-modify-write: one owner per element, no atomics, and a fixed summation order (bit-reproducible gradients).
constexpr int kReduceElems = 64, kReduceSlices = 16;

// Optional tail of the reduction: the owner of a weight element applies the optimiser to it right away (arcn_ngp_step_tail) - the
// flat parameter / moment buffers at the net's weight segment, the very adam1 of adam_ema_kernel on gradient = dW as accumulated.
struct DwAdam {
    float *param, *m, *v, *ema;   // at the first weight of the net; param == nullptr: plain reduction
    AdamArgs a;
};

__device__ __forceinline__ void dw_reduce_block(const float *__restrict__ partials, const float *__restrict__ bias_partials, const DwParams &P,
                                                int n_slots, int bx, int by, float *__restrict__ dweights, float *__restrict__ dbiases,
                                                float (&part)[kReduceSlices][kReduceElems], const DwAdam &opt) {
    int l = 0;
    while (l + 1 < P.n_layers && by >= P.quad_first[l + 1]) ++l;
    const int q = by - P.quad_first[l];
    const int N = P.dims[l + 1], K = P.dims[l];
    const int qn = (tiles16(K) + 3) / 4;
    const int mt0 = (q / qn) * 4, nt0 = (q % qn) * 4;
    const int el = threadIdx.x % kReduceElems, z = threadIdx.x / kReduceElems;
    const int e = bx * kReduceElems + el;  // 0..4095 inside the 64x64 quadrant, fragment order
    const float *src = partials + (int64_t)by * n_slots * 4096 + e;
    const int tile = e >> 8, ln = (e >> 2) & 63, r = e & 3;
    const int a = tile >> 2, b = tile & 3;
    // accumulator layout: row = 4*(ln>>4) + r (output neuron), col = ln & 15 (input neuron)
    const int row = 16 * (mt0 + a) + 4 * (ln >> 4) + r, col = 16 * (nt0 + b) + (ln & 15);
    const bool inside = row < N && col < K;  // tiles outside the layer are never written by the fused backward
    float v = 0.f;
    if (inside) {
        int sl = z;
        for (; sl + 31 * kReduceSlices < n_slots; sl += 32 * kReduceSlices) {
            float t[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) t[u] = src[(int64_t)(sl + u * kReduceSlices) * 4096];
#pragma unroll
            for (int u = 0; u < 32; ++u) v += t[u];
        }
        for (; sl < n_slots; sl += kReduceSlices) v += src[(int64_t)sl * 4096];
    }
    part[z][el] = v;
    __syncthreads();
    if (z == 0 && inside) {
        float tot = part[0][el];
#pragma unroll
        for (int k = 1; k < kReduceSlices; ++k) tot += part[k][el];
        const int64_t at = P.w_off[l] + (int64_t)row * K + col;
        float *dst = &dweights[at];
        const float g = *dst + tot;
        if (opt.param) {
            const AdamArgs &h = opt.a;
            float old = opt.param[at];
            adam1(opt.param[at], g, opt.m[at], opt.v[at], h.ema_in_param ? &old : (opt.ema ? &opt.ema[at] : nullptr), h.lr, h.b1, h.b2, h.eps, h.wd,
                  h.ema_decay, h.gscale, h.bc1, h.bc2_sqrt, h.deb_old, h.deb_new);
            *dst = h.zero_grad ? 0.f : g;
        } else {
            *dst = g;
        }
    }
    if (P.has_bias && dbiases && bias_partials && nt0 == 0 && bx == 0 && threadIdx.x < 64) {
        const float *bs = bias_partials + (int64_t)by * n_slots * 64 + threadIdx.x;
        float bv = 0.f;
        for (int k = 0; k < n_slots; ++k) bv += bs[(int64_t)k * 64];
        const int brow = 16 * mt0 + threadIdx.x;  // threadIdx.x = a*16 + i
        // several quadrants of one layer (nt0 == 0 for each row block) own different rows: still one writer per element
        if (brow < N && bv != 0.f) dbiases[P.b_off[l] + brow] += bv;
    }
}

__global__ void __launch_bounds__(kReduceElems * kReduceSlices)
mlp_dw_reduce_kernel(const float *__restrict__ partials, const float *__restrict__ bias_partials, DwParams P, int n_slots,
                     float *__restrict__ dweights, float *__restrict__ dbiases) {
    __shared__ float part[kReduceSlices][kReduceElems];
    dw_reduce_block(partials, bias_partials, P, n_slots, blockIdx.x, blockIdx.y, dweights, dbiases, part, DwAdam{});
}

// The end of the single-GPU NGP step in ONE launch (arcn_ngp_step_tail): the dW reductions of both nets with the optimiser applied by
// each element's owner, the optimiser on what else is left of the flat buffer (the table levels the scatter did not take), and the
// scatter's bin counters cleared for the next step.  Four launches of 5 - 7 us each (two reductions, optimiser, memset) were 4 % of
// the step.
struct TailNet {
    const float *partials;
    DwParams P;
    int n_slots, blocks;       // blocks = (4096 / kReduceElems) * n_layers
    int64_t w_seg;             // first weight of the net in the flat buffers
};

__global__ void __launch_bounds__(kReduceElems * kReduceSlices)
ngp_step_tail_kernel(TailNet na, TailNet nb, float *__restrict__ param, float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v,
                     float *__restrict__ ema, AdamRuns r, AdamArgs a, uint32_t *__restrict__ clear, int64_t clear_words, int clear_blocks) {
    __shared__ float part[kReduceSlices][kReduceElems];
    int bid = blockIdx.x;
    constexpr int per = 4096 / kReduceElems;
    // (two explicit branches: a net picked by reference makes the compiler copy both argument structs to scratch memory)
    if (bid < na.blocks) {
        DwAdam opt{param + na.w_seg, m + na.w_seg, v + na.w_seg, ema ? ema + na.w_seg : nullptr, a};
        dw_reduce_block(na.partials, nullptr, na.P, na.n_slots, bid % per, bid / per, grad + na.w_seg, nullptr, part, opt);
        return;
    }
    bid -= na.blocks;
    if (bid < nb.blocks) {
        DwAdam opt{param + nb.w_seg, m + nb.w_seg, v + nb.w_seg, ema ? ema + nb.w_seg : nullptr, a};
        dw_reduce_block(nb.partials, nullptr, nb.P, nb.n_slots, bid % per, bid / per, grad + nb.w_seg, nullptr, part, opt);
        return;
    }
    bid -= nb.blocks;
    if (bid < clear_blocks) {
        for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < clear_words; i += (int64_t)clear_blocks * blockDim.x) clear[i] = 0u;
        return;
    }
    bid -= clear_blocks;
    // the run of this workgroup, by constant indices (a dynamic index into the argument struct goes through scratch memory)
    int64_t lo = r.lo[0], cnt = r.n[0];
    int nb_run = r.b[0], first = 0, start = r.b[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        if (k < r.count && bid >= start) { lo = r.lo[k]; cnt = r.n[k]; nb_run = r.b[k]; first = start; }
        start += r.b[k];
    }
    adam_ema_run(param + lo, grad + lo, m + lo, v + lo, ema ? ema + lo : nullptr, cnt, bid - first, nb_run, a);
}

// ---- both NGP nets' forward in one kernel -----------------------------------------------------------------------------------------------
// arcn_mlp_fwd_lm (geometry net 32 -> 64 -> 16 on level-major hash features) followed by arcn_mlp_fwd_cat (radiance net [geo_out | SH(ray)]
// 32 -> 64 -> 64 -> 3) is the forward of EncoderMLP geometry + radiance (base_3d_model.py:233-254) in two launches with geo_out written by one
// and read back by the other.  In the transposed-MFMA form the geometry net's output tile IS the radiance net's first operand tile (the same
// K-permutation that keeps activations in registers between layers holds between the nets): one kernel, the weights of both staged once, 64
// bytes per sample less read traffic and one launch + prologue less.  The same fragments and the same MFMA order as the two kernels:
// bit-identical geo_out, sigma, saved activations and rgb.  ReLU hidden layers, linear geometry output, sigmoid radiance output compiled in
// (the NGP nets of nerf_ngp.yaml).  Measured (profiles/r6_ab_fused_nets.txt): the step 0.547 -> 0.543 ms - these kernels are bound by the
// f32 MFMA issue of waves that wait on their loads, not by their prologue.
template <int NT>
__global__ void __launch_bounds__(256)
ngp_nets_fwd_kernel(const float *__restrict__ x, int64_t x_stride, const float *__restrict__ geo_w, MlpParams G, float *__restrict__ geo_out,
                    MlpCat cat, const float *__restrict__ rad_w, MlpParams Rp, int rad_lds0, float *__restrict__ rgb, float *__restrict__ acts,
                    int64_t n_cap, int64_t n, const int32_t *n_ptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int l = 0; l < 2; ++l) stage_fragments<false>(lds + G.lds_off[l], geo_w + G.w_off[l], G.dims[l + 1], G.dims[l]);
    for (int l = 0; l < 3; ++l) stage_fragments<false>(lds + rad_lds0 + Rp.lds_off[l], rad_w + Rp.w_off[l], Rp.dims[l + 1], Rp.dims[l]);
    __syncthreads();
    const float *g0 = lds + G.lds_off[0], *g1 = lds + G.lds_off[1];
    const float *r0 = lds + rad_lds0 + Rp.lds_off[0], *r1 = lds + rad_lds0 + Rp.lds_off[1], *r2 = lds + rad_lds0 + Rp.lds_off[2];
    const int64_t cnt = dev_count(n, n_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    constexpr int SPW = 16 * NT;
    const int64_t n_tiles = ceil_div_dev(cnt, (int64_t)SPW * 4);
    // (Tiles drawn from a ticket counter instead of striding were measured in round 6, profiles/r6_exp_sched.txt / r6_exp_sched_wave.txt: per
    // workgroup - two barriers per tile - the step goes 0.534 -> 0.562 ms at 768 workgroups, worse at 512 and 1024: the barriers put the four
    // waves in lockstep, and one wave's loads under another's MFMAs is what these kernels live on; per WAVE, no barriers: 0.54 -> 0.69 ms.)
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s0 = tile * SPW * 4 + (int64_t)wave * SPW;
        if (s0 >= cnt) continue;
        f4 h[4][NT], o[4][NT];
        load_tiles_lm2<2, NT>(h, x, x_stride, s0, cnt, g, j);
        auto zero = [&](f4 (&a)[4][NT]) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) a[mt][nt] = f4{0.f, 0.f, 0.f, 0.f};
        };
        // geometry net: 2 -> 4 -> 1 tiles (its backward recomputes the hidden layer: nothing saved)
        zero(o);
        gemm_tiles<4, NT>(o, h, g0, 4, 2, lane);
        act_tiles<4, NT>(o, ARCN_ACT_RELU, 0.f);
        zero(h);
        gemm_tiles<4, NT>(h, o, g1, 1, 4, lane);
        // geo_out (n, 16) for the backward passes; sigma on the way; [geo_out | SH(ray)] as the radiance net's operand, straight from registers
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int64_t s = s0 + 16 * nt + j;
            f4 fb = {0.f, 0.f, 0.f, 0.f};
            const f4 fa = h[0][nt];
            if (s < cnt) {
                *reinterpret_cast<f4 *>(geo_out + s * 16 + 4 * g) = fa;
                fb = *reinterpret_cast<const f4 *>(cat.b_table + (int64_t)cat.b_index[s] * 16 + 4 * g);
                if (cat.head_out && g == 0) cat.head_out[s] = act_fwd(fa.x, cat.head_act, 1.0f);
            }
            h[0][nt] = cat.a_first ? fa : fb;
            h[1][nt] = cat.a_first ? fb : fa;
            h[2][nt] = f4{0.f, 0.f, 0.f, 0.f};
            h[3][nt] = f4{0.f, 0.f, 0.f, 0.f};
        }
        // radiance net: 2 -> 4 -> 4 -> 1 tiles, hidden activations saved in tile order
        zero(o);
        gemm_tiles<4, NT>(o, h, r0, 4, 2, lane);
        act_tiles<4, NT>(o, ARCN_ACT_RELU, 0.f);
        if (acts) store_tiles_frag<4, NT>(o, acts, s0, cnt, lane);
        zero(h);
        gemm_tiles<4, NT>(h, o, r1, 4, 4, lane);
        act_tiles<4, NT>(h, ARCN_ACT_RELU, 0.f);
        if (acts) store_tiles_frag<4, NT>(h, acts + pad16(n_cap) * 64, s0, cnt, lane);
        zero(o);
        gemm_tiles<4, NT>(o, h, r2, 1, 4, lane);
        act_tiles<1, NT>(o, ARCN_ACT_SIGMOID, 0.f);
        store_tiles_fast<1, NT>(o, rgb, Rp.dims[3], s0, cnt, g, j);
    }
}

// ---- the two-layer geometry nets of the NeuS-on-hash-grid + MultiVol step (BASELINE config 4), fused ----------------------------------------
// hash features (32, level-major) -> 64 (softplus beta | ReLU) -> n_out <= 32 = [sdf | features] or [log density | features], bias-free
// (configs/neus_ngp_multivol.yaml; reference sdf_model.py:42-101, base_network.py:30-44, linear_network_module.py:174-197).  trainer.
// FusedNeusNgpStep spelled each net as a chain of the generic dense products - level-major -> rows, two forward products, the activation
// derivative, the Jacobian-row product, three weight-gradient products with their reductions, three input-gradient products, the curvature
// pass: 16 launches and ~280 us for 1.25e5 foreground samples, every intermediate (n, 64) through HBM.  Here a net is ONE forward and ONE
// backward kernel in the transposed-MFMA form of the kernels above: the hidden layer never leaves registers and the backward recomputes it
// from the features (nothing is saved), all weight-gradient tiles stay in accumulators.
//   JAC (the sdf net): the forward also emits the Jacobian row of output 0,  jac = W1^T (s * W2[0]),  s = softplus'(z) = 1 - exp(-beta h)
//   (GeoNet.forward_with_grad's d sdf / d features, base_network.py:30-44), and the backward takes jac's gradient d_jac as a second
//   input:  u = W1 d_jac,  dz = dh s + beta W2[0] (s u)(1 - s),  dW1 += dz^T x + (s W2[0])^T d_jac,  dW2 += g^T h,  dW2[0] += sum_s s u
//   (ops.autograd.SdfMlpJacFn.backward's arithmetic; arcn_sdf_jac_dz2).
//   !JAC (the background density net): ReLU hidden layer, head = exp(out[0]) (TruncExp), its gradient through exp(clamp(out[0], +-15)).
template <int MT, int T, int NT>
__device__ __forceinline__ void dw_accumulate(f4 (&acc)[MT][T], const f4 (&dpre)[4][NT], const f4 (&yprev)[4][NT], float *trA, float *trB,
                                              int tr_wr, const int (&tr_rd)[4]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f4 *>(trA + mt * 256 + tr_wr) = dpre[mt][nt];
#pragma unroll
        for (int t = 0; t < T; ++t) *reinterpret_cast<f4 *>(trB + t * 256 + tr_wr) = yprev[t][nt];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float av[MT], bv[T];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = trA[mt * 256 + tr_rd[q]];
#pragma unroll
            for (int t = 0; t < T; ++t) bv[t] = trB[t * 256 + tr_rd[q]];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[t], acc[mt][t], 0, 0, 0);
        }
    }
}

template <int NT>
__device__ __forceinline__ void zero_tiles(f4 (&a)[4][NT]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[mt][nt] = f4{0.f, 0.f, 0.f, 0.f};
}

constexpr int kGeo2Frag = 8 * kFragTile;   // floats of one staged 64 x 32 (or 32 x 64) weight: 8 fragments

// softplus_beta(z) AND its derivative sigmoid(beta z) from ONE exponential: with t = beta z, e = exp(-|t|), u = 1 + e,
//   h = (max(t, 0) + log(u)) / beta,   s = 1 / u (t >= 0) or e / u (t < 0)        [= 1 - exp(-beta h), the from_y derivative]
// on the hardware exp2 / log2 / rcp (1 ulp each); the rounding of u = 1 + e is put back to first order ((e - (u - 1)) / u), so log1p keeps
// its accuracy for small e.  |error| <= ~2e-7 of the values (the libm chain expf -> log1pf -> expf it replaces: 3e-8), a fifth of the
// instructions: the sdf net's kernels were bound by that chain, not by their MFMAs (5.7 K VALU instructions around 192 MFMAs per tile);
// config 4's step 1.259 -> 1.191 ms in three alternations on one box (profiles/r6_ab_cfg4_softplus.txt).
// torch's threshold (beta z > 20: y = z) is what the formula gives there to the last bit or one ulp (log(u) < 2.1e-9).
__device__ __forceinline__ void softplus_and_slope(float z, float beta, float inv_beta, float &h, float &sl) {
    const float t = beta * z;
    const float e = __builtin_amdgcn_exp2f(-fabsf(t) * 1.44269504088896341f);
    const float u = 1.0f + e;
    const float r = __builtin_amdgcn_rcpf(u);
    const float l = __builtin_amdgcn_logf(u) * 0.693147180559945309f + (e - (u - 1.0f)) * r;
    h = (fmaxf(t, 0.f) + l) * inv_beta;
    sl = t >= 0.f ? r : e * r;
}

template <bool JAC>
__global__ void __launch_bounds__(256)
geo2_fwd_kernel(const float *__restrict__ x, int64_t x_stride, const float *__restrict__ w1, const float *__restrict__ w2, int n_out, int n_pad,
                float beta, float *__restrict__ out, float *__restrict__ head, float *__restrict__ jac, int64_t n, const int32_t *n_ptr) {
    constexpr int NT = 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *f_w1 = lds, *f_w2 = lds + kGeo2Frag, *f_w1t = lds + 2 * kGeo2Frag;
    const int MT2 = tiles16(n_out);
    stage_fragments<false>(f_w1, w1, 64, 32);
    stage_fragments<false>(f_w2, w2, n_out, 64);
    if (JAC) stage_fragments<true>(f_w1t, w1, 32, 64);
    __syncthreads();
    const int64_t cnt = dev_count(n, n_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    f4 w20[4];   // W2[0] at this lane's hidden neurons 16 t + 4 g ..
#pragma unroll
    for (int t = 0; t < 4; ++t) w20[t] = JAC ? *reinterpret_cast<const f4 *>(w2 + 16 * t + 4 * g) : f4{0.f, 0.f, 0.f, 0.f};
    const float inv_beta = 1.0f / beta;
    constexpr int SPW = 16 * NT;
    const int64_t n_tiles = ceil_div_dev(cnt, (int64_t)SPW * 4);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s0 = tile * SPW * 4 + (int64_t)wave * SPW;
        if (s0 >= cnt) continue;
        f4 h[4][NT], o[4][NT];
        if (x_stride) load_tiles_lm2<2, NT>(h, x, x_stride, s0, cnt, g, j);
        else load_tiles_fast<2, NT>(h, x, 32, s0, cnt, g, j);      // (x_stride 0: (n, 32) rows - HashGridEmbedder.forward's output, the module path)
        zero_tiles<NT>(o);
        gemm_tiles<4, NT>(o, h, f_w1, 4, 2, lane);
        f4 pj[4][NT];      // (JAC) s * W2[0]: the Jacobian row's operand, from the same exponential as the activation
        if (JAC) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float hv, sv;
                        softplus_and_slope(o[t][nt][r], beta, inv_beta, hv, sv);
                        o[t][nt][r] = hv;
                        pj[t][nt][r] = sv * w20[t][r];
                    }
        } else {
            act_tiles<4, NT>(o, ARCN_ACT_RELU, beta);
        }
        zero_tiles<NT>(h);
        gemm_tiles<4, NT>(h, o, f_w2, MT2, 4, lane);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int64_t s = s0 + 16 * nt + j;
            if (s >= cnt) continue;
            float *p = out + s * n_pad + 4 * g;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int col = 16 * t + 4 * g;
                const f4 r = h[t][nt];
                if (col + 3 < n_pad) *reinterpret_cast<f4 *>(p + 16 * t) = r;     // (n_pad is a multiple of 4: whole quads or nothing)
            }
            if (g == 0 && head) head[s] = JAC ? h[0][nt].x : expf(h[0][nt].x);
        }
        if (JAC) {
            // jac = W1^T (s * W2[0])
            zero_tiles<NT>(h);
            gemm_tiles<4, NT>(h, pj, f_w1t, 2, 4, lane);
            store_tiles_fast<2, NT>(h, jac, 32, s0, cnt, g, j);
        }
    }
}

template <bool JAC, int NT>
__global__ void __launch_bounds__(256, 2)
geo2_bwd_kernel(const float *__restrict__ x, int64_t x_stride, const float *__restrict__ w1, const float *__restrict__ w2, int n_out, float beta,
                const float *__restrict__ d_col0, int64_t ld_col0, const float *__restrict__ out_col0, int64_t ld_out, const float *__restrict__ d_feat,
                int64_t ld_feat, const float *__restrict__ d_jac, float *__restrict__ dx, int64_t dx_stride, float *__restrict__ partials,
                int n_slots, int64_t n, const int32_t *n_ptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *f_w1 = lds, *f_w1t = lds + kGeo2Frag, *f_w2t = lds + 2 * kGeo2Frag;
    const int MT2 = tiles16(n_out);
    stage_fragments<false>(f_w1, w1, 64, 32);
    stage_fragments<true>(f_w1t, w1, 32, 64);
    stage_fragments<true>(f_w2t, w2, 64, n_out);
    float *s_w20 = lds + 3 * kGeo2Frag + 8192;       // W2[0] (the Jacobian row's weights), behind the transposition tiles
    if (JAC && threadIdx.x < 64) s_w20[threadIdx.x] = w2[threadIdx.x];
    __syncthreads();
    const int64_t cnt = dev_count(n, n_ptr);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    float *trA = lds + 3 * kGeo2Frag + wave * 2048, *trB = trA + 1024;
    const int tr_wr = j * 16 + ((g ^ ((j >> 1) & 3)) << 2);
    int tr_rd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tr_rd[q] = (4 * q + g) * 16 + (((j >> 2) ^ ((2 * q + (g >> 1)) & 3)) << 2) + (j & 3);
    f4 acc1[4][2], acc2[2][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc1[a][b] = f4{0.f, 0.f, 0.f, 0.f}; acc2[b][a] = f4{0.f, 0.f, 0.f, 0.f}; }
    const float e0 = j == 0 ? 1.0f : 0.0f;      // the A operand of "row 0 += column sums": neuron i = lane & 15 of every sample
    const float inv_beta = 1.0f / beta;
    constexpr int SPW = 16 * NT;
    const int64_t n_tiles = ceil_div_dev(cnt, (int64_t)SPW * 4);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s0 = tile * SPW * 4 + (int64_t)wave * SPW;
        if (s0 >= cnt) continue;
        f4 xt[4][NT], h[4][NT], gt[4][NT], d[4][NT];
        if (x_stride) load_tiles_lm2<2, NT>(xt, x, x_stride, s0, cnt, g, j);
        else load_tiles_fast<2, NT>(xt, x, 32, s0, cnt, g, j);
        // the output gradient from its pieces: column 0 (through the head's activation), the feature columns, zeros behind them
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int64_t s = s0 + 16 * nt + j;
            const bool ok = s < cnt;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * t + 4 * g + r;
                    v[r] = 0.f;
                    if (ok && c < n_out) {
                        if (c == 0) {
                            v[r] = d_col0[s * ld_col0];
                            if (!JAC) { float pre = out_col0[s * ld_out]; pre = pre < -15.f ? -15.f : (pre > 15.f ? 15.f : pre); v[r] *= expf(pre); }
                        } else {
                            v[r] = d_feat[s * ld_feat + (c - 1)];
                        }
                    }
                }
                gt[t][nt] = f4{v[0], v[1], v[2], v[3]};
            }
            gt[2][nt] = gt[3][nt] = f4{0.f, 0.f, 0.f, 0.f};
        }
        // the hidden layer again: the fragments and the MFMA order of the forward
        zero_tiles<NT>(h);
        gemm_tiles<4, NT>(h, xt, f_w1, 4, 2, lane);
        f4 sl[4][NT];      // (JAC) the activation's slope, from the same exponential as the activation (the forward's arithmetic)
        if (JAC) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float hv, sv;
                        softplus_and_slope(h[t][nt][r], beta, inv_beta, hv, sv);
                        h[t][nt][r] = hv;
                        sl[t][nt][r] = sv;
                    }
        } else {
            act_tiles<4, NT>(h, ARCN_ACT_RELU, beta);
        }
        dw_accumulate<2, 4, NT>(acc2, gt, h, trA, trB, tr_wr, tr_rd);
        zero_tiles<NT>(d);
        gemm_tiles<4, NT>(d, gt, f_w2t, 4, MT2, lane);            // dh = W2^T g
        if (JAC) {
            f4 dj[4][NT], u[4][NT];
            load_tiles_fast<2, NT>(dj, d_jac, 32, s0, cnt, g, j);
            zero_tiles<NT>(u);
            gemm_tiles<4, NT>(u, dj, f_w1, 4, 2, lane);           // u = W1 d_jac
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f4 w20 = *reinterpret_cast<const f4 *>(s_w20 + 16 * t + 4 * g);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sv = sl[t][nt][r];
                        const float su = sv * u[t][nt][r];
                        d[t][nt][r] = d[t][nt][r] * sv + (beta * w20[r]) * su * (1.0f - sv);
                        h[t][nt][r] = sv * w20[r];        // sw: the operand of the Jacobian path's first-layer gradient
                        u[t][nt][r] = su;
                    }
                }
            }
            // dW2[0] += sum over the samples of s u: one more K pass with the constant row selector as its A operand
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int t = 0; t < 4; ++t) *reinterpret_cast<f4 *>(trB + t * 256 + tr_wr) = u[t][nt];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float bv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) bv[t] = trB[t * 256 + tr_rd[q]];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc2[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(e0, bv[t], acc2[0][t], 0, 0, 0);
                }
            }
            dw_accumulate<4, 2, NT>(acc1, h, dj, trA, trB, tr_wr, tr_rd);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[t][nt][r] = h[t][nt][r] > 0.f ? d[t][nt][r] : 0.f;
        }
        dw_accumulate<4, 2, NT>(acc1, d, xt, trA, trB, tr_wr, tr_rd);
        if (dx) {
            zero_tiles<NT>(gt);
            gemm_tiles<4, NT>(gt, d, f_w1t, 2, 4, lane);          // dx = W1^T dz
            if (dx_stride) store_tiles_lm2<2, NT>(gt, dx, dx_stride, s0, cnt, g, j);
            else store_tiles_fast<2, NT>(gt, dx, 32, s0, cnt, g, j);
        }
    }
    // the four waves' tiles summed through LDS, one partial per layer and workgroup in the order mlp_dw_reduce_kernel reads
    __syncthreads();
    float *red = lds + 3 * kGeo2Frag;
    for (int l = 0; l < 2; ++l) {
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f4 *dst = reinterpret_cast<f4 *>(red + (l == 0 ? a * 4 + b : b * 4 + a) * 256 + lane * 4);
                        const f4 v = l == 0 ? acc1[a][b] : acc2[b][a];
                        *dst = (w == 0) ? v : (*dst + v);
                    }
            }
            __syncthreads();
        }
        f4 *part = reinterpret_cast<f4 *>(partials + ((int64_t)l * n_slots + blockIdx.x) * 4096);
        const f4 *src4 = reinterpret_cast<const f4 *>(red);
        for (int e = threadIdx.x; e < 1024; e += 256) {
            const int tl = e >> 6, a = tl >> 2, b = tl & 3;
            if (l == 0 ? (b < 2) : (a < 2)) part[e] = src4[e];
        }
        __syncthreads();
    }
}

static int build_mlp_params(const arcn_mlp_desc *d, MlpParams &P, bool transposed, int *lds_floats, int *max_dim) {
    if (!d) return einval("mlp: desc is NULL");
    if (d->n_layers < 1 || d->n_layers > kMaxLayers) return einval("mlp: 1..8 layers supported");
    P.n_layers = d->n_layers;
    int woff = 0, boff = 0, loff = 0, md = 0;
    for (int l = 0; l <= d->n_layers; ++l) {
        if (d->dims[l] < 1 || d->dims[l] > 128) return einval("mlp: layer widths must be in 1..128");
        P.dims[l] = d->dims[l];
        if (d->dims[l] > md) md = d->dims[l];
    }
    for (int l = 0; l < d->n_layers; ++l) {
        P.w_off[l] = woff;
        P.b_off[l] = boff;
        P.lds_off[l] = loff;
        woff += d->dims[l] * d->dims[l + 1];
        boff += d->dims[l + 1];
        loff += tiles16(d->dims[l]) * tiles16(d->dims[l + 1]) * kFragTile;
    }
    // transposed = a backward pass: it differentiates the activations through the POST-activations the forward saved, which Sine does not
    // allow (cos(x) is not a function of sin(x)); tiny-cuda-nn's fused MLP has the same restriction
    if (transposed && (d->act_hidden == ARCN_ACT_SINE || d->act_out == ARCN_ACT_SINE))
        return einval("mlp_bwd: the Sine activation needs the pre-activations, which the fused MLP does not keep (forward / inference only)");
    P.act_hidden = d->act_hidden;
    P.act_out = d->act_out;
    P.has_bias = d->has_bias;
    P.beta = d->softplus_beta;
    *lds_floats = loff;
    *max_dim = md;
    return ARCN_OK;
}

template <typename Kern>
static int set_lds(Kern k, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
    }
    return ARCN_OK;
}

// resident workgroups re-use the staged weights across tiles: 512 = two per CU (measured best for the kernels with a runtime activation
// switch, forward and backward alike); the slim forward kernels (activations compiled in, ~100 registers) take three per CU
constexpr int kSlimGrid = 768;
inline unsigned tile_grid(int64_t n, int spb, int64_t cap = 512) {
    const int64_t b = ceil_div<int64_t>(n, spb);
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace arcn

using namespace arcn;

ARCN_EXPORT int64_t arcn_mlp_acts_floats(const arcn_mlp_desc *d, int64_t n_cap) {
    if (!d) return 0;
    int64_t s = 0;
    // rows padded to a multiple of 16: the level-major / concat entry points keep the activations in tile order (store_tiles_frag)
    for (int l = 0; l < d->n_layers - 1; ++l) s += pad16(n_cap) * d->dims[l + 1];
    return s;
}

// dpre of every layer, (n_cap, dims[l+1]) each
static int64_t arcn_mlp_dpre_floats(const arcn_mlp_desc *d, int64_t n_cap) {
    int64_t s = 0;
    for (int l = 0; l < d->n_layers; ++l) s += n_cap * d->dims[l + 1];
    return s;
}

// sample slabs of the dW kernel: a function of the CAPACITY only, so the scratch layout is fixed per allocation
static int64_t dw_slabs(int64_t n_cap) {
    int64_t slabs = ceil_div<int64_t>(n_cap, 512);
    if (slabs > 512) slabs = 512;
    if (slabs < 1) slabs = 1;
    return slabs;
}

ARCN_EXPORT int64_t arcn_mlp_scratch_floats(const arcn_mlp_desc *d, int64_t n_cap) {
    if (!d) return 0;
    int64_t quads = 0;
    for (int l = 0; l < d->n_layers; ++l) quads += ((tiles16(d->dims[l + 1]) + 3) / 4) * ((tiles16(d->dims[l]) + 3) / 4);
    // dpre of every layer + per-slab partial dW tiles (64x64 each) + per-slab partial bias sums
    return arcn_mlp_dpre_floats(d, n_cap) + dw_slabs(n_cap) * quads * (4096 + 64);
}

// x_stride = 0: x is (n, dims[0]) row-major; > 0: level-major, 2 features per level, level stride x_stride samples
// A/B switch (read once): the three-layer concat net of the NGP step recomputes its layer-0 activations in the backward instead of saving them
static bool rad_recomp() {
    static const bool on = [] { const char *e = getenv("ARCN_RAD_RECOMP"); return e && atoi(e) != 0; }();
    return on;
}

static int mlp_fwd_impl(const float *x, int64_t x_stride, const MlpCat *cat_in, const float *weights, const float *biases,
                        const arcn_mlp_desc *desc_host, float *out, float *acts, int64_t n_cap, int64_t n, const int32_t *n_ptr,
                        void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !weights || !out) return einval("mlp_fwd: missing argument");
    MlpParams P;
    int lds_floats, md;
    int rc = build_mlp_params(desc_host, P, false, &lds_floats, &md);
    if (rc) return rc;
    if (P.has_bias && !biases) return einval("mlp_fwd: biases required");
    const size_t lds_bytes = sizeof(float) * (size_t)lds_floats;
    if (lds_bytes > 144 * 1024) return einval("mlp_fwd: network too large for the LDS-resident fused kernel");
    static const int fwd_nt = 2;  // 4 waves/SIMD beat 2 with wider tiles
    MlpCat cat = {};
    if (cat_in) cat = *cat_in;
    if (x_stride || cat_in) {   // these entry points save the hidden activations in tile order: full 16-wide tiles only
        for (int l = 1; l < P.n_layers; ++l)
            if (P.dims[l] & 15) return einval("mlp_fwd_lm / mlp_fwd_cat: hidden widths must be multiples of 16");
    }
    if (!P.has_bias && (P.n_layers == 2 || P.n_layers == 3) && md <= 64) {
        const int sig = tiles16(P.dims[0]) * 1000 + tiles16(P.dims[1]) * 100 + tiles16(P.dims[2]) * 10 +
                        (P.n_layers == 3 ? tiles16(P.dims[3]) : 0);
#define ARCN_FIXED_A(T0, T1, T2, T3, XMODE, AH, AO, CAP)                                                                          \
    do {                                                                                                                         \
        if ((rc = set_lds(mlp_fwd_fixed_kernel<T0, T1, T2, T3, 2, XMODE, AH, AO>, lds_bytes))) return rc;                         \
        hipLaunchKernelGGL((mlp_fwd_fixed_kernel<T0, T1, T2, T3, 2, XMODE, AH, AO>), dim3(tile_grid(n, 128, CAP)), dim3(256), lds_bytes, \
                           as_stream(stream), x, x_stride, cat, weights, P, out, acts, n_cap, n, n_ptr);                          \
        return check_launch("mlp_fwd_fixed");                                                                                    \
    } while (0)
#define ARCN_FIXED(T0, T1, T2, T3, XMODE) ARCN_FIXED_A(T0, T1, T2, T3, XMODE, -1, -1, 512)
        // the NGP nets (ReLU hidden layers; linear geometry output, sigmoid radiance output) with their activations compiled in
        const bool relu_lin = P.act_hidden == ARCN_ACT_RELU && P.act_out == ARCN_ACT_NONE;
        const bool relu_sig = P.act_hidden == ARCN_ACT_RELU && P.act_out == ARCN_ACT_SIGMOID;
        if (cat_in) {
            if (P.dims[0] != 32) return einval("mlp_fwd_cat: the input must be 16 + 16 columns");
            if (sig == 2441 && relu_sig && rad_recomp()) {
                if ((rc = set_lds(mlp_fwd_fixed_kernel<2, 4, 4, 1, 2, 2, ARCN_ACT_RELU, ARCN_ACT_SIGMOID, true>, lds_bytes))) return rc;
                hipLaunchKernelGGL((mlp_fwd_fixed_kernel<2, 4, 4, 1, 2, 2, ARCN_ACT_RELU, ARCN_ACT_SIGMOID, true>), dim3(tile_grid(n, 128, kSlimGrid)), dim3(256), lds_bytes,
                                   as_stream(stream), x, x_stride, cat, weights, P, out, acts, n_cap, n, n_ptr);
                return check_launch("mlp_fwd_fixed");
            }
            if (sig == 2441 && relu_sig) ARCN_FIXED_A(2, 4, 4, 1, 2, ARCN_ACT_RELU, ARCN_ACT_SIGMOID, kSlimGrid);
            switch (sig) {
            case 2441: ARCN_FIXED(2, 4, 4, 1, 2);
            case 2410: ARCN_FIXED(2, 4, 1, 0, 2);
            default: break;
            }
        } else if (x_stride) {
            if (P.dims[0] & 15) return einval("mlp_fwd_lm: input width must be a multiple of 16");
            if (sig == 2410 && relu_lin) ARCN_FIXED_A(2, 4, 1, 0, 1, ARCN_ACT_RELU, ARCN_ACT_NONE, kSlimGrid);
            switch (sig) {
            case 2410: ARCN_FIXED(2, 4, 1, 0, 1);
            case 4410: ARCN_FIXED(4, 4, 1, 0, 1);
            default: break;
            }
        } else {
            switch (sig) {
            case 2410: ARCN_FIXED(2, 4, 1, 0, 0);
            case 4410: ARCN_FIXED(4, 4, 1, 0, 0);
            case 2441: ARCN_FIXED(2, 4, 4, 1, 0);
            case 3441: ARCN_FIXED(3, 4, 4, 1, 0);      // 33..48 inputs (the pvnf radiance net of NeuS-NGP: 3 + 16 + 3 + 16 = 38), ragged last tile
            case 4441: ARCN_FIXED(4, 4, 4, 1, 0);
            default: break;
            }
        }
#undef ARCN_FIXED
#undef ARCN_FIXED_A
    }
    if (cat_in) return einval("mlp_fwd_cat: only wired for bias-free nets 32 -> 64 [-> 64] -> <=16");
    if (x_stride) return einval("mlp_fwd_lm: level-major input is only wired for bias-free 2-layer nets (32|64 -> 64 -> <=16)");
    if (md <= 64 && fwd_nt == 2) {
        if ((rc = set_lds(mlp_fwd_kernel<4, 2>, lds_bytes))) return rc;
        hipLaunchKernelGGL((mlp_fwd_kernel<4, 2>), dim3(tile_grid(n, 128)), dim3(256), lds_bytes, as_stream(stream), x, weights,
                           biases, P, out, acts, n_cap, n, n_ptr);
    } else if (md <= 64) {
        if ((rc = set_lds(mlp_fwd_kernel<4, 4>, lds_bytes))) return rc;
        hipLaunchKernelGGL((mlp_fwd_kernel<4, 4>), dim3(tile_grid(n, 256)), dim3(256), lds_bytes, as_stream(stream), x, weights,
                           biases, P, out, acts, n_cap, n, n_ptr);
    } else {
        if ((rc = set_lds(mlp_fwd_kernel<8, 2>, lds_bytes))) return rc;
        hipLaunchKernelGGL((mlp_fwd_kernel<8, 2>), dim3(tile_grid(n, 128)), dim3(256), lds_bytes, as_stream(stream), x, weights,
                           biases, P, out, acts, n_cap, n, n_ptr);
    }
    return check_launch("mlp_fwd");
}

ARCN_EXPORT int arcn_mlp_fwd(const float *x, const float *weights, const float *biases, const arcn_mlp_desc *desc_host,
                             float *out, float *acts, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    return mlp_fwd_impl(x, 0, nullptr, weights, biases, desc_host, out, acts, n_cap, n, n_ptr, stream);
}

ARCN_EXPORT int arcn_mlp_fwd_lm(const float *x_lm, int64_t x_stride, const float *weights, const arcn_mlp_desc *desc_host,
                                float *out, float *acts, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    if (x_stride < n) return einval("mlp_fwd_lm: level stride smaller than n");
    return mlp_fwd_impl(x_lm, x_stride, nullptr, weights, nullptr, desc_host, out, acts, n_cap, n, n_ptr, stream);
}

ARCN_EXPORT int arcn_mlp_fwd_cat(const float *a, const float *b_table, const int32_t *b_index, int a_first, const float *weights,
                                 const arcn_mlp_desc *desc_host, float *out, float *acts, float *head_out, int head_act,
                                 int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!a || !b_table || !b_index) return einval("mlp_fwd_cat: missing argument");
    MlpCat cat = {b_table, b_index, head_out, nullptr, a_first ? 1 : 0, head_act};
    return mlp_fwd_impl(a, 0, &cat, weights, nullptr, desc_host, out, acts, n_cap, n, n_ptr, stream);
}

static int mlp_bwd_impl(const float *x, int64_t x_stride, const MlpCat *cat_in, const float *weights, const float *biases,
                        const arcn_mlp_desc *desc_host,
                        const float *out, const float *acts, const float *dout, float *dx, float *dweights,
                        float *dbiases, float *scratch, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream,
                        int defer_reduce = 0) {
    (void)biases;
    if (n <= 0) return ARCN_OK;
    if (!x || !weights || !out || !dout || !scratch) return einval("mlp_bwd: missing argument");
    MlpParams P;
    int lds_floats, md;
    int rc = build_mlp_params(desc_host, P, true, &lds_floats, &md);
    if (rc) return rc;
    if (P.n_layers > 1 && !acts) return einval("mlp_bwd: saved activations required");
    const size_t lds_bytes = sizeof(float) * (size_t)lds_floats;
    if (lds_bytes > 144 * 1024) return einval("mlp_bwd: network too large for the LDS-resident fused kernel");
    static const int bwd_nt = 2;
    MlpCat cat = {};
    if (cat_in) cat = *cat_in;
    if (x_stride || cat_in) {   // tile-order activations from arcn_mlp_fwd_lm / arcn_mlp_fwd_cat
        for (int l = 1; l < P.n_layers; ++l)
            if (P.dims[l] & 15) return einval("mlp_bwd_lm / mlp_bwd_cat: hidden widths must be multiples of 16");
    }
    if (dweights && !P.has_bias && (P.n_layers == 2 || P.n_layers == 3) && md <= 64) {
        // fused dx + dW for the tile shapes of the NGP nets; anything else takes the two-kernel path below
        const int t0 = tiles16(P.dims[0]), t1 = tiles16(P.dims[1]), t2 = tiles16(P.dims[2]);
        const int t3 = P.n_layers == 3 ? tiles16(P.dims[3]) : 0;
        const int sig = t0 * 1000 + t1 * 100 + t2 * 10 + t3;
        if (cat_in ? ((sig == 2441 || sig == 2410) && P.dims[0] == 32)
                   : x_stride ? (sig == 2410 || sig == 4410) : (sig == 2410 || sig == 2441 || sig == 3441 || sig == 4410 || sig == 4441)) {
            // + one 8 KiB transposition area per wave + the forward fragments of W_0 (<= 16 tiles) for the recomputed layer-0 activations
            const size_t fused_lds = lds_bytes + sizeof(float) * (8192 + 16 * kFragTile);
            int64_t grid = tile_grid(n, 64);
            if (grid > dw_slabs(n_cap)) grid = dw_slabs(n_cap);
            float *partials = scratch + arcn_mlp_dpre_floats(desc_host, n_cap);
            DwParams D;
            D.n_layers = P.n_layers;
            D.has_bias = 0;
            for (int l = 0; l <= P.n_layers; ++l) D.dims[l] = P.dims[l];
            for (int l = 0; l < P.n_layers; ++l) { D.w_off[l] = P.w_off[l]; D.b_off[l] = P.b_off[l]; D.quad_first[l] = l; }
            D.quad_first[P.n_layers] = P.n_layers;
#define ARCN_FUSED_A(T0, T1, T2, T3, NT, XMODE, AH, AO)                                                                         \
    do {                                                                                                                         \
        if ((rc = set_lds(mlp_bwd_fused_kernel<T0, T1, T2, T3, NT, XMODE, AH, AO>, fused_lds))) return rc;                        \
        hipLaunchKernelGGL((mlp_bwd_fused_kernel<T0, T1, T2, T3, NT, XMODE, AH, AO>), dim3((unsigned)grid), dim3(256), fused_lds, \
                           as_stream(stream), x, x_stride, cat, weights, P, out, acts, dout, dx, partials, (int)grid, n_cap, n,   \
                           n_ptr);                                                                                               \
    } while (0)
#define ARCN_FUSED(T0, T1, T2, T3, NT, XMODE) ARCN_FUSED_A(T0, T1, T2, T3, NT, XMODE, -1, -1)
            static const int fused_nt3 = 1;
            const bool relu_lin = P.act_hidden == ARCN_ACT_RELU && P.act_out == ARCN_ACT_NONE;
            const bool relu_sig = P.act_hidden == ARCN_ACT_RELU && P.act_out == ARCN_ACT_SIGMOID;
            if (cat_in) {
                if (sig == 2441 && relu_sig && rad_recomp()) {
                    if ((rc = set_lds(mlp_bwd_fused_kernel<2, 4, 4, 1, 1, 2, ARCN_ACT_RELU, ARCN_ACT_SIGMOID, true>, fused_lds))) return rc;
                    hipLaunchKernelGGL((mlp_bwd_fused_kernel<2, 4, 4, 1, 1, 2, ARCN_ACT_RELU, ARCN_ACT_SIGMOID, true>), dim3((unsigned)grid), dim3(256), fused_lds,
                                       as_stream(stream), x, x_stride, cat, weights, P, out, acts, dout, dx, partials, (int)grid, n_cap, n, n_ptr);
                } else if (sig == 2441 && relu_sig) ARCN_FUSED_A(2, 4, 4, 1, 1, 2, ARCN_ACT_RELU, ARCN_ACT_SIGMOID);
                else if (sig == 2441) ARCN_FUSED(2, 4, 4, 1, 1, 2);
                else ARCN_FUSED(2, 4, 1, 0, 2, 2);
            } else if (x_stride) {
                if (sig == 2410 && relu_lin) ARCN_FUSED_A(2, 4, 1, 0, 2, 1, ARCN_ACT_RELU, ARCN_ACT_NONE);
                else if (sig == 2410) ARCN_FUSED(2, 4, 1, 0, 2, 1);
                else ARCN_FUSED(4, 4, 1, 0, 2, 1);
            } else switch (sig) {
            case 2410: ARCN_FUSED(2, 4, 1, 0, 2, 0); break;
            case 4410: ARCN_FUSED(4, 4, 1, 0, 2, 0); break;
            case 2441: if (fused_nt3 == 2) ARCN_FUSED(2, 4, 4, 1, 2, 0); else ARCN_FUSED(2, 4, 4, 1, 1, 0); break;
            case 3441: ARCN_FUSED(3, 4, 4, 1, 1, 0); break;
            default: ARCN_FUSED(4, 4, 4, 1, 1, 0); break;
            }
#undef ARCN_FUSED
#undef ARCN_FUSED_A
            // defer_reduce: the per-workgroup partials stay in `scratch`; arcn_mlp_bwd_reduce adds them into dweights later
            if (!defer_reduce)
                hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(4096 / kReduceElems, (unsigned)P.n_layers), dim3(kReduceElems * kReduceSlices), 0, as_stream(stream), partials,
                                   static_cast<const float *>(nullptr), D, (int)grid, dweights, dbiases);
            return check_launch("mlp_bwd_fused");
        }
    }
    if (cat_in) return einval("mlp_bwd_cat: needs dweights and a bias-free net 32 -> 64 [-> 64] -> <=16");
    if (x_stride) return einval("mlp_bwd_lm: level-major input needs dweights and a bias-free 2-layer net (32|64 -> 64 -> <=16)");
    if (md <= 64 && bwd_nt == 2) {
        if ((rc = set_lds(mlp_bwd_dx_kernel<4, 2>, lds_bytes))) return rc;
        hipLaunchKernelGGL((mlp_bwd_dx_kernel<4, 2>), dim3(tile_grid(n, 128)), dim3(256), lds_bytes, as_stream(stream), weights, P,
                           out, acts, dout, dx, scratch, n_cap, n, n_ptr);
    } else if (md <= 64) {
        if ((rc = set_lds(mlp_bwd_dx_kernel<4, 4>, lds_bytes))) return rc;
        hipLaunchKernelGGL((mlp_bwd_dx_kernel<4, 4>), dim3(tile_grid(n, 256)), dim3(256), lds_bytes, as_stream(stream), weights, P,
                           out, acts, dout, dx, scratch, n_cap, n, n_ptr);
    } else {
        if ((rc = set_lds(mlp_bwd_dx_kernel<8, 2>, lds_bytes))) return rc;
        hipLaunchKernelGGL((mlp_bwd_dx_kernel<8, 2>), dim3(tile_grid(n, 128)), dim3(256), lds_bytes, as_stream(stream), weights, P,
                           out, acts, dout, dx, scratch, n_cap, n, n_ptr);
    }
    if ((rc = check_launch("mlp_bwd_dx"))) return rc;
    if (dweights) return arcn_mlp_bwd_dw(x, desc_host, acts, scratch, dweights, dbiases, n_cap, n, n_ptr, stream);
    return ARCN_OK;
}

ARCN_EXPORT int arcn_mlp_bwd(const float *x, const float *weights, const float *biases, const arcn_mlp_desc *desc_host,
                             const float *out, const float *acts, const float *dout, float *dx, float *dweights,
                             float *dbiases, float *scratch, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    return mlp_bwd_impl(x, 0, nullptr, weights, biases, desc_host, out, acts, dout, dx, dweights, dbiases, scratch, n_cap, n, n_ptr,
                        stream);
}

ARCN_EXPORT int arcn_mlp_bwd_lm(const float *x_lm, int64_t x_stride, const float *weights, const arcn_mlp_desc *desc_host,
                                const float *out, const float *acts, const float *dout, float *dx_lm, float *dweights,
                                float *scratch, int defer_reduce, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    if (x_stride < n) return einval("mlp_bwd_lm: level stride smaller than n");
    if (!dweights) return einval("mlp_bwd_lm: dweights required");
    return mlp_bwd_impl(x_lm, x_stride, nullptr, weights, nullptr, desc_host, out, acts, dout, dx_lm, dweights, nullptr, scratch, n_cap,
                        n, n_ptr, stream, defer_reduce);
}

ARCN_EXPORT int arcn_mlp_bwd_cat(const float *a, const float *b_table, const int32_t *b_index, int a_first, const float *weights,
                                 const arcn_mlp_desc *desc_host, const float *out, const float *acts, const float *dout, float *da,
                                 const float *d_head, int head_act, float *dweights, float *scratch, int defer_reduce,
                                 int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!a || !b_table || !b_index || !dweights) return einval("mlp_bwd_cat: missing argument");
    MlpCat cat = {b_table, b_index, nullptr, d_head, a_first ? 1 : 0, head_act};
    return mlp_bwd_impl(a, 0, &cat, weights, nullptr, desc_host, out, acts, dout, da, dweights, nullptr, scratch, n_cap, n, n_ptr,
                        stream, defer_reduce);
}

// second half of arcn_mlp_bwd_lm / arcn_mlp_bwd_cat called with defer_reduce = 1: sum the per-workgroup dW partials left in
// `scratch` into dweights (same n_cap and n as the backward call: they fix the number of partials).
ARCN_EXPORT int arcn_mlp_bwd_reduce(const arcn_mlp_desc *desc_host, float *scratch, float *dweights, int64_t n_cap, int64_t n,
                                    void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!scratch || !dweights) return einval("mlp_bwd_reduce: missing argument");
    MlpParams P;
    int lds_floats, md, rc;
    if ((rc = build_mlp_params(desc_host, P, true, &lds_floats, &md))) return rc;
    if (P.has_bias || P.n_layers < 2 || P.n_layers > 3 || md > 64) return einval("mlp_bwd_reduce: not a fused-backward shape");
    int64_t grid = tile_grid(n, 64);
    if (grid > dw_slabs(n_cap)) grid = dw_slabs(n_cap);
    DwParams D;
    D.n_layers = P.n_layers;
    D.has_bias = 0;
    for (int l = 0; l <= P.n_layers; ++l) D.dims[l] = P.dims[l];
    for (int l = 0; l < P.n_layers; ++l) { D.w_off[l] = P.w_off[l]; D.b_off[l] = P.b_off[l]; D.quad_first[l] = l; }
    D.quad_first[P.n_layers] = P.n_layers;
    float *partials = scratch + arcn_mlp_dpre_floats(desc_host, n_cap);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(4096 / kReduceElems, (unsigned)P.n_layers), dim3(kReduceElems * kReduceSlices), 0, as_stream(stream), partials,
                       static_cast<const float *>(nullptr), D, (int)grid, dweights, static_cast<float *>(nullptr));
    return check_launch("mlp_bwd_reduce");
}

static int tail_net(const arcn_mlp_desc *desc_host, float *scratch, int64_t n_cap, int64_t n, int64_t w_seg, TailNet &t) {
    MlpParams P;
    int lds_floats, md, rc;
    if ((rc = build_mlp_params(desc_host, P, true, &lds_floats, &md))) return rc;
    if (P.has_bias || P.n_layers < 2 || P.n_layers > 3 || md > 64) return einval("ngp_step_tail: not a fused-backward shape");
    int64_t grid = tile_grid(n, 64);
    if (grid > dw_slabs(n_cap)) grid = dw_slabs(n_cap);
    t.P.n_layers = P.n_layers;
    t.P.has_bias = 0;
    for (int l = 0; l <= P.n_layers; ++l) t.P.dims[l] = P.dims[l];
    for (int l = 0; l < P.n_layers; ++l) { t.P.w_off[l] = P.w_off[l]; t.P.b_off[l] = P.b_off[l]; t.P.quad_first[l] = l; }
    t.P.quad_first[P.n_layers] = P.n_layers;
    t.partials = scratch + arcn_mlp_dpre_floats(desc_host, n_cap);
    t.n_slots = (int)grid;
    t.blocks = (4096 / kReduceElems) * P.n_layers;
    t.w_seg = w_seg;
    return ARCN_OK;
}

/* The tail of a single-GPU training step of the packed NGP pipeline in one launch.  Replaces, with the same arithmetic,
 *   arcn_mlp_bwd_reduce(desc_a, scratch_a, grad + w_seg_a, ...), arcn_mlp_bwd_reduce(desc_b, scratch_b, grad + w_seg_b, ...)   (deferred dW sums)
 *   arcn_adam_ema_step_runs(param, grad, ..., runs = [the runs given here] + the two weight segments, zero_grad = 1)
 *   and the clearing of `clear_words` 32-bit words at `clear` (the scatter's bin counters, arcn_hashgrid_bwd_counter_block).
 * The two nets' weight segments [w_seg, w_seg + n_weights) must not overlap the runs.  Same n_cap / n as the backward calls that left
 * the partials in the scratch buffers.  Reference: the optimiser step of common/trainer/basic_trainer.py:560-577 on the MLP
 * parameters (torch.optim.Adam, common/trainer/optimizer.py:6-54) + EMA write-back (arcnerf/trainer/ema.py:29-43). */
ARCN_EXPORT int arcn_ngp_step_tail(const arcn_mlp_desc *desc_a, float *scratch_a, int64_t w_seg_a, const arcn_mlp_desc *desc_b, float *scratch_b,
                                   int64_t w_seg_b, int64_t n_cap, int64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                                   float *ema, const int64_t *runs_host, int n_runs, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, float ema_decay, float grad_scale, int step, int ema_step, uint32_t *clear,
                                   int64_t clear_words, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!desc_a || !desc_b || !scratch_a || !scratch_b || !param || !grad || !exp_avg || !exp_avg_sq || n_runs < 0 || n_runs > 4 || step < 1 ||
        (n_runs && !runs_host) || w_seg_a < 0 || w_seg_b < 0 || (clear_words > 0 && !clear))
        return einval("ngp_step_tail: missing/invalid argument");
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(ema)) & 15)
        return einval("ngp_step_tail: buffers must be 16-byte aligned");
    if (ema && ema_step < 1) return einval("ngp_step_tail: ema_step is 1-based");
    TailNet na{}, nb{};
    int rc;
    if ((rc = tail_net(desc_a, scratch_a, n_cap, n, w_seg_a, na))) return rc;
    if ((rc = tail_net(desc_b, scratch_b, n_cap, n, w_seg_b, nb))) return rc;
    AdamRuns r{};
    int total = 0;
    constexpr int kThreads = kReduceElems * kReduceSlices;
    for (int k = 0; k < n_runs; ++k) {
        const int64_t lo = runs_host[2 * k], cnt = runs_host[2 * k + 1];
        if (lo < 0 || cnt < 0 || (lo & 3)) return einval("ngp_step_tail: a run must start at a multiple of 4 floats");
        if (cnt == 0) continue;
        int64_t blocks = ceil_div<int64_t>((cnt >> 2) + 1, kThreads);
        if (blocks > 512) blocks = 512;
        r.lo[r.count] = lo; r.n[r.count] = cnt; r.b[r.count] = (int)blocks;
        total += (int)blocks;
        ++r.count;
    }
    const AdamHyper h = make_adam_hyper(lr, beta1, beta2, eps, weight_decay, ema_decay, grad_scale, step, ema_step, ema != nullptr);
    const int ema_in_param = ema == param;
    const AdamArgs a{lr, beta1, beta2, eps, weight_decay, ema_decay, grad_scale, h.bc1, h.bc2_sqrt, h.deb_old, h.deb_new, 1, ema_in_param};
    int clear_blocks = 0;
    if (clear_words > 0) {
        clear_blocks = (int)ceil_div<int64_t>(clear_words, (int64_t)kThreads * 4);
        if (clear_blocks > 64) clear_blocks = 64;
    }
    if (r.count == 0) { r.count = 1; r.b[0] = 0; }   // nothing left for the plain optimiser: no blocks behind the reductions
    const unsigned grid = (unsigned)(na.blocks + nb.blocks + clear_blocks + total);
    hipLaunchKernelGGL(ngp_step_tail_kernel, dim3(grid), dim3(kThreads), 0, as_stream(stream), na, nb, param, grad, exp_avg, exp_avg_sq,
                       ema_in_param ? static_cast<float *>(nullptr) : ema, r, a, clear, clear_words, clear_blocks);
    return check_launch("ngp_step_tail");
}

ARCN_EXPORT int arcn_mlp_bwd_dw(const float *x, const arcn_mlp_desc *desc_host, const float *acts, float *scratch,
                                float *dweights, float *dbiases, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x || !scratch || !dweights) return einval("mlp_bwd_dw: missing argument");
    MlpParams P;
    int lds_floats, md, rc;
    if ((rc = build_mlp_params(desc_host, P, true, &lds_floats, &md))) return rc;
    if (P.n_layers > 1 && !acts) return einval("mlp_bwd_dw: saved activations required");
    {
        DwParams D;
        D.n_layers = P.n_layers;
        D.has_bias = P.has_bias;
        int quads = 0;
        for (int l = 0; l <= P.n_layers; ++l) D.dims[l] = P.dims[l];
        for (int l = 0; l < P.n_layers; ++l) {
            D.w_off[l] = P.w_off[l];
            D.b_off[l] = P.b_off[l];
            D.quad_first[l] = quads;
            quads += ((tiles16(P.dims[l + 1]) + 3) / 4) * ((tiles16(P.dims[l]) + 3) / 4);
        }
        D.quad_first[P.n_layers] = quads;
        const int64_t slabs = dw_slabs(n_cap);
        float *partials = scratch + arcn_mlp_dpre_floats(desc_host, n_cap);
        float *bias_partials = partials + slabs * quads * 4096;
        hipLaunchKernelGGL(mlp_bwd_dw_kernel, dim3((unsigned)slabs, (unsigned)quads), dim3(256), 0, as_stream(stream), x, acts,
                           scratch, D, partials, bias_partials, n_cap, n, n_ptr);
        hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(4096 / kReduceElems, (unsigned)quads), dim3(kReduceElems * kReduceSlices), 0, as_stream(stream), partials,
                           bias_partials, D, (int)slabs, dweights, dbiases);
        if ((rc = check_launch("mlp_bwd_dw"))) return rc;
    }
    return ARCN_OK;
}

/* see include/arcnerf_hip.h */
static int geo2_check(const float *x_lm, int64_t x_stride, const float *w1, const float *w2, int n_out, int64_t n, const char *who) {
    (void)who;
    if (!x_lm || !w1 || !w2) return einval("geo2: missing argument");
    if (n_out < 1 || n_out > 32) return einval("geo2: 1..32 outputs");
    if (x_stride != 0 && x_stride < n) return einval("geo2: level stride smaller than n (0: row-major features)");
    if ((reinterpret_cast<uintptr_t>(x_lm) & (x_stride ? 7 : 15)) || (reinterpret_cast<uintptr_t>(w2) & 15))
        return einval("geo2: level-major features 8-byte, row-major features and the last layer 16-byte aligned");
    return ARCN_OK;
}

ARCN_EXPORT int arcn_geo2_fwd(const float *x_lm, int64_t x_stride, const float *w1, const float *w2, int n_out, int n_pad, int jac_mode, float beta,
                              float *out, float *head, float *jac, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    int rc;
    if ((rc = geo2_check(x_lm, x_stride, w1, w2, n_out, n, "geo2_fwd"))) return rc;
    if (!out || n_pad < n_out || (n_pad & 3) || n_pad > 32 || (jac_mode && !jac)) return einval("geo2_fwd: out (n, n_pad), n_pad a multiple of 4 in n_out..32; jac in Jacobian mode");
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(jac)) & 15) return einval("geo2_fwd: 16-byte aligned outputs");
    const size_t lds_bytes = sizeof(float) * 3 * kGeo2Frag;
    const unsigned grid = tile_grid(n, 128, kSlimGrid);      // (512 .. 2048 workgroups measured: 35 - 39 us, flat)
    if (jac_mode)
        hipLaunchKernelGGL(geo2_fwd_kernel<true>, dim3(grid), dim3(256), lds_bytes, as_stream(stream), x_lm, x_stride, w1, w2, n_out, n_pad, beta, out, head,
                           jac, n, n_ptr);
    else
        hipLaunchKernelGGL(geo2_fwd_kernel<false>, dim3(grid), dim3(256), lds_bytes, as_stream(stream), x_lm, x_stride, w1, w2, n_out, n_pad, beta, out, head,
                           jac, n, n_ptr);
    return check_launch("geo2_fwd");
}

static int64_t geo2_slots(int64_t n) {
    int64_t grid = tile_grid(n, 64);      // (at most 512 workgroups = two per CU; 1024 / 2048 measured: 58 -> 66 / 78 us)
    return grid;
}

ARCN_EXPORT int64_t arcn_geo2_bwd_scratch_floats(int64_t n) { return n <= 0 ? 0 : 2 * geo2_slots(n) * 4096; }

ARCN_EXPORT int arcn_geo2_bwd(const float *x_lm, int64_t x_stride, const float *w1, const float *w2, int n_out, int jac_mode, float beta,
                              const float *d_col0, int64_t ld_col0, const float *out_col0, int64_t ld_out, const float *d_feat, int64_t ld_feat,
                              const float *d_jac, float *dx, int64_t dx_stride, float *dw1, float *dw2, float *scratch, int64_t n, const int32_t *n_ptr,
                              void *stream) {
    if (n <= 0) return ARCN_OK;
    int rc;
    if ((rc = geo2_check(x_lm, x_stride, w1, w2, n_out, n, "geo2_bwd"))) return rc;
    if (!d_col0 || ld_col0 < 1 || (n_out > 1 && (!d_feat || ld_feat < n_out - 1)) || !dw1 || !dw2 || !scratch) return einval("geo2_bwd: missing argument");
    if (jac_mode ? !d_jac : (!out_col0 || ld_out < 1)) return einval("geo2_bwd: d_jac (Jacobian mode) or the forward's output column 0 (density mode) missing");
    if (dx_stride && dx_stride < n) return einval("geo2_bwd: level stride of dx smaller than n");
    if ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(d_jac)) & 15) return einval("geo2_bwd: 16-byte aligned dx / d_jac");
    const int64_t off2 = dw2 - dw1;
    if (off2 > INT32_MAX || off2 < INT32_MIN) return einval("geo2_bwd: the two weight gradients must lie within 2^31 floats of each other (one flat buffer)");
    const size_t lds_bytes = sizeof(float) * (3 * kGeo2Frag + 8192 + 64);
    const int64_t grid = geo2_slots(n);
    float *partials = scratch;
    if (jac_mode) {
        if ((rc = set_lds(geo2_bwd_kernel<true, 1>, lds_bytes))) return rc;
        hipLaunchKernelGGL((geo2_bwd_kernel<true, 1>), dim3((unsigned)grid), dim3(256), lds_bytes, as_stream(stream), x_lm, x_stride, w1, w2, n_out, beta,
                           d_col0, ld_col0, out_col0, ld_out, d_feat, ld_feat, d_jac, dx, dx_stride, partials, (int)grid, n, n_ptr);
    } else {
        if ((rc = set_lds(geo2_bwd_kernel<false, 1>, lds_bytes))) return rc;
        hipLaunchKernelGGL((geo2_bwd_kernel<false, 1>), dim3((unsigned)grid), dim3(256), lds_bytes, as_stream(stream), x_lm, x_stride, w1, w2, n_out, beta,
                           d_col0, ld_col0, out_col0, ld_out, d_feat, ld_feat, d_jac, dx, dx_stride, partials, (int)grid, n, n_ptr);
    }
    DwParams D;
    D.n_layers = 2;
    D.has_bias = 0;
    D.dims[0] = 32; D.dims[1] = 64; D.dims[2] = n_out;
    D.w_off[0] = 0; D.w_off[1] = (int32_t)off2;
    D.b_off[0] = D.b_off[1] = 0;
    D.quad_first[0] = 0; D.quad_first[1] = 1; D.quad_first[2] = 2;
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(4096 / kReduceElems, 2u), dim3(kReduceElems * kReduceSlices), 0, as_stream(stream), partials,
                       static_cast<const float *>(nullptr), D, (int)grid, dw1, static_cast<float *>(nullptr));
    return check_launch("geo2_bwd");
}

/* see include/arcnerf_hip.h */
ARCN_EXPORT int arcn_ngp_nets_fwd(const float *x_lm, int64_t x_stride, const float *geo_w, const arcn_mlp_desc *geo_desc, float *geo_out,
                                  const float *b_table, const int32_t *b_index, int a_first, const float *rad_w, const arcn_mlp_desc *rad_desc,
                                  float *rgb, float *rad_acts, float *head_out, int head_act, int64_t n_cap, int64_t n, const int32_t *n_ptr,
                                  void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!x_lm || !geo_w || !geo_out || !b_table || !b_index || !rad_w || !rgb) return einval("ngp_nets_fwd: missing argument");
    if (x_stride < n) return einval("ngp_nets_fwd: level stride smaller than n");
    MlpParams G, R;
    int lg, lr, md, rc;
    if ((rc = build_mlp_params(geo_desc, G, false, &lg, &md))) return rc;
    if ((rc = build_mlp_params(rad_desc, R, false, &lr, &md))) return rc;
    const bool geo_ok = G.n_layers == 2 && !G.has_bias && G.dims[0] == 32 && G.dims[1] == 64 && G.dims[2] == 16 && G.act_hidden == ARCN_ACT_RELU &&
                        G.act_out == ARCN_ACT_NONE;
    const bool rad_ok = R.n_layers == 3 && !R.has_bias && R.dims[0] == 32 && R.dims[1] == 64 && R.dims[2] == 64 && R.dims[3] >= 1 && R.dims[3] <= 16 &&
                        R.act_hidden == ARCN_ACT_RELU && R.act_out == ARCN_ACT_SIGMOID;
    if (!geo_ok || !rad_ok)
        return einval("ngp_nets_fwd: wired for the bias-free NGP nets (32 -> 64 ReLU -> 16 linear; 32 -> 64 -> 64 ReLU -> <= 16 sigmoid)");
    MlpCat cat = {b_table, b_index, head_out, nullptr, a_first ? 1 : 0, head_act};
    const size_t lds_bytes = sizeof(float) * (size_t)(lg + lr);
    hipLaunchKernelGGL((ngp_nets_fwd_kernel<2>), dim3(tile_grid(n, 128, kSlimGrid)), dim3(256), lds_bytes, as_stream(stream), x_lm, x_stride, geo_w, G,
                       geo_out, cat, rad_w, R, lg, rgb, rad_acts, n_cap, n, n_ptr);
    return check_launch("ngp_nets_fwd");
}
