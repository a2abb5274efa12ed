// Dense layers of the wide nets on gfx950 matrix cores: the three products a torch.nn.Linear needs in forward, backward and
// double backward, hand-written for a HUGE row count S (samples, 10^5..10^6) and small feature dimensions (3..320), in two forms:
//   exact   v_mfma_f32_16x16x4_f32 (157 TFLOP/s peak): narrow layers (<= 64 outputs), unaligned operands         (first half of the file)
//   split   every f32 operand as three bf16 planes, six v_mfma_f32_16x16x32_bf16 per product, f32 accuracy: the default for layers
//           wider than 64 - 1.4-1.7x the exact form; ReLU masks as bits, bias gradient and softplus fused in     (second half)
//
// Replaces the library GEMMs under GeoNet / RadianceNet (arcnerf/models/base_modules/geo_rad_model/linear_network_module.py:174-197,
// 318-335: nn.Linear / DenseLayer stacks, 8 x 256 with a skip concat for NeRF / NeuS / HDR-NeRF, 32 -> 64 -> 17 for NeuS on the hash
// grid) and their autograd:
//   arcn_gemm_nt   Y (S,N)  = X (S,K) . W(N,K)^T (+ bias)          forward of a layer
//   arcn_gemm_nn   dX (S,K) = dY (S,N) . W (N,K)                   input gradient
//   arcn_gemm_tn   dW (N,K) = dY (S,N)^T . X (S,K)                 weight gradient: a reduction over ALL samples
// The three are closed under differentiation (each one's gradients are the other two), which is what NeuS needs: its normals are
// d sdf / d x taken with create_graph=True and the Eikonal loss differentiates them again (base_network.py:30-44).
//
// Why not the library: the weight gradient of a 64 x 32 or 17 x 64 layer over 3e5 samples is a (tiny x tiny) output with an enormous
// reduction; hipBLASLt runs it without splitting the reduction (MT64x32x128 / MT32x32x256 tiles: 0.94 / 1.46 ms per call, 6.7 ms of a
// 17 ms NeuS-NGP step).  arcn_gemm_tn cuts the samples into slabs, one workgroup per (slab, output tile), partial sums to a scratch
// buffer, and a second kernel adds the slabs in a fixed order (deterministic, no float atomics).
//
// Orientation (as in mlp.hip): MFMA rows = output features, MFMA columns = samples (nt / nn) or input features (tn); with the K order
// k = 16 t + 4 g + ks a lane's four consecutive reduction elements are one 16-byte LDS read that feeds four MFMAs.  Operand tiles are
// staged through LDS in that fragment order, double buffered, 32 reduction elements per stage (128 B per row: whole cache lines).
#include <type_traits>
#include "common.hpp"

namespace arcn {

typedef float gf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gemm_act(float v, int act, float beta) {
    switch (act) {
    case ARCN_ACT_RELU: return v > 0.f ? v : 0.f;
    case ARCN_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    // softplus on the hardware exp2 / log2 (the libm pair costs ~5x more VALU work in the epilogue of a 256-output block than the
    // whole reduction loop): log1p(t) = log(u) t / (u - 1), u = fl(1 + t) - the rounding of 1 + t cancels (Kahan) - so the result
    // is within a few ulp for every t, like libm's, without its cost
    case ARCN_ACT_SOFTPLUS: {
        const float bv = beta * v;
        if (bv > 20.f) return v;
        const float t = __expf(bv), u = 1.f + t;
        return (u == 1.f ? t : __logf(u) * (t / (u - 1.f))) / beta;
    }
    default: return v;
    }
}

// four consecutive elements of row `row` starting at column `col` of a row-major matrix (ld floats per row), zero outside
// [0, rows) x [0, cols); one 16-byte load when the matrix allows it
template <bool ALIGNED>
__device__ __forceinline__ gf4 load_row4(const float *__restrict__ p, int64_t row, int col, int64_t ld, int64_t rows, int cols, bool aligned) {
    gf4 v = {0.f, 0.f, 0.f, 0.f};
    if (ALIGNED) {
        // every row 16-byte aligned, cols a multiple of 4.  Branch-free and select-free: an out-of-range lane reads element (0, 0); the
        // caller zeroes it when the value is USED (stash), so nothing waits for the load before the MFMA block of the current stage
        const bool ok = row < rows && col < cols;
        return *reinterpret_cast<const gf4 *>(p + (ok ? row * ld + col : 0));
    }
    if (row >= rows || col >= cols) return v;
    const float *src = p + row * ld + col;
    if (aligned && col + 3 < cols) return *reinterpret_cast<const gf4 *>(src);
    v.x = src[0];
    if (col + 1 < cols) v.y = src[1];
    if (col + 2 < cols) v.z = src[2];
    if (col + 3 < cols) v.w = src[3];
    return v;
}

// ---- out (S, No) = in (S, Ki) . M,  M[ki][no] = TRANS_W ? W[ki][no] (W row-major (Ki, No)) : W[no][ki] (W row-major (No, Ki)) --------
// 256 threads = 4 waves laid out WAVES_N (outputs) x 4 / WAVES_N (samples); a wave owns MT output tiles x NT sample tiles of 16 x 16.
template <int MT, int NT, int WAVES_N, bool TRANS_W, bool ALIGNED, bool MASK>
__global__ void __launch_bounds__(256)
gemm_rows_kernel(const float *__restrict__ in, const float *__restrict__ mask, int64_t ld_in, const float *__restrict__ W, int ld_w, const float *__restrict__ bias,
                 float *__restrict__ out, int64_t ld_out, int64_t S, const int32_t *n_ptr, int Ki, int No, int act, float beta,
                 int in_aligned, int w_aligned, int out_aligned) {
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int BN = 16 * MT * WAVES_N, BM = 16 * NT * WAVES_M, BK = 32;
    __shared__ __attribute__((aligned(16))) float Ws[2][BN * BK];
    __shared__ __attribute__((aligned(16))) float Xs[2][BM * BK];
    const int64_t cnt = dev_count(S, n_ptr);
    const int64_t s_base = (int64_t)blockIdx.x * BM;
    if (s_base >= cnt) return;
    const int n_base = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave % WAVES_N, wave_m = wave / WAVES_N;
    constexpr int XQ = BM * 8 / 256;           // float4 per thread of the sample tile
    constexpr int WQ = (BN * 8 + 255) / 256;   // ... of the weight tile (a 16-output tile has only 128 of them)
    constexpr int WF = BN * 8;
    gf4 xr[XQ], wr[WQ];
    gf4 mr[MASK ? XQ : 1];         // MASK: `in` is multiplied by (mask > 0) on the way in (ReLU backward: dpre = dy * (y > 0))
    uint32_t okx = 0u, okw = 0u;   // ALIGNED: which of the staged loads were in range (applied in stash)
    const int n_chunks = (Ki + BK - 1) / BK;

    auto fetch = [&](int c) {
        const int k0 = c * BK;
        okx = 0u;
        okw = 0u;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int f = tid + 256 * q, sl = f >> 3, kq = f & 7;
            xr[q] = load_row4<ALIGNED>(in, s_base + sl, k0 + 4 * kq, ld_in, cnt, Ki, in_aligned);
            if (MASK) mr[q] = load_row4<ALIGNED>(mask, s_base + sl, k0 + 4 * kq, ld_in, cnt, Ki, in_aligned);
            okx |= (uint32_t)((s_base + sl < cnt) && (k0 + 4 * kq < Ki)) << q;
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int f = tid + 256 * q;
            if (f >= WF) continue;
            if (!TRANS_W) {
                const int nl = f >> 3, kq = f & 7;
                wr[q] = load_row4<ALIGNED>(W, n_base + nl, k0 + 4 * kq, ld_w, No, Ki, w_aligned);
                okw |= (uint32_t)((n_base + nl < No) && (k0 + 4 * kq < Ki)) << q;
            } else {
                const int kk = f / (BN / 4), nq = f % (BN / 4);   // four consecutive OUTPUTS of reduction row k0 + kk
                wr[q] = load_row4<ALIGNED>(W, k0 + kk, n_base + 4 * nq, ld_w, Ki, No, w_aligned);
                okw |= (uint32_t)((k0 + kk < Ki) && (n_base + 4 * nq < No)) << q;
            }
        }
    };
    auto stash = [&](int buf) {
        if (ALIGNED) {
            const gf4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < XQ; ++q) xr[q] = ((okx >> q) & 1u) ? xr[q] : zero;
#pragma unroll
            for (int q = 0; q < WQ; ++q) wr[q] = ((okw >> q) & 1u) ? wr[q] : zero;
        }
        if (MASK) {
#pragma unroll
            for (int q = 0; q < XQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[q][r] = mr[q][r] > 0.f ? xr[q][r] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int f = tid + 256 * q, sl = f >> 3, kq = f & 7;
            const int t = kq >> 2, g = kq & 3;
            // slot of row j XOR-swizzled with (4 t + g): the 8 lanes that hold one row's 128 bytes land in 8 different bank groups,
            // and the fragment reads below (16 consecutive rows of one (t, g)) stay conflict-free
            *reinterpret_cast<gf4 *>(&Xs[buf][(((sl >> 4) * 2 + t) * 64 + g * 16 + ((sl & 15) ^ (4 * t + g))) * 4]) = xr[q];
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int f = tid + 256 * q;
            if (f >= WF) continue;
            if (!TRANS_W) {
                const int nl = f >> 3, kq = f & 7;
                const int t = kq >> 2, g = kq & 3;
                *reinterpret_cast<gf4 *>(&Ws[buf][(((nl >> 4) * 2 + t) * 64 + g * 16 + ((nl & 15) ^ (4 * t + g) ^ ((nl >> 4) & 3))) * 4]) = wr[q];
            } else {
                const int kk = f / (BN / 4), nq = f % (BN / 4);
                const int t = kk >> 4, g = (kk & 15) >> 2, ks = kk & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nl = 4 * nq + r;
                    // (+ the tile index in the swizzle: the transposed staging writes 4-byte words, 32 lanes of one (t, g, ks))
                    Ws[buf][(((nl >> 4) * 2 + t) * 64 + g * 16 + ((nl & 15) ^ (4 * t + g) ^ ((nl >> 4) & 3))) * 4 + ks] = wr[q][r];
                }
            }
        }
    };

    gf4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = gf4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) fetch(c + 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            gf4 a[MT], b[NT];
            const int sw = (lane & 48) + ((lane & 15) ^ (4 * t + (lane >> 4)));   // swizzled slot of lane (g, i)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                a[m] = *reinterpret_cast<const gf4 *>(&Ws[buf][(((wave_n * MT + m) * 2 + t) * 64 + (sw ^ ((wave_n * MT + m) & 3))) * 4]);
#pragma unroll
            for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const gf4 *>(&Xs[buf][(((wave_m * NT + n) * 2 + t) * 64 + sw) * 4]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][ks], b[n][ks], acc[m][n], 0, 0, 0);
        }
        if (c + 1 < n_chunks) {
            stash(buf ^ 1);     // the other buffer was last read two iterations ago (behind the barrier below)
            __syncthreads();
        }
    }
    // lane (g, j): outputs 16 mt + 4 g + 0..3 of sample 16 nt + j
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int no = n_base + (wave_n * MT + m) * 16 + 4 * g;
        if (no >= No) continue;
        gf4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            bv.x = bias[no];
            if (no + 1 < No) bv.y = bias[no + 1];
            if (no + 2 < No) bv.z = bias[no + 2];
            if (no + 3 < No) bv.w = bias[no + 3];
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int64_t s = s_base + (wave_m * NT + n) * 16 + j;
            if (s >= cnt) continue;
            gf4 v = acc[m][n] + bv;
            if (act != ARCN_ACT_NONE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gemm_act(v[r], act, beta);
            }
            float *dst = out + s * ld_out + no;
            if (out_aligned && no + 3 < No) *reinterpret_cast<gf4 *>(dst) = v;
            else {
                dst[0] = v.x;
                if (no + 1 < No) dst[1] = v.y;
                if (no + 2 < No) dst[2] = v.z;
                if (no + 3 < No) dst[3] = v.w;
            }
        }
    }
}

// ---- partial (slab, N, K) = sum over the slab's samples of A (S,N)^T . B (S,K) ------------------------------------------------------
// waves 2 x 2; a wave owns MT row tiles (features of A) x NT column tiles (features of B)
template <int MT, int NT, bool ALIGNED, bool MASK>
__global__ void __launch_bounds__(256)
gemm_tn_kernel(const float *__restrict__ A, const float *__restrict__ mask, int64_t ld_a, const float *__restrict__ B, int64_t ld_b, float *__restrict__ scratch,
               int64_t S, const int32_t *n_ptr, int N, int K, int n_slabs, int a_aligned, int b_aligned) {
    constexpr int BN = 32 * MT, BKo = 32 * NT, BS = 32;
    __shared__ __attribute__((aligned(16))) float As[2][BN * BS];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKo * BS];
    const int64_t cnt = dev_count(S, n_ptr);
    const int slab = blockIdx.x;
    const int tiles_k = (K + BKo - 1) / BKo;
    const int tn = blockIdx.y / tiles_k, tk = blockIdx.y % tiles_k;
    const int n_base = tn * BN, k_base = tk * BKo;
    // slabs are multiples of 32 samples
    const int64_t per = ((cnt + n_slabs - 1) / n_slabs + BS - 1) / BS * BS;
    const int64_t s_lo = (int64_t)slab * per, s_hi = (s_lo + per < cnt) ? s_lo + per : cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave & 1, wave_k = wave >> 1;
    constexpr int AQ = BN * 8 / 256, BQ = BKo * 8 / 256;   // float4 per thread per stage (32 samples x BN / 4)
    gf4 ar[AQ], br[BQ];
    gf4 mr[MASK ? AQ : 1];
    uint32_t oka = 0u, okb = 0u;
    gf4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = gf4{0.f, 0.f, 0.f, 0.f};
    const int64_t n_chunks = s_lo < s_hi ? (s_hi - s_lo + BS - 1) / BS : 0;

    auto fetch = [&](int64_t c) {
        const int64_t s0 = s_lo + c * BS;
        oka = 0u;
        okb = 0u;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int f = tid + 256 * q, sl = f / (BN / 4), nq = f % (BN / 4);
            ar[q] = load_row4<ALIGNED>(A, s0 + sl, n_base + 4 * nq, ld_a, s_hi, N, a_aligned);
            if (MASK) mr[q] = load_row4<ALIGNED>(mask, s0 + sl, n_base + 4 * nq, ld_a, s_hi, N, a_aligned);
            oka |= (uint32_t)((s0 + sl < s_hi) && (n_base + 4 * nq < N)) << q;
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int f = tid + 256 * q, sl = f / (BKo / 4), kq = f % (BKo / 4);
            br[q] = load_row4<ALIGNED>(B, s0 + sl, k_base + 4 * kq, ld_b, s_hi, K, b_aligned);
            okb |= (uint32_t)((s0 + sl < s_hi) && (k_base + 4 * kq < K)) << q;
        }
    };
    // fragment order [feature tile][t][g][feature in tile][ks] with sample = 16 t + 4 g + ks: a lane's 16-byte read = 4 samples
    auto stash = [&](int buf) {
        if (ALIGNED) {
            const gf4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < AQ; ++q) ar[q] = ((oka >> q) & 1u) ? ar[q] : zero;
#pragma unroll
            for (int q = 0; q < BQ; ++q) br[q] = ((okb >> q) & 1u) ? br[q] : zero;
        }
        if (MASK) {
#pragma unroll
            for (int q = 0; q < AQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) ar[q][r] = mr[q][r] > 0.f ? ar[q][r] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int f = tid + 256 * q, sl = f / (BN / 4), nq = f % (BN / 4);
            const int t = sl >> 4, g = (sl & 15) >> 2, ks = sl & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nl = 4 * nq + r;
                // feature slot XOR-swizzled with the tile index: the 32 lanes of a store spread over 16 bank groups instead of 4
                As[buf][(((nl >> 4) * 2 + t) * 64 + g * 16 + ((nl & 15) ^ ((nl >> 4) & 3))) * 4 + ks] = ar[q][r];
            }
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int f = tid + 256 * q, sl = f / (BKo / 4), kq = f % (BKo / 4);
            const int t = sl >> 4, g = (sl & 15) >> 2, ks = sl & 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kl = 4 * kq + r;
                Bs[buf][(((kl >> 4) * 2 + t) * 64 + g * 16 + ((kl & 15) ^ ((kl >> 4) & 3))) * 4 + ks] = br[q][r];
            }
        }
    };
    if (n_chunks > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int buf = (int)(c & 1);
        if (c + 1 < n_chunks) fetch(c + 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            gf4 a[MT], b[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                a[m] = *reinterpret_cast<const gf4 *>(&As[buf][(((wave_n * MT + m) * 2 + t) * 64 + (lane & 48) + ((lane & 15) ^ ((wave_n * MT + m) & 3))) * 4]);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                b[n] = *reinterpret_cast<const gf4 *>(&Bs[buf][(((wave_k * NT + n) * 2 + t) * 64 + (lane & 48) + ((lane & 15) ^ ((wave_k * NT + n) & 3))) * 4]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][ks], b[n][ks], acc[m][n], 0, 0, 0);
        }
        if (c + 1 < n_chunks) {
            stash(buf ^ 1);
            __syncthreads();
        }
    }
    // lane (g, j): rows n = 16 mt + 4 g + r, column k = 16 nt + j
    const int g = lane >> 4, j = lane & 15;
    float *dst = scratch + (int64_t)slab * N * K;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int k = k_base + (wave_k * NT + n) * 16 + j;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = n_base + (wave_n * MT + m) * 16 + 4 * g + r;
                if (row < N) dst[(int64_t)row * K + k] = acc[m][n][r];
            }
        }
}

// out[i] (+)= sum over slabs in a fixed order: 4 lanes per element take the slabs k = lane, lane + 4, ... with 4 independent
// accumulators each (16 loads in flight per element instead of one dependent chain), combined in a fixed tree
// (elements [n_first, n_elem) of a slab go to out2: the column sums that gemm_tn_split_kernel appends)
__global__ void __launch_bounds__(256) gemm_tn_reduce_kernel(const float *__restrict__ scratch, float *__restrict__ out, int64_t n_elem, int64_t stride,
                                                             int n_slabs, int accumulate, int64_t n_first, float *__restrict__ out2) {
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;   // 0..3
    __shared__ float red[4][64];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n_elem) {
        int k = part;
        for (; k + 12 < n_slabs; k += 16) {
            a0 += scratch[(int64_t)k * stride + i];
            a1 += scratch[(int64_t)(k + 4) * stride + i];
            a2 += scratch[(int64_t)(k + 8) * stride + i];
            a3 += scratch[(int64_t)(k + 12) * stride + i];
        }
        for (; k < n_slabs; k += 4) a0 += scratch[(int64_t)k * stride + i];
    }
    red[part][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (part == 0 && i < n_elem) {
        const int j = threadIdx.x & 63;
        const float s = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
        float *o = i < n_first ? out + i : out2 + (i - n_first);
        *o = accumulate ? *o + s : s;
    }
}

// ---- split-bf16 forms: f32 operands as three bf16 planes, six bf16 MFMAs per product ------------------------------------------------
// The f32 MFMA runs at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s).  An f32 value is EXACTLY hi + mid + lo with three bf16 numbers
// (8 + 8 + 8 significand bits, the exponent range of f32): hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), every subtraction
// exact.  A product a.b then needs the six partial products down to 2^-16 of it -- hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid -- each
// exact in the matrix core's f32 accumulation; what is dropped (mid.lo, lo.mid, lo.lo) is < 2^-24 |a b|, i.e. below the rounding of
// an f32 multiply.  Six v_mfma_f32_16x16x32_bf16 (16 cycles each) replace eight v_mfma_f32_16x16x4_f32 (32 cycles each) per 32
// reduction elements: 2.67 x the matrix rate at f32 accuracy (tests/test_gpu_kernels.py::test_gemm_products_vs_float64).  The cost is
// the split itself (5.5 VALU operations per element, once per staged element) and 6 instead of 4 LDS bytes per element.
// Inf / NaN inputs give NaN (inf - inf in the split) where an f32 product might give inf.
typedef uint32_t gu4 __attribute__((ext_vector_type(4)));
typedef __bf16 gbf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// eight consecutive reduction elements -> their three bf16 planes (16 bytes each)
__device__ __forceinline__ void split3(const float (&x)[8], gu4 &hi, gu4 &mid, gu4 &lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        const uint32_t h = pk_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
        const uint32_t m = pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
        hi[p] = h;
        mid[p] = m;
        lo[p] = pk_bf16(s0, s1);
    }
}

__device__ __forceinline__ gf4 mfma_bf16(const gu4 &a, const gu4 &b, const gf4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gbf8, a), __builtin_bit_cast(gbf8, b), c, 0, 0, 0);
}

// the six partial products, smallest first, of MT x 1 tiles: planes 0 / 1 / 2 = hi / mid / lo
#define ARCN_SPLIT_TERMS(OP)  OP(0, 2) OP(2, 0) OP(1, 1) OP(0, 1) OP(1, 0) OP(0, 0)

// weight operand, split once per call into MFMA fragment order: Wf[((c * tiles + m) * 3 + plane) * 64 + lane], lane (g, i) = the
// reduction elements 32 c + 4 g + 0..3 and 32 c + 16 + 4 g + 0..3 of output feature 16 m + i (zero outside the matrix).  (Which eight
// elements a lane group holds is free as long as both operands agree; this choice lets four neighbouring lanes of the sample operand's
// loads cover 64 contiguous bytes of a row.)
__global__ void __launch_bounds__(256) gemm_split_w_kernel(const float *__restrict__ W, int ld_w, int trans, int No, int Ki, gu4 *__restrict__ out) {
    const int tiles = (No + 15) / 16, chunks = (Ki + 31) / 32 + 1;   // + one all-zero stage (k >= Ki)
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= chunks * tiles) return;
    const int c = idx / tiles, m = idx % tiles;
    const int no = 16 * m + (lane & 15), k0 = 32 * c + 4 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + (e < 4 ? e : 12 + e);
        v[e] = (no < No && k < Ki) ? (trans ? W[(int64_t)k * ld_w + no] : W[(int64_t)no * ld_w + k]) : 0.f;
    }
    gu4 h, md, l;
    split3(v, h, md, l);
    out[((int64_t)idx * 3 + 0) * 64 + lane] = h;
    out[((int64_t)idx * 3 + 1) * 64 + lane] = md;
    out[((int64_t)idx * 3 + 2) * 64 + lane] = l;
}

// out (S, No) = act(in (S, Ki) . M + bias) with M given as Wf; 128 samples x 128 outputs per workgroup, waves 2 (outputs) x 2 (samples),
// a wave owns 4 x 4 tiles.  The sample operand goes global -> registers -> split -> LDS (double buffered, 24 KB per stage), the weight
// fragments go global (L2) -> registers, one stage ahead.  `in` rows must be 16-byte aligned, Ki a multiple of 4.
// (MT, NT, WAVES_N) = (2, 8, 4): 128 outputs per workgroup, three workgroups per CU; (4, 8, 4): 256 outputs -- the sample operand is
// read, split and staged once for twice the MFMAs (128 accumulator registers, two workgroups per CU).  The kernel covers the outputs
// [n_off, min(n_off + gridDim.y * BN, No)).
// MASK: 0 none, 1 the float tensor `mask` (layout of `in`; in * (mask > 0)), 2 a BIT mask: `mask` points at uint32 words in the layout
// the forward product of a ReLU layer writes through `relu_bits`: word [s / 8][f / 4] holds the 8 rows 8 (s / 8) + i x 4 features
// 4 (f / 4) + r of y as bit 4 i + r = (y > 0) (ceil(S / 8) x N / 4 words).  The backward then reads 1 bit instead of 32 per masked
// element (a third less HBM traffic for the masked products), and the 8-row x 4-feature block is exactly what ONE staging thread of
// the weight-gradient kernel masks: one 4-byte load instead of eight 16-byte ones.
template <int MASK, int MT, int NT, int WAVES_N, bool BITS_OUT>
__global__ void __launch_bounds__(256, (MASK == 1 || MT * NT > 16) ? 2 : 3)
gemm_rows_split_kernel(const float *__restrict__ in, const float *__restrict__ mask, int64_t ld_in, const gu4 *__restrict__ Wf, const float *__restrict__ bias,
                       float *__restrict__ out, uint32_t *__restrict__ relu_bits, int64_t ld_out, int64_t S, const int32_t *n_ptr, int Ki, int No, int n_off,
                       int act, float beta, int out_aligned) {
    constexpr int BN = 16 * MT * WAVES_N, BM = 128;
    static_assert(16 * NT * (4 / WAVES_N) == BM, "the staging below is written for 128 samples per workgroup");
    // a sample tile's three planes take 192 16-byte slots; tiles start 193 apart, so tile n sits n bank groups further on: the four rows of
    // a write phase are four TILES (below) and must not meet in one bank group, while a lane's read address stays base + n * constant
    constexpr int kTileSlots = 193;
    __shared__ __attribute__((aligned(16))) gu4 Xs[2][(BM / 16) * kTileSlots];
    const int64_t cnt = dev_count(S, n_ptr);
    const int64_t s_base = (int64_t)blockIdx.x * BM;
    if (s_base >= cnt) return;
    const int n_base = n_off + blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave % WAVES_N, wave_m = wave / WAVES_N;
    const int tiles = (No + 15) / 16;
    const int tile0 = n_base / 16 + wave_n * MT;
    const int n_chunks = (Ki + 31) / 32;
    // two register sets: the loads of stage c + 2 are issued before the MFMA block of stage c (two MFMA blocks to land in)
    struct Staged {
        gf4 x[2][2];
        gf4 m[MASK == 1 ? 2 : 1][2];
        uint32_t mb[2][2];
        uint32_t ok;
    };
    Staged ra, rb;
    const uint32_t *__restrict__ mbits = reinterpret_cast<const uint32_t *>(mask);
    const int mquads = Ki >> 2;

    auto fetch = [&](int c, Staged &st) {
        st.ok = 0u;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + 256 * q, sl = f >> 2, g = f & 3;
            const int col = 32 * c + 4 * g;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                st.x[q][h] = load_row4<true>(in, s_base + sl, col + 16 * h, ld_in, cnt, Ki, 1);
                if (MASK == 1) st.m[q][h] = load_row4<true>(mask, s_base + sl, col + 16 * h, ld_in, cnt, Ki, 1);
                st.ok |= (uint32_t)((s_base + sl < cnt) && (col + 16 * h < Ki)) << (2 * q + h);
            }
            if (MASK == 2) {   // the words of this row's octet and of the feature quads 8 c + g (+ 4): bits 4 (row & 7) + r
                const int64_t oct = (s_base + sl < cnt ? s_base + sl : 0) >> 3;
                const int quad = 8 * c + g;
#pragma unroll
                for (int h = 0; h < 2; ++h) st.mb[q][h] = mbits[oct * mquads + (quad + 4 * h < mquads ? quad + 4 * h : 0)];
            }
        }
    };
    auto stash = [&](int buf, const Staged &st) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + 256 * q, sl = f >> 2, g = f & 3;
            float x[8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = ((st.ok >> (2 * q + h)) & 1u) ? st.x[q][h][r] : 0.f;
                    if (MASK == 1) v = st.m[q][h][r] > 0.f ? v : 0.f;
                    if (MASK == 2) v = ((st.mb[q][h] >> (4 * (sl & 7) + r)) & 1u) ? v : 0.f;
                    x[4 * h + r] = v;
                }
            gu4 hi, md, lo;
            split3(x, hi, md, lo);
            // row sl = 8 j + n is column j of sample tile n: the eight tiles of a lane are eight CONSECUTIVE rows, i.e. one word of the ReLU
            // bit mask is a lane's own business in the epilogue (no cross-lane gather).  Slot of column j rotated by 4 g, tile n by n (its
            // base): the 16 lanes of one write phase (4 rows = 4 tiles x 4 groups) land in 16 different 16-byte bank groups
            gu4 *dst = &Xs[buf][(sl & 7) * kTileSlots + g * 16 + (((sl >> 3) + 4 * g) & 15)];
            dst[0] = hi;
            dst[64] = md;
            dst[128] = lo;
            if (MT * NT > 16) __builtin_amdgcn_sched_barrier(0);   // (one unit at a time where registers are short)
        }
    };
    // (a tile past the last one re-reads the last: its products are never stored)
    auto loadw = [&](int c, gu4 (&w)[MT][3]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int tile = tile0 + m < tiles ? tile0 + m : tiles - 1;
            const gu4 *src = Wf + ((int64_t)(c * tiles + tile) * 3) * 64 + lane;
            w[m][0] = src[0];
            w[m][1] = src[64];
            w[m][2] = src[128];
        }
    };

    // lane (g, j): outputs 16 mt + 4 g + 0..3 of sample 8 j + nt.  The bias is the accumulators' starting value: its loads travel with
    // the first operand stage instead of waiting at the end of the kernel, and the epilogue has nothing to add
    static_assert(NT == 8 && WAVES_N == 4, "row = 8 j + tile: eight sample tiles per wave, all waves along the outputs");
    const int g = lane >> 4, j = lane & 15;
    gf4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int no = n_base + (wave_n * MT + a) * 16 + 4 * g;
        gf4 bv = gf4{0.f, 0.f, 0.f, 0.f};
        if (bias && no < No) {
            bv.x = bias[no];
            if (no + 1 < No) bv.y = bias[no + 1];
            if (no + 2 < No) bv.z = bias[no + 2];
            if (no + 3 < No) bv.w = bias[no + 3];
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = bv;
    }

    auto compute = [&](int buf, const gu4 (&w)[MT][3]) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const gu4 *xb = &Xs[buf][(wave_m * NT + n) * kTileSlots + (lane & 48) + ((lane + ((lane >> 4) << 2)) & 15)];
            gu4 b[3];
            b[0] = xb[0];
            b[1] = xb[64];
            b[2] = xb[128];
#define ARCN_OP(PW, PX) _Pragma("unroll") for (int m = 0; m < MT; ++m) acc[m][n] = mfma_bf16(w[m][PW], b[PX], acc[m][n]);
            ARCN_SPLIT_TERMS(ARCN_OP)
#undef ARCN_OP
        }
    };

    // Two stages per trip with two register sets: the loads of stage c + 2 are issued before the MFMA block of stage c and consumed after
    // the MFMA block of stage c + 1.  Everything in the loop is unconditional -- a conditional consumer lets the compiler sink the loads
    // down to it, behind the MFMA block -- so a stage past the end re-reads the last stage (clamped) against the all-zero weight stage
    // that the split kernel appends (Wf stage n_chunks), and an odd stage count costs one idle MFMA block.
    // (the masked 256-output form has no registers for the second set: one set, loads one MFMA block ahead)
    constexpr bool TWO = !(MASK != 0 && MT * NT > 16);
    constexpr int D = TWO ? 2 : 1;
    Staged &r_odd = TWO ? rb : ra;
    const int last = n_chunks - 1;
    gu4 wa[MT][3];
    fetch(0, ra);
    if (TWO) fetch(last < 1 ? last : 1, rb);
    stash(0, ra);
    __syncthreads();
    for (int c = 0; c < n_chunks; c += 2) {
        loadw(c, wa);
        __builtin_amdgcn_sched_barrier(0);   // (weights first: vmcnt retires in order, and the MFMAs wait for the weights only)
        fetch(c + D < last ? c + D : last, ra);
        __builtin_amdgcn_sched_barrier(0);
        compute(0, wa);
        stash(1, r_odd);
        __syncthreads();
        loadw(c + 1, wa);
        __builtin_amdgcn_sched_barrier(0);
        fetch(c + 1 + D < last ? c + 1 + D : last, r_odd);
        __builtin_amdgcn_sched_barrier(0);
        compute(1, wa);
        stash(0, ra);
        __syncthreads();
    }
    // Epilogue with the activation as a compile-time constant (ACT < 0: whatever `act` says, at run time).  ReLU bit mask of the outputs:
    // a lane's four outputs of a sample are a nibble and its eight sample tiles are the eight rows 8 j .. 8 j + 7 of ONE word - built in
    // the lane, one 4-byte store per output tile (the four lane groups write 16 contiguous bytes).
    const int quads = No >> 2;
    const int64_t s0 = s_base + 8 * j;
    auto finish = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int no = n_base + (wave_n * MT + m) * 16 + 4 * g;
            uint32_t wd = 0u;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int64_t s = s0 + n;
                gf4 v = acc[m][n];
                if (ACT == ARCN_ACT_RELU) {
                    uint32_t nib = 0u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        nib |= (uint32_t)(v[r] > 0.f) << r;
                        v[r] = v[r] > 0.f ? v[r] : 0.f;
                    }
                    if (BITS_OUT) wd |= (s < cnt ? nib : 0u) << (4 * n);
                } else if (ACT < 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gemm_act(v[r], act, beta);
                }
                if (s >= cnt || no >= No) continue;
                float *dst = out + s * ld_out + no;
                if (out_aligned && no + 3 < No) *reinterpret_cast<gf4 *>(dst) = v;
                else {
                    dst[0] = v.x;
                    if (no + 1 < No) dst[1] = v.y;
                    if (no + 2 < No) dst[2] = v.z;
                    if (no + 3 < No) dst[3] = v.w;
                }
            }
            if (BITS_OUT && ACT == ARCN_ACT_RELU && relu_bits && s0 < cnt && no < No) relu_bits[(s0 >> 3) * quads + (no >> 2)] = wd;
        }
    };
    if (act == ARCN_ACT_RELU) finish(std::integral_constant<int, ARCN_ACT_RELU>{});
    else if (BITS_OUT || act == ARCN_ACT_NONE) finish(std::integral_constant<int, ARCN_ACT_NONE>{});
    else finish(std::integral_constant<int, -1>{});
}

// partial (slab, N, K) = sum over the slab's samples of A (S,N)^T . B (S,K), split form.  128 x 128 outputs per workgroup (waves 2 x 2,
// 4 x 4 tiles each), 32 samples per stage.  The reduction runs over SAMPLES, so a lane group's eight reduction elements are eight
// rows of the operands: a thread loads the same four features of eight consecutive samples (eight 16-byte loads; 32 lanes cover 512
// contiguous bytes of a row), which is four fragments' worth of eight-sample columns in its own registers -- the transposition costs
// nothing -- splits them and writes 16-byte fragment slots.  One LDS stage (48 KB: three workgroups per CU) with the next stage's loads
// in flight over the MFMA block.  Rows 16-byte aligned, N and K multiples of 4.
// MASK: 0 none, 1 float mask tensor, 2 ReLU bit mask (8-row x 4-feature words; see gemm_rows_split_kernel)
template <int MASK, bool COLSUM>
__global__ void __launch_bounds__(256, (MASK != 0 || COLSUM) ? 2 : 3)
gemm_tn_split_kernel(const float *__restrict__ A, const float *__restrict__ mask, int64_t ld_a, const float *__restrict__ B, int64_t ld_b, float *__restrict__ scratch,
                     int64_t S, const int32_t *n_ptr, int N, int K, int n_slabs, int want_colsum) {
    constexpr int MT = 4, NT = 4, BN = 128, BKo = 128, BS = 32;
    __shared__ __attribute__((aligned(16))) gu4 Ls[2 * 8 * 192];   // [operand][feature tile][plane][g][feature slot]
    const int64_t cnt = dev_count(S, n_ptr);
    const int slab = blockIdx.x;
    const int tiles_k = (K + BKo - 1) / BKo;
    const int tn = blockIdx.y / tiles_k, tk = blockIdx.y % tiles_k;
    const int n_base = tn * BN, k_base = tk * BKo;
    const int64_t per = ((cnt + n_slabs - 1) / n_slabs + BS - 1) / BS * BS;
    const int64_t s_lo = (int64_t)slab * per, s_hi = (s_lo + per < cnt) ? s_lo + per : cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave & 1, wave_k = wave >> 1;
    // staging role: waves 0, 1 the A operand, waves 2, 3 the B operand; unit = (feature quad, sample octet)
    const bool is_b = wave >= 2;
    const float *__restrict__ src = is_b ? B : A;
    const int64_t ld_s = is_b ? ld_b : ld_a;
    const int f_base = is_b ? k_base : n_base, f_cnt = is_b ? K : N;
    const int unit = tid & 127, nq = unit & 31, sg = unit >> 5;
    const int fcol = (f_base + 4 * nq < f_cnt) ? f_base + 4 * nq : 0;   // features past the matrix: any valid column (never stored)
    gf4 xr[8];
    gf4 mr[MASK == 1 ? 8 : 1];
    uint32_t mb = 0xffffffffu;   // MASK == 2: ONE word = this thread's 8 rows x 4 features of the stage
    const uint32_t *__restrict__ mbits = reinterpret_cast<const uint32_t *>(mask);
    const int mquads = N >> 2, mquad = fcol >> 2;   // (A-role threads: fcol is a column of dy)
    uint32_t ok = 0u;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};   // want_colsum: column sums of the (masked) A operand = the layer's bias gradient, for free
    const int64_t n_chunks = s_lo < s_hi ? (s_hi - s_lo + BS - 1) / BS : 0;

    // (uniform stage base + one 32-bit byte offset per row: half the address registers of eight 64-bit pointers)
    auto fetch = [&](int64_t c, bool live) {
        ok = 0u;
        const int64_t row0 = s_lo + c * BS;
        if (MASK == 2 && !is_b) {   // (s_lo and the stages are multiples of 32 rows; octets past the last row: any valid word)
            const int64_t oct = (row0 >> 3) + sg, oct_last = (cnt - 1) >> 3;
            mb = mbits[(oct < oct_last ? oct : oct_last) * mquads + mquad];
        }
        const char *base = reinterpret_cast<const char *>(src + row0 * ld_s);
        const char *mbase = reinterpret_cast<const char *>(mask + row0 * ld_s);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = live && row0 + 8 * sg + e < s_hi;   // (!live: the stage re-read past the end counts as all-invalid rows)
            const uint32_t off = (uint32_t)(((in ? 8 * sg + e : 0) * ld_s + fcol) * 4);
            xr[e] = *reinterpret_cast<const gf4 *>(base + off);
            if (MASK == 1) mr[e] = *reinterpret_cast<const gf4 *>((is_b ? base : mbase) + off);
            ok |= (uint32_t)in << e;
        }
    };
    auto stash = [&]() {
        gu4 *blk = &Ls[(is_b ? 8 * 192 : 0) + (nq >> 2) * 192 + sg * 16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = ((ok >> e) & 1u) ? xr[e][r] : 0.f;
                if (MASK == 1) v = (is_b || mr[e][r] > 0.f) ? v : 0.f;
                if (MASK == 2) v = __uint_as_float(__float_as_uint(v) & (uint32_t)__builtin_amdgcn_sbfe((int)mb, 4 * e + r, 1));   // bit -> 0 / ~0
                x[e] = v;
            }
            if (COLSUM) csum[r] += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
            gu4 hi, md, lo;
            split3(x, hi, md, lo);
            // feature slot rotated by the tile index: the 16 lanes of a write phase (4 tiles x 4 quads) hit 16 different bank groups
            gu4 *dst = blk + ((4 * (nq & 3) + r + (nq >> 2)) & 15);
            dst[0] = hi;
            dst[64] = md;
            dst[128] = lo;
            __builtin_amdgcn_sched_barrier(0);   // one feature at a time: interleaving the four splits costs 36 more live registers
        }
    };

    gf4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = gf4{0.f, 0.f, 0.f, 0.f};

    // output tiles in two halves of 2 x 4: 24 registers of A fragments instead of 48 (the B fragments are read twice)
    auto compute = [&]() {
#pragma unroll
        for (int mh = 0; mh < MT; mh += 2) {
            gu4 a[2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int tile = wave_n * MT + mh + m;
                const gu4 *ab = &Ls[tile * 192 + (lane & 48) + ((lane + tile) & 15)];
                a[m][0] = ab[0];
                a[m][1] = ab[64];
                a[m][2] = ab[128];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int tile = wave_k * NT + n;
                const gu4 *bb = &Ls[8 * 192 + tile * 192 + (lane & 48) + ((lane + tile) & 15)];
                gu4 b[3];
                b[0] = bb[0];
                b[1] = bb[64];
                b[2] = bb[128];
#define ARCN_OP(PW, PX) _Pragma("unroll") for (int m = 0; m < 2; ++m) acc[mh + m][n] = mfma_bf16(a[m][PW], b[PX], acc[mh + m][n]);
                ARCN_SPLIT_TERMS(ARCN_OP)
#undef ARCN_OP
            }
        }
    };

    if (n_chunks > 0) {
        fetch(0, true);
        stash();
        __syncthreads();
        for (int64_t c = 0; c < n_chunks; ++c) {
            // (unconditional, like the consumers below: see gemm_rows_split_kernel; past the end: the last stage again, as zeros)
            fetch(c + 1 < n_chunks ? c + 1 : c, c + 1 < n_chunks);
            __builtin_amdgcn_sched_barrier(0);
            compute();
            __syncthreads();
            stash();
            __syncthreads();
        }
    }
    // column sums: the four sample octets of a feature quad live in four threads (unit = quad + 32 octet) of waves 0 and 1
    float *slab_out = scratch + (int64_t)slab * ((int64_t)N * K + N);
    if (COLSUM && want_colsum && tk == 0) {
        float *red = reinterpret_cast<float *>(Ls);
        if (!is_b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[unit * 4 + r] = csum[r];
        }
        __syncthreads();
        if (tid < BN && n_base + tid < N) {
            const int q = tid >> 2, r = tid & 3;
            slab_out[(int64_t)N * K + n_base + tid] = (red[q * 4 + r] + red[(q + 32) * 4 + r]) + (red[(q + 64) * 4 + r] + red[(q + 96) * 4 + r]);
        }
    }
    // lane (g, j): rows n = 16 mt + 4 g + r, column k = 16 nt + j
    const int g = lane >> 4, j = lane & 15;
    float *dst = slab_out;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int k = k_base + (wave_k * NT + n) * 16 + j;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = n_base + (wave_n * MT + m) * 16 + 4 * g + r;
                if (row < N) dst[(int64_t)row * K + k] = acc[m][n][r];
            }
        }
}

static inline int is_aligned(const void *p, int64_t ld) { return ((reinterpret_cast<uintptr_t>(p) & 15u) == 0 && (ld & 3) == 0) ? 1 : 0; }

static int tn_slabs(int64_t S, int N, int K, int bn, int bk) {
    const int64_t tiles = (int64_t)((N + bn - 1) / bn) * ((K + bk - 1) / bk);
    int64_t slabs = 1024 / (tiles > 0 ? tiles : 1);
    if (slabs > 256) slabs = 256;                // the slab partials are summed by a second kernel: keep that sum short
    const int64_t max_slabs = (S + 255) / 256;   // at least 256 samples per slab
    if (slabs > max_slabs) slabs = max_slabs;
    if (slabs < 1) slabs = 1;
    if (slabs > 1024) slabs = 1024;
    return (int)slabs;
}

}  // namespace arcn

using namespace arcn;

static int gemm_rows(bool trans_w, const float *in, const float *mask, int64_t ld_in, const float *W, int ld_w, const float *bias, float *out, int64_t ld_out,
                     int64_t S, const int32_t *n_ptr, int Ki, int No, int act, float beta, void *stream) {
    if (S <= 0) return ARCN_OK;
    if (!in || !W || !out || Ki < 1 || No < 1) return einval("gemm: missing / invalid argument");
    const int ia = is_aligned(in, ld_in), wa = is_aligned(W, ld_w), oa = is_aligned(out, ld_out);
    // the branch-free 16-byte loads need both operands aligned and both of THEIR column counts multiples of 4
    const bool all_al = ia && wa && (Ki & 3) == 0 && (!trans_w || (No & 3) == 0);
#define ARCN_ROWS_L(MT_, NT_, WN_, TW_, AL_, MK_)                                                                                      \
    hipLaunchKernelGGL((gemm_rows_kernel<MT_, NT_, WN_, TW_, AL_, MK_>), grid, dim3(256), 0, as_stream(stream), in, mask, ld_in, W, ld_w,   \
                       bias, out, ld_out, S, n_ptr, Ki, No, act, beta, ia, wa, oa)
#define ARCN_ROWS(MT_, NT_, WN_)                                                                                                      \
    do {                                                                                                                              \
        constexpr int BN_ = 16 * MT_ * WN_, BM_ = 16 * NT_ * (4 / WN_);                                                               \
        dim3 grid((unsigned)ceil_div<int64_t>(S, BM_), (unsigned)ceil_div<int>(No, BN_));                                            \
        if (mask) {   /* masked operand: the aligned instantiations only (the callers pad) */                                         \
            if (!all_al) return einval("gemm: a masked operand needs 16-byte aligned rows and feature counts that are multiples of 4"); \
            if (trans_w) ARCN_ROWS_L(MT_, NT_, WN_, true, true, true);                                                                 \
            else ARCN_ROWS_L(MT_, NT_, WN_, false, true, true);                                                                        \
        } else if (trans_w && all_al) ARCN_ROWS_L(MT_, NT_, WN_, true, true, false);                                                   \
        else if (trans_w) ARCN_ROWS_L(MT_, NT_, WN_, true, false, false);                                                              \
        else if (all_al) ARCN_ROWS_L(MT_, NT_, WN_, false, true, false);                                                               \
        else ARCN_ROWS_L(MT_, NT_, WN_, false, false, false);                                                                          \
    } while (0)
    if (No > 64) ARCN_ROWS(4, 4, 2);        // 128 outputs x 128 samples per workgroup
    else if (No > 32) ARCN_ROWS(4, 2, 1);   // 64 outputs x 128 samples
    else if (No > 16) ARCN_ROWS(2, 2, 1);   // 32 outputs x 128 samples
    else ARCN_ROWS(1, 2, 1);                // 16 outputs x 128 samples
#undef ARCN_ROWS
#undef ARCN_ROWS_L
    return check_launch("gemm_rows");
}

/* Y (S,N) = act(X (S,K) . W (N,K)^T + bias): nn.Linear / DenseLayer forward (linear_network_module.py:174-197, linear.py). */
ARCN_EXPORT int arcn_gemm_nt(const float *x, int64_t ld_x, const float *w, const float *bias, float *y, int64_t ld_y, int64_t n_rows,
                             const int32_t *n_ptr, int K, int N, int act, float beta, void *stream) {
    return gemm_rows(false, x, nullptr, ld_x, w, K, bias, y, ld_y, n_rows, n_ptr, K, N, act, beta, stream);
}

/* dX (S,K) = (dY (S,N) * (mask > 0)) . W (N,K): input gradient of the layer; mask (same layout as dy, may be NULL) = the layer's
 * ReLU output, which folds the activation's backward into the operand load. */
ARCN_EXPORT int arcn_gemm_nn(const float *dy, const float *mask, int64_t ld_dy, const float *w, float *dx, int64_t ld_dx, int64_t n_rows,
                             const int32_t *n_ptr, int N, int K, void *stream) {
    return gemm_rows(true, dy, mask, ld_dy, w, K, nullptr, dx, ld_dx, n_rows, n_ptr, N, K, ARCN_ACT_NONE, 1.0f, stream);
}

/* bytes of the split-weight workspace of arcn_gemm_nt_split / arcn_gemm_nn_split for n_out output features and k_red reduction
 * elements (three bf16 planes in MFMA fragment order, padded to 16 x 32 tiles) */
ARCN_EXPORT int64_t arcn_gemm_split_bytes(int n_out, int k_red) {
    if (n_out < 1 || k_red < 1) return 0;
    return (int64_t)(ceil_div<int>(k_red, 32) + 1) * ceil_div<int>(n_out, 16) * 3 * 64 * 16;
}

static void split_weights(const float *W, int ld_w, bool trans_w, int No, int Ki, void *ws, void *stream) {
    const int frag_tiles = (ceil_div<int>(Ki, 32) + 1) * ceil_div<int>(No, 16);
    hipLaunchKernelGGL(gemm_split_w_kernel, dim3((unsigned)ceil_div<int>(frag_tiles, 4)), dim3(256), 0, as_stream(stream), W, ld_w, trans_w ? 1 : 0, No, Ki,
                       reinterpret_cast<gu4 *>(ws));
}

/* The weight operand of arcn_gemm_nt_split (transposed = 0: w (n_out, k_red) row-major, row stride ld_w) or of arcn_gemm_nn_split
 * (transposed = 1: w (k_red, n_out), the layer's weight used the other way round) as three bf16 planes in MFMA fragment order, written to
 * ws (>= arcn_gemm_split_bytes(n_out, k_red) bytes).  A caller that evaluates the same layer on several chunks of samples splits once
 * and passes ws_ready = 1 to the products (linear_network_module.py:174-197 runs under chunk_processing: 8 chunks per training step). */
ARCN_EXPORT int arcn_gemm_split_weights(const float *w, int ld_w, int transposed, int n_out, int k_red, void *ws, int64_t ws_bytes, void *stream) {
    if (!w || !ws || n_out < 1 || k_red < 1 || ld_w < (transposed ? n_out : k_red)) return einval("gemm_split_weights: missing / invalid argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) != 0 || ws_bytes < arcn_gemm_split_bytes(n_out, k_red))
        return einval("gemm_split_weights: workspace misaligned or smaller than arcn_gemm_split_bytes");
    split_weights(w, ld_w, transposed != 0, n_out, k_red, ws, stream);
    return check_launch("gemm_split_weights");
}

static int gemm_rows_split(bool trans_w, const float *in, const float *mask, const uint32_t *mask_bits, int64_t ld_in, const float *W, int ld_w,
                           const float *bias, float *out, uint32_t *relu_bits, int64_t ld_out, int64_t S, const int32_t *n_ptr, int Ki, int No, int act,
                           float beta, void *ws, int64_t ws_bytes, int ws_ready, void *stream) {
    if (S <= 0) return ARCN_OK;
    if (!in || !W || !out || !ws || Ki < 1 || No < 1) return einval("gemm_split: missing / invalid argument");
    if (!is_aligned(in, ld_in) || (Ki & 3) != 0 || (reinterpret_cast<uintptr_t>(ws) & 15u) != 0 || (mask && !is_aligned(mask, ld_in)))
        return einval("gemm_split: the row operand needs 16-byte aligned rows and a reduction length that is a multiple of 4");
    if (relu_bits && ((No & 3) != 0 || act != ARCN_ACT_RELU)) return einval("gemm_split: relu_bits needs a ReLU epilogue and outputs in multiples of 4");
    if ((mask || mask_bits) && relu_bits) return einval("gemm_split: relu_bits is an output of the unmasked forward product");
    const int mk = mask_bits ? 2 : (mask ? 1 : 0);
    if (mask_bits) mask = reinterpret_cast<const float *>(mask_bits);
    if (ws_bytes < arcn_gemm_split_bytes(No, Ki)) return einval("gemm_split: workspace smaller than arcn_gemm_split_bytes");
    if (!ws_ready) split_weights(W, ld_w, trans_w, No, Ki, ws, stream);   // (ws_ready: the caller did, arcn_gemm_split_weights)
    const int oa = is_aligned(out, ld_out);
    const unsigned gx = (unsigned)ceil_div<int64_t>(S, 128);
    static const int big = 1;
#define ARCN_RS(MT_, NBLK_, OFF_)                                                                                                          \
    do {                                                                                                                                   \
        dim3 grid(gx, (unsigned)(NBLK_));                                                                                                  \
        if (mk == 2)                                                                                                                       \
            hipLaunchKernelGGL((gemm_rows_split_kernel<2, MT_, 8, 4, false>), grid, dim3(256), 0, as_stream(stream), in, mask, ld_in,     \
                               reinterpret_cast<const gu4 *>(ws), bias, out, relu_bits, ld_out, S, n_ptr, Ki, No, OFF_, act, beta, oa);    \
        else if (mk == 1)                                                                                                                  \
            hipLaunchKernelGGL((gemm_rows_split_kernel<1, MT_, 8, 4, false>), grid, dim3(256), 0, as_stream(stream), in, mask, ld_in,     \
                               reinterpret_cast<const gu4 *>(ws), bias, out, relu_bits, ld_out, S, n_ptr, Ki, No, OFF_, act, beta, oa);    \
        else if (relu_bits)                                                                                                                \
            hipLaunchKernelGGL((gemm_rows_split_kernel<0, MT_, 8, 4, true>), grid, dim3(256), 0, as_stream(stream), in, mask, ld_in,      \
                               reinterpret_cast<const gu4 *>(ws), bias, out, relu_bits, ld_out, S, n_ptr, Ki, No, OFF_, act, beta, oa);    \
        else                                                                                                                               \
            hipLaunchKernelGGL((gemm_rows_split_kernel<0, MT_, 8, 4, false>), grid, dim3(256), 0, as_stream(stream), in, mask, ld_in,     \
                               reinterpret_cast<const gu4 *>(ws), bias, out, relu_bits, ld_out, S, n_ptr, Ki, No, OFF_, act, beta, oa);    \
    } while (0)
    // the outputs in blocks: 256 wide while they last, then what remains -- <= 16 outputs (257 = 256 + 1) on the exact-f32 kernel's
    // 16-output tiles (one more pass over the rows, memory bound), <= 128 as one 128-wide block, more as a partly empty 256-wide one
    int done = 0;
    if (big && No >= 256) {
        ARCN_RS(4, No / 256, 0);
        done = No / 256 * 256;
    }
    const int rem = No - done;
    if (rem > 0 && rem <= 16 && done > 0 && mk != 2 && !relu_bits) {   // (the exact kernel neither reads nor writes bit masks)
        const float *Wr = trans_w ? W + done : W + (int64_t)done * ld_w;
        const int rc = gemm_rows(trans_w, in, mask, ld_in, Wr, ld_w, bias ? bias + done : nullptr, out + done, ld_out, S, n_ptr, Ki, rem, act, beta, stream);
        if (rc != ARCN_OK) return rc;
    } else if (rem > 128 && big) {
        ARCN_RS(4, 1, done);
    } else if (rem > 0) {
        ARCN_RS(2, ceil_div<int>(rem, 128), done);
    }
#undef ARCN_RS
    return check_launch("gemm_rows_split");
}

/* arcn_gemm_nt on the bf16 matrix rate: x and w as three bf16 planes each, six MFMAs per product, f32 accuracy (see the kernel).  x rows
 * 16-byte aligned, K a multiple of 4; `ws` = arcn_gemm_split_bytes(N, K) bytes of device scratch (the split weights).  relu_bits (may be
 * NULL; act = ReLU, N % 4 == 0): ceil(n_rows / 8) x N / 4 words, word [s / 8][f / 4] bit 4 (s % 8) + (f % 4) = (y[s][f] > 0) - the mask
 * the layer's backward needs, 1/32 of the bytes of y. */
ARCN_EXPORT int arcn_gemm_nt_split(const float *x, int64_t ld_x, const float *w, const float *bias, float *y, uint32_t *relu_bits, int64_t ld_y,
                                   int64_t n_rows, const int32_t *n_ptr, int K, int N, int act, float beta, void *ws, int64_t ws_bytes, int ws_ready,
                                   void *stream) {
    return gemm_rows_split(false, x, nullptr, nullptr, ld_x, w, K, bias, y, relu_bits, ld_y, n_rows, n_ptr, K, N, act, beta, ws, ws_bytes, ws_ready, stream);
}

/* arcn_gemm_nn likewise: dy rows 16-byte aligned, N a multiple of 4; `ws` = arcn_gemm_split_bytes(K, N) bytes.  mask_bits (may be NULL,
 * then `mask` applies): the relu_bits of the layer's forward instead of its float output as the mask. */
ARCN_EXPORT int arcn_gemm_nn_split(const float *dy, const float *mask, const uint32_t *mask_bits, int64_t ld_dy, const float *w, float *dx, int64_t ld_dx,
                                   int64_t n_rows, const int32_t *n_ptr, int N, int K, void *ws, int64_t ws_bytes, int ws_ready, void *stream) {
    return gemm_rows_split(true, dy, mask, mask_bits, ld_dy, w, K, nullptr, dx, nullptr, ld_dx, n_rows, n_ptr, N, K, ARCN_ACT_NONE, 1.0f, ws, ws_bytes,
                           ws_ready, stream);
}

ARCN_EXPORT int64_t arcn_gemm_tn_scratch_floats(int64_t n_rows, int N, int K) {
    const bool small = (N <= 64 && K <= 64);
    const int slabs = small ? tn_slabs(n_rows, N, K, 64, 64) : tn_slabs(n_rows, N, K, 128, 128);
    return (int64_t)slabs * ((int64_t)N * K + N);   // (+ N per slab: the column sums of arcn_gemm_tn_split)
}

/* dW (N,K) (+)= dY (S,N)^T . X (S,K): weight gradient, reduced over all rows in a fixed order (slab partials in `scratch`, at least
 * arcn_gemm_tn_scratch_floats(n_rows, N, K) floats). */
static int gemm_tn_impl(const float *dy, const float *mask, int64_t ld_dy, const float *x, int64_t ld_x, float *dw, float *scratch,
                        int64_t scratch_floats, int64_t n_rows, const int32_t *n_ptr, int N, int K, int n_head, int accumulate, void *stream) {
    if (!dy || !x || !dw || !scratch || N < 1 || K < 1 || n_head < 1 || n_head > N) return einval("gemm_tn: missing / invalid argument");
    if (scratch_floats < arcn_gemm_tn_scratch_floats(n_rows, N, K)) return einval("gemm_tn: scratch smaller than arcn_gemm_tn_scratch_floats");
    const int aa = is_aligned(dy, ld_dy), ba = is_aligned(x, ld_x);
    const bool al = aa && ba && (N & 3) == 0 && (K & 3) == 0;
    if (mask && !al) return einval("gemm_tn: a masked operand needs 16-byte aligned rows and feature counts that are multiples of 4");
    const bool small = (N <= 64 && K <= 64);
    int slabs;
    if (n_rows <= 0) {
        slabs = 0;
    } else if (small) {
        slabs = tn_slabs(n_rows, N, K, 64, 64);
        dim3 grid((unsigned)slabs, (unsigned)(ceil_div<int>(N, 64) * ceil_div<int>(K, 64)));
        if (mask) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, true, true>), grid, dim3(256), 0, as_stream(stream), dy, mask, ld_dy, x, ld_x, scratch, n_rows, n_ptr, N, K, slabs, aa, ba);
        else if (al) hipLaunchKernelGGL((gemm_tn_kernel<2, 2, true, false>), grid, dim3(256), 0, as_stream(stream), dy, mask, ld_dy, x, ld_x, scratch, n_rows, n_ptr, N, K, slabs, aa, ba);
        else hipLaunchKernelGGL((gemm_tn_kernel<2, 2, false, false>), grid, dim3(256), 0, as_stream(stream), dy, mask, ld_dy, x, ld_x, scratch, n_rows, n_ptr, N, K, slabs, aa, ba);
    } else {
        slabs = tn_slabs(n_rows, N, K, 128, 128);
        dim3 grid((unsigned)slabs, (unsigned)(ceil_div<int>(N, 128) * ceil_div<int>(K, 128)));
        if (mask) hipLaunchKernelGGL((gemm_tn_kernel<4, 4, true, true>), grid, dim3(256), 0, as_stream(stream), dy, mask, ld_dy, x, ld_x, scratch, n_rows, n_ptr, N, K, slabs, aa, ba);
        else if (al) hipLaunchKernelGGL((gemm_tn_kernel<4, 4, true, false>), grid, dim3(256), 0, as_stream(stream), dy, mask, ld_dy, x, ld_x, scratch, n_rows, n_ptr, N, K, slabs, aa, ba);
        else hipLaunchKernelGGL((gemm_tn_kernel<4, 4, false, false>), grid, dim3(256), 0, as_stream(stream), dy, mask, ld_dy, x, ld_x, scratch, n_rows, n_ptr, N, K, slabs, aa, ba);
    }
    const int64_t n_elem = (int64_t)N * K, n_write = (int64_t)n_head * K;     // (the slabs hold N rows; the first n_head of them are summed and written)
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)ceil_div<int64_t>(n_write, 64)), dim3(256), 0, as_stream(stream), scratch, dw,
                       n_write, n_elem, slabs, accumulate, n_write, (float *)nullptr);
    return check_launch("gemm_tn");
}

ARCN_EXPORT int arcn_gemm_tn(const float *dy, const float *mask, int64_t ld_dy, const float *x, int64_t ld_x, float *dw, float *scratch,
                             int64_t scratch_floats, int64_t n_rows, const int32_t *n_ptr, int N, int K, int accumulate, void *stream) {
    return gemm_tn_impl(dy, mask, ld_dy, x, ld_x, dw, scratch, scratch_floats, n_rows, n_ptr, N, K, N, accumulate, stream);
}

/* arcn_gemm_tn whose result keeps only the first n_head of the N rows: dw (n_head, K) (+)= (dy^T x)[:n_head].  For a layer whose output is
 * padded to a multiple of 4 columns (the 1 + 16 outputs of a geometry net as 20): the product runs on the aligned 20-column operand, the
 * weight gradient of the 17 real rows lands - accumulated - in the layer's own gradient buffer. */
ARCN_EXPORT int arcn_gemm_tn_head(const float *dy, const float *mask, int64_t ld_dy, const float *x, int64_t ld_x, float *dw, float *scratch,
                                  int64_t scratch_floats, int64_t n_rows, const int32_t *n_ptr, int N, int K, int n_head, int accumulate,
                                  void *stream) {
    return gemm_tn_impl(dy, mask, ld_dy, x, ld_x, dw, scratch, scratch_floats, n_rows, n_ptr, N, K, n_head, accumulate, stream);
}

/* arcn_gemm_tn on the bf16 matrix rate (split form, see arcn_gemm_nt_split): same arguments and scratch; dy, mask and x rows 16-byte
 * aligned, N and K multiples of 4.  db (N floats, may be NULL) (+)= the column sums of dy' -- the layer's bias gradient, summed from the
 * operand as it is staged (no second pass over dy).  mask_bits (may be NULL): the forward's relu_bits as the mask. */
ARCN_EXPORT int arcn_gemm_tn_split(const float *dy, const float *mask, const uint32_t *mask_bits, int64_t ld_dy, const float *x, int64_t ld_x, float *dw,
                                   float *db, float *scratch, int64_t scratch_floats, int64_t n_rows, const int32_t *n_ptr, int N, int K, int accumulate,
                                   void *stream) {
    if (!dy || !x || !dw || !scratch || N < 1 || K < 1) return einval("gemm_tn_split: missing / invalid argument");
    const int mk = mask_bits ? 2 : (mask ? 1 : 0);
    if (scratch_floats < arcn_gemm_tn_scratch_floats(n_rows, N, K)) return einval("gemm_tn_split: scratch smaller than arcn_gemm_tn_scratch_floats");
    if (!is_aligned(dy, ld_dy) || !is_aligned(x, ld_x) || (N & 3) != 0 || (K & 3) != 0 || (mask && !is_aligned(mask, ld_dy)))
        return einval("gemm_tn_split: operands need 16-byte aligned rows and feature counts that are multiples of 4");
    int slabs = 0;
    if (n_rows > 0) {
        // one full round of workgroups: 256 CUs x 3 (unmasked: 168 registers) or x 2 (masked) resident workgroups; the generic choice
        // (1024 / tiles) is 1.33 rounds of the unmasked kernel - a third of the chip idle for the second round - and twice the partials
        const int tiles = ceil_div<int>(N, 128) * ceil_div<int>(K, 128);
        static const int slots_u = 768;
        const int slots = (mk != 0 || db) ? 512 : slots_u;
        slabs = tn_slabs(n_rows, N, K, 128, 128);
        if (slabs > slots / tiles) slabs = slots / tiles > 0 ? slots / tiles : 1;
        dim3 grid((unsigned)slabs, (unsigned)tiles);
#define ARCN_TNS(MK_, CS_)                                                                                                             \
    hipLaunchKernelGGL((gemm_tn_split_kernel<MK_, CS_>), grid, dim3(256), 0, as_stream(stream), dy, mptr, ld_dy, x, ld_x, scratch, n_rows, \
                       n_ptr, N, K, slabs, db ? 1 : 0)
        const float *mptr = mask_bits ? reinterpret_cast<const float *>(mask_bits) : mask;
        if (mk == 2 && db) ARCN_TNS(2, true);
        else if (mk == 2) ARCN_TNS(2, false);
        else if (mk == 1 && db) ARCN_TNS(1, true);
        else if (mk == 1) ARCN_TNS(1, false);
        else if (db) ARCN_TNS(0, true);
        else ARCN_TNS(0, false);
#undef ARCN_TNS
    }
    const int64_t n_first = (int64_t)N * K, n_elem = n_first + (db ? N : 0);
    // (the slab stride is N K + N whether or not db is wanted; without db the last N of each slab are not read)
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)ceil_div<int64_t>(n_elem, 64)), dim3(256), 0, as_stream(stream), scratch, dw,
                       n_elem, n_first + N, slabs, accumulate, n_first, db);
    return check_launch("gemm_tn_split");
}
