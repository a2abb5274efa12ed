// Multiresolution hash-grid encoding for gfx950: forward gather + backward scatter.
//
// Restates HashGridEmbedder.hashgrid_encode_torch (arcnerf/models/base_modules/encoding/hashgrid_encoder.py:191-249)
// with the voxel maths of Volume.get_voxel_grid_info_from_xyz (arcnerf/geometry/volume.py:486-531):
//   per level l: v = (x-min)/((max-min)/res_l); valid iff 0 <= v < res_l; c = floor(v);
//   8 corners in the order (x,y,z) in {000,010,100,110,001,011,101,111} (volume.py:156-182);
//   row = ((cx*1) ^ (cy*2654435761) ^ (cz*805459861)) mod size_l + offset_l, int64 arithmetic;
//   w = clip((x - (c*vs + min_x))/vs, 0, 1); out = sum_corner table[row] * wx*wy*wz.
// Integer rows are bit-exact with the reference (tests compare them); power-of-two level sizes take a 32-bit mask path
// (low bits of the xor are unaffected by the high product bits), the others an exact 64-bit modulo done with one f64
// multiply + fix-up (h < 2^53).
//
// Work decomposition v1: one lane per (sample, level), level fastest -> a 64-lane wave covers 4 samples x 16 levels,
// xyz loads broadcast, the (n, L*F) row-major output is written fully coalesced (8 B per lane, 512 B per wave).
// Algorithmic HBM bytes per sample (NGP config, F=2, L=16, fp32): 16*8*2*4 gathered + 12 in + 128 out = 1164 B.
#include "common.hpp"
#include "adam.hpp"

namespace arcn {

struct LevelParams {
    int32_t res;
    uint32_t size;     // rows of this level
    uint32_t mask;     // size-1 if size is a power of two else 0
    float vs[3];       // voxel size (max - min) / res per axis: the same fp32 division the reference does, done once on the host
    float rvs[3];      // 1 / vs (only for the backward's interpolation weights, which carry a tolerance, never for the cell index)
    int32_t pad;
    double inv_size;   // 1.0/size
    int64_t offset;    // first row
};

struct GridParams {
    int32_t L, F;
    float mn[3], mx[3];
    LevelParams lv[ARCN_MAX_LEVELS];
};

__device__ __forceinline__ uint32_t hash_row(uint32_t cx, uint32_t cy, uint32_t cz, const LevelParams &lp) {
    if (lp.mask) {
        uint32_t h = cx ^ (cy * 2654435761u) ^ (cz * 805459861u);
        return h & lp.mask;
    }
    uint64_t h = (uint64_t)cx ^ ((uint64_t)cy * 2654435761ull) ^ ((uint64_t)cz * 805459861ull);
    double q = floor((double)h * lp.inv_size);
    int64_t r = (int64_t)h - (int64_t)q * (int64_t)lp.size;
    if (r < 0) r += lp.size;
    else if (r >= (int64_t)lp.size) r -= lp.size;
    return (uint32_t)r;
}

struct Cell {
    uint32_t c[3];
    float w[3];
    float dw[3];  // d w / d x (0 where torch.clip blocks the gradient)
    bool valid;
};

// FASTW: interpolation weights through the precomputed reciprocal (1 ulp off the divided value; the backward's tolerance
// absorbs it).  The cell index always uses the exact division, so the rows stay bit-identical to the forward / the reference.
template <bool FASTW = false>
__device__ __forceinline__ Cell locate(const float p[3], const GridParams &g, const LevelParams &lp) {
    Cell cell;
    float v[3];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = (p[k] - g.mn[k]) / lp.vs[k];
        if (!(v[k] >= 0.f) || !(v[k] < (float)lp.res)) ok = false;
    }
    cell.valid = ok;
    if (!ok) return cell;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float cf = floorf(v[k]);
        cell.c[k] = (uint32_t)cf;
        float a = cf * lp.vs[k];
        float g0 = a + g.mn[0];  // start_point[0] for every axis (volume.py:515)
        float ww = FASTW ? (p[k] - g0) * lp.rvs[k] : (p[k] - g0) / lp.vs[k];
        cell.w[k] = ww < 0.0f ? 0.0f : (ww > 1.0f ? 1.0f : ww);
        cell.dw[k] = (ww >= 0.0f && ww <= 1.0f) ? 1.0f / lp.vs[k] : 0.0f;
    }
    return cell;
}

template <int F>
__global__ void __launch_bounds__(256) hashgrid_fwd_kernel(const float *__restrict__ xyz, const float *__restrict__ table,
                                                           GridParams g, float *__restrict__ out,
                                                           int32_t *__restrict__ hash_idx, int64_t n,
                                                           const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= cnt * g.L) return;
    const int64_t s = gid / g.L;
    const int l = (int)(gid - s * g.L);
    const LevelParams lp = g.lv[l];
    const float p[3] = {xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2]};
    const Cell cell = locate(p, g, lp);
    float acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
    if (cell.valid) {
        uint32_t rows[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // corner order of Volume.get_eight_permutation: x = (q>>1)&1, y = q&1, z = q>>2
            const uint32_t ox = (q >> 1) & 1, oy = q & 1, oz = q >> 2;
            rows[q] = hash_row(cell.c[0] + ox, cell.c[1] + oy, cell.c[2] + oz, lp);
        }
        float vals[8][F];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float *src = table + ((int64_t)rows[q] + lp.offset) * F;
            if (F == 2) {
                float2 t2 = *reinterpret_cast<const float2 *>(src);
                vals[q][0] = t2.x;
                vals[q][1 % F] = t2.y;
            } else if (F == 4) {
                float4 t4 = *reinterpret_cast<const float4 *>(src);
                vals[q][0] = t4.x; vals[q][1 % F] = t4.y; vals[q][2 % F] = t4.z; vals[q][3 % F] = t4.w;
            } else {
#pragma unroll
                for (int f = 0; f < F; ++f) vals[q][f] = src[f];
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t ox = (q >> 1) & 1, oy = q & 1, oz = q >> 2;
            float wx = ox ? cell.w[0] : 1.0f - cell.w[0];
            float wy = oy ? cell.w[1] : 1.0f - cell.w[1];
            float wz = oz ? cell.w[2] : 1.0f - cell.w[2];
            float wt = (wx * wy) * wz;
#pragma unroll
            for (int f = 0; f < F; ++f) { float a = vals[q][f] * wt; acc[f] = acc[f] + a; }
            if (hash_idx) hash_idx[gid * 8 + q] = (int32_t)((int64_t)rows[q] + lp.offset);
        }
    } else {
        if (hash_idx) {
#pragma unroll
            for (int q = 0; q < 8; ++q) hash_idx[gid * 8 + q] = -1;
        }
    }
    float *o = out + gid * F;
    if (F == 2) *reinterpret_cast<float2 *>(o) = make_float2(acc[0], acc[1 % F]);
    else if (F == 4) *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1 % F], acc[2 % F], acc[3 % F]);
    else {
#pragma unroll
        for (int f = 0; f < F; ++f) o[f] = acc[f];
    }
}

// Sum the per-level contributions to a sample's 3-vector and add it to out[3*s..]: one lane per (sample, level) with the L levels
// of a sample in consecutive lanes.  When L is a power of two (<= 64) the L lanes sit in one wavefront: butterfly reduce, the
// level-0 lane adds with a plain read-modify-write (single owner) - otherwise one float atomic per lane and component.
// Every lane of the group must call this (contribute zeros instead of returning early).
__device__ __forceinline__ void add_over_levels(float *__restrict__ out, int64_t s, int l, int L, float v[3]) {
    if ((L & (L - 1)) == 0 && L <= 64) {
        for (int off = 1; off < L; off <<= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] += __shfl_xor(v[k], off, 64);
        }
        if (l == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) out[3 * s + k] += v[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) unsafeAtomicAdd(&out[3 * s + k], v[k]);
    }
}

template <int F>
__global__ void __launch_bounds__(256) hashgrid_bwd_kernel(const float *__restrict__ xyz, const float *__restrict__ table,
                                                           const float *__restrict__ dout, GridParams g,
                                                           float *__restrict__ dtable, float *__restrict__ dxyz, int64_t n,
                                                           const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= cnt * g.L) return;
    const int64_t s = gid / g.L;
    const int l = (int)(gid - s * g.L);
    const LevelParams lp = g.lv[l];
    const float p[3] = {xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2]};
    const Cell cell = locate(p, g, lp);
    float go[F];
#pragma unroll
    for (int f = 0; f < F; ++f) go[f] = dout[gid * F + f];
    float gx[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (!cell.valid) break;
        const uint32_t ox = (q >> 1) & 1, oy = q & 1, oz = q >> 2;
        const int64_t row = (int64_t)hash_row(cell.c[0] + ox, cell.c[1] + oy, cell.c[2] + oz, lp) + lp.offset;
        float wx = ox ? cell.w[0] : 1.0f - cell.w[0];
        float wy = oy ? cell.w[1] : 1.0f - cell.w[1];
        float wz = oz ? cell.w[2] : 1.0f - cell.w[2];
        float wt = (wx * wy) * wz;
        if (dtable) {
#pragma unroll
            for (int f = 0; f < F; ++f) unsafeAtomicAdd(&dtable[row * F + f], go[f] * wt);
        }
        if (dxyz) {
            float dot = 0.f;
#pragma unroll
            for (int f = 0; f < F; ++f) dot += go[f] * table[row * F + f];
            float sx = ox ? 1.0f : -1.0f, sy = oy ? 1.0f : -1.0f, sz = oz ? 1.0f : -1.0f;
            gx[0] += dot * sx * wy * wz * cell.dw[0];
            gx[1] += dot * wx * sy * wz * cell.dw[1];
            gx[2] += dot * wx * wy * sz * cell.dw[2];
        }
    }
    if (dxyz) add_over_levels(dxyz, s, l, g.L, gx);
}

// ---- second order: the input gradient dx = J(x; table)^T dout differentiated once more ---------------------------------
// NeuS on a hash grid takes normals = d sdf / d x with create_graph=True and puts a loss on them (BaseGeoNet.forward_with_grad,
// base_network.py; the reference's torch backend differentiates hashgrid_encode_torch twice by autograd).  With the upstream
// gradient gdx (n,3) on dx and D_q = sum_k gdx_k dW_q/dp_k:
//   ddout[s,l,f]    = sum_q T[r_q,f] D_q                      gather with weights D
//   dtable[r_q,f]  += dout_f D_q                              scatter with weights D
//   d2xyz[s,j]     += sum_q <dout, T[r_q]> dD_q/dp_j          cross terms only (every a_k is piecewise linear in p_k)
// One lane per (sample, level); any output may be null.  d2xyz accumulates over the levels with atomics (caller zeroes it).
template <int F>
__global__ void __launch_bounds__(256)
hashgrid_bwd_bwd_kernel(const float *__restrict__ xyz, const float *__restrict__ gdx, const float *__restrict__ table,
                        const float *__restrict__ dout, GridParams g, float *__restrict__ ddout, float *__restrict__ dtable,
                        float *__restrict__ d2xyz, int64_t n, const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= cnt * g.L) return;
    const int64_t s = gid / g.L;
    const int l = (int)(gid - s * g.L);
    const LevelParams lp = g.lv[l];
    const float p[3] = {xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2]};
    const Cell cell = locate(p, g, lp);
    float acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
    float hsum[3] = {0.f, 0.f, 0.f};
    if (cell.valid) {
        const float gd[3] = {gdx[3 * s], gdx[3 * s + 1], gdx[3 * s + 2]};
        float go[F];
#pragma unroll
        for (int f = 0; f < F; ++f) go[f] = dout ? dout[gid * F + f] : 0.f;
        float hx[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t o[3] = {(uint32_t)((q >> 1) & 1), (uint32_t)(q & 1), (uint32_t)(q >> 2)};
            const int64_t row = (int64_t)hash_row(cell.c[0] + o[0], cell.c[1] + o[1], cell.c[2] + o[2], lp) + lp.offset;
            float a[3], sd[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a[k] = o[k] ? cell.w[k] : 1.0f - cell.w[k];
                sd[k] = o[k] ? cell.dw[k] : -cell.dw[k];
            }
            float D = gd[0] * sd[0] * a[1] * a[2];
            D = D + gd[1] * a[0] * sd[1] * a[2];
            D = D + gd[2] * a[0] * a[1] * sd[2];
            float dot = 0.f;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float t = table[row * F + f];
                acc[f] = acc[f] + t * D;
                dot += go[f] * t;
                if (dtable) unsafeAtomicAdd(&dtable[row * F + f], go[f] * D);
            }
            hx[0] += dot * sd[0] * (gd[1] * sd[1] * a[2] + gd[2] * a[1] * sd[2]);
            hx[1] += dot * sd[1] * (gd[0] * sd[0] * a[2] + gd[2] * a[0] * sd[2]);
            hx[2] += dot * sd[2] * (gd[0] * sd[0] * a[1] + gd[1] * a[0] * sd[1]);
        }
        hsum[0] = hx[0]; hsum[1] = hx[1]; hsum[2] = hx[2];
    }
    if (d2xyz) add_over_levels(d2xyz, s, l, g.L, hsum);
    if (ddout) {
#pragma unroll
        for (int f = 0; f < F; ++f) ddout[gid * F + f] = acc[f];
    }
}

// ---- forward, XCD-affine: the idea (round 1; the kernel that implements it is the cost-balanced one below) -------------------------
// MI355X: 8 XCDs, each with a private 4 MiB L2; one fine level of the table is exactly 4 MiB.  The kernel above lets every
// XCD touch all 16 levels (48.8 MB), so fine-level gathers miss L2 and are served by the Infinity Cache.  Here a workgroup
// handles ONE level for 256 consecutive samples and the level is chosen from the workgroup's position in the dispatch order
// (workgroup b runs on XCD b % 8 - observed, used for speed only, any placement is correct): XCD x works through level x for
// all its tiles, then level L-1-x, so at any time an XCD's L2 mostly holds a single level.
// LM: level-major output out[(l * n_cap + s) * F + f] (coalesced stores); otherwise the usual row-major (n, L*F).
// ---- forward, XCD-affine, cost-balanced (v2) ----------------------------------------------------------------------------------
// Measured per-level cost of the kernel above on ONE XCD (2.6e5 samples, tools/exp_gather.py --levels): 18 us for the coarse
// power-of-two levels (an instruction floor: six correctly rounded fp32 divisions and eight hashes per lane), 28 us for the
// non-power-of-two levels 1-4 (eight 64-bit modulos through f64), rising to 70 us at level 15 (every corner a new 128-byte line out
// of the XCD's L2).  Pairing level x with level 15-x therefore leaves XCD 1 with 96 us of work and XCD 7 with 42.  This kernel
//   * takes a PLAN of (level, sample-range fraction, workgroups) segments per XCD built from a per-level cost model so that
//     all XCDs finish together (a fine level may spill a sliver onto the neighbouring XCD);
//   * divides by the voxel size with the precomputed correctly rounded reciprocal + one fused residual step (Markstein:
//     q0 = n*r, e = fma(-q0, vs, n), q = fma(e, r, q0) - equal to the IEEE quotient in 1.4e9 randomised cases per level constant,
//     tools/ubench/markstein.c); the CELL INDEX, which must be bit-exact, additionally takes the true division whenever a lane's
//     quotient lies within 4e-7 relative of an integer (a wave-uniform branch, taken by ~10 % of the waves on the finest level);
//   * on the non-power-of-two levels computes the modulo once per (y, z) corner pair: with kb = bits of res + 1,
//     h = B ^ cx = (B & ~lowmask) + ((B & lowmask) ^ cx), so row = ((B & ~lowmask) mod size + ((B & lowmask) ^ cx)) mod size and
//     the second mod is one conditional subtract - 4 f64 modulos per lane instead of 8.
// Accumulation order and arithmetic are those of the kernels above: results are bit-identical.
struct FwdSeg {
    int32_t level;
    uint32_t f0, f1;      // sample-tile range of the segment as fractions (x / 65536) of the launch's tile count
    int32_t wg0, wg1;     // positions in the XCD's workgroup queue that work on it
};
constexpr int kFwdSegs = 8;
struct FwdPlan {
    FwdSeg seg[8][kFwdSegs];
    int32_t n_seg[8];
    int32_t lowbits[ARCN_MAX_LEVELS];   // non-power-of-two levels: kb (0 = use the direct 8-modulo path)
};

__device__ __forceinline__ float div_by_const(float n, float vs, float rvs) {
    const float q0 = n * rvs;
    const float e = __builtin_fmaf(-q0, vs, n);
    return __builtin_fmaf(e, rvs, q0);
}

__device__ __forceinline__ uint32_t mod_size(uint64_t h, const LevelParams &lp) {
    double q = floor((double)h * lp.inv_size);
    int64_t r = (int64_t)h - (int64_t)q * (int64_t)lp.size;
    if (r < 0) r += lp.size;
    else if (r >= (int64_t)lp.size) r -= lp.size;
    return (uint32_t)r;
}

// rows of the 8 corners of cell c (corner q: x = (q >> 1) & 1, y = q & 1, z = q >> 2).  Power-of-two levels: xor + mask.  Other levels
// with kb > 0 ((res + 1) < 2^kb <= size): the hash of a corner is B ^ cx with B = hy ^ hz shared by the two x neighbours, and
// B ^ cx = (B & ~low) + ((B & low) ^ cx) for low = 2^kb - 1 > cx - ONE 64-bit modulo per (y, z) pair and a conditional subtract per
// corner instead of eight modulos (each a f64 multiply + floor + fix-up, ~28 instructions).  kb = 0: the direct form.
__device__ __forceinline__ void corner_rows(const uint32_t c[3], const LevelParams &lp, int kb, uint32_t rows[8]) {
    if (lp.mask) {
        const uint32_t hy0 = c[1] * 2654435761u, hy1 = hy0 + 2654435761u;
        const uint32_t hz0 = c[2] * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t A[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};   // index y + 2 z
#pragma unroll
        for (int q = 0; q < 8; ++q) rows[q] = ((c[0] + ((q >> 1) & 1)) ^ A[(q & 1) + 2 * (q >> 2)]) & lp.mask;
    } else if (kb > 0) {
        const uint64_t hy0 = (uint64_t)c[1] * 2654435761ull, hy1 = hy0 + 2654435761ull;
        const uint64_t hz0 = (uint64_t)c[2] * 805459861ull, hz1 = hz0 + 805459861ull;
        const uint64_t B[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
        const uint64_t lowmask = (1ull << kb) - 1ull;
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            const uint32_t mh = mod_size(B[yz] & ~lowmask, lp);
            const uint32_t bl = (uint32_t)(B[yz] & lowmask);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                uint32_t r = mh + (bl ^ (c[0] + x));
                if (r >= lp.size) r -= lp.size;
                rows[(x << 1) + (yz & 1) + ((yz >> 1) << 2)] = r;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) rows[q] = hash_row(c[0] + ((q >> 1) & 1), c[1] + (q & 1), c[2] + (q >> 2), lp);
    }
}

// modulo sharing needs (res + 1) < 2^kb <= size (0: not applicable)
static int level_lowbits(const LevelParams &lp) {
    if (lp.mask) return 0;
    int kb = 0;
    while ((1u << kb) <= (uint32_t)(lp.res + 1)) ++kb;
    return (1ull << kb) > (uint64_t)lp.size ? 0 : kb;
}

// CORNERS: the gathered rows are kept for the second-order gathers (arcn_hashgrid_dxyz_corners / _ddout_corners), level-major in 16-byte
// quads: quad j of (sample s, level l) = rows 2j, 2j + 1 (F = 2) | rows 4j .. 4j + 3 (F = 1) at corners[((l * NJ + j) * n_cap + s) * 4],
// NJ = 2 F - store j of a wave is 64 consecutive quads, one contiguous KiB, like the level-major features; zeros outside the grid.
template <int F, bool LM, bool PAIR, bool CORNERS = false>
__global__ void __launch_bounds__(256)
hashgrid_fwd_bal_kernel(const float *__restrict__ xyz, const float *__restrict__ table, GridParams g, FwdPlan plan,
                        float *__restrict__ out, int64_t n_cap, int64_t n, const int32_t *n_ptr, float *__restrict__ corners = nullptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    int k = -1;
#pragma unroll
    for (int i = 0; i < kFwdSegs; ++i)
        if (i < plan.n_seg[xcd] && j >= plan.seg[xcd][i].wg0 && j < plan.seg[xcd][i].wg1) k = i;
    if (k < 0) return;
    const FwdSeg sg = plan.seg[xcd][k];
    const int l = sg.level;
    const LevelParams lp = g.lv[l];
    const int64_t tiles = (cnt + 255) >> 8;
    const int64_t t0 = (tiles * sg.f0) >> 16, t1 = (tiles * sg.f1) >> 16;
    const int nslot = sg.wg1 - sg.wg0, slot = j - sg.wg0;
    const int kb = plan.lowbits[l];
    const float fres = (float)lp.res;
    for (int64_t t = t0 + slot; t < t1; t += nslot) {
        const int64_t s = (t << 8) + threadIdx.x;
        if (s >= cnt) break;
        const float p[3] = {xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2]};
        float nn[3], v[3];
        bool amb = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            nn[a] = p[a] - g.mn[a];
            const float q0 = nn[a] * lp.rvs[a];
            v[a] = q0;
            const float d = fabsf(q0 - rintf(q0));
            amb = amb || !(d > 4e-7f * fmaxf(fabsf(q0), 1.0f));   // also true for NaN
        }
        if (__any(amb)) {
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = nn[a] / lp.vs[a];
        }
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (!(v[a] >= 0.f) || !(v[a] < fres)) ok = false;
        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.f;
        float vals[8][F];
        if (CORNERS) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int f = 0; f < F; ++f) vals[q][f] = 0.f;
        }
        if (ok) {
            uint32_t c[3];
            float w[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float cf = floorf(v[a]);
                c[a] = (uint32_t)cf;
                const float st = cf * lp.vs[a];
                const float g0 = st + g.mn[0];
                const float ww = div_by_const(p[a] - g0, lp.vs[a], lp.rvs[a]);
                w[a] = ww < 0.0f ? 0.0f : (ww > 1.0f ? 1.0f : ww);
            }
            uint32_t rows[8];   // corner q: x = (q>>1)&1, y = q&1, z = q>>2
            corner_rows(c, lp, kb, rows);
            const float *lt = table + lp.offset * F;
            // (a level whose first row is not 16-byte aligned - an odd row offset after dense levels of odd size, or a table segment at
            // an odd 8-byte offset of a flat parameter buffer - takes the 8-byte loads: the dwordx4 below must not straddle)
            if (PAIR && F == 2 && lp.mask && (reinterpret_cast<uintptr_t>(lt) & 15) == 0) {
                // the two x-neighbours of a (y, z) corner pair: rows r and r' = ((cx+1) ^ A) & mask.  cx even -> r' = r ^ 1: both in
                // one aligned 16-byte word (one dwordx4); cx odd -> a second 8-byte load under the lane mask
                // ALL the loads are issued before the first is consumed - the four 16-byte words, then (one branch) the four 8-byte
                // words of the odd lanes: written pair by pair the compiler waited for each word before it issued the next (the
                // selects below sat between the loads: `s_waitcnt vmcnt(0)` four times per sample and level, one load in flight per
                // wave in a kernel whose bound is the number of requests in flight)
                const bool even = (c[0] & 1u) == 0u;
                float4 t4[4];
                float2 o2[4];
#pragma unroll
                for (int yz = 0; yz < 4; ++yz) t4[yz] = *reinterpret_cast<const float4 *>(lt + (size_t)(rows[(yz & 1) + ((yz >> 1) << 2)] & ~1u) * 2);
                if (!even) {
#pragma unroll
                    for (int yz = 0; yz < 4; ++yz) o2[yz] = *reinterpret_cast<const float2 *>(lt + (size_t)rows[(yz & 1) + ((yz >> 1) << 2) + 2] * 2);
                }
#pragma unroll
                for (int yz = 0; yz < 4; ++yz) {
                    const int q0 = (yz & 1) + ((yz >> 1) << 2), q1 = q0 + 2;
                    const bool hi = (rows[q0] & 1u) != 0u;
                    vals[q0][0] = hi ? t4[yz].z : t4[yz].x;
                    vals[q0][1] = hi ? t4[yz].w : t4[yz].y;
                    vals[q1][0] = even ? (hi ? t4[yz].x : t4[yz].z) : o2[yz].x;
                    vals[q1][1] = even ? (hi ? t4[yz].y : t4[yz].w) : o2[yz].y;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float *src = lt + (size_t)rows[q] * F;
                    if (F == 2) {
                        const float2 t2 = *reinterpret_cast<const float2 *>(src);
                        vals[q][0] = t2.x;
                        vals[q][1 % F] = t2.y;
                    } else {
#pragma unroll
                        for (int f = 0; f < F; ++f) vals[q][f] = src[f];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t ox = (q >> 1) & 1, oy = q & 1, oz = q >> 2;
                const float wt = ((ox ? w[0] : 1.0f - w[0]) * (oy ? w[1] : 1.0f - w[1])) * (oz ? w[2] : 1.0f - w[2]);
#pragma unroll
                for (int f = 0; f < F; ++f) { float a = vals[q][f] * wt; acc[f] = acc[f] + a; }
            }
        }
        float *o = LM ? out + ((int64_t)l * n_cap + s) * F : out + (s * g.L + l) * F;
        if (F == 2) {
            *reinterpret_cast<float2 *>(o) = make_float2(acc[0], acc[1 % F]);
        } else {
#pragma unroll
            for (int f = 0; f < F; ++f) o[f] = acc[f];
        }
        if (CORNERS) {
            constexpr int NJ = 2 * F;
            const float *vf = &vals[0][0];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                *reinterpret_cast<float4 *>(corners + (((int64_t)l * NJ + j) * n_cap + s) * 4) = make_float4(vf[4 * j], vf[4 * j + 1], vf[4 * j + 2], vf[4 * j + 3]);
        }
    }
}

// ---- the second-order gathers of NeuS on the hash grid from the forward's corners ----------------------------------------------------------
// One lane per SAMPLE, the levels in a loop: the corner quads of a level are 64 consecutive 16-byte words per wave (streaming, coalesced),
// the per-sample operands (dout / ddout rows of L F floats) are one cache line a lane walks through.  The arithmetic per (sample, level) is
// that of hashgrid_bwd_kernel (dxyz) / hashgrid_bwd_bwd_kernel (ddout) above on the same values in the same order, and the sum over the
// levels of dxyz follows add_over_levels' butterfly (L a power of two): the results are the table forms' bit for bit.
template <int F>
__device__ __forceinline__ void load_corner_quads(const float *__restrict__ corners, int l, int64_t n_cap, int64_t s, float cv[8 * F]) {
    constexpr int NJ = 2 * F;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4 t = *reinterpret_cast<const float4 *>(corners + (((int64_t)l * NJ + j) * n_cap + s) * 4);
        cv[4 * j] = t.x; cv[4 * j + 1] = t.y; cv[4 * j + 2] = t.z; cv[4 * j + 3] = t.w;
    }
}

template <int F>
__global__ void __launch_bounds__(256)
hashgrid_dxyz_corners_kernel(const float *__restrict__ xyz, const float *__restrict__ corners, const float *__restrict__ dout, GridParams g,
                             float *__restrict__ dxyz, int64_t n_cap, int64_t n, const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cnt) return;
    const float p[3] = {xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2]};
    const bool tree = (g.L & (g.L - 1)) == 0;
    // add_over_levels' butterfly as a binary counter: st[d] = the finished sum of an aligned group of 2^d levels waiting for its right-hand
    // neighbour; level l merges upwards through its trailing one bits - (v0 + v1) + (v2 + v3) ... in the butterfly's own order
    float st[6][3];
    float seq[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < g.L; ++l) {
        const LevelParams lp = g.lv[l];
        const Cell cell = locate(p, g, lp);
        float gx[3] = {0.f, 0.f, 0.f};
        if (cell.valid) {
            float go[F];
            if (F == 2) {
                const float2 t2 = *reinterpret_cast<const float2 *>(dout + (s * g.L + l) * F);
                go[0] = t2.x; go[1 % F] = t2.y;
            } else {
#pragma unroll
                for (int f = 0; f < F; ++f) go[f] = dout[(s * g.L + l) * F + f];
            }
            float cv[8 * F];
            load_corner_quads<F>(corners, l, n_cap, s, cv);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t ox = (q >> 1) & 1, oy = q & 1, oz = q >> 2;
                float wx = ox ? cell.w[0] : 1.0f - cell.w[0];
                float wy = oy ? cell.w[1] : 1.0f - cell.w[1];
                float wz = oz ? cell.w[2] : 1.0f - cell.w[2];
                float dot = 0.f;
#pragma unroll
                for (int f = 0; f < F; ++f) dot += go[f] * cv[q * F + f];
                float sx = ox ? 1.0f : -1.0f, sy = oy ? 1.0f : -1.0f, sz = oz ? 1.0f : -1.0f;
                gx[0] += dot * sx * wy * wz * cell.dw[0];
                gx[1] += dot * wx * sy * wz * cell.dw[1];
                gx[2] += dot * wx * wy * sz * cell.dw[2];
            }
        }
        if (tree) {
            bool placed = false;
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                if (placed) continue;
                if ((l >> d) & 1) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) gx[k] = st[d][k] + gx[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) st[d][k] = gx[k];
                    placed = true;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) seq[k] += gx[k];
        }
    }
    if (tree) {
        int top = 0;
        while ((1 << top) < g.L) ++top;
#pragma unroll
        for (int d = 0; d < 6; ++d)
            if (d == top) {
#pragma unroll
                for (int k = 0; k < 3; ++k) seq[k] = st[d][k];
            }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) dxyz[3 * s + k] = 0.f + seq[k];      // (the table form adds into a cleared buffer: -0 becomes +0 there too)
}

template <int F>
__global__ void __launch_bounds__(256)
hashgrid_ddout_corners_kernel(const float *__restrict__ xyz, const float *__restrict__ gdx, const float *__restrict__ corners, GridParams g,
                              float *__restrict__ ddout, int64_t n_cap, int64_t n, const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cnt) return;
    const float p[3] = {xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2]};
    const float gd[3] = {gdx[3 * s], gdx[3 * s + 1], gdx[3 * s + 2]};
#pragma unroll 1
    for (int l = 0; l < g.L; ++l) {
        const LevelParams lp = g.lv[l];
        const Cell cell = locate(p, g, lp);
        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.f;
        if (cell.valid) {
            float cv[8 * F];
            load_corner_quads<F>(corners, l, n_cap, s, cv);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t o[3] = {(uint32_t)((q >> 1) & 1), (uint32_t)(q & 1), (uint32_t)(q >> 2)};
                float a[3], sd[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    a[k] = o[k] ? cell.w[k] : 1.0f - cell.w[k];
                    sd[k] = o[k] ? cell.dw[k] : -cell.dw[k];
                }
                float D = gd[0] * sd[0] * a[1] * a[2];
                D = D + gd[1] * a[0] * sd[1] * a[2];
                D = D + gd[2] * a[0] * a[1] * sd[2];
#pragma unroll
                for (int f = 0; f < F; ++f) acc[f] = acc[f] + cv[q * F + f] * D;
            }
        }
#pragma unroll
        for (int f = 0; f < F; ++f) ddout[(s * g.L + l) * F + f] = acc[f];
    }
}

// level-major features (L, n_cap, F) -> rows (n, ld) through an LDS tile: what the dense products (csrc/gemm.hip) read.  A workgroup moves 256
// samples: per level 256 F consecutive floats in, L F consecutive floats per sample out - both sides whole cache lines.
template <int F>
__global__ void __launch_bounds__(256)
lm_to_rows_kernel(const float *__restrict__ lm, int L, int64_t n_cap, float *__restrict__ rows, int64_t ld, int64_t n, const int32_t *n_ptr) {
    extern __shared__ float tile[];      // [256][L F + 1]
    const int64_t cnt = dev_count(n, n_ptr);
    const int64_t s0 = (int64_t)blockIdx.x * 256;
    if (s0 >= cnt) return;
    const int W = L * F, pitch = W + 1;
    const int m = (int)((cnt - s0) < 256 ? (cnt - s0) : 256);
    for (int i = threadIdx.x; i < L * 256 * F; i += 256) {
        const int l = i / (256 * F), r = i - l * 256 * F, sl = r / F, f = r - sl * F;
        if (sl < m) tile[sl * pitch + l * F + f] = lm[((int64_t)l * n_cap + s0 + sl) * F + f];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < m * W; i += 256) {
        const int sl = i / W, c = i - sl * W;
        rows[(s0 + sl) * ld + c] = tile[sl * pitch + c];
    }
}

// ---- backward, binned owner-computes scatter --------------------------------------------------------------------------------
// Measured on MI355X: the chip retires only ~18-21 G scattered fp32 global atomics per second (they are served at the memory
// side, not in the issuing XCD's L2), so the 128 atomics/sample of the plain kernel above cost 4 ms per 2^18 samples no
// matter how the work is arranged.  The scatter below issues NO global float atomics on the large levels: every table chunk
// of <= 16384 rows (128 KiB of the CU's 160 KiB LDS) is OWNED by one workgroup, which accumulates it in LDS and writes it back
// with plain coalesced stores.
//   scatter_bin_kernel    one lane per (level, sample): cell, weights, gradient.  On the hashed power-of-two levels the 8
//                         corners form 4 x-neighbour pairs whose two rows differ only in the low bits of the hash, i.e. live
//                         in the same chunk: each pair becomes ONE 16-byte record {i0 | i1 << 16, wx, g0*wy*wz, g1*wy*wz}
//                         appended to the bin of its owner chunk (LDS histogram for the rank, one global integer atomic per
//                         bin and workgroup).  On the small levels below them consecutive samples of a ray sit in one cell for
//                         long runs: each run is summed inside the wave first (segmented suffix sum) and only its head lane
//                         emits 8 single-row records.
//   scatter_accum_kernel  one workgroup per chunk streams ITS bin only (coalesced 16-byte loads, all 64 lanes busy) and
//                         accumulates in LDS.  ds_add_f32 retires ~3 cycles per active lane on gfx950 (integer LDS atomics
//                         are ~28x faster), so the large levels take a 1-bit row lock + plain 8-byte read-modify-write instead
//                         (ds_or / ds_read_b64 / ds_write_b64 / ds_and: 5.8x the rate of two ds_add_f32 when all lanes are
//                         active, tools/ubench/lds_lock.hip).  The small levels keep float atomics and are split over several
//                         workgroups per chunk, merged with global atomics on the touched rows (few rows, negligible).
// Bins have a fixed capacity (2x the mean; the hash spreads pairs evenly); a record that does not fit is applied directly
// to dtable with global atomics, so correctness never depends on the statistics.
// History (same workload, 272 K samples): plain global atomics 4.26 ms; every owner scanning all records of its level
// (3.3 GB of L2 reads, ~16 of 64 lanes in front of the float atomics) 0.79 ms; this scheme 0.33 ms.
constexpr int kChunkFloats = 32768;  // 128 KiB of accumulators per workgroup: 16384 rows at F = 2
constexpr int kTiledThreads = 1024;
constexpr int kMaxChunks = 1024;  // owner chunks per level (2^24-row levels with 16384-row chunks)

struct BinPlan {
    int32_t item_first[ARCN_MAX_LEVELS + 1];  // by dispatch position
    int32_t level_of[ARCN_MAX_LEVELS];        // dispatch position -> level
    int32_t n_chunks[ARCN_MAX_LEVELS];
    int32_t n_splits[ARCN_MAX_LEVELS];
    int32_t chunk_shift[ARCN_MAX_LEVELS];     // rows per chunk = 1 << shift
    int32_t bin_first[ARCN_MAX_LEVELS];       // index of the level's first counter
    int32_t cap[ARCN_MAX_LEVELS];             // records per bin
    int64_t rec_first[ARCN_MAX_LEVELS];       // first record of the level's bin 0
    uint32_t lock_levels;                     // consumer, bit l: exclusive row updates - 64-bit compare-and-swap (F = 2) or a row bit lock (F = 1) - else float atomics
    uint32_t active_levels;                   // levels this launch processes (arcn_hashgrid_bwd_lm_levels: a level group)
    uint32_t pair_levels;                     // hashed power-of-two levels with more than one chunk: <= 4 pair records per sample
    uint32_t nosplit_levels;                  // chunk rows > res on a hashed level: both rows of an x pair ALWAYS share the chunk
    int32_t n_bins;
    int32_t chunk_floats;                     // LDS accumulator floats per workgroup (rows per chunk * F)
    int32_t det;                              // ARCN_DETERMINISTIC=1: order-independent fixed-point accumulation (see scatter_accum_kernel)
    int32_t aux_first;                        // counters[aux_first] = bits of max |dout| (det), counters[aux_first + 1] = a bin overflowed
    int8_t lowbits[ARCN_MAX_LEVELS];          // corner_rows' kb of the level (shared 64-bit modulo on the non-power-of-two levels)
    int64_t n_recs;
};

template <int F>
__device__ __forceinline__ void overflow_add(float *__restrict__ dtable, const LevelParams &lp, uint32_t row, float a0, float a1) {
    float *d = dtable + ((int64_t)lp.offset + row) * F;
    unsafeAtomicAdd(d, a0);
    if (F > 1) unsafeAtomicAdd(d + (F > 1 ? 1 : 0), a1);
}


// place one record at position pos of its bin; beyond the bin's capacity it is applied to dtable directly
template <int F>
__device__ __forceinline__ void emit_record(uint4 *__restrict__ lrecs, float *__restrict__ dtable, const LevelParams &lp, int bin,
                                            uint32_t pos, uint32_t cap, int shift, const uint4 &rec, uint32_t *ovf_flag = nullptr) {
    if (pos < cap) {
        lrecs[(int64_t)bin * cap + pos] = rec;
    } else {
        if (ovf_flag) *ovf_flag = 1u;   // deterministic mode: these float atomics are not order-independent - say so (arcn_hashgrid_bwd_status)
        const uint32_t base_row = (uint32_t)bin << shift;
        const uint32_t i0 = rec.x & 0xffffu, i1 = rec.x >> 16;
        const float wx = __uint_as_float(rec.y), a0 = __uint_as_float(rec.z), a1 = __uint_as_float(rec.w);
        const float wl = 1.0f - wx;
        overflow_add<F>(dtable, lp, base_row + i0, a0 * wl, a1 * wl);
        if (i1 != 0xffffu) overflow_add<F>(dtable, lp, base_row + i1, a0 * wx, a1 * wx);
    }
}

// Latency, not bandwidth, bounds this kernel: a workgroup lives ~8 us (loads -> ranks in LDS -> barrier -> one global
// integer atomic per bin -> barrier -> stores) and its ~90 VGPRs allow one 1024-thread workgroup per CU.  Staging the
// records in LDS to write them out bin-sorted (coalesced) was tried and is slower: the extra barrier and scan lengthen
// exactly that chain (212 us against 140 us).
// Cell + weights for the scatter producer: the cell index from the reciprocal quotient, re-done with the true division in any wave
// where a lane's quotient lies within 4e-7 (relative) of an integer (see hashgrid_fwd_bal_kernel: bit-identical cells), the weights
// through the reciprocal (1 ulp, inside the backward's tolerance, as before).
__device__ __forceinline__ Cell locate_fastcell(const float p[3], const GridParams &g, const LevelParams &lp, bool in_range) {
    Cell cell;
    float nn[3], v[3];
    bool amb = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        nn[a] = p[a] - g.mn[a];
        const float q0 = nn[a] * lp.rvs[a];
        v[a] = q0;
        const float d = fabsf(q0 - rintf(q0));
        amb = amb || !(d > 4e-7f * fmaxf(fabsf(q0), 1.0f));
    }
    if (__any(amb && in_range)) {
#pragma unroll
        for (int a = 0; a < 3; ++a) v[a] = nn[a] / lp.vs[a];
    }
    bool ok = in_range;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (!(v[a] >= 0.f) || !(v[a] < (float)lp.res)) ok = false;
    cell.valid = ok;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float cf = floorf(v[a]);
        cell.c[a] = ok ? (uint32_t)cf : 0u;
        const float st = cf * lp.vs[a];
        const float g0 = st + g.mn[0];
        const float ww = (p[a] - g0) * lp.rvs[a];
        cell.w[a] = ww < 0.0f ? 0.0f : (ww > 1.0f ? 1.0f : ww);
        cell.dw[a] = 0.f;
    }
    return cell;
}

// Runs of consecutive samples (lanes of a wave) in the same cell, summed: va / vb (the 8 corners' contributions of every lane, F = 1: va only)
// become, in the FIRST lane of every run (`true_head`), the sums over the run.  `head` = true_head or the start of a 16-lane row: the doubling
// steps work inside rows (DPP row shifts on the VALU instead of ds_bpermute, which queued on the LDS pipeline next to the histogram atomics),
// the pieces of a run that crosses rows are joined afterwards through v_readlane.  Wave uniform control flow; every lane calls it.
template <int F>
__device__ __forceinline__ void sum_cell_runs(float (&va)[8], float (&vb)[8], bool head, bool true_head, int lane) {
    // segmented suffix sum by doubling over the 8 corners x F values
    const uint64_t all_heads = __ballot(head);
    const uint64_t above = lane == 63 ? 0ull : (all_heads & ~((2ull << lane) - 1ull));
    const int tail = above ? (int)__builtin_ctzll(above) - 1 : 63;  // <= the last lane of this lane's row
    // one doubling step with the row shift as a template constant; false once no lane has anything left to take
#define ARCN_RUN_STEP(D)                                                              \
    {                                                                                 \
        const bool take = lane + (D) <= tail;                                         \
        more = __ballot(take) != 0ull;                                                \
        if (more) {                                                                   \
            /* va += (lane i + D's va) * m, m = 1 inside the run, 0 past its end: ONE v_fmac_f32_dpp per value (m = 1 is the plain */ \
            /* sum bit for bit; a select + add were three instructions and a DPP hazard stall) */                                     \
            const float m = take ? 1.0f : 0.0f;                                       \
            _Pragma("unroll") for (int q = 0; q < 8; ++q) {                           \
                va[q] = __builtin_fmaf(dpp_zero<0x100 + (D)>(va[q]), m, va[q]);       \
                if (F > 1) vb[q] = __builtin_fmaf(dpp_zero<0x100 + (D)>(vb[q]), m, vb[q]); \
            }                                                                         \
        }                                                                             \
    }
    bool more = true;
    ARCN_RUN_STEP(1)
    if (more) ARCN_RUN_STEP(2)
    if (more) ARCN_RUN_STEP(4)
    if (more) ARCN_RUN_STEP(8)
#undef ARCN_RUN_STEP
    // join the pieces of runs that cross a row start, top row first so that a run spanning several rows chains up: lane b holds
    // the sum of its piece, the lanes of the last piece of the row below take it
    const uint64_t true_heads = __ballot(true_head);
#pragma unroll
    for (int b = 48; b >= 16; b -= 16) {
        if ((true_heads >> b) & 1ull) continue;
        const uint64_t below = all_heads & ((1ull << b) - 1ull);  // never empty: lane b-16 is a head
        const int piece = 63 - (int)__builtin_clzll(below);
        const float mj = (lane >= piece && lane < b) ? 1.0f : 0.0f;     // (the same masked multiply-add as in the steps above)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float ua = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, va[q]), b));
            va[q] = __builtin_fmaf(ua, mj, va[q]);
            if (F > 1) {
                const float ub = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vb[q]), b));
                vb[q] = __builtin_fmaf(ub, mj, vb[q]);
            }
        }
    }
}

// PLAN = true (arcn_hashgrid_bwd_plan): the half of this kernel that depends on the sample POSITIONS only - cells, run structure, rows, bins,
// ranks - run ahead of the step on the sampling stream (the samples of step i are marched two steps earlier).  It writes the index half
// {i0 | i1 << 16, wx} of every record to its final slot (ridx) and, per (level, sample), the slots its value halves belong to
// (pos4: two uint4 per sample, global record indices; kPosNone = no record, kPosOvf = the bin was full: scatter_fill_kernel applies that one to
// dtable directly).  What is left between the geometry net's backward and the chunk owners is scatter_fill_kernel: a streaming pass
// without barriers, histograms, atomics or the 64-bit modulo of the dense levels.
constexpr uint32_t kPosNone = 0xffffffffu, kPosOvf = 0xfffffffeu;
constexpr uint32_t kLposNone = 0xffffu, kLposOvf = 0xfffeu;     // 16-bit tile-local positions (< kTileSlots)
constexpr int kTileSlots = 8192;                                // records of one 1024-sample tile of one level: at most 8 per sample

template <int F, int kBinThreads, bool PLAN = false>
__global__ void __launch_bounds__(kBinThreads)
scatter_bin_kernel(const float *__restrict__ xyz, const float *__restrict__ dout, int64_t dout_lm_stride, GridParams g,
                   BinPlan plan, uint32_t *__restrict__ counters, uint4 *__restrict__ recs, float *__restrict__ dtable, int64_t n,
                   const int32_t *n_ptr, uint2 *__restrict__ ridx = nullptr, uint4 *__restrict__ pos4 = nullptr, int64_t pos_stride = 0,
                   uint32_t *__restrict__ gslot = nullptr, uint32_t *__restrict__ tile_count = nullptr, int64_t n_tiles = 0) {
    // hist: records of this tile per bin; gbase: first position of the tile's run in every bin.  Both double-buffered over the
    // tiles: the other buffer's histogram is cleared while this tile reserves its runs, so a tile costs two barriers, not four.
    __shared__ uint32_t hist2[2][kMaxChunks];
    __shared__ uint32_t gbase2[2][kMaxChunks];
    __shared__ uint32_t lbase2[PLAN ? 2 : 1][PLAN ? kMaxChunks : 1];      // PLAN: first tile-local position of every bin (exclusive scan of the tile's histogram)
    __shared__ uint32_t wsum[16];
    const int64_t cnt = dev_count(n, n_ptr);
    const int l = blockIdx.y;
    if (!((plan.active_levels >> l) & 1u)) return;
    const LevelParams lp = g.lv[l];
    const int nc = plan.n_chunks[l], shift = plan.chunk_shift[l];
    const uint32_t cmask = (1u << shift) - 1u;
    const int t = threadIdx.x, lane = t & 63;
    for (int i = threadIdx.x; i < nc; i += kBinThreads) hist2[0][i] = 0u;
    __syncthreads();
    int buf = 0;
    // Persistent over the level's tiles: a tile keeps a wave busy for ~3 us only, and one workgroup per tile left the chip at
    // ~20 % wave occupancy waiting for the dispatcher (SQ_WAVE_CYCLES / duration); gridDim.x workgroups per level loop instead.
    // inputs of the NEXT tile are requested before this tile waits for its bin reservations (software pipeline: the loads travel
    // while the reservation atomics do)
    float np_[3] = {0.f, 0.f, 0.f}, ng0 = 0.f, ng1 = 0.f;
    auto fetch = [&](int64_t s) {
        if (s < cnt) {
            np_[0] = xyz[3 * s]; np_[1] = xyz[3 * s + 1]; np_[2] = xyz[3 * s + 2];
            // level-major gradients (dout_lm_stride > 0): the lanes of a wave read consecutive 8-byte words
            if (!PLAN) {
            const float *gp = dout_lm_stride ? dout + ((int64_t)l * dout_lm_stride + s) * F : dout + (s * g.L + l) * F;
            // (non-temporal: this is the gradient's only reader - step -1.9 % in three alternations, profiles/r4_ab_nontemporal.txt)
            ng0 = __builtin_nontemporal_load(gp);
            ng1 = F > 1 ? __builtin_nontemporal_load(gp + (F > 1 ? 1 : 0)) : 0.f;
            }
        }
    };
    fetch((int64_t)blockIdx.x * kBinThreads + t);
    for (int64_t tile0 = (int64_t)blockIdx.x * kBinThreads; tile0 < cnt; tile0 += (int64_t)gridDim.x * kBinThreads, buf ^= 1) {
    uint32_t *hist = hist2[buf], *gbase = gbase2[buf];
    const int64_t s = tile0 + t;
    int sbin[8];
    uint32_t sidx[8];
    float swx[8], sa[8], sb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sbin[k] = -1;
    const float p[3] = {np_[0], np_[1], np_[2]};
    const Cell cell = locate_fastcell(p, g, lp, s < cnt);
    const float g0 = cell.valid ? ng0 : 0.f, g1 = cell.valid ? ng1 : 0.f;
    if (!PLAN && plan.det) {
        // the fixed-point scale of the consumer comes from the largest |gradient| of the launch: a max is order-independent
        float m = fmaxf(fabsf(g0), fabsf(g1));
        if (!(m == m)) m = 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if (lane == 0 && m > 0.f) atomicMax(&counters[plan.aux_first], __float_as_uint(m));
    }
    // runs of consecutive samples (lanes) in the same cell: head lanes, and for every lane the last lane of its run
    const uint32_t kxy = cell.valid ? (cell.c[0] | (cell.c[1] << 16)) : 0xffffffffu;
    const uint32_t kz = cell.valid ? cell.c[2] : (uint32_t)lane;
    // The run sums below work inside 16-lane rows first (DPP row shifts on the VALU instead of ds_bpermute, which queued on the LDS
    // pipeline next to the histogram atomics): `head` also cuts a run at every row start; the pieces of a run that crosses rows are
    // joined afterwards through v_readlane, and only the true heads emit records.
    const uint32_t pxy = (uint32_t)dpp_take_i<0x138>(-1, (int)kxy), pz = (uint32_t)dpp_take_i<0x138>(-1, (int)kz);
    const bool true_head = lane == 0 || kxy != pxy || kz != pz;
    const bool head = true_head || (lane & 15) == 0;
    const uint64_t heads = __ballot(true_head && cell.valid), valid = __ballot(cell.valid);
    // Per wave: with an average run of >= 1.5 samples the runs are summed first (fewer records, fewer row updates, and no
    // neighbouring lanes fighting over one row lock in the consumer); otherwise the x pairs go out as they are.
    const bool reduce = 3 * __popcll(heads) <= 2 * __popcll(valid);
    const bool nosplit = (plan.nosplit_levels >> l) & 1u;  // wave uniform
    const int n_slots = (!reduce && nosplit) ? 4 : 8;       // slots that can be occupied
    if (!reduce && nosplit) {
        // hashed level, chunk larger than the resolution: cx and cx+1 differ only below the chunk size, one record per pair
        if (cell.valid) {
            const uint32_t hy0 = cell.c[1] * 2654435761u, hy1 = hy0 + 2654435761u;
            const uint32_t hz0 = cell.c[2] * 805459861u, hz1 = hz0 + 805459861u;
            const float wy1 = cell.w[1], wy0 = 1.0f - wy1, wz1 = cell.w[2], wz0 = 1.0f - wz1;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const uint32_t hyz = ((pr & 1) ? hy1 : hy0) ^ ((pr >> 1) ? hz1 : hz0);
                const uint32_t r0 = (cell.c[0] ^ hyz) & lp.mask, r1 = ((cell.c[0] + 1u) ^ hyz) & lp.mask;
                const float wyz = ((pr & 1) ? wy1 : wy0) * ((pr >> 1) ? wz1 : wz0);
                sbin[pr] = (int)(r0 >> shift);
                sidx[pr] = (r0 & cmask) | ((r1 & cmask) << 16);
                swx[pr] = cell.w[0];
                sa[pr] = g0 * wyz;
                sb[pr] = g1 * wyz;
            }
        }
    } else if (!reduce) {
        if (cell.valid) {
            uint32_t rows[8];
            corner_rows(cell.c, lp, plan.lowbits[l], rows);
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const uint32_t oy = pr & 1, oz = pr >> 1;
                const uint32_t r0 = rows[oy + (oz << 2)], r1 = rows[2 + oy + (oz << 2)];
                const float wyz = (oy ? cell.w[1] : 1.0f - cell.w[1]) * (oz ? cell.w[2] : 1.0f - cell.w[2]);
                const float a0 = g0 * wyz, a1 = g1 * wyz;
                const int c0 = (int)(r0 >> shift), c1 = (int)(r1 >> shift);
                if (c0 == c1) {
                    sbin[2 * pr] = c0;
                    sidx[2 * pr] = (r0 & cmask) | ((r1 & cmask) << 16);
                    swx[2 * pr] = cell.w[0];
                    sa[2 * pr] = a0;
                    sb[2 * pr] = a1;
                } else {
                    const float wl = 1.0f - cell.w[0];
                    sbin[2 * pr] = c0;
                    sidx[2 * pr] = (r0 & cmask) | 0xffff0000u;
                    swx[2 * pr] = 0.f;
                    sa[2 * pr] = a0 * wl;
                    sb[2 * pr] = a1 * wl;
                    sbin[2 * pr + 1] = c1;
                    sidx[2 * pr + 1] = (r1 & cmask) | 0xffff0000u;
                    swx[2 * pr + 1] = 0.f;
                    sa[2 * pr + 1] = a0 * cell.w[0];
                    sb[2 * pr + 1] = a1 * cell.w[0];
                }
            }
        }
    } else {
        // the runs' sums (sum_cell_runs); only the head lane of a run emits its 8 singles
        float va[8], vb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t ox = (q >> 1) & 1, oy = q & 1, oz = q >> 2;
            float wt = 0.f;
            if (cell.valid) wt = ((ox ? cell.w[0] : 1.0f - cell.w[0]) * (oy ? cell.w[1] : 1.0f - cell.w[1])) * (oz ? cell.w[2] : 1.0f - cell.w[2]);
            va[q] = g0 * wt;
            vb[q] = g1 * wt;
        }
        if (!PLAN) sum_cell_runs<F>(va, vb, head, true_head, lane);
        if (true_head && cell.valid) {
            uint32_t rows[8];
            corner_rows(cell.c, lp, plan.lowbits[l], rows);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t r = rows[q];
                sbin[q] = (int)(r >> shift);
                sidx[q] = (r & cmask) | 0xffff0000u;
                swx[q] = 0.f;
                sa[q] = va[q];
                sb[q] = vb[q];
            }
        }
    }
    uint32_t rank[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rank[k] = (k < n_slots && sbin[k] >= 0) ? atomicAdd(&hist[sbin[k]], 1u) : 0u;
    __syncthreads();
    // reserve the tile's run in every bin: one global integer atomic per (bin, tile); clear the NEXT tile's histogram
    fetch(tile0 + (int64_t)gridDim.x * kBinThreads + t);
    uint32_t my_hist = 0u;
    for (int i = threadIdx.x; i < nc; i += kBinThreads) {
        const uint32_t h = hist[i];
        my_hist = h;
        gbase[i] = h ? atomicAdd(&counters[plan.bin_first[l] + i], h) : 0u;
        hist2[buf ^ 1][i] = 0u;
    }
    if (PLAN) {
        // tile-local positions: the tile's records sorted by bin (nc <= kBinThreads: thread i holds bin i).  The fill pass stages its values in
        // LDS in this order and copies them out as whole lines - a (bin, tile) run is contiguous in the tile AND in its bin.
        uint32_t inc = my_hist;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (nc > 64) {      // (workgroup uniform)
            if (lane == 63) wsum[t >> 6] = inc;
            __syncthreads();
            uint32_t off = 0u;
            for (int w = 0; w < (t >> 6); ++w) off += wsum[w];
            inc += off;
        }
        if (t < nc) lbase2[buf][t] = inc - my_hist;
        if (t == nc - 1) tile_count[(int64_t)l * n_tiles + tile0 / kBinThreads] = inc;
    }
    __syncthreads();
    const uint32_t cap = (uint32_t)plan.cap[l];
    if (PLAN) {
        uint32_t P[8];
        uint32_t *gs = gslot + ((int64_t)l * n_tiles + tile0 / kBinThreads) * kTileSlots;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            P[k] = kLposNone;
            if (k >= n_slots || sbin[k] < 0) continue;
            const uint32_t pos = gbase[sbin[k]] + rank[k];
            const uint32_t lpos = lbase2[buf][sbin[k]] + rank[k];
            if (pos < cap) {
                const int64_t gi = plan.rec_first[l] + (int64_t)sbin[k] * cap + pos;
                ridx[gi] = make_uint2(sidx[k], __float_as_uint(swx[k]));
                gs[lpos] = (uint32_t)gi;
                P[k] = lpos;
            } else {
                gs[lpos] = kPosNone;      // (its place in the tile stays empty: the fill pass applies this one to dtable itself)
                P[k] = kLposOvf;
                counters[plan.aux_first + 1] = 1u;
            }
        }
        // (lanes without records of their own - invalid samples, the followers of a summed run - write nothing: the fill pass decides from the
        // same cells which lanes read)
        if (cell.valid && (!reduce || true_head))
            pos4[(int64_t)l * pos_stride + s] = make_uint4(P[0] | (P[1] << 16), P[2] | (P[3] << 16), P[4] | (P[5] << 16), P[6] | (P[7] << 16));
    } else {
    uint4 *lrecs = recs + plan.rec_first[l];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= n_slots || sbin[k] < 0) continue;
        const uint4 rec = make_uint4(sidx[k], __float_as_uint(swx[k]), __float_as_uint(sa[k]), __float_as_uint(sb[k]));
        emit_record<F>(lrecs, dtable, lp, sbin[k], gbase[sbin[k]] + rank[k], cap, shift, rec, &counters[plan.aux_first + 1]);
    }
    }
    // no barrier here: the next tile counts into the other histogram and reserves into the other gbase; this tile's gbase is
    // overwritten two tiles later, behind two more barriers
    }
}

// The value half of the planned scatter (see scatter_bin_kernel<.., PLAN = true>): one lane per (level, sample) again, the SAME waves of 64
// consecutive samples - so every wave re-derives the plan's decisions (valid, run heads, `reduce`) from the same cells - and the gradient
// times the weights goes to the slots the plan reserved: vals[slot] = {a0, a1}.  Runs are summed exactly as the one-pass producer sums them
// (sum_cell_runs).  A slot the plan marked kPosOvf (its bin was full) is applied to dtable directly, with the rows recomputed here.
template <int F>
// (every argument BY VALUE: a reference to the caller's Cell or LevelParams pins that object in scratch memory for the whole kernel - ten scratch
// stores per lane and tile, 170 MB of extra HBM writes per launch, the fill pass at 150 us instead of 106)
__device__ __noinline__ void fill_overflow(float *dtable, uint32_t lv_mask, uint32_t lv_size, double lv_inv_size, int64_t lv_offset, int lowbits,
                                           uint32_t c0, uint32_t c1, uint32_t c2, float w0, int path, int k, float a0, float a1) {
    LevelParams lp;
    lp.mask = lv_mask; lp.size = lv_size; lp.inv_size = lv_inv_size; lp.offset = lv_offset; lp.res = 0; lp.pad = 0;
    Cell cell;
    cell.c[0] = c0; cell.c[1] = c1; cell.c[2] = c2; cell.w[0] = w0; cell.valid = true;
    if (path == 0) {            // hashed level, pair record pr = k
        const uint32_t hy = (cell.c[1] + (uint32_t)(k & 1)) * 2654435761u, hz = (cell.c[2] + (uint32_t)(k >> 1)) * 805459861u;
        const uint32_t hyz = hy ^ hz;
        const uint32_t r0 = (cell.c[0] ^ hyz) & lp.mask, r1 = ((cell.c[0] + 1u) ^ hyz) & lp.mask;
        const float wx = cell.w[0], wl = 1.0f - wx;
        overflow_add<F>(dtable, lp, r0, a0 * wl, a1 * wl);
        overflow_add<F>(dtable, lp, r1, a0 * wx, a1 * wx);
        return;
    }
    uint32_t rows[8];
    corner_rows(cell.c, lp, lowbits, rows);
    if (path == 3) {            // a run's single-row record of corner q = k
        overflow_add<F>(dtable, lp, rows[k], a0, a1);
    } else {                    // general path, slot 2 pr (+ 1): path 1 = an unsplit pair, path 2 = one row of a split pair
        const int pr = k >> 1;
        const uint32_t oy = pr & 1, oz = pr >> 1;
        const uint32_t r0 = rows[oy + (oz << 2)], r1 = rows[2 + oy + (oz << 2)];
        if (path == 1) {
            const float wx = cell.w[0], wl = 1.0f - wx;
            overflow_add<F>(dtable, lp, r0, a0 * wl, a1 * wl);
            overflow_add<F>(dtable, lp, r1, a0 * wx, a1 * wx);
        } else {
            overflow_add<F>(dtable, lp, (k & 1) ? r1 : r0, a0, a1);
        }
    }
}

constexpr int kFillThreads = 1024;
typedef float f2v __attribute__((ext_vector_type(2)));

template <int F>
__global__ void __launch_bounds__(kFillThreads)
scatter_fill_kernel(const float *__restrict__ xyz, const float *__restrict__ dout, int64_t dout_lm_stride, GridParams g, BinPlan plan,
                    const uint4 *__restrict__ pos4, int64_t pos_stride, const uint32_t *__restrict__ gslot, const uint32_t *__restrict__ tile_count,
                    int64_t n_tiles, float2 *__restrict__ vals, uint32_t *__restrict__ counters, float *__restrict__ dtable, int64_t n,
                    const int32_t *n_ptr) {
    // Scattered stores are what the one-pass producer and the first form of this pass both cost: ~10 M 8- or 16-byte stores to as many
    // different lines are ~65 us of L2 write requests whatever is computed around them (this pass without its stores: 38 us; measured,
    // DESIGN.md).  So the values of a tile are staged in LDS at the tile-local positions the plan assigned - the tile's records sorted by bin -
    // and copied out by consecutive lanes: a (bin, tile) run is contiguous in both orders, a store instruction then covers a few whole lines.
    extern __shared__ __attribute__((aligned(16))) f2v stage[];      // 2 x kTileSlots
    const int64_t cnt = dev_count(n, n_ptr);
    const int l = blockIdx.y;
    if (!((plan.active_levels >> l) & 1u)) return;
    const LevelParams lp = g.lv[l];
    const int lane = threadIdx.x & 63;
    const bool nosplit = (plan.nosplit_levels >> l) & 1u;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));      // (HIP's uint4 struct, assigned under a condition inside a lambda, stayed in scratch memory)
    float np_[3] = {0.f, 0.f, 0.f}, ng0 = 0.f, ng1 = 0.f;
    u4v nP = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    uint32_t n_rec_next = 0u;
    const int64_t stride = (int64_t)gridDim.x * kFillThreads;
    auto fetch = [&](int64_t t0) {       // inputs, slots and record count of the tile at t0, one trip ahead
        const int64_t sn = t0 + threadIdx.x;
        if (t0 < cnt) n_rec_next = tile_count[(int64_t)l * n_tiles + t0 / kFillThreads];
        if (sn < cnt) {
            np_[0] = xyz[3 * sn]; np_[1] = xyz[3 * sn + 1]; np_[2] = xyz[3 * sn + 2];
            const float *gp = dout_lm_stride ? dout + ((int64_t)l * dout_lm_stride + sn) * F : dout + (sn * g.L + l) * F;
            ng0 = __builtin_nontemporal_load(gp);
            ng1 = F > 1 ? __builtin_nontemporal_load(gp + (F > 1 ? 1 : 0)) : 0.f;
            nP = reinterpret_cast<const u4v *>(pos4)[(int64_t)l * pos_stride + sn];
        }
    };
    int buf = 0;
    fetch((int64_t)blockIdx.x * kFillThreads);
    for (int64_t tile0 = (int64_t)blockIdx.x * kFillThreads; tile0 < cnt; tile0 += stride, buf ^= 1) {
        const int64_t s = tile0 + threadIdx.x;
        const float p[3] = {np_[0], np_[1], np_[2]};
        float g0 = ng0, g1 = ng1;
        const u4v Pw = nP;
        const uint32_t n_rec = n_rec_next;
        // the global slots of the tile's records, in tile order: consecutive lanes, consecutive words (needed behind the barrier)
        const uint32_t *gs = gslot + ((int64_t)l * n_tiles + tile0 / kFillThreads) * kTileSlots;
        uint32_t G[kTileSlots / kFillThreads];
#pragma unroll
        for (int j = 0; j < kTileSlots / kFillThreads; ++j) {
            const uint32_t i = threadIdx.x + j * kFillThreads;
            G[j] = i < n_rec ? gs[i] : kPosNone;
        }
        fetch(tile0 + stride);
        f2v *st = stage + buf * kTileSlots;
        const Cell cell = locate_fastcell(p, g, lp, s < cnt);
        if (!cell.valid) { g0 = 0.f; g1 = 0.f; }
        if (plan.det) {
            float m = fmaxf(fabsf(g0), fabsf(g1));
            if (!(m == m)) m = 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            if (lane == 0 && m > 0.f) atomicMax(&counters[plan.aux_first], __float_as_uint(m));
        }
        const uint32_t kxy = cell.valid ? (cell.c[0] | (cell.c[1] << 16)) : 0xffffffffu;
        const uint32_t kz = cell.valid ? cell.c[2] : (uint32_t)lane;
        const uint32_t pxy = (uint32_t)dpp_take_i<0x138>(-1, (int)kxy), pz = (uint32_t)dpp_take_i<0x138>(-1, (int)kz);
        const bool true_head = lane == 0 || kxy != pxy || kz != pz;
        const bool head = true_head || (lane & 15) == 0;
        const uint64_t heads = __ballot(true_head && cell.valid), valid = __ballot(cell.valid);
        const bool reduce = 3 * __popcll(heads) <= 2 * __popcll(valid);
        const uint32_t Ps[8] = {Pw.x & 0xffffu, Pw.x >> 16, Pw.y & 0xffffu, Pw.y >> 16, Pw.z & 0xffffu, Pw.z >> 16, Pw.w & 0xffffu, Pw.w >> 16};
        auto put = [&](uint32_t lpos, float a0, float a1, int path, int k) {
            if (lpos < kLposOvf) st[lpos] = f2v{a0, a1};
            else if (lpos == kLposOvf) fill_overflow<F>(dtable, lp.mask, lp.size, lp.inv_size, lp.offset, plan.lowbits[l], cell.c[0], cell.c[1], cell.c[2], cell.w[0], path, k, a0, a1);
        };
        if (!reduce && nosplit) {
            if (cell.valid) {
                const float wy1 = cell.w[1], wy0 = 1.0f - wy1, wz1 = cell.w[2], wz0 = 1.0f - wz1;
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const float wyz = ((pr & 1) ? wy1 : wy0) * ((pr >> 1) ? wz1 : wz0);
                    put(Ps[pr], g0 * wyz, g1 * wyz, 0, pr);
                }
            }
        } else if (!reduce) {
            if (cell.valid) {
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const uint32_t oy = pr & 1, oz = pr >> 1;
                    const float wyz = (oy ? cell.w[1] : 1.0f - cell.w[1]) * (oz ? cell.w[2] : 1.0f - cell.w[2]);
                    const float a0 = g0 * wyz, a1 = g1 * wyz;
                    if (Ps[2 * pr + 1] == kLposNone) {
                        put(Ps[2 * pr], a0, a1, 1, 2 * pr);
                    } else {
                        const float wl = 1.0f - cell.w[0];
                        put(Ps[2 * pr], a0 * wl, a1 * wl, 2, 2 * pr);
                        put(Ps[2 * pr + 1], a0 * cell.w[0], a1 * cell.w[0], 2, 2 * pr + 1);
                    }
                }
            }
        } else {
            float va[8], vb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t ox = (q >> 1) & 1, oy = q & 1, oz = q >> 2;
                float wt = 0.f;
                if (cell.valid) wt = ((ox ? cell.w[0] : 1.0f - cell.w[0]) * (oy ? cell.w[1] : 1.0f - cell.w[1])) * (oz ? cell.w[2] : 1.0f - cell.w[2]);
                va[q] = g0 * wt;
                vb[q] = g1 * wt;
            }
            sum_cell_runs<F>(va, vb, head, true_head, lane);
            if (true_head && cell.valid) {
#pragma unroll
                for (int q = 0; q < 8; ++q) put(Ps[q], va[q], vb[q], 3, q);
            }
        }
        // ONE barrier per tile: the stage is double-buffered - the next tile writes the other half, and this half is written again two tiles on,
        // behind the next tile's barrier, which every wave reaches only after its copy-out here
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kTileSlots / kFillThreads; ++j) {
            const uint32_t i = threadIdx.x + j * kFillThreads;
            if (i < n_rec && G[j] != kPosNone) *reinterpret_cast<f2v *>(vals + G[j]) = st[i];
        }
    }
}

// Producer of the SECOND-ORDER table scatter (arcn_hashgrid_bwd_bwd): dtable[r_q] += dout * D_q with the derivative weights
// D_q = sum_k gdx_k dW_q/dp_k.  D has no (x pair) x (yz) product structure, so every corner goes out as a single-row record
// {row, wx = 0, dout0 * D, dout1 * D}: 8 records per (sample, level), the bins of a plan built for twice the samples.  Same
// ranking scheme as scatter_bin_kernel (LDS histogram, one global integer atomic per bin and tile), same consumer.
template <int F>
__global__ void __launch_bounds__(1024)
scatter_bin_dir_kernel(const float *__restrict__ xyz, const float *__restrict__ gdx, const float *__restrict__ dout, GridParams g,
                       BinPlan plan, uint32_t *__restrict__ counters, uint4 *__restrict__ recs, float *__restrict__ dtable, int64_t n,
                       const int32_t *n_ptr) {
    __shared__ uint32_t hist2[2][kMaxChunks];
    __shared__ uint32_t gbase2[2][kMaxChunks];
    const int64_t cnt = dev_count(n, n_ptr);
    const int l = blockIdx.y;
    if (!((plan.active_levels >> l) & 1u)) return;
    const LevelParams lp = g.lv[l];
    const int nc = plan.n_chunks[l], shift = plan.chunk_shift[l];
    const uint32_t cmask = (1u << shift) - 1u;
    for (int i = threadIdx.x; i < nc; i += 1024) hist2[0][i] = 0u;
    __syncthreads();
    int buf = 0;
    for (int64_t tile0 = (int64_t)blockIdx.x * 1024; tile0 < cnt; tile0 += (int64_t)gridDim.x * 1024, buf ^= 1) {
        uint32_t *hist = hist2[buf], *gbase = gbase2[buf];
        const int64_t s = tile0 + threadIdx.x;
        const int lane = threadIdx.x & 63;
        int sbin[8];
        uint32_t sidx[8], rank[8];
        float sa[8], sb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { sbin[q] = -1; sa[q] = 0.f; sb[q] = 0.f; }
        Cell cell;
        cell.valid = false;
        cell.c[0] = cell.c[1] = cell.c[2] = 0u;
        if (s < cnt) {
            const float p[3] = {xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2]};
            cell = locate(p, g, lp);
            if (cell.valid) {
                const float gd[3] = {gdx[3 * s], gdx[3 * s + 1], gdx[3 * s + 2]};
                const float *gp = dout + (s * g.L + l) * F;
                const float g0 = gp[0], g1 = F > 1 ? gp[F > 1 ? 1 : 0] : 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t o[3] = {(uint32_t)((q >> 1) & 1), (uint32_t)(q & 1), (uint32_t)(q >> 2)};
                    float a[3], sd[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        a[k] = o[k] ? cell.w[k] : 1.0f - cell.w[k];
                        sd[k] = o[k] ? cell.dw[k] : -cell.dw[k];
                    }
                    float D = gd[0] * sd[0] * a[1] * a[2];
                    D = D + gd[1] * a[0] * sd[1] * a[2];
                    D = D + gd[2] * a[0] * a[1] * sd[2];
                    sa[q] = g0 * D;
                    sb[q] = g1 * D;
                }
                if (plan.det) {   // deterministic mode: the consumer's fixed-point scale comes from the largest contribution (order-independent)
                    float m = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) m = fmaxf(m, fmaxf(fabsf(sa[q]), fabsf(sb[q])));
                    if (m == m && m > 0.f) atomicMax(&counters[plan.aux_first], __float_as_uint(m));
                }
            }
        }
        // Runs of consecutive samples in one cell (a ray's neighbours on the coarse and middle levels) are summed in the wave first, like the
        // first-order producer does: without it every sample sent 8 single-row records per level, neighbouring lanes of the consumer fought over
        // the same rows (compare-and-swap retries) and the NeuS-on-hash-grid step's first + second order pass took 245 us for 125 K points
        const uint32_t kxy = cell.valid ? (cell.c[0] | (cell.c[1] << 16)) : 0xffffffffu;
        const uint32_t kz = cell.valid ? cell.c[2] : (uint32_t)lane;
        const uint32_t pxy = (uint32_t)dpp_take_i<0x138>(-1, (int)kxy), pz = (uint32_t)dpp_take_i<0x138>(-1, (int)kz);
        const bool true_head = lane == 0 || kxy != pxy || kz != pz;
        const bool head = true_head || (lane & 15) == 0;
        const uint64_t heads = __ballot(true_head && cell.valid), valid = __ballot(cell.valid);
        const bool reduce = 3 * __popcll(heads) <= 2 * __popcll(valid);      // wave uniform: an average run of >= 1.5 samples
        if (reduce) sum_cell_runs<F>(sa, sb, head, true_head, lane);
        if (cell.valid && (!reduce || true_head)) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t r = hash_row(cell.c[0] + ((q >> 1) & 1), cell.c[1] + (q & 1), cell.c[2] + (q >> 2), lp);
                sbin[q] = (int)(r >> shift);
                sidx[q] = (r & cmask) | 0xffff0000u;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) rank[q] = sbin[q] >= 0 ? atomicAdd(&hist[sbin[q]], 1u) : 0u;
        __syncthreads();
        for (int i = threadIdx.x; i < nc; i += 1024) {
            const uint32_t h = hist[i];
            gbase[i] = h ? atomicAdd(&counters[plan.bin_first[l] + i], h) : 0u;
            hist2[buf ^ 1][i] = 0u;
        }
        __syncthreads();
        const uint32_t cap = (uint32_t)plan.cap[l];
        uint4 *lrecs = recs + plan.rec_first[l];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (sbin[q] < 0) continue;
            const uint4 rec = make_uint4(sidx[q], 0u, __float_as_uint(sa[q]), __float_as_uint(sb[q]));
            emit_record<F>(lrecs, dtable, lp, sbin[q], gbase[sbin[q]] + rank[q], cap, shift, rec, &counters[plan.aux_first + 1]);
        }
    }
}

// a record is read exactly once: a non-temporal load keeps the 230 MB of records from displacing the table's lines on their way through
// (step -2.3 % in three alternations, scatter 0.220 -> 0.208 ms, the gather of the next step -2 us; profiles/r4_ab_nontemporal.txt.  The
// same hint on the record STORES doubles the scatter: scattered 16-byte writes lose their write combining)
__device__ __forceinline__ uint4 ld_rec(const uint4 *p) {
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Optimiser fused into the consumer (arcn_hashgrid_bwd_lm_adam): the owner of a chunk holds the chunk's COMPLETE gradient in LDS, so it
// applies Adam (+ EMA write-back) to its rows right there - parameter and moments read and written once, the gradient never goes to
// HBM (the separate pass re-reads it and clears it: 28 B/param; here 24 B/param and no launch) - while the other owners are still in
// their LDS phase.  Levels with one owner per chunk only (fuse_levels); the caller runs the plain kernel on everything else.
struct AdamFuse {
    float *param, *m, *v;      // table segment of the flat parameter buffer / exp_avg / exp_avg_sq (row 0 of level 0)
    AdamHyper h;
    uint32_t fuse_levels;      // 0 = not fused
    int32_t ema_in_param;      // EMA shadow aliased onto the parameter (else no EMA)
};

// Two workgroups per CU: the 8192-row chunk (64 KiB of accumulators) was sized for it in round 2, but at 96 VGPRs the 16 waves of ONE workgroup
// filled the register file (4 waves x 96 of 512 per SIMD lane) and the second never came.  8 waves per SIMD = 64 VGPRs: the optimiser phase
// takes one float4 per array and trip instead of four (55 VGPRs, no spill; four: 63 + 5 spilled dwords, same speed; the bound alone with four:
// 31 spilled dwords, slower) - consumer 100 -> 91 us, scatter 0.195 -> 0.188 ms per launch, four alternations.
// SPLIT: the records of the planned scatter - index halves (ridx, written by the plan pass) and value halves (vals, by the fill pass)
__device__ __forceinline__ uint4 ld_rec_split(const uint2 *pi, const float2 *pv) {
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    const u2v a = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(pi));
    const u2v b = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(pv));
    return make_uint4(a.x, a.y, b.x, b.y);
}

template <int F, bool SPLIT = false>
__global__ void __launch_bounds__(kTiledThreads, 8)
scatter_accum_kernel(const uint4 *__restrict__ recs, const uint32_t *__restrict__ counters, GridParams g, BinPlan plan,
                     float *__restrict__ dtable, AdamFuse fz, const uint2 *__restrict__ ridx = nullptr, const float2 *__restrict__ vals = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    int k = 0;
    while (k + 1 < g.L && (int)blockIdx.x >= plan.item_first[k + 1]) ++k;
    const int item = blockIdx.x - plan.item_first[k];
    const int l = plan.level_of[k];
    if (!((plan.active_levels >> l) & 1u)) return;
    const int ns = plan.n_splits[l];
    const int chunk = item / ns, split = item - chunk * ns;
    const LevelParams lp = g.lv[l];
    const uint32_t cap = (uint32_t)plan.cap[l];
    uint32_t cnt = counters[plan.bin_first[l] + chunk];
    const bool bin_overflowed = cnt > cap;      // the producer applied the records past the capacity to dtable directly (emit_record)
    cnt = cnt < cap ? cnt : cap;
    const uint32_t per = (cnt + ns - 1) / ns;
    const uint32_t lo = per * split, hi = (lo + per < cnt) ? lo + per : cnt;
    const bool fused = (fz.fuse_levels >> l) & 1u;       // (Adam moves every row, touched or not: an empty bin still has work)
    if (lo >= hi && !fused) return;
    const uint32_t row_lo = (uint32_t)chunk << plan.chunk_shift[l];
    const uint32_t row_hi_raw = row_lo + (1u << plan.chunk_shift[l]);
    const int n_rows = (int)((row_hi_raw < lp.size ? row_hi_raw : lp.size) - row_lo);
    if (plan.det) {
        // DETERMINISTIC mode (ARCN_DETERMINISTIC=1).  Float sums depend on the order in which the records of a row arrive - the order
        // of the bins' reservation atomics and of the compare-and-swap winners, different in every run (identical runs of the full
        // training differed by +-1.3 dB PSNR at 10k iterations).  Here every contribution is converted to 64-bit FIXED
        // POINT (scaled by a power of two taken from the launch's largest |gradient|: max * 2^k < 2^38, so 2^24 terms cannot
        // overflow) and accumulated with 64-bit integer LDS atomics: integer addition is associative, the sums are the same bits
        // whatever the order, and the conversion of a float times a power of two is exact down to max * 2^-37.  One owner per chunk
        // (no split merges), 16 bytes of LDS per row.
        unsigned long long *iacc = reinterpret_cast<unsigned long long *>(acc);
        for (int i = threadIdx.x; i < n_rows * F; i += kTiledThreads) iacc[i] = 0ull;
        __syncthreads();
        const uint32_t mbits = counters[plan.aux_first];
        int e2 = (int)((mbits >> 23) & 0xffu) - 127;
        if (e2 < -126) e2 = -126;
        int k2 = 37 - e2;
        if (k2 > 127) k2 = 127;
        if (k2 < -126) k2 = -126;
        const float scale = __uint_as_float((uint32_t)(k2 + 127) << 23);
        const int64_t rbase = plan.rec_first[l] + (int64_t)chunk * cap + lo;
        const uint4 *prd = recs + rbase;
        // four records per thread and trip, all loads issued before the first atomic (a bin is ~32 records per thread: one load in
        // flight per thread would run at memory latency)
        const uint32_t lend = hi - lo;
        const uint4 none = make_uint4(0xffffffffu, 0u, 0u, 0u);
        for (uint32_t i0g = threadIdx.x; i0g < lend; i0g += kTiledThreads * 4) {
            uint4 rr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                rr[u] = (i0g + u * kTiledThreads < lend) ? (SPLIT ? ld_rec_split(ridx + rbase + i0g + u * kTiledThreads, vals + rbase + i0g + u * kTiledThreads)
                                                                  : prd[i0g + u * kTiledThreads]) : none;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint4 r = rr[u];
                const uint32_t i0 = r.x & 0xffffu, i1 = r.x >> 16;
                if (i0 == 0xffffu) continue;
                const float wx = __uint_as_float(r.y), a0 = __uint_as_float(r.z), a1 = __uint_as_float(r.w);
                const float wl = 1.0f - wx;
                atomicAdd(iacc + i0 * F, (unsigned long long)__float2ll_rn((a0 * wl) * scale));
                if (F > 1) atomicAdd(iacc + i0 * F + (F > 1 ? 1 : 0), (unsigned long long)__float2ll_rn((a1 * wl) * scale));
                if (i1 != 0xffffu) {
                    atomicAdd(iacc + i1 * F, (unsigned long long)__float2ll_rn((a0 * wx) * scale));
                    if (F > 1) atomicAdd(iacc + i1 * F + (F > 1 ? 1 : 0), (unsigned long long)__float2ll_rn((a1 * wx) * scale));
                }
            }
        }
        __syncthreads();
        const double inv = (double)__uint_as_float((uint32_t)(127 - k2) << 23);
        float *dstd = dtable + ((int64_t)lp.offset + row_lo) * F;
        for (int j = threadIdx.x; j < n_rows * F; j += kTiledThreads) {
            const long long v = (long long)iacc[j];
            if (v != 0) dstd[j] += (float)((double)v * inv);
        }
        return;
    }
    uint32_t *locks = reinterpret_cast<uint32_t *>(acc + plan.chunk_floats);
    {
        float4 *acc4 = reinterpret_cast<float4 *>(acc);
        for (int i = threadIdx.x; i < (n_rows * F + 3) / 4; i += kTiledThreads) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = threadIdx.x; i < plan.chunk_floats / F / 32; i += kTiledThreads) locks[i] = 0u;
    __syncthreads();
    const bool locked = (plan.lock_levels >> l) & 1u;
    const int64_t rbase = plan.rec_first[l] + (int64_t)chunk * cap + lo;
    const uint4 *pr = recs + rbase;
    const uint2 *pri = ridx + rbase;
    const float2 *prv = vals + rbase;
    auto ld = [&](uint32_t at) { return SPLIT ? ld_rec_split(pri + at, prv + at) : ld_rec(pr + at); };
    const uint32_t len = hi > lo ? hi - lo : 0u;
    // A bin is only ~32 records per thread: with one load in flight per thread the loop would be bound by memory latency.
    // Each trip takes kUnroll records per thread (the next group's loads are issued before the current group is applied).
    constexpr int kUnroll = 4;
    const uint32_t trips = (len + kTiledThreads * kUnroll - 1) / (kTiledThreads * kUnroll);
    const uint4 none = make_uint4(0xffffffffu, 0u, 0u, 0u);  // i0 = i1 = 0xffff: nothing to do
    uint32_t i = threadIdx.x;
    uint4 nxt[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) nxt[u] = (i + u * kTiledThreads < len) ? ld(i + u * kTiledThreads) : none;
    for (uint32_t trip = 0; trip < trips; ++trip) {
        uint4 cur[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) cur[u] = nxt[u];
        i += kTiledThreads * kUnroll;
        if (trip + 1 < trips) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) nxt[u] = (i + u * kTiledThreads < len) ? ld(i + u * kTiledThreads) : none;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint4 r = cur[u];
            const uint32_t i0 = r.x & 0xffffu, i1 = r.x >> 16;
            const bool have = i0 != 0xffffu;
            const float wx = __uint_as_float(r.y), a0 = __uint_as_float(r.z), a1 = __uint_as_float(r.w);
            const float wl = 1.0f - wx;
            if (locked && F == 2) {
                // lock-free: read the 8-byte row, add, 64-bit compare-and-swap; two LDS operations per row update when nobody
                // interferes (the bit lock needs four: or, read, write, and).  The comparison is bitwise, so NaNs and signed
                // zeros are handled; a lost race just retries with the value the swap returned.
                unsigned long long *rows = reinterpret_cast<unsigned long long *>(acc);
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const uint32_t tgt = side ? i1 : i0;
                    const float wt = side ? wx : wl;
                    if (have && tgt != 0xffffu) {
                        const float va = a0 * wt, vb = a1 * wt;
                        unsigned long long *cell = rows + tgt;
                        // first guess: the row is still zero (true for its first update, and then ONE operation does it); a wrong
                        // guess costs what the explicit read would have cost, the failed swap returns the current value
                        unsigned long long seen = 0ull;
                        while (true) {
                            float2 v = __builtin_bit_cast(float2, seen);
                            v.x += va;
                            v.y += vb;
                            unsigned long long expect = seen;
                            if (__hip_atomic_compare_exchange_strong(cell, &expect, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                                break;
                            seen = expect;
                        }
                    }
                }
            } else if (locked) {
                int todo = have ? (i1 != 0xffffu ? 2 : 1) : 0;
                uint32_t tgt = i0;
                float va = a0 * wl, vb = a1 * wl;
                // wave-uniform loop condition: the holder's update and release must stay INSIDE the loop (with a per-lane exit
                // the compiler may sink them past it, where the holder would wait for the spinning lanes of its own wave)
                while (__ballot(todo > 0)) {
                    const uint32_t bit = 1u << (tgt & 31u);
                    uint32_t *lk = locks + (tgt >> 5);
                    uint32_t prev = bit;
                    if (todo > 0) prev = __hip_atomic_fetch_or(lk, bit, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!(prev & bit)) {
                        if (F == 2) {
                            float2 *cell = reinterpret_cast<float2 *>(acc) + tgt;
                            float2 v = *cell;
                            v.x += va;
                            v.y += vb;
                            *cell = v;
                        } else {
                            acc[tgt] += va;
                        }
                        __hip_atomic_fetch_and(lk, ~bit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        --todo;
                        tgt = i1;
                        va = a0 * wx;
                        vb = a1 * wx;
                    }
                }
            } else if (have) {
                float *d0 = acc + i0 * F;
                __hip_atomic_fetch_add(d0, a0 * wl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (F > 1) __hip_atomic_fetch_add(d0 + (F > 1 ? 1 : 0), a1 * wl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (i1 != 0xffffu) {
                    float *d1 = acc + i1 * F;
                    __hip_atomic_fetch_add(d1, a0 * wx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (F > 1) __hip_atomic_fetch_add(d1 + (F > 1 ? 1 : 0), a1 * wx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();
    if (fused) {
        const int64_t base = ((int64_t)lp.offset + row_lo) * F;
        float *P = fz.param + base, *M = fz.m + base, *V = fz.v + base;
        const AdamHyper h = fz.h;
        const int nf = n_rows * F;
        if (bin_overflowed) {
            // this chunk's bin ran over: part of its gradient sits in dtable (the producer's direct atomics, complete since the producer
            // kernel has ended).  Fold it into the accumulator and leave dtable clear - nothing else reads a fused level's dtable rows.
            float *D = dtable + base;
            for (int j = threadIdx.x; j < nf; j += kTiledThreads) {
                acc[j] += D[j];
                D[j] = 0.f;
            }
            __syncthreads();
        }
        if ((base & 3) == 0) {
            const int n4 = nf >> 2;
            constexpr int kB = 1;      // (registers: see the kernel's launch bounds)
            for (int j0 = threadIdx.x; j0 < n4; j0 += kTiledThreads * kB) {
                float4 p4[kB], m4[kB], v4[kB];
#pragma unroll
                for (int u = 0; u < kB; ++u) {
                    const int j = j0 + u * kTiledThreads;
                    if (j < n4) {
                        p4[u] = reinterpret_cast<const float4 *>(P)[j];
                        m4[u] = reinterpret_cast<const float4 *>(M)[j];
                        v4[u] = reinterpret_cast<const float4 *>(V)[j];
                    }
                }
#pragma unroll
                for (int u = 0; u < kB; ++u) {
                    const int j = j0 + u * kTiledThreads;
                    if (j < n4) {
                        const float4 g4 = reinterpret_cast<const float4 *>(acc)[j];
                        float pp[4] = {p4[u].x, p4[u].y, p4[u].z, p4[u].w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
                        float mm[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w}, vv[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float old = pp[k];
                            adam1(pp[k], gg[k], mm[k], vv[k], fz.ema_in_param ? &old : nullptr, h.lr, h.b1, h.b2, h.eps, h.wd, h.ema_decay,
                                  h.gscale, h.bc1, h.bc2_sqrt, h.deb_old, h.deb_new);
                        }
                        reinterpret_cast<float4 *>(P)[j] = make_float4(pp[0], pp[1], pp[2], pp[3]);
                        reinterpret_cast<float4 *>(M)[j] = make_float4(mm[0], mm[1], mm[2], mm[3]);
                        reinterpret_cast<float4 *>(V)[j] = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    }
                }
            }
            for (int j = (n4 << 2) + threadIdx.x; j < nf; j += kTiledThreads) {
                float old = P[j];
                adam1(P[j], acc[j], M[j], V[j], fz.ema_in_param ? &old : nullptr, h.lr, h.b1, h.b2, h.eps, h.wd, h.ema_decay, h.gscale, h.bc1,
                      h.bc2_sqrt, h.deb_old, h.deb_new);
            }
        } else {
            for (int j = threadIdx.x; j < nf; j += kTiledThreads) {
                float old = P[j];
                adam1(P[j], acc[j], M[j], V[j], fz.ema_in_param ? &old : nullptr, h.lr, h.b1, h.b2, h.eps, h.wd, h.ema_decay, h.gscale, h.bc1,
                      h.bc2_sqrt, h.deb_old, h.deb_new);
            }
        }
        return;
    }
    float *dst = dtable + ((int64_t)lp.offset + row_lo) * F;
    if (ns == 1) {
        // exclusive owner: plain coalesced 16-byte stores (+= so that callers accumulating over several launches stay correct);
        // the global loads of a batch are all issued before the first add, or the loop would run at memory latency
        const int n4 = (n_rows * F) >> 2;  // level offsets and chunk sizes are multiples of 4 floats except on a ragged tail
        const bool aligned = (((int64_t)lp.offset + row_lo) * F) % 4 == 0;
        if (aligned) {
            float4 *dst4 = reinterpret_cast<float4 *>(dst);
            const float4 *acc4 = reinterpret_cast<const float4 *>(acc);
            constexpr int kBatch = 8;
            for (int j0 = threadIdx.x; j0 < n4; j0 += kTiledThreads * kBatch) {
                float4 v[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int j = j0 + u * kTiledThreads;
                    if (j < n4) v[u] = dst4[j];
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int j = j0 + u * kTiledThreads;
                    if (j < n4) {
                        const float4 a = acc4[j];
                        v[u].x += a.x; v[u].y += a.y; v[u].z += a.z; v[u].w += a.w;
                        dst4[j] = v[u];
                    }
                }
            }
            for (int j = n4 * 4 + threadIdx.x; j < n_rows * F; j += kTiledThreads) dst[j] += acc[j];
        } else {
            for (int j = threadIdx.x; j < n_rows * F; j += kTiledThreads) dst[j] += acc[j];
        }
    } else {
        for (int j = threadIdx.x; j < n_rows * F; j += kTiledThreads) {
            const float v = acc[j];
            if (v != 0.f) unsafeAtomicAdd(dst + j, v);
        }
    }
}

static int build_params(const arcn_hashgrid_desc *d, GridParams &g) {
    if (!d) return einval("hashgrid: desc is NULL");
    if (d->n_levels < 1 || d->n_levels > ARCN_MAX_LEVELS) return einval("hashgrid: n_levels out of range");
    if (!(d->n_feat == 1 || d->n_feat == 2 || d->n_feat == 4)) return einval("hashgrid: n_feat must be 1, 2 or 4");
    g.L = d->n_levels;
    g.F = d->n_feat;
    for (int k = 0; k < 3; ++k) { g.mn[k] = d->min_xyz[k]; g.mx[k] = d->max_xyz[k]; }
    for (int l = 0; l < g.L; ++l) {
        int64_t size = d->offsets[l + 1] - d->offsets[l];
        if (size <= 0 || size > 0xffffffffll || d->resolutions[l] <= 0) return einval("hashgrid: bad level table");
        LevelParams &lp = g.lv[l];
        lp.res = d->resolutions[l];
        for (int k = 0; k < 3; ++k) {
            const float ext = d->max_xyz[k] - d->min_xyz[k];
            lp.vs[k] = ext / (float)lp.res;
            lp.rvs[k] = 1.0f / lp.vs[k];
        }
        lp.size = (uint32_t)size;
        lp.mask = ((size & (size - 1)) == 0) ? (uint32_t)(size - 1) : 0u;
        lp.pad = 0;
        lp.inv_size = 1.0 / (double)size;
        lp.offset = d->offsets[l];
    }
    return ARCN_OK;
}

// Owner chunks, bins and dispatch order of the v3 backward for a capacity of n samples.
static int build_bin_plan(const GridParams &g, int64_t n, BinPlan &plan) {
    // 8192-row owner chunks (64 KiB of LDS, two consumer workgroups per CU): 704 owners on 512 slots instead of 352 on 256 -
    // step 0.680 -> 0.664 ms against the 16384-row chunks, scatter 0.213 -> 0.196 ms per launch (A/B in one session)
    constexpr int chunk_floats = 16384;
    static_assert(chunk_floats <= kChunkFloats, "owner chunk larger than the consumer's LDS accumulator");
    plan.chunk_floats = chunk_floats;
    { static const int det = [] { const char *e = getenv("ARCN_DETERMINISTIC"); return e ? atoi(e) : 0; }(); plan.det = det ? 1 : 0; }
    const int rows_cap = chunk_floats / g.F;
    int bins = 0;
    int64_t recs = 0;
    plan.active_levels = 0xffffffffu;      // (arcn_hashgrid_bwd_lm_levels narrows it to a level group)
    plan.lock_levels = 0u;
    plan.pair_levels = 0u;
    plan.nosplit_levels = 0u;
    for (int l = 0; l < g.L; ++l) {
        const int64_t size = g.lv[l].size;
        int shift = 0;
        while ((1 << shift) < rows_cap && ((int64_t)1 << shift) < size) ++shift;
        const int64_t nc = (size + ((int64_t)1 << shift) - 1) >> shift;
        if (nc > kMaxChunks) return einval("hashgrid_bwd: level too large for the binned scatter");
        // hashed power-of-two levels: both rows of an x pair share a chunk -> 4 pair records per sample, samples interleaved;
        // the small levels below them: runs of samples in one cell are summed in the producer -> at most 8 singles per sample
        const bool paired = g.lv[l].mask != 0 && nc > 1;
        if (paired) plan.pair_levels |= 1u << l;
        if (g.lv[l].mask != 0 && ((int64_t)1 << shift) > g.lv[l].res) plan.nosplit_levels |= 1u << l;
        if (paired) plan.lock_levels |= 1u << l;
        const bool locked = (plan.lock_levels >> l) & 1u;
        const int64_t mean = (paired ? 4 : 8) * n / nc;
        int64_t cap = nc == 1 ? 8 * n : 2 * mean + 1024;
        // deterministic mode must not take the overflow path (global float atomics): the dense (un-hashed) levels, whose bins are
        // regions of space and fill unevenly, get room for two records per sample in EVERY bin (their runs are merged first: ~n / 8
        // records per level in the bench workload); the hashed levels spread evenly, 2x the mean is hundreds of sigmas away
        if (plan.det && nc > 1 && g.lv[l].mask == 0 && cap < 2 * n + 1024) cap = 2 * n + 1024;
        if (cap > 8 * n) cap = 8 * n;
        if (cap > 0x7fffffffll) return einval("hashgrid_bwd: too many samples for the binned scatter");
        // one exclusive owner per chunk (plain-store flush); splits only where float atomics make the items long
        int ns = (locked || plan.det) ? 1 : (int)(64 / nc);
        if (ns < 1) ns = 1;
        while (ns > 1 && cap / ns < 8192) ns >>= 1;
        plan.n_chunks[l] = (int)nc;
        plan.n_splits[l] = ns;
        plan.chunk_shift[l] = shift;
        plan.bin_first[l] = bins;
        plan.cap[l] = (int)cap;
        plan.rec_first[l] = recs;
        bins += (int)nc;
        recs += nc * cap;
    }
    plan.n_bins = bins;
    plan.aux_first = bins;      // two words behind the bin counters (zeroed with them)
    plan.n_recs = recs;
    for (int l = 0; l < ARCN_MAX_LEVELS; ++l) plan.lowbits[l] = l < g.L ? (int8_t)level_lowbits(g.lv[l]) : 0;
    // dispatch: levels with the longest items (largest bins per workgroup) first, ties finest first
    int order[ARCN_MAX_LEVELS];
    for (int l = 0; l < g.L; ++l) order[l] = g.L - 1 - l;
    for (int a = 1; a < g.L; ++a)
        for (int b = a; b > 0 && plan.cap[order[b]] / plan.n_splits[order[b]] > plan.cap[order[b - 1]] / plan.n_splits[order[b - 1]]; --b) {
            const int tmp = order[b];
            order[b] = order[b - 1];
            order[b - 1] = tmp;
        }
    int items = 0, k = 0;
    for (; k < g.L; ++k) {
        const int l = order[k];
        plan.item_first[k] = items;
        plan.level_of[k] = l;
        items += plan.n_chunks[l] * plan.n_splits[l];
    }
    for (; k <= ARCN_MAX_LEVELS; ++k) {
        plan.item_first[k] = items;
        if (k < ARCN_MAX_LEVELS) plan.level_of[k] = 0;
    }
    return ARCN_OK;
}

// a level the scatter's chunk owners may apply the optimiser to: one owner per chunk, and its rows start and end on 16-byte boundaries of the
// flat buffers - the caller's plain Adam pass on the REST of the buffer works in float4 (arcn_adam_ema_step_runs: every run 4-float aligned)
static inline bool level_fusable(const GridParams &g, const BinPlan &plan, int l) {
    const int64_t lo = (int64_t)g.lv[l].offset * g.F, hi = ((int64_t)g.lv[l].offset + g.lv[l].size) * g.F;
    return plan.n_splits[l] == 1 && ((plan.active_levels >> l) & 1u) && (lo & 3) == 0 && (hi & 3) == 0;
}

static inline int64_t bin_counter_floats(const BinPlan &plan) { return ((int64_t)plan.n_bins + 2 + 63) / 64 * 64; }

// the plan workspace of the planned scatter (one per batch in flight): [bin counters | index halves, 8 B per record slot | tile-local positions,
// eight 16-bit words per (level, sample) with the level stride n | global slots in tile order, kTileSlots words per (level, 1024-sample tile) |
// records per (level, tile)]
constexpr int kFillWgs = 32;       // persistent workgroups per level of scatter_fill_kernel, as many as the plan pass
static inline int64_t plan_ridx_first(const BinPlan &plan) { return bin_counter_floats(plan); }
static inline int64_t plan_pos_first(const BinPlan &plan) { return (plan_ridx_first(plan) + 2 * plan.n_recs + 3) / 4 * 4; }
static inline int64_t plan_tiles(int64_t n) { return (n + kFillThreads - 1) / kFillThreads; }
static inline int64_t plan_gslot_first(const BinPlan &plan, const GridParams &g, int64_t n) { return plan_pos_first(plan) + (int64_t)g.L * n * 4; }
static inline int64_t plan_count_first(const BinPlan &plan, const GridParams &g, int64_t n) { return plan_gslot_first(plan, g, n) + (int64_t)g.L * plan_tiles(n) * kTileSlots; }
static inline int64_t plan_ws_floats_for(const BinPlan &plan, const GridParams &g, int64_t n) { return plan_count_first(plan, g, n) + (int64_t)g.L * plan_tiles(n); }

}  // namespace arcn

using namespace arcn;

ARCN_EXPORT int arcn_hashgrid_fwd(const float *xyz, const float *table, const arcn_hashgrid_desc *desc_host, float *out,
                                  int32_t *hash_idx, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!xyz || !table || !out) return einval("hashgrid_fwd: missing argument");
    GridParams g;
    int rc = build_params(desc_host, g);
    if (rc) return rc;
    dim3 grid((unsigned)ceil_div<int64_t>(n * g.L, 256));
    switch (g.F) {
    case 1: hipLaunchKernelGGL(hashgrid_fwd_kernel<1>, grid, dim3(256), 0, as_stream(stream), xyz, table, g, out, hash_idx, n, n_ptr); break;
    case 2: hipLaunchKernelGGL(hashgrid_fwd_kernel<2>, grid, dim3(256), 0, as_stream(stream), xyz, table, g, out, hash_idx, n, n_ptr); break;
    default: hipLaunchKernelGGL(hashgrid_fwd_kernel<4>, grid, dim3(256), 0, as_stream(stream), xyz, table, g, out, hash_idx, n, n_ptr); break;
    }
    return check_launch("hashgrid_fwd");
}

ARCN_EXPORT int64_t arcn_hashgrid_bwd_workspace_floats(const arcn_hashgrid_desc *desc_host, int64_t n);

static int hashgrid_bwd_impl(const float *xyz, const float *table, const float *dout, int64_t dout_lm_stride,
                             const arcn_hashgrid_desc *desc_host, float *dtable, float *dxyz, float *workspace,
                             int64_t workspace_floats, int64_t n, const int32_t *n_ptr, void *stream, const AdamFuse *fuse = nullptr,
                             uint32_t *fused_levels_out = nullptr, bool counters_clear = false, uint32_t level_mask = 0xffffffffu,
                             float *plan_ws = nullptr, int64_t plan_ws_floats = 0) {
    if (n <= 0) return ARCN_OK;
    if (!xyz || !dout || (!dtable && !dxyz) || (dxyz && !table)) return einval("hashgrid_bwd: missing argument");
    GridParams g;
    int rc = build_params(desc_host, g);
    if (rc) return rc;
    if (workspace && workspace_floats < arcn_hashgrid_bwd_workspace_floats(desc_host, n))
        return einval("hashgrid_bwd: workspace smaller than arcn_hashgrid_bwd_workspace_floats(desc, n)");
    if (workspace && dtable && !dxyz && g.F <= 2) {
        // v3: bin the corner-pair records per owner chunk, then accumulate every bin in LDS
        BinPlan plan;
        rc = build_bin_plan(g, n, plan);
        if (rc) return rc;
        plan.active_levels &= level_mask;      // arcn_hashgrid_bwd_lm_levels: the workgroups of the other levels leave at once
        uint32_t *counters = reinterpret_cast<uint32_t *>(workspace);
        uint4 *recs = reinterpret_cast<uint4 *>(workspace + bin_counter_floats(plan));
        // planned scatter (arcn_hashgrid_bwd_plan ran on these samples): the plan's counters, index halves and slots; the value halves go
        // where the records would have gone
        const uint2 *p_ridx = nullptr;
        const uint4 *p_pos = nullptr;
        const uint32_t *p_gslot = nullptr, *p_count = nullptr;
        if (plan_ws) {
            if (plan_ws_floats < plan_ws_floats_for(plan, g, n)) return einval("hashgrid_bwd_planned: plan workspace smaller than arcn_hashgrid_plan_workspace_floats(desc, n)");
            counters = reinterpret_cast<uint32_t *>(plan_ws);
            p_ridx = reinterpret_cast<const uint2 *>(plan_ws + plan_ridx_first(plan));
            p_pos = reinterpret_cast<const uint4 *>(plan_ws + plan_pos_first(plan));
            p_gslot = reinterpret_cast<const uint32_t *>(plan_ws + plan_gslot_first(plan, g, n));
            p_count = reinterpret_cast<const uint32_t *>(plan_ws + plan_count_first(plan, g, n));
        }
        // (the whole 256-byte padded block: ONE fill launch, an odd size is two; none when the caller's previous pass left it clear)
        hipError_t e = (counters_clear || plan_ws) ? hipSuccess : hipMemsetAsync(counters, 0, sizeof(uint32_t) * (size_t)bin_counter_floats(plan), as_stream(stream));
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
        const size_t lds = plan.det ? sizeof(unsigned long long) * (size_t)plan.chunk_floats
                                    : sizeof(float) * (size_t)plan.chunk_floats + sizeof(uint32_t) * (size_t)(plan.chunk_floats / 32);
        e = g.F == 1
            ? hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_accum_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
            : hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_accum_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
        constexpr int bin_wgs = 32;   // persistent producer workgroups per level (16 / 64 measured slower)
        // producer workgroup size = samples per tile.  The kernel needs ~124 VGPRs, so a 1024-thread workgroup owns a whole CU;
        // two 512-thread workgroups per CU (covering each other's barrier phases) measured SLOWER, 0.262 vs 0.249 ms for the
        // whole scatter (twice the per-bin global atomics, half the run length per bin)
        constexpr int bin_threads = 1024;
        int64_t bx = ceil_div<int64_t>(n, bin_threads);
        const int64_t wgs = (int64_t)bin_wgs * (1024 / bin_threads);
        if (bx > wgs) bx = wgs;  // persistent workgroups per level
        dim3 bgrid((unsigned)bx, (unsigned)g.L);
        dim3 agrid((unsigned)plan.item_first[g.L]);
#define ARCN_BIN(F_, T_) hipLaunchKernelGGL((scatter_bin_kernel<F_, T_>), bgrid, dim3(T_), 0, st_bin, xyz, dout, dout_lm_stride, g, plan, counters, recs, dtable, n, n_ptr, (uint2 *)nullptr, (uint4 *)nullptr, (int64_t)0, (uint32_t *)nullptr, (uint32_t *)nullptr, (int64_t)0)
#define ARCN_ACC(F_) hipLaunchKernelGGL(scatter_accum_kernel<F_>, agrid, dim3(kTiledThreads), lds, st_acc, recs, counters, g, plan, dtable, fz, (const uint2 *)nullptr, (const float2 *)nullptr)
        hipStream_t st_bin = as_stream(stream), st_acc = as_stream(stream);
        AdamFuse fz{};
        if (fuse && !plan.det) {
            fz = *fuse;
            fz.fuse_levels = 0u;
            for (int l = 0; l < g.L; ++l)
                if (level_fusable(g, plan, l)) fz.fuse_levels |= 1u << l;
        }
        if (fused_levels_out) *fused_levels_out = fz.fuse_levels;
        if (plan_ws) {
            e = g.F == 1
                ? hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_accum_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                : hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_accum_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
            float2 *vals = reinterpret_cast<float2 *>(recs);
            int64_t fx = ceil_div<int64_t>(n, kFillThreads);
            if (fx > kFillWgs) fx = kFillWgs;
            dim3 fgrid((unsigned)fx, (unsigned)g.L);
            const size_t flds = 2 * (size_t)kTileSlots * sizeof(float2);
            e = g.F == 1 ? hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_fill_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds)
                         : hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_fill_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds);
            if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
            if (g.F == 1) {
                hipLaunchKernelGGL(scatter_fill_kernel<1>, fgrid, dim3(kFillThreads), flds, st_bin, xyz, dout, dout_lm_stride, g, plan, p_pos, n, p_gslot, p_count, plan_tiles(n), vals, counters, dtable, n, n_ptr);
                hipLaunchKernelGGL((scatter_accum_kernel<1, true>), agrid, dim3(kTiledThreads), lds, st_acc, recs, counters, g, plan, dtable, fz, p_ridx, (const float2 *)vals);
            } else {
                hipLaunchKernelGGL(scatter_fill_kernel<2>, fgrid, dim3(kFillThreads), flds, st_bin, xyz, dout, dout_lm_stride, g, plan, p_pos, n, p_gslot, p_count, plan_tiles(n), vals, counters, dtable, n, n_ptr);
                hipLaunchKernelGGL((scatter_accum_kernel<2, true>), agrid, dim3(kTiledThreads), lds, st_acc, recs, counters, g, plan, dtable, fz, p_ridx, (const float2 *)vals);
            }
            return check_launch("hashgrid_bwd_planned");
        }
        if (g.F == 1) {
            ARCN_BIN(1, 1024);
            ARCN_ACC(1);
        } else {
            ARCN_BIN(2, 1024);
            ARCN_ACC(2);
        }
#undef ARCN_ACC
#undef ARCN_BIN
        return check_launch("hashgrid_bwd_binned");
    }
    if (dout_lm_stride) return einval("hashgrid_bwd_lm: needs a workspace, dtable only, n_feat 1 or 2");
    dim3 grid((unsigned)ceil_div<int64_t>(n * g.L, 256));
    switch (g.F) {
    case 1: hipLaunchKernelGGL(hashgrid_bwd_kernel<1>, grid, dim3(256), 0, as_stream(stream), xyz, table, dout, g, dtable, dxyz, n, n_ptr); break;
    case 2: hipLaunchKernelGGL(hashgrid_bwd_kernel<2>, grid, dim3(256), 0, as_stream(stream), xyz, table, dout, g, dtable, dxyz, n, n_ptr); break;
    default: hipLaunchKernelGGL(hashgrid_bwd_kernel<4>, grid, dim3(256), 0, as_stream(stream), xyz, table, dout, g, dtable, dxyz, n, n_ptr); break;
    }
    return check_launch("hashgrid_bwd");
}

ARCN_EXPORT int arcn_hashgrid_bwd(const float *xyz, const float *table, const float *dout, const arcn_hashgrid_desc *desc_host,
                                  float *dtable, float *dxyz, float *workspace, int64_t workspace_floats, int64_t n,
                                  const int32_t *n_ptr, void *stream) {
    return hashgrid_bwd_impl(xyz, table, dout, 0, desc_host, dtable, dxyz, workspace, workspace_floats, n, n_ptr, stream);
}

ARCN_EXPORT int arcn_hashgrid_bwd_bwd(const float *xyz, const float *gdx, const float *table, const float *dout,
                                      const arcn_hashgrid_desc *desc_host, float *ddout, float *dtable, float *d2xyz,
                                      float *workspace, int64_t workspace_floats, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!xyz || !gdx || !table || !dout || (!ddout && !dtable && !d2xyz)) return einval("hashgrid_bwd_bwd: missing argument");
    GridParams g;
    int rc = build_params(desc_host, g);
    if (rc) return rc;
    // binned table scatter: 8 single-row records per (sample, level), bins sized by a plan for 2n samples
    const bool binned = workspace && dtable && g.F <= 2;
    if (binned) {
        if (workspace_floats < arcn_hashgrid_bwd_workspace_floats(desc_host, 2 * n))
            return einval("hashgrid_bwd_bwd: workspace smaller than arcn_hashgrid_bwd_workspace_floats(desc, 2 * n)");
        BinPlan plan;
        rc = build_bin_plan(g, 2 * n, plan);
        if (rc) return rc;
        uint32_t *counters = reinterpret_cast<uint32_t *>(workspace);
        uint4 *recs = reinterpret_cast<uint4 *>(workspace + bin_counter_floats(plan));
        hipError_t e = hipMemsetAsync(counters, 0, sizeof(uint32_t) * (size_t)bin_counter_floats(plan), as_stream(stream));   // (the whole 256-byte padded block: ONE fill launch, an odd size is two)
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
        const size_t lds = plan.det ? sizeof(unsigned long long) * (size_t)plan.chunk_floats
                                    : sizeof(float) * (size_t)plan.chunk_floats + sizeof(uint32_t) * (size_t)(plan.chunk_floats / 32);
        e = g.F == 1
            ? hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_accum_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
            : hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_accum_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
        int64_t bx = ceil_div<int64_t>(n, 1024);
        if (bx > 32) bx = 32;
        dim3 bgrid((unsigned)bx, (unsigned)g.L), agrid((unsigned)plan.item_first[g.L]);
        if (g.F == 1) {
            hipLaunchKernelGGL(scatter_bin_dir_kernel<1>, bgrid, dim3(1024), 0, as_stream(stream), xyz, gdx, dout, g, plan, counters, recs, dtable, n, n_ptr);
            hipLaunchKernelGGL(scatter_accum_kernel<1>, agrid, dim3(kTiledThreads), lds, as_stream(stream), recs, counters, g, plan, dtable, AdamFuse{}, (const uint2 *)nullptr, (const float2 *)nullptr);
        } else {
            hipLaunchKernelGGL(scatter_bin_dir_kernel<2>, bgrid, dim3(1024), 0, as_stream(stream), xyz, gdx, dout, g, plan, counters, recs, dtable, n, n_ptr);
            hipLaunchKernelGGL(scatter_accum_kernel<2>, agrid, dim3(kTiledThreads), lds, as_stream(stream), recs, counters, g, plan, dtable, AdamFuse{}, (const uint2 *)nullptr, (const float2 *)nullptr);
        }
        if ((rc = check_launch("hashgrid_bwd_bwd_binned"))) return rc;
        if (!ddout && !d2xyz) return ARCN_OK;
    }
    float *dt_plain = binned ? nullptr : dtable;
    dim3 grid((unsigned)ceil_div<int64_t>(n * g.L, 256));
    switch (g.F) {
    case 1: hipLaunchKernelGGL(hashgrid_bwd_bwd_kernel<1>, grid, dim3(256), 0, as_stream(stream), xyz, gdx, table, dout, g, ddout, dt_plain, d2xyz, n, n_ptr); break;
    case 2: hipLaunchKernelGGL(hashgrid_bwd_bwd_kernel<2>, grid, dim3(256), 0, as_stream(stream), xyz, gdx, table, dout, g, ddout, dt_plain, d2xyz, n, n_ptr); break;
    default: hipLaunchKernelGGL(hashgrid_bwd_bwd_kernel<4>, grid, dim3(256), 0, as_stream(stream), xyz, gdx, table, dout, g, ddout, dt_plain, d2xyz, n, n_ptr); break;
    }
    return check_launch("hashgrid_bwd_bwd");
}

// First- and second-order table gradients of ONE batch in one consumer pass (NeuS on the hash grid: the table receives d loss / d enc
// through the encoding AND d loss / d normal through the encoding's input gradient).  Both producers append their records to the same
// bins - scatter_bin_dir_kernel 8 single-row records per (sample, level), scatter_bin_kernel its pair / run records - of a plan sized
// for 3 n samples, and every owner chunk is accumulated and written back ONCE: one scatter_accum pass (the dominant kernel of the
// scatter: 114 us for the 16-level table whatever the sample count) instead of two.
static int hashgrid_bwd_first_second_impl(const float *xyz, const float *dout, const float *gdx, const float *dout_dx,
                                          const arcn_hashgrid_desc *desc_host, float *dtable, float *workspace, int64_t workspace_floats,
                                          int64_t n, void *stream, const AdamFuse *fuse, uint32_t *fused_levels_out) {
    if (n <= 0) return ARCN_OK;
    if (!xyz || !dout || !gdx || !dout_dx || !dtable || !workspace) return einval("hashgrid_bwd_first_second: missing argument");
    GridParams g;
    int rc = build_params(desc_host, g);
    if (rc) return rc;
    if (g.F > 2) return einval("hashgrid_bwd_first_second: n_feat 1 or 2");
    if (workspace_floats < arcn_hashgrid_bwd_workspace_floats(desc_host, 3 * n))
        return einval("hashgrid_bwd_first_second: workspace smaller than arcn_hashgrid_bwd_workspace_floats(desc, 3 * n)");
    BinPlan plan;
    rc = build_bin_plan(g, 3 * n, plan);
    if (rc) return rc;
    uint32_t *counters = reinterpret_cast<uint32_t *>(workspace);
    uint4 *recs = reinterpret_cast<uint4 *>(workspace + bin_counter_floats(plan));
    hipError_t e = hipMemsetAsync(counters, 0, sizeof(uint32_t) * (size_t)bin_counter_floats(plan), as_stream(stream));
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
    const size_t lds = plan.det ? sizeof(unsigned long long) * (size_t)plan.chunk_floats
                                : sizeof(float) * (size_t)plan.chunk_floats + sizeof(uint32_t) * (size_t)(plan.chunk_floats / 32);
    e = g.F == 1
        ? hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_accum_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
        : hipFuncSetAttribute(reinterpret_cast<const void *>(scatter_accum_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
    int64_t bx = ceil_div<int64_t>(n, 1024);
    if (bx > 32) bx = 32;
    dim3 bgrid((unsigned)bx, (unsigned)g.L), agrid((unsigned)plan.item_first[g.L]);
    const int32_t *no_count = nullptr;
    AdamFuse fz{};
    if (fuse && !plan.det) {      // the chunk owners apply the optimiser to the levels they own alone (see hashgrid_bwd_impl)
        fz = *fuse;
        fz.fuse_levels = 0u;
        for (int l = 0; l < g.L; ++l)
            if (level_fusable(g, plan, l)) fz.fuse_levels |= 1u << l;
    }
    if (fused_levels_out) *fused_levels_out = fz.fuse_levels;
    if (g.F == 1) {
        hipLaunchKernelGGL(scatter_bin_dir_kernel<1>, bgrid, dim3(1024), 0, as_stream(stream), xyz, gdx, dout_dx, g, plan, counters, recs, dtable, n, no_count);
        hipLaunchKernelGGL((scatter_bin_kernel<1, 1024>), bgrid, dim3(1024), 0, as_stream(stream), xyz, dout, (int64_t)0, g, plan, counters, recs, dtable, n, no_count, (uint2 *)nullptr, (uint4 *)nullptr, (int64_t)0, (uint32_t *)nullptr, (uint32_t *)nullptr, (int64_t)0);
        hipLaunchKernelGGL(scatter_accum_kernel<1>, agrid, dim3(kTiledThreads), lds, as_stream(stream), recs, counters, g, plan, dtable, fz, (const uint2 *)nullptr, (const float2 *)nullptr);
    } else {
        hipLaunchKernelGGL(scatter_bin_dir_kernel<2>, bgrid, dim3(1024), 0, as_stream(stream), xyz, gdx, dout_dx, g, plan, counters, recs, dtable, n, no_count);
        hipLaunchKernelGGL((scatter_bin_kernel<2, 1024>), bgrid, dim3(1024), 0, as_stream(stream), xyz, dout, (int64_t)0, g, plan, counters, recs, dtable, n, no_count, (uint2 *)nullptr, (uint4 *)nullptr, (int64_t)0, (uint32_t *)nullptr, (uint32_t *)nullptr, (int64_t)0);
        hipLaunchKernelGGL(scatter_accum_kernel<2>, agrid, dim3(kTiledThreads), lds, as_stream(stream), recs, counters, g, plan, dtable, fz, (const uint2 *)nullptr, (const float2 *)nullptr);
    }
    return check_launch("hashgrid_bwd_first_second");
}

ARCN_EXPORT int arcn_hashgrid_bwd_first_second(const float *xyz, const float *dout, const float *gdx, const float *dout_dx,
                                               const arcn_hashgrid_desc *desc_host, float *dtable, float *workspace, int64_t workspace_floats,
                                               int64_t n, void *stream) {
    return hashgrid_bwd_first_second_impl(xyz, dout, gdx, dout_dx, desc_host, dtable, workspace, workspace_floats, n, stream, nullptr, nullptr);
}

ARCN_EXPORT int arcn_hashgrid_bwd_lm(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host,
                                     float *dtable, float *workspace, int64_t workspace_floats, int64_t n, const int32_t *n_ptr,
                                     void *stream) {
    if (dout_stride < n) return einval("hashgrid_bwd_lm: level stride smaller than n");
    if (!workspace) return einval("hashgrid_bwd_lm: workspace required");
    return hashgrid_bwd_impl(xyz, nullptr, dout_lm, dout_stride, desc_host, dtable, nullptr, workspace, workspace_floats, n, n_ptr, stream);
}

ARCN_EXPORT int arcn_hashgrid_bwd_lm_levels(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host,
                                            float *dtable, float *workspace, int64_t workspace_floats, int64_t n, const int32_t *n_ptr,
                                            uint32_t level_mask, int counters_clear, void *stream) {
    if (dout_stride < n) return einval("hashgrid_bwd_lm_levels: level stride smaller than n");
    if (!workspace) return einval("hashgrid_bwd_lm_levels: workspace required");
    if (level_mask == 0u) return ARCN_OK;
    return hashgrid_bwd_impl(xyz, nullptr, dout_lm, dout_stride, desc_host, dtable, nullptr, workspace, workspace_floats, n, n_ptr, stream,
                             nullptr, nullptr, counters_clear != 0, level_mask);
}

ARCN_EXPORT int64_t arcn_hashgrid_bwd_fusable_levels(const arcn_hashgrid_desc *desc_host, int64_t n) {
    if (!desc_host || n <= 0) return 0;
    GridParams g;
    if (build_params(desc_host, g) || g.F > 2) return 0;
    BinPlan plan;
    if (build_bin_plan(g, n, plan) || plan.det) return 0;
    int64_t mask = 0;
    for (int l = 0; l < g.L; ++l)
        if (level_fusable(g, plan, l)) mask |= (int64_t)1 << l;
    return mask;
}

static int make_adam_fuse(AdamFuse &fz, const char *who, const float *workspace, float *table, float *exp_avg, float *exp_avg_sq, float lr, float beta1,
                          float beta2, float eps, float weight_decay, float ema_decay, float grad_scale, int step, int ema_step,
                          uint32_t *fused_levels_host) {
    static thread_local char msg[160];
    if (!workspace || !table || !exp_avg || !exp_avg_sq || !fused_levels_host || step < 1) {
        snprintf(msg, sizeof(msg), "%s: missing / invalid argument", who);
        return einval(msg);
    }
    if ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) {
        snprintf(msg, sizeof(msg), "%s: parameter and moment buffers must be 16-byte aligned", who);
        return einval(msg);
    }
    const bool ema = ema_decay >= 0.f;
    if (ema && ema_step < 1) {
        snprintf(msg, sizeof(msg), "%s: ema_step is 1-based", who);
        return einval(msg);
    }
    fz.param = table; fz.m = exp_avg; fz.v = exp_avg_sq;
    fz.h = make_adam_hyper(lr, beta1, beta2, eps, weight_decay, ema ? ema_decay : 0.f, grad_scale, step, ema_step, ema);
    fz.ema_in_param = ema ? 1 : 0;
    *fused_levels_host = 0u;
    return ARCN_OK;
}

ARCN_EXPORT int arcn_hashgrid_bwd_lm_adam(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host,
                                          float *dtable, float *table, float *exp_avg, float *exp_avg_sq, float lr, float beta1, float beta2,
                                          float eps, float weight_decay, float ema_decay, float grad_scale, int step, int ema_step,
                                          float *workspace, int64_t workspace_floats, int counters_clear, int64_t n, const int32_t *n_ptr,
                                          uint32_t *fused_levels_host, void *stream) {
    if (dout_stride != 0 && dout_stride < n) return einval("hashgrid_bwd_lm_adam: level stride smaller than n");
    AdamFuse fz{};
    int rc = make_adam_fuse(fz, "hashgrid_bwd_lm_adam", workspace, table, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, ema_decay, grad_scale,
                            step, ema_step, fused_levels_host);
    if (rc) return rc;
    return hashgrid_bwd_impl(xyz, nullptr, dout_lm, dout_stride, desc_host, dtable, nullptr, workspace, workspace_floats, n, n_ptr, stream, &fz,
                             fused_levels_host, counters_clear != 0);
}

ARCN_EXPORT int64_t arcn_hashgrid_plan_workspace_floats(const arcn_hashgrid_desc *desc_host, int64_t n) {
    if (!desc_host || n <= 0) return 0;
    GridParams g;
    if (build_params(desc_host, g) || g.F > 2) return 0;
    BinPlan plan;
    if (build_bin_plan(g, n, plan) || plan.n_recs >= (int64_t)kPosOvf) return 0;
    return plan_ws_floats_for(plan, g, n);
}

/* The position-only half of the binned scatter for the samples xyz (n slots, *n_ptr of them in use), into plan_ws: to be followed - any time
 * later, on any stream ordered behind this one - by ONE arcn_hashgrid_bwd_lm_planned / _adam_planned on the same xyz, n and n_ptr. */
ARCN_EXPORT int arcn_hashgrid_bwd_plan(const float *xyz, const arcn_hashgrid_desc *desc_host, float *plan_ws, int64_t plan_ws_floats, int64_t n,
                                       const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!xyz || !plan_ws) return einval("hashgrid_bwd_plan: missing argument");
    GridParams g;
    int rc = build_params(desc_host, g);
    if (rc) return rc;
    if (g.F > 2) return einval("hashgrid_bwd_plan: n_feat 1 or 2");
    BinPlan plan;
    rc = build_bin_plan(g, n, plan);
    if (rc) return rc;
    if (plan.n_recs >= (int64_t)kPosOvf) return einval("hashgrid_bwd_plan: too many record slots for 32-bit slot indices");
    if (plan_ws_floats < plan_ws_floats_for(plan, g, n)) return einval("hashgrid_bwd_plan: workspace smaller than arcn_hashgrid_plan_workspace_floats(desc, n)");
    uint32_t *counters = reinterpret_cast<uint32_t *>(plan_ws);
    uint2 *ridx = reinterpret_cast<uint2 *>(plan_ws + plan_ridx_first(plan));
    uint4 *pos4 = reinterpret_cast<uint4 *>(plan_ws + plan_pos_first(plan));
    hipError_t e = hipMemsetAsync(counters, 0, sizeof(uint32_t) * (size_t)bin_counter_floats(plan), as_stream(stream));
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
    int64_t bx = ceil_div<int64_t>(n, 1024);
    if (bx > 32) bx = 32;
    dim3 bgrid((unsigned)bx, (unsigned)g.L);
    const float *no_dout = nullptr;
    uint4 *no_recs = nullptr;
    float *no_dtable = nullptr;
    uint32_t *gslot = reinterpret_cast<uint32_t *>(plan_ws + plan_gslot_first(plan, g, n));
    uint32_t *tcount = reinterpret_cast<uint32_t *>(plan_ws + plan_count_first(plan, g, n));
    if (g.F == 1) hipLaunchKernelGGL((scatter_bin_kernel<1, 1024, true>), bgrid, dim3(1024), 0, as_stream(stream), xyz, no_dout, (int64_t)0, g, plan, counters, no_recs, no_dtable, n, n_ptr, ridx, pos4, n, gslot, tcount, plan_tiles(n));
    else hipLaunchKernelGGL((scatter_bin_kernel<2, 1024, true>), bgrid, dim3(1024), 0, as_stream(stream), xyz, no_dout, (int64_t)0, g, plan, counters, no_recs, no_dtable, n, n_ptr, ridx, pos4, n, gslot, tcount, plan_tiles(n));
    return check_launch("hashgrid_bwd_plan");
}

/* arcn_hashgrid_bwd_lm on a batch whose plan exists: fill pass + chunk owners */
ARCN_EXPORT int arcn_hashgrid_bwd_lm_planned(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host,
                                             float *dtable, float *plan_ws, int64_t plan_ws_floats, float *workspace, int64_t workspace_floats,
                                             int64_t n, const int32_t *n_ptr, void *stream) {
    if (dout_stride != 0 && dout_stride < n) return einval("hashgrid_bwd_lm_planned: level stride smaller than n");
    if (!workspace || !plan_ws) return einval("hashgrid_bwd_lm_planned: workspace and plan workspace required");
    return hashgrid_bwd_impl(xyz, nullptr, dout_lm, dout_stride, desc_host, dtable, nullptr, workspace, workspace_floats, n, n_ptr, stream, nullptr, nullptr,
                             false, 0xffffffffu, plan_ws, plan_ws_floats);
}

/* arcn_hashgrid_bwd_lm_adam on a batch whose plan exists */
ARCN_EXPORT int arcn_hashgrid_bwd_lm_adam_planned(const float *xyz, const float *dout_lm, int64_t dout_stride, const arcn_hashgrid_desc *desc_host,
                                                  float *dtable, float *table, float *exp_avg, float *exp_avg_sq, float lr, float beta1, float beta2,
                                                  float eps, float weight_decay, float ema_decay, float grad_scale, int step, int ema_step,
                                                  float *plan_ws, int64_t plan_ws_floats, float *workspace, int64_t workspace_floats, int64_t n,
                                                  const int32_t *n_ptr, uint32_t *fused_levels_host, void *stream) {
    if (dout_stride != 0 && dout_stride < n) return einval("hashgrid_bwd_lm_adam_planned: level stride smaller than n");
    if (!plan_ws) return einval("hashgrid_bwd_lm_adam_planned: plan workspace required");
    AdamFuse fz{};
    int rc = make_adam_fuse(fz, "hashgrid_bwd_lm_adam_planned", workspace, table, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, ema_decay,
                            grad_scale, step, ema_step, fused_levels_host);
    if (rc) return rc;
    return hashgrid_bwd_impl(xyz, nullptr, dout_lm, dout_stride, desc_host, dtable, nullptr, workspace, workspace_floats, n, n_ptr, stream, &fz,
                             fused_levels_host, false, 0xffffffffu, plan_ws, plan_ws_floats);
}

ARCN_EXPORT int arcn_hashgrid_bwd_first_second_adam(const float *xyz, const float *dout, const float *gdx, const float *dout_dx,
                                                    const arcn_hashgrid_desc *desc_host, float *dtable, float *table, float *exp_avg,
                                                    float *exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay,
                                                    float ema_decay, float grad_scale, int step, int ema_step, float *workspace,
                                                    int64_t workspace_floats, int64_t n, uint32_t *fused_levels_host, void *stream) {
    AdamFuse fz{};
    int rc = make_adam_fuse(fz, "hashgrid_bwd_first_second_adam", workspace, table, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, ema_decay,
                            grad_scale, step, ema_step, fused_levels_host);
    if (rc) return rc;
    return hashgrid_bwd_first_second_impl(xyz, dout, gdx, dout_dx, desc_host, dtable, workspace, workspace_floats, n, stream, &fz, fused_levels_host);
}

ARCN_EXPORT int64_t arcn_hashgrid_bwd_counter_words(const arcn_hashgrid_desc *desc_host, int64_t n) {
    if (!desc_host || n <= 0) return 0;
    GridParams g;
    if (build_params(desc_host, g) || g.F > 2) return 0;
    BinPlan plan;
    if (build_bin_plan(g, n, plan)) return 0;
    return bin_counter_floats(plan);
}

ARCN_EXPORT int64_t arcn_hashgrid_bwd_status_offset(const arcn_hashgrid_desc *desc_host, int64_t n) {
    if (!desc_host || n <= 0) return -1;
    GridParams g;
    if (build_params(desc_host, g)) return -1;
    BinPlan plan;
    if (build_bin_plan(g, n, plan)) return -1;
    return plan.aux_first;
}

ARCN_EXPORT int arcn_deterministic(void) {
    static const int det = [] { const char *e = getenv("ARCN_DETERMINISTIC"); return e ? atoi(e) : 0; }();
    return det ? 1 : 0;
}

ARCN_EXPORT int64_t arcn_hashgrid_bwd_workspace_floats(const arcn_hashgrid_desc *desc_host, int64_t n) {
    if (!desc_host || n <= 0) return 0;
    GridParams g;
    if (build_params(desc_host, g)) return 0;
    if (g.F > 2) return 0;  // n_feat 4 always takes the plain kernel
    BinPlan plan;
    if (build_bin_plan(g, n, plan)) return 0;
    return bin_counter_floats(plan) + plan.n_recs * 4;  // bin counters + 16-byte records
}

// Cost model of one level of the forward gather on one XCD, in arbitrary units (us per 2.6e5 samples measured on MI355X with the
// bench's ray batch, tools/exp_gather.py --levels): an instruction floor, the extra modulo work of the non-power-of-two levels, and
// a memory term that grows with the resolution (fewer samples per cell -> fewer lanes of a wave share a line).
static float fwd_level_cost(const LevelParams &lp) {
    // one-XCD time of the level minus the launch overhead, pair loads on (tools/exp_gather.py --levels with ARCN_GATHER_ONE_XCD=1)
    static const float res_pts[] = {64.f, 80.f, 111.f, 153.f, 212.f, 294.f, 406.f, 561.f, 776.f, 1072.f, 1482.f, 2047.f, 8192.f};
    static const float mem_pts[] = {0.0f, 1.0f, 2.9f, 6.4f, 10.5f, 16.9f, 24.4f, 32.6f, 38.0f, 40.9f, 42.6f, 42.5f, 43.0f};
    static const float floor_cost = 10.0f;
    static const float mod_cost = 3.0f;
    const float r = (float)lp.res;
    float mem = 0.f;
    const int np = (int)(sizeof(res_pts) / sizeof(res_pts[0]));
    if (r >= res_pts[np - 1]) mem = mem_pts[np - 1];
    else
        for (int i = 0; i + 1 < np; ++i)
            if (r >= res_pts[i] && r < res_pts[i + 1]) {
                mem = mem_pts[i] + (mem_pts[i + 1] - mem_pts[i]) * (r - res_pts[i]) / (res_pts[i + 1] - res_pts[i]);
                break;
            }
    return floor_cost + (lp.mask ? 0.f : mod_cost) + mem;
}

// Levels in descending cost are poured into the 8 XCD queues, each filled to 1/8 of the total; a level that does not fit is cut
// (by sample range) and continues on the next XCD.  Workgroups per segment follow its share of the XCD's cost.
static int build_fwd_plan(const GridParams &g, int64_t n, FwdPlan &plan, int &wg_per_xcd) {
    int order[ARCN_MAX_LEVELS];
    float cost[ARCN_MAX_LEVELS];
    float total = 0.f;
    for (int l = 0; l < g.L; ++l) {
        order[l] = l;
        cost[l] = fwd_level_cost(g.lv[l]);
        total += cost[l];
        plan.lowbits[l] = level_lowbits(g.lv[l]);
    }
    for (int l = g.L; l < ARCN_MAX_LEVELS; ++l) plan.lowbits[l] = 0;
    for (int i = 0; i < g.L; ++i)
        for (int k = i + 1; k < g.L; ++k)
            if (cost[order[k]] > cost[order[i]]) { int t = order[i]; order[i] = order[k]; order[k] = t; }
    const int64_t tiles = ceil_div<int64_t>(n, 256);
    // (one 256-sample tile per workgroup at the bench's batch: 2048 against 1024 resident-loop workgroups per XCD, 62.5 against 64.2 us alone)
    static const int max_wg = 2048;
    int64_t want = ceil_div<int64_t>(tiles * g.L, 8);    // one tile per workgroup when there is little work
    wg_per_xcd = (int)(want < max_wg ? want : max_wg);
    if (wg_per_xcd < 1) wg_per_xcd = 1;
    static const int one_xcd = [] { const char *e = getenv("ARCN_GATHER_ONE_XCD"); return e ? atoi(e) : 0; }();  // calibration aid
    const float share = one_xcd ? total * 1.01f : total / 8.f;
    for (int x = 0; x < 8; ++x) plan.n_seg[x] = 0;
    // the 8 most expensive levels (the fine ones: every one of them fills a 4 MiB L2 by itself) get an XCD each and are never cut -
    // two fine levels in one L2 evict each other (measured: +6 us on the XCDs that held a cut level); the cheaper levels, whose
    // touched rows are a small part of their table, are the filler that evens the queues out
    float load[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add_seg = [&](int x, int l, float f0, float f1, float c) -> int {
        if (plan.n_seg[x] >= kFwdSegs) return -1;
        FwdSeg &sg = plan.seg[x][plan.n_seg[x]++];
        sg.level = l;
        sg.f0 = (uint32_t)(f0 * 65536.f + 0.5f);
        sg.f1 = f1 >= 1.f ? 65536u : (uint32_t)(f1 * 65536.f + 0.5f);
        sg.wg0 = 0;
        sg.wg1 = (int32_t)(c * 1024.f);   // the cost, until the queues are complete
        load[x] += c;
        return 0;
    };
    const int whole = one_xcd ? 0 : (g.L < 8 ? g.L : 8);
    for (int i = 0; i < whole; ++i)
        if (add_seg(i, order[i], 0.f, 1.f, cost[order[i]])) return einval("hashgrid_fwd_xcd: plan overflow");
    for (int i = whole; i < g.L; ++i) {
        const int l = order[i];
        float left = cost[l], f0 = 0.f;
        int guard = 0;
        while (left > 1e-6f * total && guard++ < 64) {
            int x = 0;   // the least loaded queue that can still take a segment
            float best = 1e30f;
            for (int q = 0; q < (one_xcd ? 1 : 8); ++q) {
                static const float odd_bias = 1.0f;
                const float eff = load[q] / ((q & 1) ? odd_bias : 2.0f - odd_bias);
                if (plan.n_seg[q] < kFwdSegs && eff < best) { best = eff; x = q; }
            }
            if (best > 1e29f) return einval("hashgrid_fwd_xcd: plan overflow");
            // the odd XCDs finish a given queue ~4 us later than the even ones, for any plan (round 2): with one tile per workgroup a smaller
            // share evens them out - 0.92: 59.9 us alone against 62.5 (0.617 of 8 TB/s), 78.5 against 81.4 us in the step; 0.96 is WORSE
            // (64.7 us: where the cuts of the filler levels land matters), 0.90 - 0.94 all within 1 us
            static const float odd_scale = 0.92f;
            const float room = share * ((x & 1) ? odd_scale : 2.0f - odd_scale) - load[x];
            const float take = (room <= 1e-4f * total || left <= room) ? left : room;
            const float f1 = (take >= left) ? 1.f : f0 + (1.f - f0) * (take / left);
            if (add_seg(x, l, f0, f1, take)) return einval("hashgrid_fwd_xcd: plan overflow");
            f0 = f1;
            left -= take;
        }
    }
    // workgroups of an XCD's queue: one per segment, the rest in proportion to the segments' cost (small launches: a queue is never
    // shorter than its segment list)
    int max_seg = 1;
    for (int q = 0; q < 8; ++q) max_seg = plan.n_seg[q] > max_seg ? plan.n_seg[q] : max_seg;
    if (wg_per_xcd < max_seg) wg_per_xcd = max_seg;
    for (int q = 0; q < 8; ++q) {
        const int ns = plan.n_seg[q];
        if (ns == 0) continue;
        float sum = 0.f;
        int big = 0;
        for (int i = 0; i < ns; ++i) {
            sum += (float)plan.seg[q][i].wg1;
            if (plan.seg[q][i].wg1 > plan.seg[q][big].wg1) big = i;
        }
        const int extra = wg_per_xcd - ns;
        int cnt[kFwdSegs], used = 0;
        for (int i = 0; i < ns; ++i) {
            cnt[i] = 1 + (int)((float)extra * ((float)plan.seg[q][i].wg1 / (sum > 0.f ? sum : 1.f)));
            used += cnt[i];
        }
        cnt[big] += wg_per_xcd - used;   // rounding remainder (>= 0) to the most expensive segment
        int pos = 0;
        for (int i = 0; i < ns; ++i) {
            plan.seg[q][i].wg0 = pos;
            plan.seg[q][i].wg1 = pos + cnt[i];
            pos += cnt[i];
        }
    }
    return ARCN_OK;
}

// XCD-affine forward (n_feat 1 or 2).  level_major = 0: out (n, L*F) row-major like arcn_hashgrid_fwd;
// level_major = 1: out[(l * n_cap + s) * F + f].  corners (optional): the gathered rows in level-major quads (hashgrid_fwd_bal_kernel).
static int hashgrid_fwd_xcd_impl(const char *who, const float *xyz, const float *table, const arcn_hashgrid_desc *desc_host, float *out,
                                 int level_major, float *corners, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    auto bad = [&](const char *what) {
        char msg[160];
        snprintf(msg, sizeof(msg), "%s: %s", who, what);
        return einval(msg);
    };
    if (!xyz || !table || !out || n_cap < n) return bad("missing/invalid argument");
    GridParams g;
    int rc = build_params(desc_host, g);
    if (rc) return rc;
    if (g.F > 2) return bad("n_feat 1 or 2");
    FwdPlan plan;
    int wg_per_xcd = 0;
    rc = build_fwd_plan(g, n, plan, wg_per_xcd);
    if (rc) return rc;
    {   // calibration aid (tools/pmc_gather.sh): only one XCD's queue runs
        static const int only = [] { const char *e = getenv("ARCN_GATHER_ONLY_XCD"); return e ? atoi(e) : -1; }();
        if (only >= 0) for (int x = 0; x < 8; ++x) if (x != only) plan.n_seg[x] = 0;
    }
    dim3 grid((unsigned)(8 * wg_per_xcd));
#define ARCN_BAL(F_, LM_, P_, C_) hipLaunchKernelGGL((hashgrid_fwd_bal_kernel<F_, LM_, P_, C_>), grid, dim3(256), 0, as_stream(stream), xyz, table, g, plan, out, n_cap, n, n_ptr, corners)
    if (corners) {
        if (g.F == 1) { if (level_major) ARCN_BAL(1, true, false, true); else ARCN_BAL(1, false, false, true); }
        else { if (level_major) ARCN_BAL(2, true, true, true); else ARCN_BAL(2, false, true, true); }
    } else {
        if (g.F == 1) { if (level_major) ARCN_BAL(1, true, false, false); else ARCN_BAL(1, false, false, false); }
        else { if (level_major) ARCN_BAL(2, true, true, false); else ARCN_BAL(2, false, true, false); }
    }
#undef ARCN_BAL
    return check_launch(who);
}

ARCN_EXPORT int arcn_hashgrid_fwd_xcd(const float *xyz, const float *table, const arcn_hashgrid_desc *desc_host, float *out,
                                      int level_major, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    return hashgrid_fwd_xcd_impl("hashgrid_fwd_xcd", xyz, table, desc_host, out, level_major, nullptr, n_cap, n, n_ptr, stream);
}

/* arcn_hashgrid_fwd_xcd that also keeps the gathered rows (level-major quads, 8 F n_cap L floats) */
ARCN_EXPORT int arcn_hashgrid_fwd_corners(const float *xyz, const float *table, const arcn_hashgrid_desc *desc_host, float *out, int level_major,
                                          float *corners, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n > 0 && !corners) return einval("hashgrid_fwd_corners: missing argument");
    return hashgrid_fwd_xcd_impl("hashgrid_fwd_corners", xyz, table, desc_host, out, level_major, corners, n_cap, n, n_ptr, stream);
}

/* dxyz (n, 3) = d <dout, enc(x)> / d x (what arcn_hashgrid_bwd adds to a cleared dxyz) from the forward's corners instead of the table: the
 * same arithmetic in the same order on the same values - bit-identical - as a streaming read */
ARCN_EXPORT int arcn_hashgrid_dxyz_corners(const float *xyz, const float *corners, const float *dout, const arcn_hashgrid_desc *desc_host,
                                           float *dxyz, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!xyz || !corners || !dout || !dxyz || n_cap < n) return einval("hashgrid_dxyz_corners: missing/invalid argument");
    GridParams g;
    int rc = build_params(desc_host, g);
    if (rc) return rc;
    if (g.F > 2) return einval("hashgrid_dxyz_corners: n_feat 1 or 2");
    dim3 grid((unsigned)ceil_div<int64_t>(n, 256));
    if (g.F == 1) hipLaunchKernelGGL(hashgrid_dxyz_corners_kernel<1>, grid, dim3(256), 0, as_stream(stream), xyz, corners, dout, g, dxyz, n_cap, n, n_ptr);
    else hipLaunchKernelGGL(hashgrid_dxyz_corners_kernel<2>, grid, dim3(256), 0, as_stream(stream), xyz, corners, dout, g, dxyz, n_cap, n, n_ptr);
    return check_launch("hashgrid_dxyz_corners");
}

/* ddout (n, L F) = d <gdx, J(x; table)^T dout> / d dout (the ddout of arcn_hashgrid_bwd_bwd) from the forward's corners: bit-identical,
 * streaming */
ARCN_EXPORT int arcn_hashgrid_ddout_corners(const float *xyz, const float *gdx, const float *corners, const arcn_hashgrid_desc *desc_host,
                                            float *ddout, int64_t n_cap, int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!xyz || !gdx || !corners || !ddout || n_cap < n) return einval("hashgrid_ddout_corners: missing/invalid argument");
    GridParams g;
    int rc = build_params(desc_host, g);
    if (rc) return rc;
    if (g.F > 2) return einval("hashgrid_ddout_corners: n_feat 1 or 2");
    dim3 grid((unsigned)ceil_div<int64_t>(n, 256));
    if (g.F == 1) hipLaunchKernelGGL(hashgrid_ddout_corners_kernel<1>, grid, dim3(256), 0, as_stream(stream), xyz, gdx, corners, g, ddout, n_cap, n, n_ptr);
    else hipLaunchKernelGGL(hashgrid_ddout_corners_kernel<2>, grid, dim3(256), 0, as_stream(stream), xyz, gdx, corners, g, ddout, n_cap, n, n_ptr);
    return check_launch("hashgrid_ddout_corners");
}

/* level-major features lm[(l * n_cap + s) * F + f] (what arcn_hashgrid_fwd_xcd writes with level_major = 1) -> rows[s * ld + l * F + f] */
ARCN_EXPORT int arcn_hashgrid_lm_to_rows(const float *lm, int n_levels, int n_feat, int64_t n_cap, float *rows, int64_t ld, int64_t n,
                                         const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!lm || !rows || n_cap < n || n_levels < 1 || n_levels > ARCN_MAX_LEVELS || (n_feat != 1 && n_feat != 2) || ld < n_levels * n_feat)
        return einval("hashgrid_lm_to_rows: missing/invalid argument");
    dim3 grid((unsigned)ceil_div<int64_t>(n, 256));
    const size_t lds = (size_t)256 * (n_levels * n_feat + 1) * sizeof(float);
    if (n_feat == 1) hipLaunchKernelGGL(lm_to_rows_kernel<1>, grid, dim3(256), lds, as_stream(stream), lm, n_levels, n_cap, rows, ld, n, n_ptr);
    else hipLaunchKernelGGL(lm_to_rows_kernel<2>, grid, dim3(256), lds, as_stream(stream), lm, n_levels, n_cap, rows, ld, n, n_ptr);
    return check_launch("hashgrid_lm_to_rows");
}
