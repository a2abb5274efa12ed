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

namespace arcn {

struct LevelParams {
    int32_t res;
    uint32_t size;     // rows of this level
    uint32_t mask;     // size-1 if size is a power of two else 0
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

__device__ __forceinline__ Cell locate(const float p[3], const GridParams &g, int res) {
    Cell cell;
    float v[3], vs[3];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        vs[k] = (g.mx[k] - g.mn[k]) / (float)res;
        v[k] = (p[k] - g.mn[k]) / vs[k];
        if (!(v[k] >= 0.f) || !(v[k] < (float)res)) ok = false;
    }
    cell.valid = ok;
    if (!ok) return cell;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float cf = floorf(v[k]);
        cell.c[k] = (uint32_t)cf;
        float a = cf * vs[k];
        float g0 = a + g.mn[0];  // start_point[0] for every axis (volume.py:515)
        float ww = (p[k] - g0) / vs[k];
        cell.w[k] = ww < 0.0f ? 0.0f : (ww > 1.0f ? 1.0f : ww);
        cell.dw[k] = (ww >= 0.0f && ww <= 1.0f) ? 1.0f / vs[k] : 0.0f;
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
    const Cell cell = locate(p, g, lp.res);
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
    } else if (hash_idx) {
#pragma unroll
        for (int q = 0; q < 8; ++q) hash_idx[gid * 8 + q] = -1;
    }
    float *o = out + gid * F;
    if (F == 2) *reinterpret_cast<float2 *>(o) = make_float2(acc[0], acc[1 % F]);
    else if (F == 4) *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1 % F], acc[2 % F], acc[3 % F]);
    else {
#pragma unroll
        for (int f = 0; f < F; ++f) o[f] = acc[f];
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
    const Cell cell = locate(p, g, lp.res);
    if (!cell.valid) return;
    float go[F];
#pragma unroll
    for (int f = 0; f < F; ++f) go[f] = dout[gid * F + f];
    float gx[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
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
    if (dxyz) {
#pragma unroll
        for (int k = 0; k < 3; ++k) unsafeAtomicAdd(&dxyz[3 * s + k], gx[k]);
    }
}

// ---- backward, owner-computes scatter through LDS -------------------------------------------------------------
// Measured on MI355X: the chip retires only ~18-21 G scattered fp32 global atomics per second (they are served at the
// memory side, not in the issuing XCD's L2), so the 128 atomics/sample of the plain kernel above cost 4 ms per 2^18
// samples — 70 % of a training step — no matter how the work is arranged.  This kernel issues NO global atomics on the
// fine levels: every workgroup OWNS a slice of <= kChunkRows rows of one level's dtable, held in LDS (128 KiB of the
// CU's 160 KiB).  It scans the samples, recomputes each sample's cell and the 8 hashed rows for its level, and
// accumulates the corners that fall into its slice with LDS atomics (ds_add_f32); at the end the slice is written back
// with plain coalesced stores.  A corner hashes into a given slice with probability 1/n_chunks, so the LDS traffic is
// tiny and the kernel is VALU-bound on the (cheap, integer) hashing; the price is that each of a level's n_chunks
// workgroups re-derives every sample's cell.  Small (coarse) levels have few chunks but heavy same-row traffic: they are
// additionally split over disjoint sample ranges, each split accumulating privately in LDS and merging its non-zero
// rows with a few global atomics (n_splits * rows, negligible).
// The cell index uses a reciprocal multiply with an exact-division fallback whenever the result is within 2e-3 of an
// integer (error bound of the fast path 4e-4), so rows stay bit-identical to the forward/reference.
constexpr int kChunkRows = 16384;  // x F(=2) floats = 128 KiB
constexpr int kTiledThreads = 1024;

struct TiledPlan {
    int32_t item_first[ARCN_MAX_LEVELS + 1];  // prefix of work items per level
    int32_t n_chunks[ARCN_MAX_LEVELS];
    int32_t n_splits[ARCN_MAX_LEVELS];
    int32_t chunk_rows[ARCN_MAX_LEVELS];
};

// Prep pass of the owner-computes scatter: one 32-byte record per (level, sample), level-major so the owners of level l
// stream it contiguously.  Everything that is identical for the ~32 owner workgroups of a level is done ONCE here: the
// exact cell (fp32 divide, bit-identical to the forward), the trilinear weights and the incoming gradient.
//   word0 = cx | cy << 16, word1 = cz | valid << 16, w[3], g[2], pad
struct __attribute__((aligned(16))) ScatterRec {
    uint32_t cxy, czv;
    float w0, w1, w2, g0, g1, pad;
};

template <int F>
__global__ void __launch_bounds__(256)
scatter_prep_kernel(const float *__restrict__ xyz, const float *__restrict__ dout, GridParams g, ScatterRec *__restrict__ recs,
                    int64_t n_cap, int64_t n, const int32_t *n_ptr) {
    const int64_t cnt = dev_count(n, n_ptr);
    const int l = blockIdx.y;
    const LevelParams lp = g.lv[l];
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cnt; s += (int64_t)gridDim.x * blockDim.x) {
        const float p[3] = {xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2]};
        const Cell cell = locate(p, g, lp.res);
        ScatterRec r;
        r.cxy = cell.valid ? (cell.c[0] | (cell.c[1] << 16)) : 0u;
        r.czv = cell.valid ? (cell.c[2] | (1u << 16)) : 0u;
        r.w0 = cell.valid ? cell.w[0] : 0.f;
        r.w1 = cell.valid ? cell.w[1] : 0.f;
        r.w2 = cell.valid ? cell.w[2] : 0.f;
        r.g0 = dout[(s * g.L + l) * F];
        r.g1 = F > 1 ? dout[(s * g.L + l) * F + (F > 1 ? 1 : 0)] : 0.f;
        r.pad = 0.f;
        recs[(int64_t)l * n_cap + s] = r;
    }
}

template <int F>
__global__ void __launch_bounds__(kTiledThreads)
hashgrid_bwd_tiled_kernel(const ScatterRec *__restrict__ recs, int64_t n_cap, GridParams g, TiledPlan plan,
                          float *__restrict__ dtable, int64_t n, const int32_t *n_ptr) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int64_t cnt = dev_count(n, n_ptr);
    // work items are laid out finest level first: the unsplit fine-level owners (all the same length) fill the first
    // round of CUs, the shorter split items of the coarse levels pack into the second one
    int k = 0;
    while (k + 1 < g.L && (int)blockIdx.x >= plan.item_first[k + 1]) ++k;
    const int item = blockIdx.x - plan.item_first[k];
    const int l = g.L - 1 - k;
    const int chunk = item / plan.n_splits[l], split = item % plan.n_splits[l];
    const LevelParams lp = g.lv[l];
    const uint32_t row_lo = (uint32_t)chunk * (uint32_t)plan.chunk_rows[l];
    const uint32_t row_hi_raw = row_lo + (uint32_t)plan.chunk_rows[l];
    const uint32_t row_hi = row_hi_raw < lp.size ? row_hi_raw : lp.size;
    const int n_rows = (int)(row_hi - row_lo);
    const uint32_t n_rows_u = (uint32_t)n_rows;
    for (int i = threadIdx.x; i < n_rows * F; i += kTiledThreads) acc[i] = 0.f;
    __syncthreads();
    const int64_t per = (cnt + plan.n_splits[l] - 1) / plan.n_splits[l];
    const int64_t s_lo = per * split, s_hi = (s_lo + per < cnt) ? s_lo + per : cnt;
    // Inner loop: VALU-issue bound, so it only unpacks, hashes (two xors + mask per corner on power-of-two levels) and
    // visits the few in-slice corners.  LDS float atomics cost ~3 cycles per ACTIVE lane on gfx950 (measured; integer ones
    // are ~28x cheaper), hence the set-bit loop instead of 16 predicated full-wave ds_add_f32.
    const ScatterRec *pr = recs + (int64_t)l * n_cap + s_lo + (int64_t)threadIdx.x;
    int remaining = (s_lo + (int64_t)threadIdx.x < s_hi)
                        ? (int)((s_hi - s_lo - (int64_t)threadIdx.x + kTiledThreads - 1) / kTiledThreads) : 0;
    ScatterRec nxt = {};
    if (remaining > 0) nxt = *pr;
    for (; remaining > 0; --remaining) {
        const ScatterRec r = nxt;
        pr += kTiledThreads;
        if (remaining > 1) nxt = *pr;  // register prefetch of the next record
        const uint32_t cx = r.cxy & 0xffffu, cy = r.cxy >> 16, cz = r.czv & 0xffffu;
        uint32_t hit = 0;
        uint32_t hx0 = 0, hx1 = 0, hy0 = 0, hy1 = 0, hz0 = 0, hz1 = 0;
        if (lp.mask) {
            hx0 = cx; hx1 = cx + 1u;
            hy0 = cy * 2654435761u; hy1 = hy0 + 2654435761u;
            hz0 = cz * 805459861u; hz1 = hz0 + 805459861u;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t rr = ((((q >> 1) & 1) ? hx1 : hx0) ^ ((q & 1) ? hy1 : hy0) ^ ((q >> 2) ? hz1 : hz0)) & lp.mask;
                hit |= ((rr - row_lo) < n_rows_u ? 1u : 0u) << q;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t rr = hash_row(cx + ((q >> 1) & 1), cy + (q & 1), cz + (q >> 2), lp);
                hit |= ((rr - row_lo) < n_rows_u ? 1u : 0u) << q;
            }
        }
        if (!(r.czv >> 16)) hit = 0;
        while (hit) {
            const int q = __builtin_ctz(hit);
            hit &= hit - 1;
            const uint32_t ox = (q >> 1) & 1, oy = q & 1, oz = q >> 2;
            uint32_t rr;
            if (lp.mask) rr = ((ox ? hx1 : hx0) ^ (oy ? hy1 : hy0) ^ (oz ? hz1 : hz0)) & lp.mask;
            else rr = hash_row(cx + ox, cy + oy, cz + oz, lp);
            const float wt = ((ox ? r.w0 : 1.0f - r.w0) * (oy ? r.w1 : 1.0f - r.w1)) * (oz ? r.w2 : 1.0f - r.w2);
            float *dsta = acc + (rr - row_lo) * F;
            __hip_atomic_fetch_add(dsta, r.g0 * wt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (F > 1) __hip_atomic_fetch_add(dsta + (F > 1 ? 1 : 0), r.g1 * wt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    float *dst = dtable + ((int64_t)lp.offset + row_lo) * F;
    if (plan.n_splits[l] == 1) {
        // exclusive owner: plain coalesced stores (+= so that callers accumulating over several launches stay correct)
        for (int i = threadIdx.x; i < n_rows * F; i += kTiledThreads) dst[i] += acc[i];
    } else {
        for (int i = threadIdx.x; i < n_rows * F; i += kTiledThreads) {
            const float v = acc[i];
            if (v != 0.f) unsafeAtomicAdd(dst + i, v);
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
        lp.size = (uint32_t)size;
        lp.mask = ((size & (size - 1)) == 0) ? (uint32_t)(size - 1) : 0u;
        lp.pad = 0;
        lp.inv_size = 1.0 / (double)size;
        lp.offset = d->offsets[l];
    }
    return ARCN_OK;
}

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

ARCN_EXPORT int arcn_hashgrid_bwd(const float *xyz, const float *table, const float *dout,
                                  const arcn_hashgrid_desc *desc_host, float *dtable, float *dxyz, float *workspace,
                                  int64_t n, const int32_t *n_ptr, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!xyz || !dout || (!dtable && !dxyz) || (dxyz && !table)) return einval("hashgrid_bwd: missing argument");
    GridParams g;
    int rc = build_params(desc_host, g);
    if (rc) return rc;
    if (workspace && dtable && !dxyz && g.F <= 2) {
        // owner-computes scatter through LDS (workspace only selects the path; nothing is stored in it)
        TiledPlan plan;
        int items = 0;
        for (int k = 0; k < g.L; ++k) {
            const int l = g.L - 1 - k;  // item_first is indexed by dispatch position k, the per-level fields by level
            const int64_t size = g.lv[l].size;
            const int rows_cap = kChunkRows * 2 / g.F;  // 128 KiB of floats
            int nc = (int)((size + rows_cap - 1) / rows_cap);
            int cr = (int)((size + nc - 1) / nc);
            int ns = 32 / nc;  // about 32 workgroups per level
            if (!g.lv[l].mask) ns *= 2;  // non-power-of-two levels pay an exact 64-bit modulo per corner: shorter items
            if (ns < 1) ns = 1;
            // never split so finely that a split has fewer than 4096 samples
            while (ns > 1 && n / ns < 4096) ns >>= 1;
            plan.item_first[k] = items;
            plan.n_chunks[l] = nc;
            plan.n_splits[l] = ns;
            plan.chunk_rows[l] = cr;
            items += nc * ns;
        }
        plan.item_first[g.L] = items;
        for (int l = g.L + 1; l <= ARCN_MAX_LEVELS; ++l) plan.item_first[l] = items;
        const size_t lds = sizeof(float) * (size_t)kChunkRows * 2;
        hipError_t e = g.F == 1
            ? hipFuncSetAttribute(reinterpret_cast<const void *>(hashgrid_bwd_tiled_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
            : hipFuncSetAttribute(reinterpret_cast<const void *>(hashgrid_bwd_tiled_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return ARCN_ELAUNCH; }
        dim3 pgrid((unsigned)items);
        ScatterRec *recs = reinterpret_cast<ScatterRec *>(workspace);
        int64_t tb = ceil_div<int64_t>(n, 256);
        if (tb > 1024) tb = 1024;
        dim3 tgrid((unsigned)tb, (unsigned)g.L);
        if (g.F == 1) {
            hipLaunchKernelGGL(scatter_prep_kernel<1>, tgrid, dim3(256), 0, as_stream(stream), xyz, dout, g, recs, n, n, n_ptr);
            hipLaunchKernelGGL(hashgrid_bwd_tiled_kernel<1>, pgrid, dim3(kTiledThreads), lds, as_stream(stream), recs, n, g, plan, dtable, n, n_ptr);
        } else {
            hipLaunchKernelGGL(scatter_prep_kernel<2>, tgrid, dim3(256), 0, as_stream(stream), xyz, dout, g, recs, n, n, n_ptr);
            hipLaunchKernelGGL(hashgrid_bwd_tiled_kernel<2>, pgrid, dim3(kTiledThreads), lds, as_stream(stream), recs, n, g, plan, dtable, n, n_ptr);
        }
        return check_launch("hashgrid_bwd_tiled");
    }
    dim3 grid((unsigned)ceil_div<int64_t>(n * g.L, 256));
    switch (g.F) {
    case 1: hipLaunchKernelGGL(hashgrid_bwd_kernel<1>, grid, dim3(256), 0, as_stream(stream), xyz, table, dout, g, dtable, dxyz, n, n_ptr); break;
    case 2: hipLaunchKernelGGL(hashgrid_bwd_kernel<2>, grid, dim3(256), 0, as_stream(stream), xyz, table, dout, g, dtable, dxyz, n, n_ptr); break;
    default: hipLaunchKernelGGL(hashgrid_bwd_kernel<4>, grid, dim3(256), 0, as_stream(stream), xyz, table, dout, g, dtable, dxyz, n, n_ptr); break;
    }
    return check_launch("hashgrid_bwd");
}

ARCN_EXPORT int64_t arcn_hashgrid_bwd_workspace_floats(const arcn_hashgrid_desc *desc_host, int64_t n) {
    if (!desc_host || n <= 0) return 0;
    return n * (int64_t)desc_host->n_levels * 8;  // one 32-byte record per (level, sample)
}
