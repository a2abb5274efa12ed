// Occupancy density grid in Morton order + its packed bitfield: the `_bitfield_func` maintenance kernels for gfx950.
//
// Replaces arcnerf/ops/src/bitfield_func/bitfield_func_kernel.cu K6-K10 (K5, the sampler over this layout, lives in
// volume.hip next to K3: same marching loop).  All of it is HBM-bound integer / byte work over an n_grid^3 float grid
// (8 MiB at 128^3): one coalesced pass per kernel, nothing synchronises, every kernel runs on the caller's stream.
#include "common.hpp"
#include "morton.hpp"

namespace arcn {

// ---- K6 generate_grid_samples (bitfield_func_kernel.cu:141-181) -------------------------------------------------------
// Sample i probes up to 10 pseudo-random cells (uint32 wrap-around LCG on (i, ema_step)) and keeps the first one whose
// density exceeds `thresh` (or the 10th); the position is a jittered point of that cell in [0,1)^3.
__global__ void __launch_bounds__(256)
grid_samples_kernel(uint32_t n_elements, Pcg32 rng, uint32_t step, const float *__restrict__ grid_in,
                    float *__restrict__ positions, int32_t *__restrict__ indices, uint32_t n_grid, float thresh) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_elements) return;
    rng.advance((int64_t)(uint32_t)(i * 4u));
    const uint32_t n_cells = n_grid * n_grid * n_grid;
    const uint32_t base = (i + step * n_elements) * 56924617u + 96925573u;
    uint32_t idx = 0;
    for (uint32_t j = 0; j < 10; ++j) {
        idx = (base + j * 19349663u) % n_cells;
        if (grid_in[idx] > thresh) break;
    }
    const float x = (float)morton_compact(idx), y = (float)morton_compact(idx >> 1), z = (float)morton_compact(idx >> 2);
    const float r0 = rng.next_float(), r1 = rng.next_float(), r2 = rng.next_float();
    const float ng = (float)n_grid;
    positions[3 * (int64_t)i + 0] = (x + r0) / ng;
    positions[3 * (int64_t)i + 1] = (y + r1) / ng;
    positions[3 * (int64_t)i + 2] = (z + r2) / ng;
    indices[i] = (int32_t)idx;
}

// ---- K12 generate_grid_samples_multivol (multivol_func_kernel.cu:148-206) ---------------------------------------------
// K6 over the cascade: a level is drawn first (redrawn while it is the inner one when that is excluded), the cell probe adds the
// level's slot, and the jittered point is mapped to that level's volume (inner box scaled 2^level about its centre).
__global__ void __launch_bounds__(256)
grid_samples_multivol_kernel(uint32_t n_elements, const float *__restrict__ aabb, Pcg32 rng, uint32_t step,
                             const float *__restrict__ grid_in, float *__restrict__ positions, int32_t *__restrict__ indices,
                             uint32_t n_cascades, uint32_t n_grid, float thresh, int inclusive) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_elements) return;
    rng.advance((int64_t)(uint32_t)(i * 4u));
    uint32_t level = 0;
    if (inclusive) {
        level = (uint32_t)(rng.next_float() * (float)n_cascades) % n_cascades;
    } else {
        while (level == 0) level = (uint32_t)(rng.next_float() * (float)n_cascades) % n_cascades;
    }
    const uint32_t n_cells = n_grid * n_grid * n_grid;
    const uint32_t base = (i + step * n_elements) * 56924617u + 96925573u;
    const uint32_t slot = (inclusive ? level : level - 1) * n_cells;
    uint32_t idx = 0;
    for (uint32_t j = 0; j < 10; ++j) {
        idx = (base + j * 19349663u) % n_cells + slot;
        if (grid_in[idx] > thresh) break;
    }
    const uint32_t cell = idx % n_cells;
    const float c[3] = {(float)morton_compact(cell), (float)morton_compact(cell >> 1), (float)morton_compact(cell >> 2)};
    float r[3];
    r[0] = rng.next_float();
    r[1] = rng.next_float();
    r[2] = rng.next_float();
    const float scale = scalbnf(1.0f, (int)level), ng = (float)n_grid;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float center = (aabb[k] + aabb[3 + k]) / 2.0f;
        const float len = aabb[3 + k] - aabb[k];
        float p = (c[k] + r[k]) / ng;
        p = p - 0.5f;
        p = p * len;
        p = p * scale;
        positions[3 * (int64_t)i + k] = p + center;
    }
    indices[i] = (int32_t)idx;
}

// ---- K7 splat_grid_samples (bitfield_func_kernel.cu:215-229) ----------------------------------------------------------
// max over the samples of each cell; non-negative floats order like their bit patterns, so an integer atomic max is exact
__global__ void __launch_bounds__(256)
splat_kernel(uint32_t n, const int32_t *__restrict__ indices, const float *__restrict__ density, float *__restrict__ grid_tmp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicMax(reinterpret_cast<unsigned int *>(&grid_tmp[(uint32_t)indices[i]]), __float_as_uint(density[i]));
}

// ---- K8 ema_grid_samples_nerf (bitfield_func_kernel.cu:257-272) -------------------------------------------------------
__device__ __forceinline__ float ema_one(float prev, float importance, float decay) {
    const float dec = prev * decay;
    return (prev < 0.f) ? prev : fmaxf(dec, importance);
}

__global__ void __launch_bounds__(256)
ema_kernel(uint32_t n, float decay, float *__restrict__ grid, const float *__restrict__ grid_tmp) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;  // one float4 per lane
    const uint32_t i = q * 4u;
    if (i + 3 < n) {
        const float4 imp = *reinterpret_cast<const float4 *>(grid_tmp + i);
        float4 v = *reinterpret_cast<float4 *>(grid + i);
        v.x = ema_one(v.x, imp.x, decay);
        v.y = ema_one(v.y, imp.y, decay);
        v.z = ema_one(v.z, imp.z, decay);
        v.w = ema_one(v.w, imp.w, decay);
        *reinterpret_cast<float4 *>(grid + i) = v;
    } else {
        for (uint32_t k = i; k < n; ++k) grid[k] = ema_one(grid[k], grid_tmp[k], decay);
    }
}

// ---- K9 grid_to_bitfield (bitfield_func_kernel.cu:301-321) ------------------------------------------------------------
// one byte (8 consecutive Morton cells) per lane; threshold = min(opa_thres, mean density).  mean_dev, when given, is read
// on the device so that the caller's mean() needs no host round trip.
__global__ void __launch_bounds__(256)
grid_to_bitfield_kernel(uint32_t n_bytes, const float *__restrict__ grid, uint8_t *__restrict__ bitfield, float mean_host,
                        const float *__restrict__ mean_dev, float opa_thres) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bytes) return;
    const float mean = mean_dev ? mean_dev[0] : mean_host;
    const float thresh = opa_thres < mean ? opa_thres : mean;
    const float4 a = *reinterpret_cast<const float4 *>(grid + (int64_t)i * 8);
    const float4 b = *reinterpret_cast<const float4 *>(grid + (int64_t)i * 8 + 4);
    uint32_t bits = 0;
    bits |= a.x > thresh ? 1u : 0u;
    bits |= a.y > thresh ? 2u : 0u;
    bits |= a.z > thresh ? 4u : 0u;
    bits |= a.w > thresh ? 8u : 0u;
    bits |= b.x > thresh ? 16u : 0u;
    bits |= b.y > thresh ? 32u : 0u;
    bits |= b.z > thresh ? 64u : 0u;
    bits |= b.w > thresh ? 128u : 0u;
    bitfield[i] = (uint8_t)bits;
}

// ---- K10 count_bitfield (bitfield_func_kernel.cu:350-366) --------------------------------------------------------------
// The reference's test is `byte && (1 << j)` — a LOGICAL and — so each non-zero byte adds 8 x 1.0f to the float counter,
// whatever its population.  Reproduced: 8 per non-zero byte; per-wave ballot, one float atomic per workgroup.  Every
// partial sum is an integer below 2^24 (n_grid <= 256), so the float result does not depend on the order of the adds.
__global__ void __launch_bounds__(256)
count_bitfield_kernel(uint32_t n_bytes, const uint8_t *__restrict__ bitfield, float *__restrict__ counter) {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t local = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes; i += gridDim.x * blockDim.x)
        local += bitfield[i] != 0 ? 8u : 0u;
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) local += __shfl_down(local, dlt, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&s_cnt, local);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(counter, (float)s_cnt);
}

static inline bool pow2_grid(int n_grid) { return n_grid > 0 && n_grid <= 1024 && (n_grid & (n_grid - 1)) == 0; }


// ---- cells of an occupancy refresh (VolumeBound.optimize after its warm-up, volume_bound.py:178-190) ------------------------------
// n / 4 cells drawn uniformly without repetition + the first n / 4 occupied cells in flat-index order, and for every selected cell a point
// jittered uniformly inside its voxel.  The uniform subset is the image of [0, n / 4) under a seeded bijection of the (power-of-two)
// cell range (two rounds of odd multiply + add, xor-shift, odd multiply, xor-shift: geometry/volume.py mix_permutation); it is MARKED
// in a byte map (at its position on the Z-curve) and both halves then leave the map / the bitfield IN ORDER through one ordered
// compaction (block counts, scan of the 512 block totals, ranked writes) - ordered cells keep the lanes of the hash gather that follows
// spatial neighbours (a refresh in permutation order spent 0.82 us per 1000 points in the gather, the training step's ray-ordered
// samples take 0.36).
struct RefreshPerm {
    uint64_t a[2], c[2];
    int s1, s2;       // xor-shift distances
    uint64_t mask;    // n_cells - 1
};

__device__ __forceinline__ uint64_t refresh_perm(uint64_t x, const RefreshPerm &p) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        x = (x * p.a[r] + p.c[r]) & p.mask;
        x ^= x >> p.s1;
        x = (x * 0x9E3779B1ull) & p.mask;
        x ^= x >> p.s2;
    }
    return x;
}

// the uniform cells are marked at their MORTON position: the compaction then hands them out along the Z-curve, 64 consecutive points of
// the gather sit in one compact block of voxels (a run of flat indices is a line along z - the slowest axis of the dense level tables)
__global__ void __launch_bounds__(256) refresh_mark_kernel(uint8_t *__restrict__ sel, int64_t n_s, RefreshPerm p, int lg) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_s) return;
    const uint32_t cell = (uint32_t)refresh_perm((uint64_t)i, p), m = (1u << lg) - 1u;
    sel[morton3d(cell >> (2 * lg), (cell >> lg) & m, cell & m)] = 1;
}

constexpr int kRefreshPer = 16;                       // cells per thread
constexpr int kRefreshBlock = 256 * kRefreshPer;      // cells per workgroup

// flags of a thread's 16 consecutive cells as bits (two 8-byte loads per map)
__device__ __forceinline__ uint32_t refresh_flags16(const uint8_t *__restrict__ m, int64_t c0, int64_t n_cells) {
    uint32_t f = 0u;
    if (c0 + kRefreshPer <= n_cells) {
        const uint64_t lo = *reinterpret_cast<const uint64_t *>(m + c0), hi = *reinterpret_cast<const uint64_t *>(m + c0 + 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f |= (uint32_t)(((lo >> (8 * k)) & 0xffull) != 0ull) << k;
            f |= (uint32_t)(((hi >> (8 * k)) & 0xffull) != 0ull) << (8 + k);
        }
    } else {
        for (int k = 0; k < kRefreshPer; ++k)
            if (c0 + k < n_cells && m[c0 + k]) f |= 1u << k;
    }
    return f;
}

__global__ void __launch_bounds__(256) refresh_count_kernel(const uint8_t *__restrict__ sel, const uint8_t *__restrict__ occ, int64_t n_cells,
                                                            int32_t *__restrict__ blk) {   // blk[2][gridDim.x]
    __shared__ int32_t red[2][4];
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * kRefreshPer;
    int a = __popc(refresh_flags16(sel, c0, n_cells)), b = __popc(refresh_flags16(occ, c0, n_cells));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        blk[gridDim.x + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// one workgroup: exclusive scan of the two rows of block totals in place; n_valid = uniform cells + min(occupied, n_s)
__global__ void __launch_bounds__(1024) refresh_scan_kernel(int32_t *__restrict__ blk, int n_blocks, int64_t n_s, int32_t *__restrict__ n_valid) {
    __shared__ int32_t s_wave[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n_blocks + 1023) / 1024;
    int32_t tot[2];
    for (int row = 0; row < 2; ++row) {
        int32_t *b = blk + (int64_t)row * n_blocks;
        const int lo = tid * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
        int32_t sum = 0;
        for (int k = lo; k < hi; ++k) sum += b[k];
        int32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_wave[row][wv] = incl;
        __syncthreads();
        int32_t base = 0, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) base += s_wave[row][w]; total += s_wave[row][w]; }
        int32_t run = base + incl - sum;
        for (int k = lo; k < hi; ++k) { const int32_t c = b[k]; b[k] = run; run += c; }
        tot[row] = total;
    }
    if (tid == 0) *n_valid = tot[0] + (tot[1] < (int32_t)n_s ? tot[1] : (int32_t)n_s);
}

// ranked writes: cells_out[0 .. n_uniform) = marked cells, cells_out[n_s + r] = r-th occupied cell (r < n_s), both in flat order, and
// the jittered point of each: centre of the voxel + (u - 0.5) * voxel size, u from a pcg32 stream advanced to 3 x the output slot
__global__ void __launch_bounds__(256) refresh_write_kernel(const uint8_t *__restrict__ sel, const uint8_t *__restrict__ occ, int64_t n_cells,
                                                            const int32_t *__restrict__ blk, int n_blocks, int64_t n_s, int n_grid, float vs0, float vs1,
                                                            float vs2, float mn0, float mn1, float mn2, Pcg32 rng, int64_t *__restrict__ cells_out,
                                                            float *__restrict__ pts_out) {
    __shared__ int32_t s_wave[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + tid) * kRefreshPer;
    const uint32_t f[2] = {refresh_flags16(sel, c0, n_cells), refresh_flags16(occ, c0, n_cells)};
    int32_t rank[2];
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        const int32_t cnt = __popc(f[row]);
        int32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_wave[row][wv] = incl;
        rank[row] = incl - cnt;
    }
    __syncthreads();
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        for (int w = 0; w < wv; ++w) rank[row] += s_wave[row][w];
        rank[row] += blk[(int64_t)row * n_blocks + blockIdx.x];
    }
    const float mn[3] = {mn0, mn1, mn2}, vsz[3] = {vs0, vs1, vs2};
    const int lg = 31 - __clz(n_grid);                  // n_grid is a power of two: the index split is shifts and masks
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        uint32_t bits = f[row];
        int64_t r = rank[row];
        if (!bits || (row == 1 && r >= n_s)) continue;
        // a thread's selected cells take CONSECUTIVE output slots: one jump of the stream to the first, then it is read in order
        Pcg32 g = rng;
        g.advance(((row ? n_s : 0) + r) * 3);
        while (bits) {
            const int k = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            if (row == 1 && r >= n_s) break;
            const int64_t slot = (row ? n_s : 0) + r;
            uint32_t cell = (uint32_t)(c0 + k);
            if (row == 0)      // a position on the Z-curve -> flat index x n^2 + y n + z
                cell = (morton_compact(cell) << (2 * lg)) | (morton_compact(cell >> 1) << lg) | morton_compact(cell >> 2);
            cells_out[slot] = (int64_t)cell;
            const float idx3[3] = {(float)(cell >> (2 * lg)), (float)((cell >> lg) & (uint32_t)(n_grid - 1)), (float)(cell & (uint32_t)(n_grid - 1))};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float p = idx3[a] * vsz[a];      // the reference's order: index * size + half a voxel + origin, then the jitter
                p = p + 0.5f * vsz[a];
                p = p + mn[a];
                const float u = g.next_float();
                pts_out[slot * 3 + a] = p + (u - 0.5f) * vsz[a];
            }
            ++r;
        }
    }
}

}  // namespace arcn

using namespace arcn;

ARCN_EXPORT int arcn_generate_grid_samples(const float *density_grid, int ema_step, int n_elements, int n_grid, float thresh,
                                           uint64_t rng_state, uint64_t rng_inc, float *positions, int32_t *indices,
                                           void *stream) {
    if (n_elements <= 0) return ARCN_OK;  // linear_kernel launches nothing for an empty request (include/common.h:40-47)
    if (!density_grid || !positions || !indices) return einval("generate_grid_samples: missing argument");
    if (!pow2_grid(n_grid)) return einval("generate_grid_samples: n_grid must be a power of two <= 1024");
    Pcg32 rng{rng_state, rng_inc};
    hipLaunchKernelGGL(grid_samples_kernel, dim3(ceil_div<uint32_t>((uint32_t)n_elements, 256)), dim3(256), 0, as_stream(stream),
                       (uint32_t)n_elements, rng, (uint32_t)ema_step, density_grid, positions, indices, (uint32_t)n_grid, thresh);
    return check_launch("generate_grid_samples");
}

ARCN_EXPORT int arcn_splat_grid_samples(const float *density, const int32_t *indices, int n_samples, float *density_grid_tmp,
                                        void *stream) {
    if (n_samples <= 0) return ARCN_OK;
    if (!density || !indices || !density_grid_tmp) return einval("splat_grid_samples: missing argument");
    hipLaunchKernelGGL(splat_kernel, dim3(ceil_div<uint32_t>((uint32_t)n_samples, 256)), dim3(256), 0, as_stream(stream),
                       (uint32_t)n_samples, indices, density, density_grid_tmp);
    return check_launch("splat_grid_samples");
}

ARCN_EXPORT int arcn_ema_grid_samples_nerf(const float *density_grid_tmp, int n_elements, float decay, float *density_grid,
                                           void *stream) {
    if (n_elements <= 0) return ARCN_OK;
    if (!density_grid_tmp || !density_grid) return einval("ema_grid_samples_nerf: missing argument");
    if (((uintptr_t)density_grid_tmp | (uintptr_t)density_grid) & 15) return einval("ema_grid_samples_nerf: grids must be 16-byte aligned");
    const uint32_t quads = ceil_div<uint32_t>((uint32_t)n_elements, 4);
    hipLaunchKernelGGL(ema_kernel, dim3(ceil_div<uint32_t>(quads, 256)), dim3(256), 0, as_stream(stream), (uint32_t)n_elements,
                       decay, density_grid, density_grid_tmp);
    return check_launch("ema_grid_samples_nerf");
}

ARCN_EXPORT int arcn_update_bitfield(const float *density_grid, float density_grid_mean, const float *density_grid_mean_dev,
                                     uint8_t *bitfield, float opa_thres, int n_grid, void *stream) {
    if (!density_grid || !bitfield) return einval("update_bitfield: missing argument");
    if (!pow2_grid(n_grid) || n_grid < 2) return einval("update_bitfield: n_grid must be a power of two in [2, 1024]");
    if ((uintptr_t)density_grid & 15) return einval("update_bitfield: the grid must be 16-byte aligned");
    const uint32_t n_bytes = (uint32_t)n_grid * (uint32_t)n_grid * (uint32_t)n_grid / 8u;
    hipLaunchKernelGGL(grid_to_bitfield_kernel, dim3(ceil_div<uint32_t>(n_bytes, 256)), dim3(256), 0, as_stream(stream), n_bytes,
                       density_grid, bitfield, density_grid_mean, density_grid_mean_dev, opa_thres);
    return check_launch("update_bitfield");
}

ARCN_EXPORT int arcn_generate_grid_samples_multivol(const float *density_grid, int ema_step, int n_elements, const float *aabb,
                                                    int n_cascade, int n_grid, float thresh, int inclusive, uint64_t rng_state,
                                                    uint64_t rng_inc, float *positions, int32_t *indices, void *stream) {
    if (n_elements <= 0) return ARCN_OK;
    if (!density_grid || !aabb || !positions || !indices) return einval("generate_grid_samples_multivol: missing argument");
    if (!pow2_grid(n_grid)) return einval("generate_grid_samples_multivol: n_grid must be a power of two <= 1024");
    if (n_cascade < 1 || (!inclusive && n_cascade < 2) || (int64_t)n_grid * n_grid * n_grid * n_cascade > 0xffffffffLL)
        return einval("generate_grid_samples_multivol: bad n_cascade");
    Pcg32 rng{rng_state, rng_inc};
    hipLaunchKernelGGL(grid_samples_multivol_kernel, dim3(ceil_div<uint32_t>((uint32_t)n_elements, 256)), dim3(256), 0,
                       as_stream(stream), (uint32_t)n_elements, aabb, rng, (uint32_t)ema_step, density_grid, positions, indices,
                       (uint32_t)n_cascade, (uint32_t)n_grid, thresh, inclusive);
    return check_launch("generate_grid_samples_multivol");
}

ARCN_EXPORT int arcn_update_bitfield_multivol(const float *density_grid, float density_grid_mean, const float *density_grid_mean_dev,
                                              uint8_t *bitfield, float opa_thres, int n_grid, int n_cascade, int inclusive,
                                              void *stream) {
    if (!density_grid || !bitfield) return einval("update_bitfield_multivol: missing argument");
    if (!pow2_grid(n_grid) || n_grid < 2) return einval("update_bitfield_multivol: n_grid must be a power of two in [2, 1024]");
    const int levels = inclusive ? n_cascade : n_cascade - 1;
    if (levels < 1 || (int64_t)n_grid * n_grid * n_grid * levels > 0xffffffffLL) return einval("update_bitfield_multivol: bad n_cascade");
    if ((uintptr_t)density_grid & 15) return einval("update_bitfield_multivol: the grid must be 16-byte aligned");
    const uint32_t n_bytes = (uint32_t)n_grid * (uint32_t)n_grid * (uint32_t)n_grid / 8u * (uint32_t)levels;
    hipLaunchKernelGGL(grid_to_bitfield_kernel, dim3(ceil_div<uint32_t>(n_bytes, 256)), dim3(256), 0, as_stream(stream), n_bytes,
                       density_grid, bitfield, density_grid_mean, density_grid_mean_dev, opa_thres);
    return check_launch("update_bitfield_multivol");
}

ARCN_EXPORT int arcn_count_bitfield(const uint8_t *bitfield, float *counter, int n_grid, void *stream) {
    if (!bitfield || !counter) return einval("count_bitfield: missing argument");
    if (n_grid <= 0) return einval("count_bitfield: n_grid must be positive");
    const uint32_t n_bytes = (uint32_t)((int64_t)n_grid * n_grid * n_grid / 8);
    if (n_bytes == 0) return ARCN_OK;
    const uint32_t wgs = ceil_div<uint32_t>(n_bytes, 256);
    hipLaunchKernelGGL(count_bitfield_kernel, dim3(wgs < 1024 ? wgs : 1024), dim3(256), 0, as_stream(stream), n_bytes, bitfield,
                       counter);
    return check_launch("count_bitfield");
}

/* Cells and sample points of one occupancy refresh after the warm-up (VolumeBound.optimize, volume_bound.py:178-190): n_s = n_grid^3 / 4
 * cells drawn uniformly without repetition (the image of [0, n_s) under the seeded bijection perm_a / perm_c, geometry/volume.py
 * mix_permutation; n_grid a power of two) followed by the first n_s occupied cells of the boolean bitfield in flat-index order, both
 * halves ordered (the uniform half along the Z-curve, the occupied half by flat index); cells_out (2 n_s int64), n_valid = n_s + min(occupied, n_s) (device int32; entries behind it are not
 * written), pts_out (2 n_s, 3) = voxel centre + uniform jitter of one voxel (per-axis voxel sizes; pcg32 stream rng_state / rng_inc).  workspace: n_grid^3 +
 * 8 * (n_grid^3 / 4096 + 1) bytes.  No host synchronisation. */
ARCN_EXPORT int arcn_refresh_cells_points(const uint8_t *bitfield_bool, int n_grid, const uint64_t *perm_a, const uint64_t *perm_c,
                                          const float *voxel_size_host, const float *min_xyz_host, uint64_t rng_state, uint64_t rng_inc, int64_t *cells_out, float *pts_out,
                                          int32_t *n_valid, uint8_t *workspace, int64_t workspace_bytes, void *stream) {
    if (!bitfield_bool || !perm_a || !perm_c || !voxel_size_host || !min_xyz_host || !cells_out || !pts_out || !n_valid || !workspace)
        return einval("refresh_cells_points: missing argument");
    if (n_grid < 16 || n_grid > 1024 || (n_grid & (n_grid - 1))) return einval("refresh_cells_points: n_grid must be a power of two in 16 .. 1024");
    const int64_t n_cells = (int64_t)n_grid * n_grid * n_grid, n_s = n_cells / 4;
    const int n_blocks = (int)ceil_div<int64_t>(n_cells, kRefreshBlock);
    const int64_t need = n_cells + 8 * ((int64_t)n_blocks + 1);
    if (workspace_bytes < need || ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(bitfield_bool)) & 7u))
        return einval("refresh_cells_points: workspace too small, or workspace / bitfield not 8-byte aligned");
    int k = 0;
    while ((1ll << k) < n_cells) ++k;
    RefreshPerm p;
    p.mask = (uint64_t)n_cells - 1ull;
    for (int r = 0; r < 2; ++r) { p.a[r] = perm_a[r] & p.mask; p.c[r] = perm_c[r] & p.mask; if (!(p.a[r] & 1ull)) return einval("refresh_cells_points: even multiplier"); }
    p.s1 = (k + 1) / 2 > 1 ? (k + 1) / 2 : 1;
    p.s2 = k / 3 > 1 ? k / 3 : 1;
    uint8_t *sel = workspace;
    int32_t *blk = reinterpret_cast<int32_t *>(workspace + n_cells);
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(sel, 0, (size_t)n_cells, st) != hipSuccess) return check_launch("memset");
    int lg = 0;
    while ((1 << lg) < n_grid) ++lg;
    hipLaunchKernelGGL(refresh_mark_kernel, dim3((unsigned)ceil_div<int64_t>(n_s, 256)), dim3(256), 0, st, sel, n_s, p, lg);
    hipLaunchKernelGGL(refresh_count_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, sel, bitfield_bool, n_cells, blk);
    hipLaunchKernelGGL(refresh_scan_kernel, dim3(1), dim3(1024), 0, st, blk, n_blocks, n_s, n_valid);
    Pcg32 rng{rng_state, rng_inc};
    hipLaunchKernelGGL(refresh_write_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, sel, bitfield_bool, n_cells, blk, n_blocks, n_s, n_grid,
                       voxel_size_host[0], voxel_size_host[1], voxel_size_host[2], min_xyz_host[0], min_xyz_host[1], min_xyz_host[2], rng, cells_out, pts_out);
    return check_launch("refresh_cells_points");
}

