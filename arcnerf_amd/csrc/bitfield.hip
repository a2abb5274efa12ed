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
