// 3-D Morton (Z-order) index of the `_bitfield_func` occupancy layout (arcnerf/ops/include/volume_func.h:141-168):
// bit 3k of the index is bit k of x, bit 3k+1 of y, bit 3k+2 of z; 10 bits per axis.
#pragma once
#include <cstdint>

namespace arcn {

__host__ __device__ __forceinline__ uint32_t morton_spread(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__host__ __device__ __forceinline__ uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return morton_spread(x) | (morton_spread(y) << 1) | (morton_spread(z) << 2);
}

__host__ __device__ __forceinline__ uint32_t morton_compact(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

}  // namespace arcn
