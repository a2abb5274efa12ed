// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, 256 CUs in 8 XCDs).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/arcnerf_hip.h"

#define ARCN_EXPORT extern "C" __attribute__((visibility("default")))

namespace arcn {

constexpr int kWave = 64;

void set_error(const char *msg);

// Launch check: records the hip error string for arcn_last_error().
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        (void)what;
        return ARCN_ELAUNCH;
    }
    return ARCN_OK;
}

inline int einval(const char *msg) {
    set_error(msg);
    return ARCN_EINVAL;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) {
    return (a + b - 1) / b;
}

__device__ __forceinline__ int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Device-side element count: *n_ptr (clamped to capacity n) when given, else n.
__device__ __forceinline__ int64_t dev_count(int64_t n, const int32_t *n_ptr) {
    if (n_ptr) {
        int64_t v = (int64_t)(*n_ptr);
        return v < n ? v : n;
    }
    return n;
}

// ---- pcg32 (arcnerf/ops/include/pcg32.h:38-165), bit-for-bit ---------------------------------
struct Pcg32 {
    uint64_t state, inc;
    static constexpr uint64_t kMult = 0x5851f42d4c957f2dULL;

    __host__ __device__ uint32_t next_uint() {
        uint64_t old = state;
        state = old * kMult + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    __host__ __device__ float next_float() {
        union { uint32_t u; float f; } x;
        x.u = (next_uint() >> 9) | 0x3f800000u;
        return x.f - 1.0f;
    }
    __host__ __device__ void advance(int64_t delta_) {
        uint64_t cur_mult = kMult, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
        uint64_t delta = (uint64_t)delta_;
        while (delta > 0) {
            if (delta & 1) {
                acc_mult *= cur_mult;
                acc_plus = acc_plus * cur_mult + cur_plus;
            }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta /= 2;
        }
        state = acc_mult * state + acc_plus;
    }
    __host__ __device__ void seed(uint64_t initstate, uint64_t initseq) {
        state = 0u;
        inc = (initseq << 1u) | 1u;
        next_uint();
        state += initstate;
        next_uint();
    }
};

// ---- activations (get_activation, base_modules/activation.py:24-50; TruncExp ops/trunc_exp.py:7-37) ----
__device__ __forceinline__ float act_fwd(float v, int act, float beta) {
    switch (act) {
    case ARCN_ACT_RELU: return v > 0.f ? v : 0.f;
    case ARCN_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case ARCN_ACT_TRUNCEXP: return expf(v);
    case ARCN_ACT_SOFTPLUS: { float bv = beta * v; return bv > 20.f ? v : log1pf(expf(bv)) / beta; }
    default: return v;
    }
}

// derivative wrt the pre-activation given pre-activation v and post-activation y
__device__ __forceinline__ float act_grad(float v, float y, int act, float beta) {
    switch (act) {
    case ARCN_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case ARCN_ACT_SIGMOID: return y * (1.0f - y);
    case ARCN_ACT_TRUNCEXP: { float c = v < -15.f ? -15.f : (v > 15.f ? 15.f : v); return expf(c); }
    case ARCN_ACT_SOFTPLUS: { float bv = beta * v; return bv > 20.f ? 1.f : 1.0f / (1.0f + expf(-bv)); }
    default: return 1.f;
    }
}

// ---- wave-level scans (64 lanes) ---------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// inclusive prefix product across the wave
__device__ __forceinline__ float wave_incl_prod(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_up(v, d, 64);
        if (lane >= d) v = v * o;
    }
    return v;
}

// inclusive prefix sum
__device__ __forceinline__ float wave_incl_sum(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_up(v, d, 64);
        if (lane >= d) v = v + o;
    }
    return v;
}

// inclusive suffix sum (lane i gets sum of lanes >= i)
__device__ __forceinline__ float wave_incl_suffix_sum(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float o = __shfl_down(v, d, 64);
        if (lane + d < 64) v = v + o;
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

}  // namespace arcn
