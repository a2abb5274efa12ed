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
    case ARCN_ACT_SQUAREPLUS: { float X = 10.f * v; return 0.5f * (X + sqrtf(X * X + 4.f)) / 10.f; }
    case ARCN_ACT_SINE: return sinf(v);
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
    case ARCN_ACT_SQUAREPLUS: { float Y = 10.f * y, t = Y * Y; return t / (t + 1.f); }
    case ARCN_ACT_SINE: return cosf(v);
    default: return 1.f;
    }
}

// second derivative wrt the pre-activation (the derivative of act_grad as a function of v)
__device__ __forceinline__ float act_grad2(float v, int act, float beta) {
    switch (act) {
    case ARCN_ACT_SIGMOID: { float y = 1.0f / (1.0f + expf(-v)); return y * (1.0f - y) * (1.0f - 2.0f * y); }
    case ARCN_ACT_TRUNCEXP: return (v > -15.f && v < 15.f) ? expf(v) : 0.f;
    case ARCN_ACT_SOFTPLUS: { float bv = beta * v; if (bv > 20.f) return 0.f; float sg = 1.0f / (1.0f + expf(-bv)); return beta * sg * (1.0f - sg); }
    case ARCN_ACT_SQUAREPLUS: { float X = 10.f * v, Y = 0.5f * (X + sqrtf(X * X + 4.f)), t = Y * Y + 1.f; return 10.f * 2.f * Y * Y * Y / (t * t * t); }
    case ARCN_ACT_SINE: return -sinf(v);
    default: return 0.f;      // identity, ReLU
    }
}

// ---- wave-level scans (64 lanes) ---------------------------------------------------------------
// All of these run on the VALU's DPP cross-lane path: a shift inside a 16-lane row is a modifier on the consuming instruction,
// and the three row totals cross rows through v_readlane (SGPRs).  The ds_bpermute forms (__shfl*) they replace go through the
// LDS pipeline at ~6 issue cycles per wave each: the fused compositor issued 59 of them per ray and, with 8320 rays in flight,
// spent a quarter of its run time queued on that pipeline.  Every lane of the wave must be active at the call.
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// DPP controls (gfx9 encoding): row_shl:n 0x100+n (lane i reads lane i+n of its row), row_shr:n 0x110+n (lane i reads lane i-n),
// wave_shl:1 0x130, wave_shr:1 0x138, row_mirror 0x140, row_half_mirror 0x141.  A lane whose source falls outside the row (or the
// wave) keeps `fill`.
template <int CTRL>
__device__ __forceinline__ float dpp_take(float fill, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// the same with zero for a source outside the row (bound_ctrl): foldable into the DPP form of the VOP2 instruction that consumes it
template <int CTRL>
__device__ __forceinline__ float dpp_zero(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_take_i(int fill, int v) {
    return __builtin_amdgcn_update_dpp(fill, v, CTRL, 0xf, 0xf, false);
}
template <int LANE>
__device__ __forceinline__ float lane_value(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), LANE));
}

// lane i <- lane i-1 (lane 0 <- fill) and lane i <- lane i+1 (lane 63 <- fill)
__device__ __forceinline__ float wave_from_below(float v, float fill) { return dpp_take<0x138>(fill, v); }
__device__ __forceinline__ float wave_from_above(float v, float fill) { return dpp_take<0x130>(fill, v); }

// inclusive prefix product across the wave
__device__ __forceinline__ float wave_incl_prod(float v) {
    v = v * dpp_take<0x111>(1.f, v);
    v = v * dpp_take<0x112>(1.f, v);
    v = v * dpp_take<0x114>(1.f, v);
    v = v * dpp_take<0x118>(1.f, v);
    const float r0 = lane_value<15>(v), r1 = lane_value<31>(v), r2 = lane_value<47>(v);
    const int row = lane_id() >> 4;
    const float r01 = r0 * r1;
    const float below = row == 0 ? 1.f : (row == 1 ? r0 : (row == 2 ? r01 : r01 * r2));
    return below * v;
}

// inclusive prefix sum
__device__ __forceinline__ float wave_incl_sum(float v) {
    v = v + dpp_take<0x111>(0.f, v);
    v = v + dpp_take<0x112>(0.f, v);
    v = v + dpp_take<0x114>(0.f, v);
    v = v + dpp_take<0x118>(0.f, v);
    const float r0 = lane_value<15>(v), r1 = lane_value<31>(v), r2 = lane_value<47>(v);
    const int row = lane_id() >> 4;
    const float r01 = r0 + r1;
    const float below = row == 0 ? 0.f : (row == 1 ? r0 : (row == 2 ? r01 : r01 + r2));
    return below + v;
}

// inclusive suffix sum (lane i gets sum of lanes >= i)
__device__ __forceinline__ float wave_incl_suffix_sum(float v) {
    v = v + dpp_take<0x101>(0.f, v);
    v = v + dpp_take<0x102>(0.f, v);
    v = v + dpp_take<0x104>(0.f, v);
    v = v + dpp_take<0x108>(0.f, v);
    const float r1 = lane_value<16>(v), r2 = lane_value<32>(v), r3 = lane_value<48>(v);
    const int row = lane_id() >> 4;
    const float r23 = r2 + r3;
    const float above = row == 3 ? 0.f : (row == 2 ? r3 : (row == 1 ? r23 : r23 + r1));
    return above + v;
}

// sum over the wave, the same value in every lane
__device__ __forceinline__ float wave_sum(float v) {
    v = v + dpp_take<0xB1>(0.f, v);   // quad_perm [1,0,3,2]
    v = v + dpp_take<0x4E>(0.f, v);   // quad_perm [2,3,0,1]
    v = v + dpp_take<0x141>(0.f, v);  // the other quad of the half row
    v = v + dpp_take<0x140>(0.f, v);  // the other half of the row
    return (lane_value<0>(v) + lane_value<16>(v)) + (lane_value<32>(v) + lane_value<48>(v));
}

__device__ __forceinline__ int wave_sum_i(int v) {
    v = v + dpp_take_i<0xB1>(0, v);
    v = v + dpp_take_i<0x4E>(0, v);
    v = v + dpp_take_i<0x141>(0, v);
    v = v + dpp_take_i<0x140>(0, v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

}  // namespace arcn
