// Pinhole camera -> world-space ray of a pixel (arcnerf/render/ray_helper.py:12-153 get_rays, geometry/projection.py:8-66 pixel_to_world):
// shared by the ray generation of volume.hip and the training-batch fetch of batch.hip.
#pragma once

#include "common.hpp"

namespace arcn {

struct CamParams {
    float fx, skew, cx, fy, cy;
    float r[3][4];  // c2w rows
};

__device__ __forceinline__ void ray_dir(const CamParams &c, float pi, float pj, bool normalise, float out[3]) {
    float cam[3];
    cam[0] = (pi - (c.skew * (pj - c.cy) / c.fy) - c.cx) / c.fx * 1.0f;
    cam[1] = (pj - c.cy) / c.fy * 1.0f;
    cam[2] = 1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float w = c.r[k][0] * cam[0];
        w = w + c.r[k][1] * cam[1];
        w = w + c.r[k][2] * cam[2];
        w = w + c.r[k][3];
        out[k] = w - c.r[k][3];
    }
    if (normalise) {
        const float nrm = sqrtf(out[0] * out[0] + out[1] * out[1] + out[2] * out[2]) + 1e-8f;
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = out[k] / nrm;
    }
}

// origin and direction of pixel (pi, pj) as the reference returns them: normalised, or warped to NDC (get_ndc_rays)
__device__ __forceinline__ void pixel_ray(const CamParams &c, int W, int H, float pi, float pj, bool normalise, bool ndc,
                                          float ndc_near, float o[3], float d[3]) {
    o[0] = c.r[0][3]; o[1] = c.r[1][3]; o[2] = c.r[2][3];
    ray_dir(c, pi, pj, normalise && !ndc, d);
    if (ndc) {
        const float t = -(ndc_near + o[2]) / d[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = o[k] + t * d[k];
        const float ax = -1.0f / ((float)W / (2.0f * c.fx)), ay = -1.0f / ((float)H / (2.0f * c.fy));
        const float no[3] = {ax * o[0] / o[2], ay * o[1] / o[2], 1.0f + 2.0f * ndc_near / o[2]};
        const float nd[3] = {ax * (d[0] / d[2] - o[0] / o[2]), ay * (d[1] / d[2] - o[1] / o[2]), -2.0f * ndc_near / o[2]};
#pragma unroll
        for (int k = 0; k < 3; ++k) { o[k] = no[k]; d[k] = nd[k]; }
    }
}

// camera `view` of (n, 3, 3) intrinsics / (n, 4, 4) poses, row-major device floats
__device__ __forceinline__ CamParams load_camera(const float *__restrict__ K, const float *__restrict__ c2w) {
    CamParams c;
    c.fx = K[0]; c.skew = K[1]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int m = 0; m < 4; ++m) c.r[k][m] = c2w[4 * k + m];
    return c;
}

}  // namespace arcn
