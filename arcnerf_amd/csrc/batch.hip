// The training batch from a dataset kept as IMAGES + CAMERAS (arcnerf/trainer/pipeline.py:243-300 fetch_step_ray_sample /
// fetch_step_bkg_color on what arcnerf_trainer.py:188-219 concat_train_batch collects).
//
// The reference materialises, per training pixel, img 3 + mask 1 + rays_o 3 + rays_d 3 + rays_r 1 floats (44 B; 2.8 GB for the 100
// Lego views), CROPS those tensors (step_crop_center_image), SHUFFLES them by a randperm gather (step_ray_sample) and slices n_rays
// rows per iteration, then blends a random background colour into the target (5 torch kernels).  Here the dataset stays what it was on
// disk - RGBA bytes (or float img + mask) and one camera per view - the crop is a window, the shuffle a permutation of ray ids, and one
// lane per ray of ONE launch turns an id into (view, y, x), the pixel's ray through that view's camera (the arithmetic of get_rays,
// camera.hpp), the mip-nerf radius, the colour, the mask and the blended target: 8 B of id + 4..16 B of pixel read per ray.
#include "camera.hpp"
#include "common.hpp"

namespace arcn {

struct BatchView {
    int n_img, H, W, y0, x0, Hc, Wc;
};

__global__ void __launch_bounds__(256)
fetch_train_batch_kernel(const uint8_t *__restrict__ rgba, const float *__restrict__ img, const float *__restrict__ mask,
                         const float *__restrict__ K, const float *__restrict__ c2w, BatchView v, const int64_t *__restrict__ ids, int64_t n,
                         int center_pixel, int normalise, const float *__restrict__ bkg_rand, float b0, float b1, float b2, int blend,
                         float *__restrict__ rays_o, float *__restrict__ rays_d, float *__restrict__ rays_r, float *__restrict__ img_out,
                         float *__restrict__ mask_out, float *__restrict__ bkg_out, int64_t *__restrict__ src_out, int32_t *__restrict__ bad) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t per = (int64_t)v.Hc * v.Wc;
    int64_t id = ids[r];
    if (id < 0 || id >= per * v.n_img) {      // reported, and clamped so that the launch stays inside the dataset
        if (bad) atomicAdd(bad, 1);
        id = 0;
    }
    const int64_t view = id / per, rem = id % per;
    const int y = v.y0 + (int)(rem / v.Wc), x = v.x0 + (int)(rem % v.Wc);
    const int64_t src = (view * v.H + y) * v.W + x;
    if (src_out) src_out[r] = src;
    if (rays_o) {
        const CamParams c = load_camera(K + 9 * view, c2w + 16 * view);
        const float off = center_pixel ? 0.5f : 0.0f;
        float o[3], d[3];
        pixel_ray(c, v.W, v.H, (float)x + off, (float)y + off, normalise, false, 1.0f, o, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) { rays_o[3 * r + k] = o[k]; rays_d[3 * r + k] = d[k]; }
        if (rays_r) {   // |d(x, y) - d(x + 1, y)| * 2 / sqrt(12); the last column takes column W-3's value (ray_helper.py:106-116)
            const int xa = x < v.W - 1 ? x : v.W - 3;
            float ao[3], a[3], bo[3], b[3];
            pixel_ray(c, v.W, v.H, (float)xa + off, (float)y + off, normalise, false, 1.0f, ao, a);
            pixel_ray(c, v.W, v.H, (float)(xa + 1) + off, (float)y + off, normalise, false, 1.0f, bo, b);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float df = a[k] - b[k]; acc += df * df; }
            rays_r[r] = sqrtf(acc) * 2.0f / sqrtf(12.0f);
        }
    }
    if (!img_out) return;
    float col[3], m = 1.0f;
    if (rgba) {     // NeRF.read_image_list (nerf_dataset.py:107-119): bytes.astype(float32) / 255.0
        const uint32_t px = reinterpret_cast<const uint32_t *>(rgba)[src];
        col[0] = (float)(px & 0xffu) / 255.0f;
        col[1] = (float)((px >> 8) & 0xffu) / 255.0f;
        col[2] = (float)((px >> 16) & 0xffu) / 255.0f;
        m = (float)(px >> 24) / 255.0f;
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] = img[3 * src + k];
        if (mask) m = mask[src];
    }
    const float bk[3] = {bkg_rand ? bkg_rand[3 * r] : b0, bkg_rand ? bkg_rand[3 * r + 1] : b1, bkg_rand ? bkg_rand[3 * r + 2] : b2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t = col[k];
        if (blend) {    // img * mask + (1 - mask) * bkg_color (pipeline.py:297), torch's operation order
            const float a = col[k] * m;
            const float w = (1.0f - m) * bk[k];
            t = a + w;
        }
        img_out[3 * r + k] = t;
        if (bkg_out && blend) bkg_out[3 * r + k] = bk[k];
    }
    if (mask_out) mask_out[r] = m;
}

}  // namespace arcn

using namespace arcn;

ARCN_EXPORT int arcn_fetch_train_batch(const uint8_t *rgba, const float *img, const float *mask, const float *intrinsic, const float *c2w,
                                       int n_img, int H, int W, int y0, int x0, int Hc, int Wc, const int64_t *ids, int64_t n,
                                       int center_pixel, int normalize_rays_d, const float *bkg_rand, const float *bkg_const_host,
                                       float *rays_o, float *rays_d, float *rays_r, float *img_out, float *mask_out, float *bkg_out,
                                       int64_t *src_out, int32_t *bad_ids, void *stream) {
    if (n <= 0) return ARCN_OK;
    if (!ids || n_img <= 0 || H <= 0 || W <= 0) return einval("fetch_train_batch: missing ids / empty dataset");
    if (y0 < 0 || x0 < 0 || Hc <= 0 || Wc <= 0 || y0 + Hc > H || x0 + Wc > W) return einval("fetch_train_batch: the crop window leaves the image");
    if ((rays_o != nullptr) != (rays_d != nullptr)) return einval("fetch_train_batch: rays_o and rays_d come together");
    if (rays_o && (!intrinsic || !c2w)) return einval("fetch_train_batch: rays need intrinsic (n_img,3,3) and c2w (n_img,4,4)");
    if (rays_r && !rays_o) return einval("fetch_train_batch: the ray radius comes with the rays");
    if (rays_r && W < 3) return einval("fetch_train_batch: the ray radius needs W >= 3");
    if (img_out && !rgba && !img) return einval("fetch_train_batch: colours need rgba (n_img,H,W,4) bytes or img (n_img,H,W,3) floats");
    if (rgba && img) return einval("fetch_train_batch: rgba bytes OR float img (+ mask), not both");
    if ((mask_out || bkg_out) && !img_out) return einval("fetch_train_batch: mask / bkg outputs come with img_out");
    if (bkg_rand && bkg_const_host) return einval("fetch_train_batch: random background colours OR one constant colour");
    const int has_mask = rgba != nullptr || mask != nullptr;
    const int blend = has_mask && (bkg_rand || bkg_const_host);      // (the reference blends only when the data has a mask, pipeline.py:281-283)
    if (bkg_out && !blend) return einval("fetch_train_batch: bkg_out without a mask and a background colour");
    if (!rays_o && !img_out && !src_out) return einval("fetch_train_batch: nothing to write");
    float b[3] = {0.f, 0.f, 0.f};
    if (bkg_const_host) { b[0] = bkg_const_host[0]; b[1] = bkg_const_host[1]; b[2] = bkg_const_host[2]; }
    BatchView v{n_img, H, W, y0, x0, Hc, Wc};
    hipLaunchKernelGGL(fetch_train_batch_kernel, dim3((unsigned)ceil_div<int64_t>(n, 256)), dim3(256), 0, as_stream(stream), rgba, img, mask,
                       intrinsic, c2w, v, ids, n, center_pixel, normalize_rays_d, bkg_rand, b[0], b[1], b[2], blend, rays_o, rays_d, rays_r,
                       img_out, mask_out, bkg_out, src_out, bad_ids);
    return check_launch("fetch_train_batch");
}
