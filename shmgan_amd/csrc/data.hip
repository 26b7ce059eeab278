// Input pipeline step of datasetLoader.py:47-60 on the device: tf.image.resize(bilinear, half-pixel centres,
// no antialias) of a decoded uint8 RGB image to image_size x image_size, x / 255, tf.image.flip_up_down.
// The decoded file is uploaded as bytes (3 B/pixel instead of 12) and everything after the decode is one kernel.
#include "common.h"

__global__ void resize_bilinear_u8_kernel(const unsigned char* __restrict__ src, int hin, int win, int c, float* __restrict__ dst, int ho, int wo,
                                          float hs, float ws, float scale, int flip_ud) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // (oy, ox)
    if (idx >= (size_t)ho * wo) return;
    const int ox = (int)(idx % wo), oy = (int)(idx / wo);
    // ResizeBilinear with half_pixel_centers: in = (out + 0.5) * scale - 0.5; lower = max(floor(in), 0);
    // upper = min(ceil(in), size - 1); lerp = in - floor(in)
    const float fy = ((float)oy + 0.5f) * hs - 0.5f, fx = ((float)ox + 0.5f) * ws - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), y1 = min((int)ceilf(fy), hin - 1);
    const int x0 = max((int)flx, 0), x1 = min((int)ceilf(fx), win - 1);
    const float ly = fy - fly, lx = fx - flx;
    const int oyy = flip_ud ? ho - 1 - oy : oy;
    float* o = dst + ((size_t)oyy * wo + ox) * c;
    for (int k = 0; k < c; ++k) {
        const float tl = src[((size_t)y0 * win + x0) * c + k], tr = src[((size_t)y0 * win + x1) * c + k];
        const float bl = src[((size_t)y1 * win + x0) * c + k], br = src[((size_t)y1 * win + x1) * c + k];
        const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
        o[k] = (top + (bot - top) * ly) * scale;
    }
}

extern "C" int shm_resize_bilinear_u8(const unsigned char* src, int hin, int win, int c, float* dst, int ho, int wo, float scale, int flip_ud,
                                      void* stream) {
    SHM_REQUIRE(src && dst && hin > 0 && win > 0 && ho > 0 && wo > 0 && c > 0, SHM_E_SHAPE, "shm_resize_bilinear_u8: bad arguments");
    const size_t n = (size_t)ho * wo;
    hipLaunchKernelGGL(resize_bilinear_u8_kernel, dim3(shm_cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)stream, src, hin, win, c, dst, ho, wo,
                       (float)hin / (float)ho, (float)win / (float)wo, scale, flip_ud);
    SHM_LAUNCH_CHECK("shm_resize_bilinear_u8");
    return SHM_OK;
}
