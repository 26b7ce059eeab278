// Elementwise kernels of the SpecSeg mask network (inference only; SpecSeg.py:27-98, called at
// SHM.py:492) and the specular loss it feeds (SHM.py:792-806).  The 3x3 conv + ReLU layers and the
// 2x2 stride-2 Conv2DTranspose run on the tap GEMM (conv_igemm.hip); everything here is HBM-bound.
#include "common.h"

static int grid_cap(size_t n, int per_block = 256, int cap = 8192) {
    long g = (long)((n + per_block - 1) / per_block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------ channel pack (1 -> 16 pitch)
__global__ void pack_channels_kernel(const float* __restrict__ src, int ldsrc, int c0, int nc, float* __restrict__ dst, int lddst4, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i / lddst4;
        int q = (int)(i % lddst4) * 4;
        const float* s = src + p * ldsrc + c0;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (q + j < nc) ? s[q + j] : 0.f;
        *(f32x4*)(dst + i * 4) = v;
    }
}

extern "C" int shm_pack_channels(const float* src, int ldsrc, int c0, int nc, float* dst, int lddst, size_t npix, void* stream) {
    SHM_REQUIRE(lddst % 4 == 0 && nc <= lddst && c0 >= 0 && c0 + nc <= ldsrc, SHM_E_SHAPE, "shm_pack_channels: bad channel window");
    size_t total = npix * (lddst / 4);
    if (total == 0) return SHM_OK;
    hipLaunchKernelGGL(pack_channels_kernel, dim3(grid_cap(total)), dim3(256), 0, (hipStream_t)stream, src, ldsrc, c0, nc, dst, lddst / 4, total);
    SHM_LAUNCH_CHECK("shm_pack_channels");
    return SHM_OK;
}

// ------------------------------------------------------------ BatchNormalization (inference)
// out = (a - mean) * gamma / sqrt(var + eps) + beta ; optionally also the 2x2 max pool of out.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ a, int lda, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ var, float eps, float* __restrict__ out, int ldo,
                                                       size_t npix, int c4) {
    const size_t total = npix * c4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i / c4;
        int c = (int)(i % c4) * 4;
        f32x4 v = *(const f32x4*)(a + p * lda + c);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sc = gamma[c + j] / sqrtf(var[c + j] + eps);
            r[j] = (v[j] - mean[c + j]) * sc + beta[c + j];
        }
        *(f32x4*)(out + p * ldo + c) = r;
    }
}

extern "C" int shm_bn_apply(const float* a, int lda, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                            float* out, int ldo, size_t npix, int c, void* stream) {
    SHM_REQUIRE(c % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0, SHM_E_SHAPE, "shm_bn_apply: channels/pitch must be multiples of 4");
    if (npix == 0 || c == 0) return SHM_OK;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_cap(npix * (c / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, gamma, beta, mean, var, eps, out, ldo,
                       npix, c / 4);
    SHM_LAUNCH_CHECK("shm_bn_apply");
    return SHM_OK;
}

// ------------------------------------------------------------------------------- MaxPooling2D
__global__ void maxpool2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int h, int w, int c4, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int cl = (int)(i % c4);
        size_t q = i / c4;
        int wo = w >> 1, ho = h >> 1;
        int ox = (int)(q % wo);
        size_t t = q / wo;
        int oy = (int)(t % ho);
        size_t n = t / ho;
        const float* b = x + ((n * h + 2 * oy) * w + 2 * ox) * ldx + cl * 4;
        f32x4 v0 = *(const f32x4*)b, v1 = *(const f32x4*)(b + ldx), v2 = *(const f32x4*)(b + (size_t)w * ldx), v3 = *(const f32x4*)(b + (size_t)(w + 1) * ldx);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = fmaxf(fmaxf(v0[j], v1[j]), fmaxf(v2[j], v3[j]));
        *(f32x4*)(y + q * ldy + cl * 4) = r;
    }
}

extern "C" int shm_maxpool2_fwd(const float* x, int ldx, float* y, int ldy, int batch, int h, int w, int c, void* stream) {
    SHM_REQUIRE(c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, SHM_E_SHAPE, "shm_maxpool2_fwd: channels/pitch must be multiples of 4");
    SHM_REQUIRE(h % 2 == 0 && w % 2 == 0, SHM_E_SHAPE, "shm_maxpool2_fwd: odd size %dx%d", h, w);
    size_t total = (size_t)batch * (h / 2) * (w / 2) * (c / 4);
    if (total == 0) return SHM_OK;
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_cap(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, h, w, c / 4, total);
    SHM_LAUNCH_CHECK("shm_maxpool2_fwd");
    return SHM_OK;
}

// --------------------------------------------------------- Conv2D(1, 1x1, activation='sigmoid')
__global__ __launch_bounds__(256) void head_sigmoid_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, size_t npix, int c) {
    const int lanes_c = c >> 2, PP = 256 / lanes_c;
    const int pp = threadIdx.x / lanes_c, cl = threadIdx.x % lanes_c;
    f32x4 wv = *(const f32x4*)(w + cl * 4);
    const float b = bias ? bias[0] : 0.f;
    for (size_t p = (size_t)blockIdx.x * PP + pp; p < npix; p += (size_t)gridDim.x * PP) {
        f32x4 xv = *(const f32x4*)(x + p * ldx + cl * 4);
        float s = xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        for (int o = lanes_c >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (cl == 0) y[p] = 1.f / (1.f + expf(-(s + b)));
    }
}

extern "C" int shm_head_sigmoid_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, size_t npix, int c, void* stream) {
    const int l = c / 4;
    SHM_REQUIRE(c % 4 == 0 && l >= 1 && l <= 64 && (l & (l - 1)) == 0 && ldx % 4 == 0, SHM_E_SHAPE, "shm_head_sigmoid_fwd: channels %d unsupported", c);
    if (npix == 0) return SHM_OK;
    int PP = 256 / l;
    long blocks = ((long)npix + PP - 1) / PP;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(head_sigmoid_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, w, bias, y, npix, c);
    SHM_LAUNCH_CHECK("shm_head_sigmoid_fwd");
    return SHM_OK;
}

// ------------------------------------------------------------------- specular loss (logged only)
// loss[k] = sum over (b, p, ch) of (mask[b,p] * (cyc_k_yuv[b,p,ch] - ds_k[b,p,ch]))^2, with
// cyc_k_yuv = concat(cyc_y[k*B + b], cbcr[b]).  The caller divides by B*S*S*3 (reduce_mean).
struct SpecPtrs {
    const float* ds[5];
};

__global__ __launch_bounds__(256) void spec_loss_kernel(const float* __restrict__ cyc_y, const float* __restrict__ cbcr, SpecPtrs ds,
                                                        const float* __restrict__ mask, double* __restrict__ loss, size_t n) {
    const int k = blockIdx.y;
    const float* d = ds.ds[k];
    const float* cy = cyc_y + (size_t)k * n;
    double acc = 0.0;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        float m = mask[p];
        float e0 = m * cy[p] - m * d[p * 3];
        float e1 = m * cbcr[p * 2] - m * d[p * 3 + 1];
        float e2 = m * cbcr[p * 2 + 1] - m * d[p * 3 + 2];
        acc += (double)(e0 * e0) + (double)(e1 * e1) + (double)(e2 * e2);
    }
    acc = shm_wave_sum(acc);
    __shared__ double sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss + k, sm[0] + sm[1] + sm[2] + sm[3]);
}

extern "C" int shm_spec_loss(const float* cyc_y, const float* cbcr, const float* const* ds, const float* mask, double* loss, int batch, size_t npix,
                             void* stream) {
    SHM_REQUIRE(cyc_y && cbcr && ds && mask && loss, SHM_E_SHAPE, "shm_spec_loss: null pointer");
    int r = shm_zero(loss, 5 * sizeof(double), stream);
    if (r) return r;
    size_t n = (size_t)batch * npix;
    if (n == 0) return SHM_OK;
    SpecPtrs P;
    for (int k = 0; k < 5; ++k) P.ds[k] = ds[k];
    hipLaunchKernelGGL(spec_loss_kernel, dim3(grid_cap(n, 256, 256), 5), dim3(256), 0, (hipStream_t)stream, cyc_y, cbcr, P, mask, loss, n);
    SHM_LAUNCH_CHECK("shm_spec_loss");
    return SHM_OK;
}
