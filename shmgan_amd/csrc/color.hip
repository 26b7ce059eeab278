// Colour conversion, per-image standardisation, generator/discriminator input assembly,
// discriminator-head losses and the clip+Adam update.  All HBM-bound, one pass each.
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// tf.image.rgb_to_yuv kernel (column j of yuv = sum_i rgb[i] * K[i][j])
#define R2Y_00 0.299f
#define R2Y_01 -0.14714119f
#define R2Y_02 0.61497538f
#define R2Y_10 0.587f
#define R2Y_11 -0.28886916f
#define R2Y_12 -0.51496512f
#define R2Y_20 0.114f
#define R2Y_21 0.43601035f
#define R2Y_22 -0.10001026f
// tf.image.yuv_to_rgb kernel
#define Y2R_V_R 1.13988303f
#define Y2R_U_G -0.394642334f
#define Y2R_V_G -0.58062185f
#define Y2R_U_B 2.03206185f

__device__ __forceinline__ void rgb2yuv(float r, float g, float b, float& y, float& u, float& v) {
    y = r * R2Y_00 + g * R2Y_10 + b * R2Y_20;
    u = r * R2Y_01 + g * R2Y_11 + b * R2Y_21;
    v = r * R2Y_02 + g * R2Y_12 + b * R2Y_22;
}

__device__ __forceinline__ double block_sum_d(double v) {
    __shared__ double ws[4];
    v = shm_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    return ws[0] + ws[1] + ws[2] + ws[3];
}

// ---------------------------------------------------------------- rgb -> yuv + standardise
__global__ __launch_bounds__(256) void yuv_stats_kernel(const float* __restrict__ rgb, double* __restrict__ acc, size_t npix) {
    const int b = blockIdx.y;
    const float* base = rgb + (size_t)b * npix * 3;
    double s = 0.0, q = 0.0;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
        float y, u, v;
        rgb2yuv(base[p * 3], base[p * 3 + 1], base[p * 3 + 2], y, u, v);
        s += (double)y + (double)u + (double)v;
        q += (double)y * y + (double)u * u + (double)v * v;
    }
    s = block_sum_d(s);
    q = block_sum_d(q);
    if (threadIdx.x == 0) {
        atomicAdd(&acc[2 * b], s);
        atomicAdd(&acc[2 * b + 1], q);
    }
}

__global__ __launch_bounds__(256) void yuv_scale_kernel(const float* __restrict__ rgb, float* __restrict__ yuv, const double* __restrict__ acc, float* __restrict__ scale_out, size_t npix) {
    const int b = blockIdx.y;
    const double cnt = (double)npix * 3.0;
    double mean = acc[2 * b] / cnt;
    double var = acc[2 * b + 1] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    float scale = fmaxf((float)sqrt(var), 1.0f / 256.0f);   // rsqrt(65536), SHM.py:1280,1293
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[b] = scale;
    const float* base = rgb + (size_t)b * npix * 3;
    float* ob = yuv + (size_t)b * npix * 3;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
        float y, u, v;
        rgb2yuv(base[p * 3], base[p * 3 + 1], base[p * 3 + 2], y, u, v);
        ob[p * 3] = y / scale;
        ob[p * 3 + 1] = u / scale;
        ob[p * 3 + 2] = v / scale;
    }
}

static int grid1d(size_t n, int per_block = 256, int cap = 4096) {
    long g = (long)((n + per_block - 1) / per_block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int shm_rgb2yuv_std(const float* rgb, float* yuv, double* acc, float* scale_out, int batch, size_t npix, void* stream) {
    if (batch == 0 || npix == 0) return SHM_OK;
    hipStream_t st = (hipStream_t)stream;
    int r = shm_zero(acc, (size_t)batch * 2 * sizeof(double), stream);
    if (r) return r;
    dim3 grid(grid1d(npix, 256, 512), batch);
    // statistics pass: every block ends in two f64 atomics on its sample's two sums -- 32 blocks per sample, not 256 (the adds on one
    // address serialize: 39 us per launch for 6 MB of pixels)
    dim3 grids(grid1d(npix, 256, 32), batch);
    hipLaunchKernelGGL(yuv_stats_kernel, grids, dim3(256), 0, st, rgb, acc, npix);
    SHM_LAUNCH_CHECK("shm_rgb2yuv_std(stats)");
    hipLaunchKernelGGL(yuv_scale_kernel, grid, dim3(256), 0, st, rgb, yuv, (const double*)acc, scale_out, npix);
    SHM_LAUNCH_CHECK("shm_rgb2yuv_std(scale)");
    return SHM_OK;
}

__global__ void avg_cbcr_kernel(const float* __restrict__ y0, const float* __restrict__ y1, const float* __restrict__ y2, const float* __restrict__ y3,
                                const float* __restrict__ y4, float* __restrict__ out, size_t n) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    out[p * 2] = (y0[p * 3 + 1] + y1[p * 3 + 1] + y2[p * 3 + 1] + y3[p * 3 + 1] + y4[p * 3 + 1]) / 5.0f;
    out[p * 2 + 1] = (y0[p * 3 + 2] + y1[p * 3 + 2] + y2[p * 3 + 2] + y3[p * 3 + 2] + y4[p * 3 + 2]) / 5.0f;
}

extern "C" int shm_avg_cbcr(const float* y0, const float* y1, const float* y2, const float* y3, const float* y4, float* out, size_t n, void* stream) {
    if (n == 0) return SHM_OK;
    hipLaunchKernelGGL(avg_cbcr_kernel, dim3(shm_cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)stream, y0, y1, y2, y3, y4, out, n);
    SHM_LAUNCH_CHECK("shm_avg_cbcr");
    return SHM_OK;
}

// ------------------------------------------------------------------- generator input planes
struct Five {
    const float* p[5];
};

// writes the 10 real channels + zero padding up to the pitch ldo (16 floats or 32 bf16: one 64-byte LDS row)
template <typename T>
__global__ void build_gen_input_kernel(Five ys, const float* __restrict__ gen_y, int flags, int mode, T* __restrict__ out, int ldo, int batch, size_t npix) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // (img, p)
    const int nimg = mode ? 5 * batch : batch;
    if (idx >= (size_t)nimg * npix) return;
    size_t p = idx % npix;
    int img = (int)(idx / npix);
    int k = mode ? img / batch : 4;           // target view (one-hot position)
    int b = mode ? img - k * batch : img;
    size_t src = (size_t)b * npix + p;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        bool fl = (flags >> j) & 1;
        if (mode == 0)
            v[j] = fl ? 0.f : ys.p[j][src * 3];
        else
            v[j] = (j == k) ? 0.f : (fl ? gen_y[src] : ys.p[j][src * 3]);
    }
    v[5 + k] = 1.f;
    T* o = out + idx * ldo;
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(o + 4 * q, (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
    for (int q = 4; q < (ldo >> 2); ++q) st4(o + 4 * q, (f32x4){0.f, 0.f, 0.f, 0.f});
}

extern "C" int shm_build_gen_input(const float* y0, const float* y1, const float* y2, const float* y3, const float* y4, const float* gen_y,
                                   int flags_mask, int mode, void* out, int ldo, int batch, size_t npix, int dtype, void* stream) {
    SHM_REQUIRE(mode == 0 || (mode == 1 && gen_y), SHM_E_SHAPE, "shm_build_gen_input: bad mode/gen_y");
    SHM_REQUIRE(ldo >= 16 && ldo % 4 == 0, SHM_E_SHAPE, "shm_build_gen_input: pitch %d must be >= 16 and a multiple of 4", ldo);
    size_t total = (size_t)(mode ? 5 * batch : batch) * npix;
    if (total == 0) return SHM_OK;
    Five f{{y0, y1, y2, y3, y4}};
    SHM_DISPATCH(dtype, "shm_build_gen_input",
                 hipLaunchKernelGGL(build_gen_input_kernel<T>, dim3(shm_cdiv((long)total, 256)), dim3(256), 0, (hipStream_t)stream, f, gen_y, flags_mask, mode,
                                    (T*)out, ldo, batch, npix));
    SHM_LAUNCH_CHECK("shm_build_gen_input");
    return SHM_OK;
}

template <typename T>
__global__ void cyc_input_bwd_kernel(const T* __restrict__ dcyc, int ld, int flags, float* __restrict__ dgen_y, int batch, size_t npix) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // (b, p)
    if (idx >= (size_t)batch * npix) return;
    float s = 0.f;
    for (int k = 0; k < 5; ++k) {
        const T* row = dcyc + ((size_t)k * batch * npix + idx) * ld;
        for (int j = 0; j < 5; ++j)
            if (j != k && ((flags >> j) & 1)) s += (float)row[j];
    }
    dgen_y[idx] += s;
}

extern "C" int shm_cyc_input_bwd(const void* dcyc, int ld, int flags_mask, float* dgen_y, int batch, size_t npix, int dtype, void* stream) {
    size_t total = (size_t)batch * npix;
    if (total == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_cyc_input_bwd",
                 hipLaunchKernelGGL(cyc_input_bwd_kernel<T>, dim3(shm_cdiv((long)total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dcyc, ld, flags_mask,
                                    dgen_y, batch, npix));
    SHM_LAUNCH_CHECK("shm_cyc_input_bwd");
    return SHM_OK;
}

// -------------------------------------------------------------------------- yuv -> rgb
template <typename T>
__device__ __forceinline__ void store_rgb_pad(T* o, int ld, float r, float g, float b) {
    st4(o, (f32x4){r, g, b, 0.f});
    for (int q = 1; q < (ld >> 2); ++q) st4(o + 4 * q, (f32x4){0.f, 0.f, 0.f, 0.f});
}

// The same rows when they are 64 bytes (the MFMA staging pitch: 16 floats / 32 bf16), written by the whole wave: lane l of
// store j covers bytes [1024 j + 16 l, +16) of the wave's 64 consecutive rows, i.e. row 16 j + l / 4, chunk l % 4 -- four fully
// coalesced 1 KiB stores instead of 4 (fp32) or 8 (bf16) per lane that each touch 64 different lines.  Every lane of the wave
// must call it (idx = this lane's row, live = idx < n); r, g, b of dead lanes are ignored.
template <typename T>
__device__ __forceinline__ void store_rgb_rows64(T* base, size_t idx, size_t n, float r, float g, float b) {
    const int lane = threadIdx.x & 63;
    const size_t row0 = idx - lane;                        // the wave's first row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int src = 16 * j + (lane >> 2);
        const float rr = __shfl(r, src, 64), gg = __shfl(g, src, 64), bb = __shfl(b, src, 64);
        u32x4 v = {0u, 0u, 0u, 0u};
        if ((lane & 3) == 0) {
            if constexpr (sizeof(T) == 4) {
                v = (u32x4){__builtin_bit_cast(unsigned, rr), __builtin_bit_cast(unsigned, gg), __builtin_bit_cast(unsigned, bb), 0u};
            } else {
                const unsigned short hr = __builtin_bit_cast(unsigned short, (T)rr), hg = __builtin_bit_cast(unsigned short, (T)gg),
                                     hb = __builtin_bit_cast(unsigned short, (T)bb);
                v = (u32x4){(unsigned)hr | ((unsigned)hg << 16), (unsigned)hb, 0u, 0u};
            }
        }
        if (row0 + src < n) *(u32x4*)((char*)base + (row0 + src) * 64 + (lane & 3) * 16) = v;
    }
}

template <typename T>
__global__ void yuv2rgb_kernel(const float* __restrict__ ych, const float* __restrict__ cbcr, const float* __restrict__ noise, float* __restrict__ rgb,
                               T* __restrict__ dpad, int ldp, int nimg, int batch, size_t npix) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)nimg * npix;
    const bool rows64 = dpad && (size_t)ldp * sizeof(T) == 64 && (((size_t)dpad & 15) == 0);
    const bool live = idx < total;
    if (!live && !rows64) return;
    float r = 0.f, g = 0.f, bl = 0.f;
    if (live) {
        size_t p = idx % npix;
        int b = (int)((idx / npix) % batch);
        float y = ych[idx];
        float u = cbcr[((size_t)b * npix + p) * 2], v = cbcr[((size_t)b * npix + p) * 2 + 1];
        r = y + Y2R_V_R * v;
        g = y + Y2R_U_G * u + Y2R_V_G * v;
        bl = y + Y2R_U_B * u;
        rgb[idx * 3] = r;
        rgb[idx * 3 + 1] = g;
        rgb[idx * 3 + 2] = bl;
        if (dpad && noise) {
            r += noise[idx * 3];
            g += noise[idx * 3 + 1];
            bl += noise[idx * 3 + 2];
        }
    }
    if (rows64) {
        store_rgb_rows64(dpad, idx, total, r, g, bl);
        return;
    }
    if (dpad) store_rgb_pad(dpad + idx * ldp, ldp, r, g, bl);
}

extern "C" int shm_yuv2rgb(const float* ych, const float* cbcr, const float* noise, float* rgb, void* dpad, int ldp, int nimg, int batch, size_t npix,
                           int dtype, void* stream) {
    SHM_REQUIRE(!dpad || (ldp >= 4 && ldp % 4 == 0), SHM_E_SHAPE, "shm_yuv2rgb: pitch %d must be a multiple of 4", ldp);
    SHM_REQUIRE(batch > 0 && nimg % batch == 0, SHM_E_SHAPE, "shm_yuv2rgb: nimg %d not a multiple of batch %d", nimg, batch);
    size_t total = (size_t)nimg * npix;
    if (total == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_yuv2rgb",
                 hipLaunchKernelGGL(yuv2rgb_kernel<T>, dim3(shm_cdiv((long)total, 256)), dim3(256), 0, (hipStream_t)stream, ych, cbcr, noise, rgb, (T*)dpad, ldp,
                                    nimg, batch, npix));
    SHM_LAUNCH_CHECK("shm_yuv2rgb");
    return SHM_OK;
}

template <typename T>
__global__ void pack_rgb16_kernel(const float* __restrict__ rgb, const float* __restrict__ noise, T* __restrict__ dpad, int ldp, size_t n) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool rows64 = (size_t)ldp * sizeof(T) == 64 && (((size_t)dpad & 15) == 0);
    const bool live = idx < n;
    if (!live && !rows64) return;
    float r = 0.f, g = 0.f, b = 0.f;
    if (live) {
        r = rgb[idx * 3], g = rgb[idx * 3 + 1], b = rgb[idx * 3 + 2];
        if (noise) {
            r += noise[idx * 3];
            g += noise[idx * 3 + 1];
            b += noise[idx * 3 + 2];
        }
    }
    if (rows64)
        store_rgb_rows64(dpad, idx, n, r, g, b);
    else
        store_rgb_pad(dpad + idx * ldp, ldp, r, g, b);
}

extern "C" int shm_pack_rgb16(const float* rgb, const float* noise, void* dpad, int ldp, size_t n, int dtype, void* stream) {
    SHM_REQUIRE(ldp >= 4 && ldp % 4 == 0, SHM_E_SHAPE, "shm_pack_rgb16: pitch %d must be a multiple of 4", ldp);
    if (n == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_pack_rgb16",
                 hipLaunchKernelGGL(pack_rgb16_kernel<T>, dim3(shm_cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)stream, rgb, noise, (T*)dpad, ldp, n));
    SHM_LAUNCH_CHECK("shm_pack_rgb16");
    return SHM_OK;
}

template <typename T>
__global__ void rgb16_to_dy_kernel(const T* __restrict__ d16, int ld, float* __restrict__ dy, size_t n, int acc) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    f32x4 v = ld4(d16 + idx * ld);
    float s = v[0] + v[1] + v[2];
    dy[idx] = acc ? dy[idx] + s : s;
}

extern "C" int shm_rgb16_to_dy(const void* d16, int ld, float* dy, size_t n, int accumulate, int dtype, void* stream) {
    SHM_REQUIRE(ld >= 4 && ld % 4 == 0, SHM_E_SHAPE, "shm_rgb16_to_dy: pitch %d must be a multiple of 4", ld);
    if (n == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_rgb16_to_dy",
                 hipLaunchKernelGGL(rgb16_to_dy_kernel<T>, dim3(shm_cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)d16, ld, dy, n, accumulate));
    SHM_LAUNCH_CHECK("shm_rgb16_to_dy");
    return SHM_OK;
}

// ------------------------------------------------------------- discriminator head losses
// one block (64 threads) per D-batch sample; raw loss slots (sums over the batch):
//  0 mse(rf_D1,T)  1 sum_k mse(rf_D3k,T)  2 mean(rf_D1^2)  3 sum_k mean(rf_D3k^2)
//  4 mse(rf_D2,T)  5 sum_k mse(rf_D4k,T)  6 ce_D1  7 sum_k ce_D3k  8 sum_k ce_D4k
__global__ __launch_bounds__(64) void dhead_losses_kernel(const float* __restrict__ rf, const float* __restrict__ cls, double* __restrict__ loss,
                                                          float* __restrict__ drf_d, float* __restrict__ dcls_d, float* __restrict__ drf_g,
                                                          int batch, int np, float T, int xent_mode) {
    const int i = blockIdx.x;
    int group, k = 0;        // 0:D1 1:D3 2:D2 3:D4
    if (i < batch) group = 0;
    else if (i < 6 * batch) { group = 1; k = (i - batch) / batch; }
    else if (i < 7 * batch) group = 2;
    else { group = 3; k = (i - 7 * batch) / batch; }
    const float invB = 1.0f / batch, invnp = 1.0f / np;
    // D-loss coefficient on mse(.,T) (D2: 2/6, D4: 1/6) or mean(.^2) (D1: 2/6, D3: 1/6)
    const float cD = (group == 0 || group == 2) ? (2.0f / 6.0f) : (1.0f / 6.0f);
    const bool real = group >= 2;
    double smse = 0.0, ssq = 0.0;
    for (int p = threadIdx.x; p < np; p += 64) {
        float v = rf[(size_t)i * np + p];
        smse += (double)(v - T) * (double)(v - T);
        ssq += (double)v * (double)v;
        drf_d[(size_t)i * np + p] = cD * 2.0f * (real ? (v - T) : v) * invnp * invB;
        if (i < 6 * batch) drf_g[(size_t)i * np + p] = (1.0f / 6.0f) * 2.0f * (v - T) * invnp * invB;
    }
    smse = shm_wave_sum(smse);
    ssq = shm_wave_sum(ssq);
    if (threadIdx.x == 0) {
        // softmax cross entropy over 5 logits
        float z[5], mx = -3.0e38f;
        for (int j = 0; j < 5; ++j) { z[j] = cls[(size_t)i * 5 + j]; mx = fmaxf(mx, z[j]); }
        double se = 0.0;
        for (int j = 0; j < 5; ++j) se += exp((double)(z[j] - mx));
        double lse = log(se) + mx;
        int lab = (group == 0 || group == 2) ? 4 : k;
        float lw = (group == 0) ? T : 1.0f;                 // D1 labels = [0,0,0,0,T]
        float coef = (group == 0 || group == 1) ? (1.0f / 6.0f) : (group == 3 ? 10.5f : 0.0f);
        double ce = -(double)lw * ((double)z[lab] - lse);
        for (int j = 0; j < 5; ++j) {
            float sm = (float)exp((double)z[j] - lse);
            // SHM_XENT_TF_FUSED: tf.nn.softmax_cross_entropy_with_logits returns backprop = softmax - labels and its registered
            // gradient is grad_loss * backprop (exact only when the labels sum to 1; D1's label row [0,0,0,0,T] does not);
            // SHM_XENT_INTENDED: the true derivative of -sum(labels * log_softmax) = sum(labels) * softmax - labels
            const float onehot = (j == lab ? 1.0f : 0.0f);
            const float g = xent_mode == SHM_XENT_TF_FUSED ? (sm - lw * onehot) : lw * (sm - onehot);
            dcls_d[(size_t)i * 5 + j] = coef * g * invB;
        }
        double m = smse * invnp, q = ssq * invnp;
        if (group == 0) { atomicAdd(&loss[0], m); atomicAdd(&loss[2], q); atomicAdd(&loss[6], ce); }
        else if (group == 1) { atomicAdd(&loss[1], m); atomicAdd(&loss[3], q); atomicAdd(&loss[7], ce); }
        else if (group == 2) { atomicAdd(&loss[4], m); }
        else { atomicAdd(&loss[5], m); atomicAdd(&loss[8], ce); }
    }
}

extern "C" int shm_dhead_losses(const float* rf, const float* cls, double* loss, float* drf_d, float* dcls_d, float* drf_g, int batch, int np, float target, int xent_mode, void* stream) {
    SHM_REQUIRE(xent_mode == SHM_XENT_TF_FUSED || xent_mode == SHM_XENT_INTENDED, SHM_E_SHAPE, "shm_dhead_losses: xent_mode %d is neither SHM_XENT_TF_FUSED nor SHM_XENT_INTENDED", xent_mode);
    if (batch == 0) return SHM_OK;
    int r = shm_zero(loss, 16 * sizeof(double), stream);
    if (r) return r;
    hipLaunchKernelGGL(dhead_losses_kernel, dim3(12 * batch), dim3(64), 0, (hipStream_t)stream, rf, cls, loss, drf_d, dcls_d, drf_g, batch, np, target, xent_mode);
    SHM_LAUNCH_CHECK("shm_dhead_losses");
    return SHM_OK;
}

// ------------------------------------------------------------------------- clip + Adam
__global__ void adam_clip_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g, size_t n,
                                 float alpha, float b1, float b2, float eps, float gscale, const unsigned* __restrict__ abort_word) {
    // a kernel of this step gave up (shm_set_abort_words): its gradients are built on unfinished sums -- apply nothing
    if (abort_word && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float gg = fminf(fmaxf(g[i] * gscale, -1.0f), 1.0f);
        float mm = m[i] + (gg - m[i]) * (1.0f - b1);
        float vv = v[i] + (gg * gg - v[i]) * (1.0f - b2);
        m[i] = mm;
        v[i] = vv;
        w[i] = w[i] - alpha * mm / (sqrtf(vv) + eps);
    }
}

extern "C" int shm_adam_clip(float* w, float* m, float* v, const float* g, size_t n, float alpha, float beta1, float beta2, float eps, float gscale, void* stream) {
    if (n == 0) return SHM_OK;
    hipLaunchKernelGGL(adam_clip_kernel, dim3(grid1d(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, w, m, v, g, n, alpha, beta1, beta2, eps, gscale,
                       shm_abort_dev_word());
    SHM_LAUNCH_CHECK("shm_adam_clip");
    return SHM_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Step-level random draws on the device (SHM.py:352 GaussianNoise(0.1), SHM.py:363 Dropout(0.2)): counter-based
// Philox-4x32-10 (Salmon et al. 2011), one counter per group of four outputs, key = (seed, stream).  The reference draws
// these inside Keras layers from TF's global generator; the values differ from any TF run by construction (the parity tests
// inject explicit draws), the distributions are what the layers specify.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

__global__ void randn_kernel(float* __restrict__ out, size_t n, float stddev, unsigned seed_lo, unsigned seed_hi, unsigned stream_id) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // group of four outputs
    if (i * 4 >= n) return;
    unsigned r[4];
    philox4x32_10((unsigned)i, (unsigned)(i >> 32), stream_id, 0u, seed_lo, seed_hi, r);
    float v[4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {                                              // Box-Muller on two pairs of uniforms in (0, 1]
        const float u1 = ((float)(r[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(r[2 * p + 1] >> 8) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * logf(u1)) * stddev;
        float sn, cs;
        sincosf(6.28318530717958647692f * u2, &sn, &cs);
        v[2 * p] = rad * cs;
        v[2 * p + 1] = rad * sn;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i * 4 + e < n) out[i * 4 + e] = v[e];
}

__global__ void keep_mask_kernel(float* __restrict__ out, size_t n, float rate, unsigned seed_lo, unsigned seed_hi, unsigned stream_id) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 >= n) return;
    unsigned r[4];
    philox4x32_10((unsigned)i, (unsigned)(i >> 32), stream_id, 1u, seed_lo, seed_hi, r);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i * 4 + e < n) out[i * 4 + e] = (float)(r[e] >> 8) * (1.0f / 16777216.0f) >= rate ? 1.f : 0.f;
}

extern "C" int shm_randn(float* out, size_t n, float stddev, unsigned long long seed, unsigned stream_id, void* stream) {
    SHM_REQUIRE(out || n == 0, SHM_E_SHAPE, "shm_randn: null pointer");
    if (n == 0) return SHM_OK;
    hipLaunchKernelGGL(randn_kernel, dim3(shm_cdiv((long)((n + 3) / 4), 256)), dim3(256), 0, (hipStream_t)stream, out, n, stddev, (unsigned)seed,
                       (unsigned)(seed >> 32), stream_id);
    SHM_LAUNCH_CHECK("shm_randn");
    return SHM_OK;
}

extern "C" int shm_keep_mask(float* out, size_t n, float rate, unsigned long long seed, unsigned stream_id, void* stream) {
    SHM_REQUIRE(out || n == 0, SHM_E_SHAPE, "shm_keep_mask: null pointer");
    SHM_REQUIRE(rate >= 0.f && rate < 1.f, SHM_E_SHAPE, "shm_keep_mask: rate %g outside [0,1)", (double)rate);
    if (n == 0) return SHM_OK;
    hipLaunchKernelGGL(keep_mask_kernel, dim3(shm_cdiv((long)((n + 3) / 4), 256)), dim3(256), 0, (hipStream_t)stream, out, n, rate, (unsigned)seed,
                       (unsigned)(seed >> 32), stream_id);
    SHM_LAUNCH_CHECK("shm_keep_mask");
    return SHM_OK;
}
