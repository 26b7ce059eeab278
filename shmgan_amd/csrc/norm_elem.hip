// HBM-bound kernels around the convolutions: InstanceNormalization statistics / apply /
// backward (fused with LeakyReLU' and the AveragePooling2D gradient), pooling, the
// 1-output-channel layers (generator head, PatchGAN logits, Dense(5)), dropout mask.
//
// Layout: NHWC with a channel pitch; every thread moves 16 bytes (4 channels of one
// pixel); a block covers PP = 256/(C/4) pixels per iteration, so a wave reads whole
// contiguous channel rows.  Per-(sample,channel) sums are accumulated in fp64 per thread,
// combined through LDS and added with one f64 atomic per (block, channel).
#include "common.h"

#include <stdarg.h>

// ---------------------------------------------------------------------------- core API
static thread_local char g_err[512] = "";

void shm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_kernel[128] = "";

void shm_set_last_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

// ---- dispatch tuning (shm_set_tuning / shm_get_tuning) -------------------------------------------------
#include <atomic>
#include <stdlib.h>
#include <string.h>
namespace {
struct TuneDef {
    const char* key;
    const char* env;       // initial value (read once) -- keeps the ablation tools' environment knobs alive
    int dflt, lo, hi;
};
const TuneDef kTune[SHM_TUNE_COUNT] = {
    {"tapgemm.variant", "SHM_TAPGEMM_VARIANT", 0, 0, SHM_TG_COUNT - 1},
    {"tapgemm.halo_min_blocks", "SHM_TAPGEMM_HALO_MIN", 1024, 0, 1 << 30},
    {"tapgemm.small_grid_blocks", "SHM_TAPGEMM_SMALLM", 1024, 0, 1 << 30},
    {"tapgemm.phase4_min_blocks", "SHM_TAPGEMM_PHASE4_MIN", 256, 0, 1 << 30},
    {"wgrad.variant", "SHM_WGRAD_VARIANT", 0, 0, 3},
    {"wgrad.blocks", "SHM_WGRAD_BLOCKS", 0, 0, 1 << 20},
    {"wgrad.bf16_rows", "SHM_WGRAD_BF16_ROWS", 0, 0, 4},
    {"stats.fusion", "SHM_STATS_FUSION", 1, 0, 1},
    {"elem.reverse", "SHM_ELEM_REVERSE", 1, 0, 1},
    {"elem.reduce_blocks", "SHM_ELEM_REDUCE_BLOCKS", 0, 0, 1 << 20},
    {"elem.nt_loads", "SHM_ELEM_NT", 0, 0, 1},
    {"elem.chunk_mb", "SHM_ELEM_CHUNK_MB", 0, 0, 1 << 20},
    {"elem.interleave", "SHM_ELEM_INTERLEAVE", 1, 0, 1},
    {"elem.stream_blocks", "SHM_ELEM_STREAM_BLOCKS", 32768, 256, 1 << 20},
    {"elem.apply_blocks", "SHM_ELEM_APPLY_BLOCKS", 4096, 256, 1 << 20},
    {"tapgemm.wreg16", "SHM_TAPGEMM_WREG16", 2, 0, 2},
    {"wgrad.bf16_wide", "SHM_WGRAD_BF16_WIDE", 0, 0, 4},
    {"wgrad.f32_split", "SHM_WGRAD_F32_SPLIT", 0, 0, 1},
    {"tapgemm.flat_epilogue", "SHM_TAPGEMM_FLAT_EPILOGUE", 0, 0, 1},
    {"elem.fused_bwd", "SHM_ELEM_FUSED_BWD", 1, 0, 1},
    {"elem.fused_max_slices", "SHM_ELEM_FUSED_MAX_SLICES", 256, 1, 512},
    {"conv.f32_split", "SHM_CONV_F32_SPLIT", 0, 0, 1},
    {"elem.fused_test_stall", "SHM_ELEM_FUSED_TEST_STALL", 0, 0, 1},
    {"elem.fused_hold", "SHM_ELEM_FUSED_HOLD", 0, 0, 2},
    {"elem.fused_gvariant", "SHM_ELEM_FUSED_GVARIANT", 0, 0, 1},
};
std::atomic<int> g_tune[SHM_TUNE_COUNT];
std::atomic<int> g_tune_init{0};
void tune_init() {
    if (g_tune_init.load(std::memory_order_acquire) == 2) return;
    int expect = 0;
    if (g_tune_init.compare_exchange_strong(expect, 1)) {
        for (int i = 0; i < SHM_TUNE_COUNT; ++i) {
            const char* e = getenv(kTune[i].env);
            int v = e ? atoi(e) : kTune[i].dflt;
            if (v < kTune[i].lo || v > kTune[i].hi) v = kTune[i].dflt;
            g_tune[i].store(v);
        }
        g_tune_init.store(2, std::memory_order_release);
    } else {
        while (g_tune_init.load(std::memory_order_acquire) != 2) {
        }
    }
}
int tune_find(const char* key) {
    if (!key) return -1;
    for (int i = 0; i < SHM_TUNE_COUNT; ++i)
        if (strcmp(key, kTune[i].key) == 0) return i;
    return -1;
}
}  // namespace

int shm_tune(int id) {
    tune_init();
    return g_tune[id].load(std::memory_order_relaxed);
}

extern "C" int shm_set_tuning(const char* key, int value) {
    tune_init();
    if (key && strcmp(key, "reset") == 0) {            // every knob back to its built-in default
        for (int i = 0; i < SHM_TUNE_COUNT; ++i) g_tune[i].store(kTune[i].dflt);
        return SHM_OK;
    }
    const int i = tune_find(key);
    SHM_REQUIRE(i >= 0, SHM_E_SHAPE, "shm_set_tuning: unknown key '%s'", key ? key : "(null)");
    if (value < 0) value = kTune[i].dflt;                // negative = default
    SHM_REQUIRE(value >= kTune[i].lo && value <= kTune[i].hi, SHM_E_SHAPE, "shm_set_tuning: %s = %d outside [%d, %d]", key, value,
                kTune[i].lo, kTune[i].hi);
    g_tune[i].store(value);
    return SHM_OK;
}

extern "C" int shm_get_tuning(const char* key, int* value) {
    tune_init();
    const int i = tune_find(key);
    SHM_REQUIRE(i >= 0 && value, SHM_E_SHAPE, "shm_get_tuning: unknown key '%s'", key ? key : "(null)");
    *value = g_tune[i].load();
    return SHM_OK;
}

extern "C" const char* shm_last_error(void) { return g_err; }
extern "C" const char* shm_last_kernel(void) { return g_kernel; }
extern "C" int shm_version(void) { return 200; }

extern "C" int shm_zero(void* p, size_t bytes, void* stream) {
    if (bytes == 0) return SHM_OK;
    hipError_t e = hipMemsetAsync(p, 0, bytes, (hipStream_t)stream);
    SHM_REQUIRE(e == hipSuccess, SHM_E_HIP, "shm_zero: %s", hipGetErrorString(e));
    return SHM_OK;
}

__global__ void cvt_f64_f32_kernel(const double* __restrict__ s, float* __restrict__ d, size_t n, int acc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (acc ? d[i] : 0.f) + (float)s[i];
}

extern "C" int shm_cvt_f64_f32(const double* src, float* dst, size_t n, int accumulate, void* stream) {
    if (n == 0) return SHM_OK;
    hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3(shm_cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n, accumulate);
    SHM_LAUNCH_CHECK("shm_cvt_f64_f32");
    return SHM_OK;
}

// f32 -> activation dtype copy (bf16 operand copies of the fp32 master weights)
template <typename T>
__global__ void cast_f32_kernel(const float* __restrict__ s, T* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (T)s[i];
}

extern "C" int shm_cast_f32(const float* src, void* dst, size_t n, int dtype, void* stream) {
    if (n == 0) return SHM_OK;
    long blocks = (long)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    SHM_DISPATCH(dtype, "shm_cast_f32", hipLaunchKernelGGL(cast_f32_kernel<T>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, n));
    SHM_LAUNCH_CHECK("shm_cast_f32");
    return SHM_OK;
}

// ------------------------------------------------------------------ pixel-chunk skeleton
// thread -> (pp, cl): pixel slot and 4-channel lane.  PP pixel slots per block iteration.
struct PixMap {
    int lanes_c, PP, pp, cl;
    bool active;
    __device__ PixMap(int c) {
        lanes_c = c >> 2;
        PP = 256 / lanes_c;
        pp = threadIdx.x / lanes_c;
        cl = threadIdx.x - pp * lanes_c;
        active = pp < PP;
    }
};

// blocks = 0: the target of the passes without a per-block prologue or reduction (InstanceNorm apply, its pooling forms),
// "elem.stream_blocks", default 32768: short blocks keep the addresses in flight a narrow band that sweeps through the tensors
// (tools/probes/elem_probe.hip: a 2-read / 1-write pass over 3 x 671 MB runs at 5.5 TB/s with 4k blocks of 160 KB each and at 7.0 TB/s with
// 16k blocks of 40 KB; shm_in_apply on the same tensor 5.16 -> 5.85 TB/s in fp32, 5.24 -> 6.01 in bf16).  The passes that start with
// a per-block prologue and end in an LDS reduction + atomics (InstanceNorm backward) keep 4096: they get SLOWER with more blocks.
static int pix_chunks(long npix_per_sample, int batch, int c, int blocks = 0) {
    // enough blocks to fill the chip, but at least 8 (streaming target) / 16 pixel iterations per thread so the
    // per-block LDS reduction + f64 atomics (one per channel and block) stay a small fraction
    if (blocks == 0) blocks = shm_tune(SHM_TUNE_ELEM_STREAM_BLOCKS);
    const int min_iter = blocks > 4096 ? 8 : 16;
    int lanes_c = c / 4;
    int PP = 256 / lanes_c;
    long want = (blocks + batch - 1) / batch;
    long maxc = npix_per_sample / ((long)PP * min_iter);
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

#define SHM_CHECK_C(c, who) SHM_REQUIRE((c) % 4 == 0 && (c) >= 4 && (c) <= 1024, SHM_E_SHAPE, "%s: channels %d must be a multiple of 4 in [4,1024]", who, (c))

// Combine per-thread double[NV][4] partials over the PP pixel slots, then one atomic per
// (channel, value).  dst index = base + (ch * NV + v) when interleaved, or v*c + ch otherwise.
template <int NV>
__device__ __forceinline__ void block_reduce_atomic(double (&v)[NV][4], const PixMap& pm, double* dst, int c, bool interleaved) {
    __shared__ double red[256 * 4];
    for (int q = 0; q < NV; ++q) {
        __syncthreads();
        if (pm.active) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(pm.pp * pm.lanes_c + pm.cl) * 4 + e] = v[q][e];
        }
        __syncthreads();
        if (pm.active && pm.pp == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double s = 0.0;
                for (int p = 0; p < pm.PP; ++p) s += red[(p * pm.lanes_c + pm.cl) * 4 + e];
                int ch = pm.cl * 4 + e;
                if (ch < c) atomicAdd(&dst[interleaved ? ch * NV + q : q * c + ch], s);
            }
        }
    }
}

// ------------------------------------------------------------------------- IN statistics
template <typename T>
__global__ __launch_bounds__(256) void in_stats_kernel(const T* __restrict__ a, int lda, double* __restrict__ stats, int hw, int c, int chunk) {
    PixMap pm(c);
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * chunk, p1 = min(hw, p0 + chunk);
    double v[2][4] = {};
    if (pm.active) {
        const T* base = a + (size_t)n * hw * lda + pm.cl * 4;
        for (int p = p0 + pm.pp; p < p1; p += pm.PP) {
            f32x4 x = ld4(base + (size_t)p * lda);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[0][e] += (double)x[e];
                v[1][e] += (double)x[e] * (double)x[e];
            }
        }
    }
    block_reduce_atomic<2>(v, pm, stats + (size_t)n * c * 2, c, true);
}

// part != null: the sums were accumulated over `nslot` slot copies part[slot][total][2]; the copies are zeroed
// again as they are consumed, so the scratch is zero whenever no call is in flight (no memset per launch)
// nt != null: also the float table [batch][4][c] = (mean, inv, beta, ring) of the consumers that normalise on the fly (common.h)
__global__ void in_finalize_kernel(double* __restrict__ stats, double* __restrict__ part, int nslot, int total, int hw, double eps, float* __restrict__ nt,
                                   const float* __restrict__ beta, int c) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    double s, q;
    if (part) {
        s = q = 0.0;
        for (int k = 0; k < nslot; ++k) {
            double* pk = part + ((size_t)k * total + i) * 2;
            s += pk[0];
            q += pk[1];
            pk[0] = 0.0;
            pk[1] = 0.0;
        }
    } else {
        s = stats[2 * i];
        q = stats[2 * i + 1];
    }
    double mean = s / hw;
    double var = q / hw - mean * mean;
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + eps);
    stats[2 * i] = mean;
    stats[2 * i + 1] = inv;
    if (nt) {
        const int n = i / c, ch = i - n * c;
        float* t = nt + (size_t)n * SHM_NT_PLANES * c + ch;
        t[0] = (float)mean;
        t[c] = (float)inv;
        t[2 * c] = beta[ch];
        t[3 * c] = (float)mean - beta[ch] / (float)inv;
    }
}

int shm_in_finalize_internal(double* stats, double* part, int nslot, int total, int hw, double eps, float* nt, const float* beta, int c, hipStream_t st) {
    hipLaunchKernelGGL(in_finalize_kernel, dim3(shm_cdiv((long)total, 256)), dim3(256), 0, st, stats, part, nslot, total, hw, eps, nt, beta, c);
    SHM_LAUNCH_CHECK("shm_in_finalize");
    return SHM_OK;
}

// the table alone, from finalized statistics (mean, inv)
__global__ void in_norm_table_kernel(const double* __restrict__ stats, const float* __restrict__ beta, float* __restrict__ nt, int total, int c) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = i / c, ch = i - n * c;
    float* t = nt + (size_t)n * SHM_NT_PLANES * c + ch;
    t[0] = (float)stats[2 * i];
    t[c] = (float)stats[2 * i + 1];
    t[2 * c] = beta[ch];
    t[3 * c] = (float)stats[2 * i] - beta[ch] / (float)stats[2 * i + 1];
}

extern "C" int shm_in_norm_table(const double* stats, const float* beta, float* nt, int batch, int c, void* stream) {
    SHM_REQUIRE(stats && beta && nt, SHM_E_SHAPE, "shm_in_norm_table: null pointer");
    SHM_REQUIRE(c % 4 == 0 && c > 0, SHM_E_SHAPE, "shm_in_norm_table: channels %d must be a positive multiple of 4", c);
    if (batch == 0) return SHM_OK;
    hipLaunchKernelGGL(in_norm_table_kernel, dim3(shm_cdiv((long)batch * c, 256)), dim3(256), 0, (hipStream_t)stream, stats, beta, nt, batch * c, c);
    SHM_LAUNCH_CHECK("shm_in_norm_table");
    return SHM_OK;
}

extern "C" int shm_in_stats(const void* a, int lda, double* stats, int batch, int hw, int c, float eps, int dtype, void* stream) {
    SHM_CHECK_C(c, "shm_in_stats");
    SHM_REQUIRE(lda % 4 == 0 && lda >= c, SHM_E_SHAPE, "shm_in_stats: bad pitch %d", lda);
    hipStream_t st = (hipStream_t)stream;
    if (batch == 0 || hw == 0) return SHM_OK;
    int r = shm_zero(stats, (size_t)batch * c * 2 * sizeof(double), stream);
    if (r) return r;
    int nch = pix_chunks(hw, batch, c, 4096);
    int chunk = shm_cdiv(hw, nch);
    SHM_DISPATCH(dtype, "shm_in_stats",
                 hipLaunchKernelGGL(in_stats_kernel<T>, dim3(shm_cdiv(hw, chunk), batch), dim3(256), 0, st, (const T*)a, lda, stats, hw, c, chunk));
    SHM_LAUNCH_CHECK("shm_in_stats");
    hipLaunchKernelGGL(in_finalize_kernel, dim3(shm_cdiv((long)batch * c, 256)), dim3(256), 0, st, stats, (double*)nullptr, 0, batch * c, hw, (double)eps,
                       (float*)nullptr, (const float*)nullptr, c);
    SHM_LAUNCH_CHECK("shm_in_stats(finalize)");
    return SHM_OK;
}

// rev: walk the tensor back to front.  The producing convolution wrote the samples in ascending order, so the LAST ones are
// still in the 256 MiB Infinity Cache: reading them first turns up to 256 MiB of this pass's reads into cache hits (front to
// back, an LRU cache smaller than the tensor yields none), and it leaves sample 0 written last -- where the
// consuming convolution starts.
template <typename T>
__global__ __launch_bounds__(256) void in_apply_kernel(const T* __restrict__ a, int lda, const double* __restrict__ stats, const float* __restrict__ beta,
                                                       T* __restrict__ out, int ldo, int hw, int c, int chunk, int rev) {
    PixMap pm(c);
    if (!pm.active) return;
    const int n = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
    const int bx = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int p0 = bx * chunk, p1 = min(hw, p0 + chunk);
    float mean[4], inv[4], bt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int ch = pm.cl * 4 + e;
        mean[e] = (float)stats[((size_t)n * c + ch) * 2];
        inv[e] = (float)stats[((size_t)n * c + ch) * 2 + 1];
        bt[e] = beta[ch];
    }
    const T* base = a + (size_t)n * hw * lda + pm.cl * 4;
    T* ob = out + (size_t)n * hw * ldo + pm.cl * 4;
    constexpr int U = sizeof(T) == 2 ? 8 : 4;
    int p = p0 + pm.pp;
    for (; p + (U - 1) * pm.PP < p1; p += U * pm.PP) {
        f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld4(base + (size_t)(p + u * pm.PP) * lda);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = shm_in_norm(x[u][e], mean[e], inv[e], bt[e]);
            st4(ob + (size_t)(p + u * pm.PP) * ldo, y);
        }
    }
    for (; p < p1; p += pm.PP) {
        f32x4 x = ld4(base + (size_t)p * lda);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = shm_in_norm(x[e], mean[e], inv[e], bt[e]);
        st4(ob + (size_t)p * ldo, y);
    }
}

extern "C" int shm_in_apply(const void* a, int lda, const double* stats, const float* beta, void* out, int ldo, int batch, int hw, int c, int dtype,
                            void* stream) {
    SHM_CHECK_C(c, "shm_in_apply");
    SHM_REQUIRE(lda % 4 == 0 && ldo % 4 == 0, SHM_E_SHAPE, "shm_in_apply: bad pitch");
    if (batch == 0 || hw == 0) return SHM_OK;
    int nch = pix_chunks(hw, batch, c);
    int chunk = shm_cdiv(hw, nch);
    SHM_DISPATCH(dtype, "shm_in_apply",
                 hipLaunchKernelGGL(in_apply_kernel<T>, dim3(shm_cdiv(hw, chunk), batch), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, stats, beta,
                                    (T*)out, ldo, hw, c, chunk, shm_tune(SHM_TUNE_ELEM_REVERSE)));
    SHM_LAUNCH_CHECK("shm_in_apply");
    return SHM_OK;
}

// InstanceNorm apply + AveragePooling2D(2) in one pass (the second block of every encoder level feeds both the skip and the
// pool): a thread normalises the four pixels of a 2 x 2 quad for its four channels, writes them, and writes their mean -- the
// pooled tensor is formed from the values as stored (rounded to T), in avgpool2_kernel's order, so it is bit-identical to
// shm_in_apply followed by shm_avgpool2_fwd; the separate pooling pass (a full read of the normalised tensor) is gone.
// OUT = false (shm_in_pool): only the pooled tensor is written -- the skip connection's consumers normalise the stored activation
// on the fly (shm_conv2d_in_fwd_norm / shm_conv2d_wgrad_norm); the pooled values are the same bits as with OUT = true.
template <typename T, bool OUT = true>
__global__ __launch_bounds__(256) void in_apply_pool_kernel(const T* __restrict__ a, int lda, const double* __restrict__ stats, const float* __restrict__ beta,
                                                            T* __restrict__ out, int ldo, T* __restrict__ pooled, int ldp, int h, int w, int c, int chunk,
                                                            int rev) {
    PixMap pm(c);
    if (!pm.active) return;
    const int n = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
    const int bx = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int wo = w >> 1, hq = (h >> 1) * wo;
    const int q0 = bx * chunk, q1 = min(hq, q0 + chunk);
    float mean[4], inv[4], bt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int ch = pm.cl * 4 + e;
        mean[e] = (float)stats[((size_t)n * c + ch) * 2];
        inv[e] = (float)stats[((size_t)n * c + ch) * 2 + 1];
        bt[e] = beta[ch];
    }
    const T* base = a + (size_t)n * h * w * lda + pm.cl * 4;
    T* ob = out + (size_t)n * h * w * ldo + pm.cl * 4;
    T* pb = pooled + (size_t)n * hq * ldp + pm.cl * 4;
    constexpr int U = 2;
    auto quad = [&](int q, f32x4 (&x)[4]) {
        const int oy = q / wo, ox = q - oy * wo;
        const size_t p = (size_t)(2 * oy) * w + 2 * ox;
        x[0] = ld4(base + p * lda);
        x[1] = ld4(base + (p + 1) * lda);
        x[2] = ld4(base + (p + w) * lda);
        x[3] = ld4(base + (p + w + 1) * lda);
    };
    auto finish = [&](int q, const f32x4 (&x)[4]) {
        const int oy = q / wo, ox = q - oy * wo;
        const size_t p = (size_t)(2 * oy) * w + 2 * ox;
        f32x4 y[4], s;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) y[t][e] = rnd_as((const T*)nullptr, shm_in_norm(x[t][e], mean[e], inv[e], bt[e]));
        if constexpr (OUT) {
            st4(ob + p * ldo, y[0]);
            st4(ob + (p + 1) * ldo, y[1]);
            st4(ob + (p + w) * ldo, y[2]);
            st4(ob + (p + w + 1) * ldo, y[3]);
        }
        s = ((y[0] + y[1]) + y[2]) + y[3];
        st4(pb + (size_t)q * ldp, s * 0.25f);
    };
    int q = q0 + pm.pp;
    for (; q + (U - 1) * pm.PP < q1; q += U * pm.PP) {
        f32x4 x[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) quad(q + u * pm.PP, x[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) finish(q + u * pm.PP, x[u]);
    }
    for (; q < q1; q += pm.PP) {
        f32x4 x[4];
        quad(q, x);
        finish(q, x);
    }
}

extern "C" int shm_in_apply_pool(const void* a, int lda, const double* stats, const float* beta, void* out, int ldo, void* pooled, int ldp, int batch,
                                 int h, int w, int c, int dtype, void* stream) {
    SHM_CHECK_C(c, "shm_in_apply_pool");
    SHM_REQUIRE(lda % 4 == 0 && ldo % 4 == 0 && ldp % 4 == 0, SHM_E_SHAPE, "shm_in_apply_pool: bad pitch");
    SHM_REQUIRE(h % 2 == 0 && w % 2 == 0, SHM_E_SHAPE, "shm_in_apply_pool: odd size %dx%d", h, w);
    SHM_REQUIRE(a && stats && beta && out && pooled, SHM_E_SHAPE, "shm_in_apply_pool: null pointer");
    if (batch == 0 || h * w == 0) return SHM_OK;
    const int hq = (h / 2) * (w / 2);
    int nch = pix_chunks(hq, batch, c);
    int chunk = shm_cdiv(hq, nch);
    SHM_DISPATCH(dtype, "shm_in_apply_pool",
                 hipLaunchKernelGGL((in_apply_pool_kernel<T, true>), dim3(shm_cdiv(hq, chunk), batch), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, stats, beta,
                                    (T*)out, ldo, (T*)pooled, ldp, h, w, c, chunk, shm_tune(SHM_TUNE_ELEM_REVERSE)));
    SHM_LAUNCH_CHECK("shm_in_apply_pool");
    return SHM_OK;
}

extern "C" int shm_in_pool(const void* a, int lda, const double* stats, const float* beta, void* pooled, int ldp, int batch, int h, int w, int c, int dtype,
                           void* stream) {
    SHM_CHECK_C(c, "shm_in_pool");
    SHM_REQUIRE(lda % 4 == 0 && ldp % 4 == 0, SHM_E_SHAPE, "shm_in_pool: bad pitch");
    SHM_REQUIRE(h % 2 == 0 && w % 2 == 0, SHM_E_SHAPE, "shm_in_pool: odd size %dx%d", h, w);
    SHM_REQUIRE(a && stats && beta && pooled, SHM_E_SHAPE, "shm_in_pool: null pointer");
    if (batch == 0 || h * w == 0) return SHM_OK;
    const int hq = (h / 2) * (w / 2);
    int nch = pix_chunks(hq, batch, c);
    int chunk = shm_cdiv(hq, nch);
    SHM_DISPATCH(dtype, "shm_in_pool",
                 hipLaunchKernelGGL((in_apply_pool_kernel<T, false>), dim3(shm_cdiv(hq, chunk), batch), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, stats,
                                    beta, (T*)nullptr, 0, (T*)pooled, ldp, h, w, c, chunk, shm_tune(SHM_TUNE_ELEM_REVERSE)));
    SHM_LAUNCH_CHECK("shm_in_pool");
    return SHM_OK;
}

// --------------------------------------------------------------------------- IN backward
struct InBwdArgs {               // g1, g2, a, dz: tensors of the kernels' element type T
    const void* g1;
    const void* g2;
    const void* a;
    const double* stats;
    double* red;
    void* dz;
    double* dbias;
    int ldg1, ldg2, lda, lddz;
    int h, w, c, chunk;
    float slope;
    int rev;
    const float* r1_dz;          // rank-1 gradient (R1 kernels): d_out[n, p, ch] = r1_dz[n * hw + p] * r1_w[ch] -- the generator head's
    const float* r1_w;           // input gradient, formed on the fly instead of being written by the head and read twice here
    // RAW apply kernels (shm_in_bwd_apply): the sums come from the epilogues of the launches that wrote g1 / g2 (gsum), as slot
    // copies [gslots][batch][c][2]: gred = (sum g1, sum g1 * a), gredp = (sum g2, sum g2 * pooled) or null; dstage = f64 [batch][c]
    // staging of the bias gradient
    const double* gred;
    const double* gredp;
    const float* beta;
    double* dstage;
    int gslots;
    int nt;                      // apply pass: g1 is read for the last time -> non-temporal loads
    int n0, nbatch;              // sample chunking (in_bwd_impl): this launch covers samples [n0, n0 + gridDim.y) of nbatch
    int interleave;              // apply pass: tiles of pixels dealt round-robin over a sample's blocks ("elem.interleave")
    int fold;                    // one-pass kernels: the launch's last group folds the staged bias gradient into dbias itself
};

// G2 is a template parameter: a run-time `if (k.g2)` between the loads makes hipcc wait for each load
// before the branch (s_waitcnt vmcnt(0) + s_cbranch per pixel), which serialises the whole stream
// (measured 2.0 TB/s instead of 5+).
template <typename TG, bool G2, bool R1 = false>
__device__ __forceinline__ f32x4 in_bwd_dout(const InBwdArgs& k, int n, int p, int cl, const f32x4& wv = f32x4{0.f, 0.f, 0.f, 0.f}) {
    if constexpr (R1) {
        const float d = k.r1_dz[(size_t)n * k.h * k.w + p];
        return wv * d;
    }
    f32x4 g = k.nt ? ld4nt((const TG*)k.g1 + ((size_t)n * k.h * k.w + p) * k.ldg1 + cl * 4)
                   : ld4((const TG*)k.g1 + ((size_t)n * k.h * k.w + p) * k.ldg1 + cl * 4);
    if constexpr (G2) {
        int y = p / k.w, x = p - y * k.w;
        size_t q = ((size_t)n * (k.h >> 1) + (y >> 1)) * (k.w >> 1) + (x >> 1);
        f32x4 u = ld4((const TG*)k.g2 + q * k.ldg2 + cl * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] += 0.25f * u[e];
    }
    return g;
}

// The reduce pass walks the tensor back to front when k.rev is set (the input-gradient product that wrote g1 went front to back:
// its last samples are still in the Infinity Cache), the apply pass that follows front to back again (it starts where the
// reduce pass ended).
template <typename T, typename TG, bool G2, bool R1 = false>
__global__ __launch_bounds__(256) void in_bwd_reduce_kernel(const InBwdArgs k) {
    PixMap pm(k.c);
    f32x4 wr = {0.f, 0.f, 0.f, 0.f};
    if constexpr (R1) {
        if (pm.active) wr = *(const f32x4*)(k.r1_w + pm.cl * 4);
    }
    const int n = k.n0 + (k.rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y), hw = k.h * k.w;
    const int bx = k.rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int p0 = bx * k.chunk, p1 = min(hw, p0 + k.chunk);
    double v[2][4] = {};
    if (pm.active) {
        float mean[4], inv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mean[e] = (float)k.stats[((size_t)n * k.c + pm.cl * 4 + e) * 2];
            inv[e] = (float)k.stats[((size_t)n * k.c + pm.cl * 4 + e) * 2 + 1];
        }
        // U pixels per iteration: the kernel is bound by bytes in flight, not by arithmetic -- 4 pixels of
        // 16-byte loads in fp32, 8 pixels of 8-byte loads in bf16 keep the same 8-12 x 16 B outstanding
        // per thread; the per-pixel partial sums are combined in fp32 before the fp64 accumulation
        constexpr int U = sizeof(T) == 2 ? 8 : 4;
        int p = p0 + pm.pp;
        for (; p + (U - 1) * pm.PP < p1; p += U * pm.PP) {
            f32x4 g[U], x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g[u] = in_bwd_dout<TG, G2, R1>(k, n, p + u * pm.PP, pm.cl, wr);
                x[u] = ld4((const T*)k.a + ((size_t)n * hw + p + u * pm.PP) * k.lda + pm.cl * 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sg = 0.f, sx = 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float xh = (x[u][e] - mean[e]) * inv[e];
                    sg += g[u][e];
                    sx += g[u][e] * xh;
                }
                v[0][e] += (double)sg;
                v[1][e] += (double)sx;
            }
        }
        for (; p < p1; p += pm.PP) {
            f32x4 g = in_bwd_dout<TG, G2, R1>(k, n, p, pm.cl, wr);
            f32x4 x = ld4((const T*)k.a + ((size_t)n * hw + p) * k.lda + pm.cl * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xh = (x[e] - mean[e]) * inv[e];
                v[0][e] += (double)g[e];
                v[1][e] += (double)g[e] * (double)xh;
            }
        }
    }
    block_reduce_atomic<2>(v, pm, k.red + (size_t)n * k.c * 2, k.c, true);
}

// bf16 form of the reduce pass with EIGHT channels (16 bytes) per thread: with four (8-byte loads) the pass reached 2.2-2.8 TB/s
// where its fp32 twin, whose four channels are 16 bytes, reaches 4.0 (rocprofv3, profiles/r02_*): the loads per wave are what
// limits a read-only stream.  Same sums, same scratch layout as in_bwd_reduce_kernel.
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x8 ld8(const bf16_t* p) {
    const uint4 u = *(const uint4*)p;
    f32x8 r;
    r[0] = __uint_as_float(u.x << 16);
    r[1] = __uint_as_float(u.x & 0xffff0000u);
    r[2] = __uint_as_float(u.y << 16);
    r[3] = __uint_as_float(u.y & 0xffff0000u);
    r[4] = __uint_as_float(u.z << 16);
    r[5] = __uint_as_float(u.z & 0xffff0000u);
    r[6] = __uint_as_float(u.w << 16);
    r[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
__device__ __forceinline__ f32x8 ld8(const float* p) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <typename TG, bool G2, bool R1 = false>
__global__ __launch_bounds__(256) void in_bwd_reduce8_kernel(const InBwdArgs k) {
    __shared__ double red[256 * 8];
    const int lanes_c = k.c >> 3, PP = 256 / lanes_c;
    const int pp = threadIdx.x / lanes_c, cl = threadIdx.x - pp * lanes_c;
    const bool active = pp < PP;
    const int n = k.n0 + (k.rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y), hw = k.h * k.w;
    const int bx = k.rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int p0 = bx * k.chunk, p1 = min(hw, p0 + k.chunk);
    double v[2][8] = {};
    if (active) {
        float mean[8], inv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mean[e] = (float)k.stats[((size_t)n * k.c + cl * 8 + e) * 2];
            inv[e] = (float)k.stats[((size_t)n * k.c + cl * 8 + e) * 2 + 1];
        }
        f32x8 wr8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (R1) wr8 = ld8(k.r1_w + cl * 8);
        auto dout = [&](int p) {
            if constexpr (R1) {
                const float d = k.r1_dz[(size_t)n * hw + p];
                return wr8 * d;
            }
            f32x8 g = ld8((const TG*)k.g1 + ((size_t)n * hw + p) * k.ldg1 + cl * 8);
            if constexpr (G2) {
                const int y = p / k.w, x = p - y * k.w;
                const size_t q = ((size_t)n * (k.h >> 1) + (y >> 1)) * (k.w >> 1) + (x >> 1);
                const f32x8 u = ld8((const TG*)k.g2 + q * k.ldg2 + cl * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] += 0.25f * u[e];
            }
            return g;
        };
        constexpr int U = 4;
        int p = p0 + pp;
        // k.interleave ("elem.interleave"): a sample's blocks take tiles of U * PP pixels round-robin (back to front under k.rev) instead of
        // one contiguous chunk each: see in_bwd_apply_kernel
        const int tile = U * PP, ntiles = hw / tile;
        int pstep = tile, pend = p1;
        if (k.interleave) {
            p = (k.rev ? ntiles - 1 - (int)blockIdx.x : (int)blockIdx.x) * tile + pp;
            pstep = (k.rev ? -(int)gridDim.x : (int)gridDim.x) * tile;
            pend = ntiles * tile;
        }
        for (; p >= 0 && p + (U - 1) * PP < pend; p += pstep) {
            f32x8 g[U], x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g[u] = dout(p + u * PP);
                x[u] = ld8((const bf16_t*)k.a + ((size_t)n * hw + p + u * PP) * k.lda + cl * 8);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float sg = 0.f, sx = 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float xh = (x[u][e] - mean[e]) * inv[e];
                    sg += g[u][e];
                    sx += g[u][e] * xh;
                }
                v[0][e] += (double)sg;
                v[1][e] += (double)sx;
            }
        }
        int ptail = p, ptend = p1;
        if (k.interleave) {                  // the pixels beyond the last whole tile: block 0, one at a time
            ptail = blockIdx.x == 0 ? pend + pp : hw;
            ptend = hw;
        }
        p = ptail;
        for (; p < ptend; p += PP) {
            const f32x8 g = dout(p);
            const f32x8 x = ld8((const bf16_t*)k.a + ((size_t)n * hw + p) * k.lda + cl * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (x[e] - mean[e]) * inv[e];
                v[0][e] += (double)g[e];
                v[1][e] += (double)g[e] * (double)xh;
            }
        }
    }
    // combine over the PP pixel slots, then one atomic per (channel, value): red[(n*c + ch)*2 + q]
    double* dst = k.red + (size_t)n * k.c * 2;
    for (int q = 0; q < 2; ++q) {
        __syncthreads();
        if (active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(pp * lanes_c + cl) * 8 + e] = v[q][e];
        }
        __syncthreads();
        if (active && pp == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                double s = 0.0;
                for (int t = 0; t < PP; ++t) s += red[(t * lanes_c + cl) * 8 + e];
                atomicAdd(&dst[(cl * 8 + e) * 2 + q], s);
            }
        }
    }
}

// RAW: the two means come from gsum slot sums (InBwdArgs::gred / gredp) instead of the reduce pass's `red`:
//   sum g     = sum g1 + sum g2                      (g2 is the gradient of the 2x2 average pool: each value reaches 4 pixels x 1/4)
//   sum g*xh  = inv * (sum g1*a - mean * sum g1)  +  (sum g2*pooled - beta * sum g2)      (pooled = avgpool(xh) + beta)
template <typename T, typename TG, bool G2, bool R1 = false, bool RAW = false>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const InBwdArgs k) {
    PixMap pm(k.c);
    f32x4 wr = {0.f, 0.f, 0.f, 0.f};
    if constexpr (R1) {
        if (pm.active) wr = *(const f32x4*)(k.r1_w + pm.cl * 4);
    }
    const int n = k.n0 + blockIdx.y, hw = k.h * k.w;
    const int p0 = blockIdx.x * k.chunk, p1 = min(hw, p0 + k.chunk);
    // interleaved pixel mapping ("elem.interleave"): bf16 activations only -- a compile-time property of the instantiation, because the
    // float32 pass gains nothing from it and loses 5 % to the extra loop bookkeeping when it is a run-time option (4.12 -> 4.34 ms per step)
    constexpr bool IL = sizeof(T) == 2;
    // RAW: the two means of every channel, formed ONCE per block from the slot copies (thread ch sums channel ch's slots: with every
    // thread summing the slots of its own four channels the pass spent a third of its time re-reading 64 doubles per thread)
    __shared__ float sm12[RAW ? 2048 : 2];
    if constexpr (RAW) {
        for (int ch = threadIdx.x; ch < k.c; ch += 256) {
            const size_t i = ((size_t)n * k.c + ch) * 2;
            const size_t sstride = (size_t)k.nbatch * k.c * 2;
            double sg = 0.0, sga = 0.0, pg = 0.0, pgx = 0.0;
            for (int sl = 0; sl < k.gslots; ++sl) {
                sg += k.gred[sl * sstride + i];
                sga += k.gred[sl * sstride + i + 1];
            }
            if (k.gredp) {
                for (int sl = 0; sl < k.gslots; ++sl) {
                    pg += k.gredp[sl * sstride + i];
                    pgx += k.gredp[sl * sstride + i + 1];
                }
            }
            const double bt = k.gredp ? (double)k.beta[ch] : 0.0;
            sm12[ch * 2] = (float)((sg + pg) / hw);
            sm12[ch * 2 + 1] = (float)((k.stats[i + 1] * (sga - k.stats[i] * sg) + (pgx - bt * pg)) / hw);
        }
        __syncthreads();
    }
    double v[1][4] = {};
    if (pm.active) {
        float mean[4], inv[4], m1[4], m2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            size_t i = ((size_t)n * k.c + pm.cl * 4 + e) * 2;
            mean[e] = (float)k.stats[i];
            inv[e] = (float)k.stats[i + 1];
            if constexpr (RAW) {
                m1[e] = sm12[(pm.cl * 4 + e) * 2];
                m2[e] = sm12[(pm.cl * 4 + e) * 2 + 1];
            } else {
                m1[e] = (float)(k.red[i] / hw);
                m2[e] = (float)(k.red[i + 1] / hw);
            }
        }
        constexpr int U = sizeof(T) == 2 ? 8 : 4;
        int p = p0 + pm.pp;
        // k.interleave (experiment "elem.interleave"): the blocks of a sample take tiles of U * PP pixels round-robin instead of one
        // contiguous chunk each -- at any instant the chip then reads a narrow band of the tensors instead of ~2000 separate places
        const int tile = U * pm.PP;
        int pstep = tile, pend = p1;
        if constexpr (IL)
            if (k.interleave) {
                p = blockIdx.x * tile + pm.pp;
                pstep = gridDim.x * tile;
                pend = hw - hw % tile;
            }
        for (; p + (U - 1) * pm.PP < (IL ? pend : p1); p += (IL ? pstep : tile)) {
            f32x4 g[U], x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g[u] = in_bwd_dout<TG, G2, R1>(k, n, p + u * pm.PP, pm.cl, wr);
                x[u] = ld4((const T*)k.a + ((size_t)n * hw + p + u * pm.PP) * k.lda + pm.cl * 4);
            }
            float sd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xh = (x[u][e] - mean[e]) * inv[e];
                    float da = inv[e] * (g[u][e] - m1[e] - xh * m2[e]);
                    d[e] = x[u][e] > 0.f ? da : da * k.slope;
                    sd[e] += d[e];
                }
                st4((T*)k.dz + ((size_t)n * hw + p + u * pm.PP) * k.lddz + pm.cl * 4, d);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[0][e] += (double)sd[e];
        }
        int ptend = p1;
        if constexpr (IL)
            if (k.interleave) {              // the pixels beyond the last whole tile: block 0, one at a time
                p = blockIdx.x == 0 ? pend + pm.pp : hw;
                ptend = hw;
            }
        for (; p < (IL ? ptend : p1); p += pm.PP) {
            f32x4 g = in_bwd_dout<TG, G2, R1>(k, n, p, pm.cl, wr);
            f32x4 x = ld4((const T*)k.a + ((size_t)n * hw + p) * k.lda + pm.cl * 4);
            f32x4 d;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xh = (x[e] - mean[e]) * inv[e];
                float da = inv[e] * (g[e] - m1[e] - xh * m2[e]);
                d[e] = x[e] > 0.f ? da : da * k.slope;
                v[0][e] += (double)d[e];
            }
            st4((T*)k.dz + ((size_t)n * hw + p) * k.lddz + pm.cl * 4, d);
        }
    }
    // bias gradient: staged per sample in red[2*batch*c + n*c + ch] -- one f64 atomic address per (n, ch)
    // instead of per ch (4096 blocks on 64 addresses cost 90-210 us per launch), folded by dbias_fold_kernel
    if (k.dbias) block_reduce_atomic<1>(v, pm, (RAW ? k.dstage : k.red + (size_t)k.nbatch * k.c * 2) + (size_t)n * k.c, k.c, true);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// bf16 InstanceNorm backward in ONE pass over HBM (round 5).  The two-pass form reads g and a twice (reduce, apply): the
// second read of a 34-335 MB tensor pair comes from HBM again (l2_hit 0.02-0.26 in profiles/r05_bf16_traffic_pmc.json).  Here a block
// KEEPS its slice of g and a (and of the pooled gradient g2) in registers between the two phases: 8 pixels x 8 channels x 2-3 tensors per
// thread as raw bf16 (64-96 VGPRs), and the blocks of one SAMPLE meet at a per-sample barrier between the phases:
//   phase 1: partial sums (sum g, sum g * xhat) of the slice -> LDS combine -> the block's own row of a partials table (plain coherent stores:
//            no float atomics anywhere -- the first version's 192 f64 atomics per block, 2 M per launch, WERE the launch: 1.36 ms against 0.38)
//   barrier: arrival counter of the sample; the LAST ARRIVER adds the rows in block order (bitwise reproducible), publishes the two means of every
//            channel and raises the release flags (details at the code).  The blocks of a sample are consecutive block ids (blockIdx.y = sample)
//            and at most 256 of them (the launcher falls back to two passes otherwise): a whole sample is resident long before the chip is
//            full (>= 512 blocks fit), blocks are dispatched in id order, so the earliest incomplete sample always completes.  A block that
//            would spin for more than ~1 s raises the scratch's timeout word and goes on (wrong numbers, never a hang).
//   phase 2: every block applies from its registers with the published means, stores dz and writes its row of bias-gradient partials; the last
//            block to leave adds those rows in block order into the per-sample staging and clears the sample's counters, flags and means
//            (scratch: zero on entry, zero on return -- the partial rows are rewritten in full by every launch and need no clearing).
// HBM traffic: g + a read once, dz written once = 3 tensor passes instead of 5.  Phase stamps (tools/probes/in_bwd_fused_stamps.py, n = 40 at
// 256 x 256 x 64, median per block): slices loaded 3.1 us, rows written 2.4, arrival to release 13.8 (of which ~5 waiting for the sample's last
// block), phase 2 stores 1.2, departure 2.3: 23 us per block with 768 resident.  Starting the samples of the first resident generation a fraction
// of a period apart changed nothing (the launch is bound by that latency chain times the residency, not by a memory phase all blocks share).
constexpr int SHM_FUSED_FLAGS = 16, SHM_FUSED_SYNC_WORDS = 32 * (SHM_FUSED_FLAGS + 2);
// scratch (float64 units): partials f32 [batch][c / CB][bpi][3 CB] (sum g, sum g * xhat interleaved, then sum dz) | means f32 [batch][c][2] | sync u32
// [batch][c / CB][SYNC_WORDS] | timeout word.  CB = min(c, 64) channels per barrier group, bpi = blocks per group = h * w * CB / 16384.
static size_t fused_row_doubles(int batch, size_t bpi, int c) { return ((size_t)batch * bpi * 3 * c + 1) / 2; }        // fp32 rows: [batch][c / CB][bpi][3 CB]
static size_t fused_scratch_doubles(int batch, int hw, int c) {
    const int cb = c < 64 ? c : 64;
    const size_t bpi = (size_t)hw * cb / 16384;
    return fused_row_doubles(batch, bpi, c) + (size_t)batch * c + (size_t)batch * (c / cb) * (SHM_FUSED_SYNC_WORDS / 2) + 1;
}
__device__ __forceinline__ f32x8 unpack8(const shm_u32x4 u) {
    f32x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r[2 * i] = __uint_as_float(u[i] << 16);
        r[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
    return r;
}
// coherent (device-scope, L2-bypassing) accesses without fences: see the barrier below
template <typename V>
__device__ __forceinline__ V coh_load(const V* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename V>
__device__ __forceinline__ void coh_store(V* p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The launch's last group folds the staged bias gradient (round 6: dbias_fold_kernel was one more launch behind each of the step's ~44 one-pass
// calls).  `staging` = f64 [batch][c], written by every group's last departer (its CB channels of its sample, coherent stores, acknowledged before
// the group takes a ticket at `ticket`); the group whose ticket is the last adds the samples in dbias_fold_kernel's order -- four interleaved partial
// sums, (s0 + s1) + (s2 + s3): the same bits as the separate launch -- into dbias, and leaves staging and ticket zero.  Called by all 256 threads of a
// group's last departer, after its staging stores.
__device__ __forceinline__ void fused_fold_dbias(double* __restrict__ staging, double* __restrict__ dbias, unsigned* __restrict__ ticket, int batch, int c,
                                                 unsigned ngroups, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *s_flag = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == ngroups;
    __syncthreads();
    if (!*s_flag) return;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        double sg[4] = {0.0, 0.0, 0.0, 0.0};
        int i = 0;
        for (; i + 8 <= batch; i += 8) {             // eight loads in flight (one at a time, 160 samples were 160 round trips)
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = coh_load(staging + (size_t)(i + j) * c + ch);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sg[j & 3] += v[j];                   // (i is a multiple of 8: (i + j) & 3 == j & 3)
                coh_store(staging + (size_t)(i + j) * c + ch, 0.0);
            }
        }
        for (; i < batch; ++i) {
            sg[i & 3] += coh_load(staging + (size_t)i * c + ch);
            coh_store(staging + (size_t)i * c + ch, 0.0);
        }
        dbias[ch] += (sg[0] + sg[1]) + (sg[2] + sg[3]);
    }
    if (threadIdx.x == 0) coh_store(ticket, 0u);
}

template <bool G2>
__global__ __launch_bounds__(256, G2 ? 3 : 4) void in_bwd_fused8_kernel(const InBwdArgs k, float* __restrict__ fpart, float* __restrict__ fres,
                                                                         unsigned* __restrict__ fsync, unsigned* __restrict__ ferr,
                                                                         unsigned* __restrict__ abort_dev, unsigned* __restrict__ abort_host,
                                                                         const unsigned arrivals) {
    // arrivals: blocks a group's barrier waits for = gridDim.x (one more under "elem.fused_test_stall": the timeout path under test)
    constexpr int U = 8;
    __shared__ double red[256 * 8];
    __shared__ float sm12[128], smi[128];
    __shared__ int s_last;
    // A barrier GROUP is (sample, block of CB = min(c, 64) channels): its blocks are the 16384 / CB-pixel slices of the map, blockIdx.x.  (With the
    // whole channel range in one group the 256- and 512-channel levels had short slices, rows of 3 c values and a last arriver adding
    // bpi * c / 256 values per thread: slower than the two passes at n = 20.)  A pixel's CB channels are one 128-byte line (c >= 64).
    const int CB = k.c < 64 ? k.c : 64, c0 = blockIdx.y * CB;
    const int lanes_c = CB >> 3, PP = 256 / lanes_c;                   // CB is a power of two (launcher): every thread is active
    const int pp = threadIdx.x / lanes_c, cl = threadIdx.x - pp * lanes_c;
    const int n = k.rev ? (int)gridDim.z - 1 - (int)blockIdx.z : (int)blockIdx.z, hw = k.h * k.w;
    const int gidx = n * gridDim.y + blockIdx.y;                       // the group
    const int pbase = blockIdx.x * (U * PP) + pp;                      // pixel of slot u: pbase + u * PP (hw % (U * PP) == 0)
    const int bpi = gridDim.x, c3 = 3 * CB;
    float* const prow0 = fpart + (size_t)gidx * bpi * c3;              // the group's rows; row b = [CB][2] sums, then [CB] bias-gradient partials
    float* const prow = prow0 + (size_t)blockIdx.x * c3;               // (fp32: a row holds sums over one slice, and the means are fp32 in the end)

#ifdef SHM_FUSED_STAMP
    unsigned long long* const stamp = (unsigned long long*)(ferr + 2) + ((size_t)gidx * gridDim.x + blockIdx.x) * 12;
#define FSTAMP(i) do { if (threadIdx.x == 0) stamp[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif
    FSTAMP(0);
    // Pixel of slot u.  Plain form: pbase + u * PP (a slice is 256 consecutive pixels).  Pooled form (CB = 64, PP = 32): a slice is a tile of
    // R rows x Wt = min(w, 128) columns, R * Wt = 256, and a thread owns two 2 x 2 QUADS of it (quad pp and pp + 32 of the tile's 64): one pooled
    // gradient value serves four pixels, two loads instead of eight -- with eight the form needs 204 registers and two blocks per CU.
    int qbase = 0, wt2 = 1;                                              // pooled form: pixel index of the tile's origin, quads per tile row
    if constexpr (G2) {
        const int wt = k.w < 128 ? k.w : 128, tpr = k.w / wt, ty = blockIdx.x / tpr, tx = blockIdx.x - ty * tpr;
        wt2 = wt >> 1;
        qbase = ty * (256 / wt) * k.w + tx * wt;
    }
    auto pix = [&](int u, int base) {
        if constexpr (G2) {
            const int q = pp + 32 * (u >> 2), qy = q / wt2, qx = q - qy * wt2;
            return base + (2 * qy + ((u >> 1) & 1)) * k.w + 2 * qx + (u & 1);
        } else {
            return base + u * PP;
        }
    };
    shm_u32x4 gq[U], aq[U];
    [[maybe_unused]] shm_u32x4 hq[2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = pix(u, G2 ? qbase : pbase);
        const size_t off = (size_t)n * hw + p;
        gq[u] = *(const shm_u32x4*)((const bf16_t*)k.g1 + off * k.ldg1 + c0 + cl * 8);
        aq[u] = *(const shm_u32x4*)((const bf16_t*)k.a + off * k.lda + c0 + cl * 8);
        if constexpr (G2) {
            if ((u & 3) == 0) {
                const int y = p / k.w, x = p - y * k.w;
                const size_t q = ((size_t)n * (k.h >> 1) + (y >> 1)) * (k.w >> 1) + (x >> 1);
                hq[u >> 2] = *(const shm_u32x4*)((const bf16_t*)k.g2 + q * k.ldg2 + c0 + cl * 8);
            }
        }
    }
    float mean[8], inv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mean[e] = (float)k.stats[((size_t)n * k.c + c0 + cl * 8 + e) * 2];
        inv[e] = (float)k.stats[((size_t)n * k.c + c0 + cl * 8 + e) * 2 + 1];
    }
    if (pp == 0) {                    // phase 2 takes them from LDS again: sixteen registers less across the barrier
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            smi[(cl * 8 + e) * 2] = mean[e];
            smi[(cl * 8 + e) * 2 + 1] = inv[e];
        }
    }
    auto gval = [&](int u) {
        f32x8 g = unpack8(gq[u]);
        if constexpr (G2) {
            const f32x8 h = unpack8(hq[u >> 2]);
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] += 0.25f * h[e];
        }
        return g;
    };
    // Combine NV x 8 per-thread values over the PP pixel slots and store them: value j of channel ch lands at dst[ch * stride + j].  Every thread
    // parks its values in LDS (fp32, [j][pp][c]), thread t < NV * c adds the PP terms of one output (consecutive threads read consecutive words).
    // (The first version -- eight threads adding 32 LDS doubles each, twice -- took 21 of a block's 46 us; xor-shuffles spilled 116 registers.)
    float* const redf = (float*)red;
    auto park = [&](const float (&v)[8], int j) {
        *(f32x4*)&redf[j * 2048 + threadIdx.x * 8] = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)&redf[j * 2048 + threadIdx.x * 8 + 4] = f32x4{v[4], v[5], v[6], v[7]};
    };
    auto finish = [&](int nv, float* dst, int stride) {
        __syncthreads();
        for (int t = threadIdx.x; t < CB * nv; t += 256) {
            const int j = t / CB, ch = t - j * CB;
            float sum = 0.f;
            for (int q = 0; q < PP; ++q) sum += redf[j * 2048 + q * CB + ch];
            coh_store(&dst[ch * stride + j], sum);
        }
    };
    // Sums of `npairs` consecutive float PAIRS of every row of the sample, in a fixed order (bitwise reproducible): thread (rg, q) adds pair q of
    // rows rg, rg + RG, ... (sixteen 8-byte coherent loads in flight), the RG partial sums meet in LDS; on return thread v < 2 * npairs calls
    // total(v).  (One thread per value walking all 256 rows, eight loads at a time, made the barrier 45 us long.)
    auto rowsum = [&](const float* base, int npairs) {
        const int P = npairs < 256 ? npairs : 256, RG = 256 / P, rg = threadIdx.x / P;
        __syncthreads();
        for (int q = threadIdx.x % P; q < npairs; q += P) {
            double s0 = 0.0, s1 = 0.0;
            const unsigned long long* col = (const unsigned long long*)base + q;
            const size_t rs = (size_t)c3 / 2;                                // row stride in pairs (c3 is even)
            int b = rg;
            for (; b + 15 * RG < bpi; b += 16 * RG) {
                unsigned long long t[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) t[j] = coh_load(col + (size_t)(b + j * RG) * rs);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    s0 += (double)__uint_as_float((unsigned)t[j]);
                    s1 += (double)__uint_as_float((unsigned)(t[j] >> 32));
                }
            }
            for (; b < bpi; b += RG) {
                const unsigned long long t = coh_load(col + (size_t)b * rs);
                s0 += (double)__uint_as_float((unsigned)t);
                s1 += (double)__uint_as_float((unsigned)(t >> 32));
            }
            red[(rg * npairs + q) * 2] = s0;
            red[(rg * npairs + q) * 2 + 1] = s1;
        }
        __syncthreads();
        return RG;
    };
    auto total = [&](int v, int npairs, int RG) {
        double s = 0.0;
        for (int r = 0; r < RG; ++r) s += red[(r * npairs + (v >> 1)) * 2 + (v & 1)];
        return s;
    };
    // ---- phase 1
    {
        float sg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef SHM_FUSED_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FSTAMP(1);
#endif
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x8 g = gval(u), x = unpack8(aq[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (x[e] - mean[e]) * inv[e];
                sg[e] += g[e];
                sx[e] += g[e] * xh;
            }
        }
        __syncthreads();
        park(sg, 0);
        park(sx, 1);
        finish(2, prow, 2);
    }
    // ---- the sample's barrier.  No agent-scope FENCES: on a multi-XCD chip a release fence is buffer_wbl2 (write the XCD's dirty L2 lines back)
    // and every acquire -- one per poll -- a buffer_inv of the L2 (0.65 ms per launch with ~1000 resident blocks; even ONE acq_rel arrival, one
    // release flag store and one acquire fence behind the poll loop per block: 1 191 us against 291 for the n = 40 level, step 31.7 against 22.8 ms).
    // Everything the blocks exchange
    // moves through device-scope relaxed atomics (loads, stores, the two counters), which are performed at the device's coherence point: a block
    // waits for the acknowledgement of its stores (vmcnt) and then counts its arrival.  And no crowd on one address: 256 blocks polling the arrival
    // counter queue their reads in front of the arrivals themselves (measured: 41 us per sample).  The LAST ARRIVER (it alone knows every row is
    // in) publishes the means and raises SHM_FUSED_FLAGS copies of the release flag, each in its own 128-byte line; block b polls copy b % 16.
    unsigned* const sy = fsync + (size_t)gidx * SHM_FUSED_SYNC_WORDS;     // [0] arrivals, [32] departures, [64 + 32 j] flag copy j
    float* const res = fres + ((size_t)n * k.c + c0) * 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FSTAMP(2);
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(sy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == arrivals;
    __syncthreads();
    FSTAMP(3);
    if (s_last) {
        FSTAMP(8);
        const int RG = rowsum(prow0, CB);
        FSTAMP(9);
        for (int v = threadIdx.x; v < 2 * CB; v += 256) {
            const float r = (float)(total(v, CB, RG) / hw);
            sm12[v] = r;
            coh_store(res + v, r);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the means are written before a flag goes up
        __syncthreads();
        FSTAMP(10);
        if (threadIdx.x < SHM_FUSED_FLAGS) coh_store(sy + 64 + 32 * threadIdx.x, 1u);
    } else {
        if (threadIdx.x == 0) {
            const unsigned* const flag = sy + 64 + 32 * (blockIdx.x % SHM_FUSED_FLAGS);
            int spins = 0;
            while (coh_load(flag) == 0u) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (arrivals == gridDim.x ? 1 << 20 : 1 << 10)) {          // (the stalled test form gives up after ~1 ms)
                    // gave up: this launch goes on with wrong means.  The scratch's own word names the buffer; the caller's abort words
                    // (shm_set_abort_words) make it fatal: shm_adam_clip applies nothing while the device word is set, and the host word
                    // (mapped host memory) lets the trainer see it without a synchronisation
                    __hip_atomic_fetch_or(ferr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (abort_dev) __hip_atomic_fetch_or(abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (abort_host) __hip_atomic_store(abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CB * 2; i += 256) sm12[i] = coh_load(res + i);
    }
    __syncthreads();
    FSTAMP(4);
    // ---- phase 2 (the raw slices pass through an opaque copy: otherwise hipcc keeps phase 1's UNPACKED values alive across the barrier and spills)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        asm volatile("" : "+v"(gq[u]), "+v"(aq[u]));
        if constexpr (G2)
            if ((u & 3) == 0) asm volatile("" : "+v"(hq[u >> 2]));
    }
    int pb2 = G2 ? qbase : pbase;     // (opaque as well: the store addresses are formed here, not carried from the loads at the top)
    asm volatile("" : "+v"(pb2));
    // d = inv * (g - m1 - xhat * m2) with xhat = (x - mean) * inv, as three constants per channel: d = A g - (B x + C)
    float cA[8], cB[8], cC[8], sd[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float m1 = sm12[(cl * 8 + e) * 2], m2 = sm12[(cl * 8 + e) * 2 + 1], mu = smi[(cl * 8 + e) * 2], iv = smi[(cl * 8 + e) * 2 + 1];
        cA[e] = iv;
        cB[e] = iv * iv * m2;
        cC[e] = iv * m1 - cB[e] * mu;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const f32x8 g = gval(u), x = unpack8(aq[u]);
        typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float da = cA[e] * g[e] - (cB[e] * x[e] + cC[e]);
            const float d = x[e] > 0.f ? da : da * k.slope;
            sd[e] += d;
            o[e] = (bf16_t)d;
        }
        *(bf16x8_t*)((bf16_t*)k.dz + ((size_t)n * hw + pix(u, pb2)) * k.lddz + c0 + cl * 8) = o;
    }
    FSTAMP(5);
    if (k.dbias) {
        __syncthreads();
        park(sd, 0);
        finish(1, prow + 2 * CB, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this block's row is written before its departure is counted
    }
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(sy + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x;
    __syncthreads();
    FSTAMP(6);
    if (s_last) {                     // every block of the sample is through: the bias-gradient staging of the sample, then a clean scratch
        if (k.dbias) {
            const int RG = rowsum(prow0 + 2 * CB, CB / 2);
            for (int ch = threadIdx.x; ch < CB; ch += 256) coh_store(k.red + (size_t)k.nbatch * k.c * 2 + (size_t)n * k.c + c0 + ch, total(ch, CB / 2, RG));
        }
        for (int i = threadIdx.x; i < CB * 2; i += 256) coh_store(res + i, 0.f);
        if (threadIdx.x < SHM_FUSED_FLAGS + 2) coh_store(sy + 32 * threadIdx.x, 0u);
        // fold = 1: this launch also folds the staged bias gradient (no dbias_fold_kernel behind it); ferr[1] is the launch's group ticket
        if (k.dbias && k.fold) fused_fold_dbias(k.red + (size_t)k.nbatch * k.c * 2, k.dbias, ferr + 1, k.nbatch, k.c, gridDim.y * gridDim.z, &s_last);
    }
#ifdef SHM_FUSED_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FSTAMP(7);
#endif
#undef FSTAMP
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the same one-pass protocol with HALF the registers per byte of traffic.  in_bwd_fused8_kernel is bound by residency x latency (a
// block holds its 64 KiB of g and a for ~23 us; 4 blocks per CU fill the register file: 768-1024 x 96 KiB of traffic per 23 us = 3.2-4.3 TB/s,
// LABNOTES 11.6).  Here a block holds ONLY g (the gradient is dead after this kernel; the activation stays in HBM and the Infinity Cache): a slice
// is 16 pixel slots per thread (32768 / CB pixels), g raw bf16 in 64 registers, and `a` is streamed through 16-byte transient registers twice --
// phase 1 for sum g * (x - mean), phase 2 for the apply (the second read, 20-30 us after the first, was meant to hit the Infinity Cache; the
// counters of profiles/r06_* say it comes from HBM: 1.38 x the algorithmic bytes -- the kernel is residency bound and faster all the same).  The
// same 4 blocks per CU now cover 192 KiB of traffic each, and a barrier group has HALF the blocks (128 on the 256 x 256 x 64 maps: the last
// arriver's row sums, 5.9 of 13.3 us there, halve; the 512 x 512 maps of BASELINE configs[3] get 512-block groups, which twice fit the chip).
// The register budget decides the form: 64 (g) + 16 (sums) + 8 (means) leave room for TWO transient loads of `a` per batch at four blocks per CU
// (<2, 2, 4>: 8 spills; eight batches per phase, each a round trip); eight per batch need three blocks per CU (<8, 8, 3>).  hipcc has to be held
// to the batches by data dependences (below).  Measured, n = 40 / 160 at 256 x 256 x 64: 287 / 1072 us (in_bwd_fused8_kernel) -> 252 / 879 (<2, 2, 4>),
// 262 / 920 (<8, 8, 3>); on maps with fewer blocks per group the extra round trips lose (128 x 128 x 128: 122 -> 147): see the launcher.
// Sums: sum g and sum g * (x - mean) per slice in fp32 (the centring keeps the second one free of the cancellation a raw sum g * x would meet
// when |mean| >> 1 / inv), times inv at the end; everything else -- rows, last arriver, flags, timeout and abort words, departure -- as above.
#define FG_LOAD(rs, base, voff, soff) __builtin_bit_cast(shm_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0))
#define FG_STORE(v, rs, base, voff) __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0)
template <int AB, int AB2, int BPC>          // transient `a` loads in flight per batch in phase 1 / phase 2; blocks per CU the register budget is cut for
__global__ __launch_bounds__(256, BPC) void in_bwd_fusedg_kernel(const InBwdArgs k, float* __restrict__ fpart, float* __restrict__ fres,
                                                               unsigned* __restrict__ fsync, unsigned* __restrict__ ferr,
                                                               unsigned* __restrict__ abort_dev, unsigned* __restrict__ abort_host, const unsigned arrivals) {
    constexpr int U = 16;
    __shared__ double red[256 * 8];
    __shared__ float sm12[128], smi[128];
    __shared__ int s_last;
    const int CB = k.c < 64 ? k.c : 64, c0 = blockIdx.y * CB;
    const int lanes_c = CB >> 3, PP = 256 / lanes_c;
    const int pp = threadIdx.x / lanes_c, cl = threadIdx.x - pp * lanes_c;
    const int n = k.rev ? (int)gridDim.z - 1 - (int)blockIdx.z : (int)blockIdx.z, hw = k.h * k.w;
    const int gidx = n * gridDim.y + blockIdx.y;
    const int pbase = blockIdx.x * (U * PP) + pp;
    const int bpi = gridDim.x, c3 = 3 * CB;
    float* const prow0 = fpart + (size_t)gidx * bpi * c3;
    float* const prow = prow0 + (size_t)blockIdx.x * c3;
    // buffer accesses: one descriptor per tensor and SAMPLE (a sample is below 4 GiB: launcher), the lane's byte offset in ONE register, the pixel
    // slot as a scalar offset -- with flat 64-bit addresses hipcc kept sixteen address pairs alive beside the 64 registers of g and spilled 216
    const unsigned samp_g = (unsigned)hw * (unsigned)k.ldg1 * 2u, samp_a = (unsigned)hw * (unsigned)k.lda * 2u, samp_z = (unsigned)hw * (unsigned)k.lddz * 2u;
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((char*)k.g1 + (size_t)n * samp_g, 0, samp_g, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((char*)k.a + (size_t)n * samp_a, 0, samp_a, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((char*)k.dz + (size_t)n * samp_z, 0, samp_z, 0x00020000);
    const unsigned og = (unsigned)(pbase * k.ldg1 + c0 + cl * 8) * 2u, oa = (unsigned)(pbase * k.lda + c0 + cl * 8) * 2u;
    const unsigned sg_step = (unsigned)(PP * k.ldg1) * 2u, sa_step = (unsigned)(PP * k.lda) * 2u, sz_step = (unsigned)(PP * k.lddz) * 2u;      // scalars
    shm_u32x4 gq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) gq[u] = FG_LOAD(rsg, (const char*)k.g1 + (size_t)n * samp_g, og, (unsigned)u * sg_step);
    float* const redf = (float*)red;
    auto park = [&](const float (&v)[8], int j) {
        *(f32x4*)&redf[j * 2048 + threadIdx.x * 8] = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)&redf[j * 2048 + threadIdx.x * 8 + 4] = f32x4{v[4], v[5], v[6], v[7]};
    };
    auto finish = [&](int nv, float* dst, int stride) {
        __syncthreads();
        for (int t = threadIdx.x; t < CB * nv; t += 256) {
            const int j = t / CB, ch = t - j * CB;
            float sum = 0.f;
            for (int q = 0; q < PP; ++q) sum += redf[j * 2048 + q * CB + ch];
            coh_store(&dst[ch * stride + j], sum);
        }
    };
    auto rowsum = [&](const float* base, int npairs) {
        const int P = npairs < 256 ? npairs : 256, RG = 256 / P, rg = threadIdx.x / P;
        __syncthreads();
        for (int q = threadIdx.x % P; q < npairs; q += P) {
            double s0 = 0.0, s1 = 0.0;
            const unsigned long long* col = (const unsigned long long*)base + q;
            const size_t rs = (size_t)c3 / 2;
            int b = rg;
            for (; b + 15 * RG < bpi; b += 16 * RG) {
                unsigned long long t[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) t[j] = coh_load(col + (size_t)(b + j * RG) * rs);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    s0 += (double)__uint_as_float((unsigned)t[j]);
                    s1 += (double)__uint_as_float((unsigned)(t[j] >> 32));
                }
            }
            for (; b < bpi; b += RG) {
                const unsigned long long t = coh_load(col + (size_t)b * rs);
                s0 += (double)__uint_as_float((unsigned)t);
                s1 += (double)__uint_as_float((unsigned)(t >> 32));
            }
            red[(rg * npairs + q) * 2] = s0;
            red[(rg * npairs + q) * 2 + 1] = s1;
        }
        __syncthreads();
        return RG;
    };
    auto total = [&](int v, int npairs, int RG) {
        double s = 0.0;
        for (int r = 0; r < RG; ++r) s += red[(r * npairs + (v >> 1)) * 2 + (v & 1)];
        return s;
    };
    // ---- phase 1: `a` passes through sixteen-byte transients; (sum g, inv * sum g * (x - mean)) of the slice -> the block's row
    {
        float mean[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) mean[e] = (float)k.stats[((size_t)n * k.c + c0 + cl * 8 + e) * 2];
        if (pp == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                smi[(cl * 8 + e) * 2] = mean[e];
                smi[(cl * 8 + e) * 2 + 1] = (float)k.stats[((size_t)n * k.c + c0 + cl * 8 + e) * 2 + 1];
            }
        }
        float sg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unsigned oab = oa;
        // `a` in batches of AB transient loads (the register budget of four blocks per CU: 64 for g + 16 sums + 8 means leave ~32); the
        // sched_barrier keeps hipcc from hoisting the next batch's loads over this batch's arithmetic (it would: 168 spills)
#pragma unroll
        for (int b = 0; b < U / AB; ++b) {
            shm_u32x4 aq[AB];
#pragma unroll
            for (int u = 0; u < AB; ++u) aq[u] = FG_LOAD(rsa, (const char*)k.a + (size_t)n * samp_a, oab, (unsigned)(b * AB + u) * sa_step);
            // (the g-only half of the arithmetic -- unpack, sum g -- is pure register work: left visible, LLVM hoists it for all sixteen slots to
            // the top of the kernel, 128 live floats; the opaque copy ties it to its batch)
#pragma unroll
            for (int u = 0; u < AB; ++u) asm volatile("" : "+v"(gq[b * AB + u]));
#pragma unroll
            for (int u = 0; u < AB; ++u) {
                const f32x8 x = unpack8(aq[u]), g = unpack8(gq[b * AB + u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sg[e] += g[e];
                    sx[e] += g[e] * (x[e] - mean[e]);
                }
            }
            // the next batch's loads wait (as far as hipcc can see) for this batch's sums: otherwise all sixteen are hoisted to the top
            asm volatile("" : "+v"(oab) : "v"(sg[0]), "v"(sg[1]), "v"(sg[2]), "v"(sg[3]), "v"(sg[4]), "v"(sg[5]), "v"(sg[6]), "v"(sg[7]), "v"(sx[0]), "v"(sx[1]), "v"(sx[2]), "v"(sx[3]),
                         "v"(sx[4]), "v"(sx[5]), "v"(sx[6]), "v"(sx[7]));
        }
        __syncthreads();                    // smi is written
#pragma unroll
        for (int e = 0; e < 8; ++e) sx[e] *= smi[(cl * 8 + e) * 2 + 1];
        park(sg, 0);
        park(sx, 1);
        finish(2, prow, 2);
    }
    // ---- the group's barrier (in_bwd_fused8_kernel: relaxed device-scope atomics, the last arriver adds the rows in block order and raises the flags)
    unsigned* const sy = fsync + (size_t)gidx * SHM_FUSED_SYNC_WORDS;
    float* const res = fres + ((size_t)n * k.c + c0) * 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(sy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == arrivals;
    __syncthreads();
    if (s_last) {
        const int RG = rowsum(prow0, CB);
        for (int v = threadIdx.x; v < 2 * CB; v += 256) {
            const float r = (float)(total(v, CB, RG) / hw);
            sm12[v] = r;
            coh_store(res + v, r);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < SHM_FUSED_FLAGS) coh_store(sy + 64 + 32 * threadIdx.x, 1u);
    } else {
        if (threadIdx.x == 0) {
            const unsigned* const flag = sy + 64 + 32 * (blockIdx.x % SHM_FUSED_FLAGS);
            int spins = 0;
            while (coh_load(flag) == 0u) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (arrivals == gridDim.x ? 1 << 20 : 1 << 10)) {
                    __hip_atomic_fetch_or(ferr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (abort_dev) __hip_atomic_fetch_or(abort_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (abort_host) __hip_atomic_store(abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CB * 2; i += 256) sm12[i] = coh_load(res + i);
    }
    __syncthreads();
    // ---- phase 2: d = A g - (B x + C) from the held g and a second read of a
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(gq[u]));
    int pb2 = pbase;
    asm volatile("" : "+v"(pb2));
    const unsigned oa2 = (unsigned)(pb2 * k.lda + c0 + cl * 8) * 2u, oz = (unsigned)(pb2 * k.lddz + c0 + cl * 8) * 2u;
    float cA[8], cB[8], cC[8], sd[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float m1 = sm12[(cl * 8 + e) * 2], m2 = sm12[(cl * 8 + e) * 2 + 1], mu = smi[(cl * 8 + e) * 2], iv = smi[(cl * 8 + e) * 2 + 1];
        cA[e] = iv;
        cB[e] = iv * iv * m2;
        cC[e] = iv * m1 - cB[e] * mu;
    }
    unsigned oa2b = oa2;
#pragma unroll
    for (int b = 0; b < U / AB2; ++b) {
        shm_u32x4 aq[AB2];
#pragma unroll
        for (int u = 0; u < AB2; ++u) aq[u] = FG_LOAD(rsa, (const char*)k.a + (size_t)n * samp_a, oa2b, (unsigned)(b * AB2 + u) * sa_step);
#pragma unroll
        for (int u = 0; u < AB2; ++u) asm volatile("" : "+v"(gq[b * AB2 + u]));
#pragma unroll
        for (int u = 0; u < AB2; ++u) {
            const f32x8 g = unpack8(gq[b * AB2 + u]), x = unpack8(aq[u]);
            typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
            bf16x8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float da = cA[e] * g[e] - (cB[e] * x[e] + cC[e]);
                const float d = x[e] > 0.f ? da : da * k.slope;
                sd[e] += d;
                o[e] = (bf16_t)d;
            }
            // the slot offset goes into the VECTOR offset: behind a 16-byte buffer store whose soffset is an SGPR hipcc leaves no wait states in front of
            // a VALU write of the store's data registers, and the MI355X needs them (common.h, round 4; tools/check_isa_hazards.py flags the form)
            FG_STORE(__builtin_bit_cast(shm_u32x4, o), rsz, (char*)k.dz + (size_t)n * samp_z, oz + (unsigned)(b * AB2 + u) * sz_step);
        }
        asm volatile("" : "+v"(oa2b) : "v"(sd[0]), "v"(sd[1]), "v"(sd[2]), "v"(sd[3]), "v"(sd[4]), "v"(sd[5]), "v"(sd[6]), "v"(sd[7]));
    }
    if (k.dbias) {
        __syncthreads();
        park(sd, 0);
        finish(1, prow + 2 * CB, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(sy + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x;
    __syncthreads();
    if (s_last) {
        if (k.dbias) {
            const int RG = rowsum(prow0 + 2 * CB, CB / 2);
            for (int ch = threadIdx.x; ch < CB; ch += 256) coh_store(k.red + (size_t)k.nbatch * k.c * 2 + (size_t)n * k.c + c0 + ch, total(ch, CB / 2, RG));
        }
        for (int i = threadIdx.x; i < CB * 2; i += 256) coh_store(res + i, 0.f);
        if (threadIdx.x < SHM_FUSED_FLAGS + 2) coh_store(sy + 32 * threadIdx.x, 0u);
        // fold = 1: this launch also folds the staged bias gradient (no dbias_fold_kernel behind it); ferr[1] is the launch's group ticket
        if (k.dbias && k.fold) fused_fold_dbias(k.red + (size_t)k.nbatch * k.c * 2, k.dbias, ferr + 1, k.nbatch, k.c, gridDim.y * gridDim.z, &s_last);
    }
}

// shm_in_bwd_apply's last launch: fold the staged bias gradient (dbias[ch] += sum over samples) and clear the gsum slot copies the
// apply pass consumed -- "zero on entry, zero on return" for every f64 scratch, no memset in front of a launch.
// keep != null: the per-sample sums are also copied out ([nslot = batch][c]: shm_in_bwd_keep_dz_sums)
__global__ __launch_bounds__(256) void gsum_finish_kernel(double* __restrict__ part, double* __restrict__ dbias, int nslot, int c, double* __restrict__ clr1,
                                                          size_t n1, double* __restrict__ clr2, size_t n2, double* __restrict__ keep) {
    __shared__ double red[4][64];
    if (dbias && blockIdx.x * 64 < (unsigned)c) {
        const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
        const int ch = blockIdx.x * 64 + cl;
        double s = 0.0;
        if (ch < c)
            for (int i = g; i < nslot; i += 4) {
                const double v = part[(size_t)i * c + ch];
                s += v;
                if (keep) keep[(size_t)i * c + ch] = v;
                part[(size_t)i * c + ch] = 0.0;
            }
        red[g][cl] = s;
        __syncthreads();
        if (g == 0 && ch < c) dbias[ch] += (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += stride) clr1[i] = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) clr2[i] = 0.0;
}

// Stand-alone gsum: (sum g, sum g * aux) per (sample, channel) into slot 0 of red -- what the convolution epilogues produce for
// the launches they can take it in (conv_igemm.hip); the *_gsum entry points fall back to this pass otherwise.
template <typename TG, typename T>
__global__ __launch_bounds__(256) void gsum_reduce_kernel(const TG* __restrict__ g, int ldg, const T* __restrict__ aux, int ldaux, double* __restrict__ red,
                                                          int hw, int c, int chunk) {
    PixMap pm(c);
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * chunk, p1 = min(hw, p0 + chunk);
    double v[2][4] = {};
    if (pm.active) {
        constexpr int U = 4;
        int p = p0 + pm.pp;
        for (; p + (U - 1) * pm.PP < p1; p += U * pm.PP) {
            f32x4 gv[U], x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                gv[u] = ld4(g + ((size_t)n * hw + p + u * pm.PP) * ldg + pm.cl * 4);
                x[u] = ld4(aux + ((size_t)n * hw + p + u * pm.PP) * ldaux + pm.cl * 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sg = 0.f, sx = 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    sg += gv[u][e];
                    sx += gv[u][e] * x[u][e];
                }
                v[0][e] += (double)sg;
                v[1][e] += (double)sx;
            }
        }
        for (; p < p1; p += pm.PP) {
            const f32x4 gv = ld4(g + ((size_t)n * hw + p) * ldg + pm.cl * 4);
            const f32x4 x = ld4(aux + ((size_t)n * hw + p) * ldaux + pm.cl * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[0][e] += (double)gv[e];
                v[1][e] += (double)gv[e] * (double)x[e];
            }
        }
    }
    block_reduce_atomic<2>(v, pm, red + (size_t)n * c * 2, c, true);
}

int shm_gsum_reduce_internal(const void* g, int ldg, const void* aux, int ldaux, double* red, int batch, int hw, int c, int dtype, hipStream_t st) {
    SHM_CHECK_C(c, "gsum reduce");
    SHM_REQUIRE(ldg % 4 == 0 && ldaux % 4 == 0, SHM_E_SHAPE, "gsum reduce: bad pitch");
    if (batch == 0 || hw == 0) return SHM_OK;
    const int chunk = shm_cdiv(hw, pix_chunks(hw, batch, c, 1024));
    const dim3 grid(shm_cdiv(hw, chunk), batch);
    SHM_DISPATCH_G(dtype, "gsum reduce", hipLaunchKernelGGL((gsum_reduce_kernel<TG, T>), grid, dim3(256), 0, st, (const TG*)g, ldg, (const T*)aux, ldaux, red, hw, c, chunk));
    SHM_LAUNCH_CHECK("gsum reduce");
    return SHM_OK;
}

// dbias[ch] += sum over slots of part[slot*c + ch]
// `clear` != null: also zero the 2*nslot*c reduction sums in front of `part` (shm_in_bwd's scratch is zero on return).
// Block = 64 channels x 4 slot groups (a serial loop over the slots per channel was latency bound: 10 us per launch).
__global__ __launch_bounds__(256) void dbias_fold_kernel(double* __restrict__ part, double* __restrict__ dbias, int nslot, int c,
                                                         double* __restrict__ clear, double* __restrict__ keep) {
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (ch < c)
        for (int i = g; i < nslot; i += 4) {
            const double v = part[(size_t)i * c + ch];
            s += v;
            if (keep) keep[(size_t)i * c + ch] = v;
            part[(size_t)i * c + ch] = 0.0;
            if (clear) {
                clear[((size_t)i * c + ch) * 2] = 0.0;
                clear[((size_t)i * c + ch) * 2 + 1] = 0.0;
            }
        }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && ch < c) dbias[ch] += (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// shm_in_bwd_keep_dz_sums: the next shm_in_bwd / shm_in_bwd_apply / shm_in_bwd_rank1 call of this thread also copies out the per-sample
// channel sums of dz ([batch][c] float64) it stages on the way to the bias gradient (the second term of a SHM_NORM_SCALED weight
// gradient needs them per sample).  One-shot.
static thread_local double* g_keep_dzsum = nullptr;
extern "C" int shm_in_bwd_keep_dz_sums(double* dst) {
    g_keep_dzsum = dst;
    return SHM_OK;
}
// shm_in_bwd_fused_scratch: scratch of the one-pass bf16 form (in_bwd_fused8_kernel) for the next shm_in_bwd call of this thread: n_doubles >=
// SHM_IN_BWD_FUSED_DOUBLES(batch, h * w, c) float64 (fused_scratch_doubles above), zero on entry, zero on return.  One-shot; without it (or on shapes the one-pass form does
// not take) shm_in_bwd runs its two passes.
static thread_local double* g_fused_scratch = nullptr;
static thread_local size_t g_fused_doubles = 0;
extern "C" int shm_in_bwd_fused_scratch(double* scratch, size_t n_doubles) {
    g_fused_scratch = scratch;
    g_fused_doubles = scratch ? n_doubles : 0;
    return SHM_OK;
}

// shm_set_abort_words: where a kernel that had to give up (today: a barrier of in_bwd_fused8_kernel that timed out) says so for THIS thread's
// later calls.  dev_word: u32 in device memory, OR-ed to non-zero; shm_adam_clip reads it on the device and applies NOTHING while it is set, so a
// gradient built on unfinished sums never reaches the weights, however far the host has run ahead.  host_word: u32 in mapped (pinned) host
// memory, set to 1 by the same kernel: the caller polls it without a device synchronisation.  Both stay set until the caller clears them.
// NULLs disarm.  Persistent per thread (like the tuning table this is configuration, not data-path state).
static thread_local unsigned* g_abort_dev = nullptr;
static thread_local unsigned* g_abort_host = nullptr;
extern "C" int shm_set_abort_words(unsigned* dev_word, unsigned* host_word) {
    g_abort_dev = dev_word;
    g_abort_host = host_word;
    return SHM_OK;
}
const unsigned* shm_abort_dev_word() { return g_abort_dev; }
// shm_set_clock_probe: measurement hook (bench.py's north-star ceiling).  While set on this thread, the ping-pong convolution kernel
// (tapgemm_pp_bf16_kernel) writes, from one wave of its middle block, dev2[0] = s_memtime ticks (shader clock) and dev2[1] = s_memrealtime ticks
// (100 MHz) spent in its patch loop: dev2[0] / dev2[1] x 0.1 = the clock in GHz the kernel held.  NULL (default) disarms; no other kernel reads it.
static thread_local unsigned long long* g_clock_probe = nullptr;
extern "C" int shm_set_clock_probe(unsigned long long* dev2) {
    g_clock_probe = dev2;
    return SHM_OK;
}
unsigned long long* shm_clock_probe() { return g_clock_probe; }

// Blocks of in_bwd_fused8_kernel<G2> the current device holds at once (CUs x occupancy; queried once per device and form).  The kernel's
// barrier only completes if a whole group is resident, and two such launches may run side by side (two streams), each stuck with LESS than a
// group resident only while free slots remain -- so a group is limited to HALF of this figure (advisor, round 5: a CPX partition, a CU mask or a
// smaller part holds far fewer than the 1024 / 768 blocks of a whole MI355X, and the launcher used to assume them).  0 if the query fails.
static int fused_resident_blocks(int form) {          // 0: in_bwd_fused8_kernel<false>, 1: <true> (pooled), 2: in_bwd_fusedg_kernel<2, 2, 4>, 3: <8, 8, 3>
    static int cache[16][4];                 // 0 = not asked yet, -1 = the query failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    int v = __atomic_load_n(&cache[dev][form], __ATOMIC_RELAXED);
    if (v == 0) {
        int cus = 0, per_cu = 0;
        hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess)
            e = form == 1   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, in_bwd_fused8_kernel<true>, 256, 0)
                : form == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, in_bwd_fused8_kernel<false>, 256, 0)
                : form == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, in_bwd_fusedg_kernel<2, 2, 4>, 256, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, in_bwd_fusedg_kernel<8, 8, 3>, 256, 0);
        if (e != hipSuccess) (void)hipGetLastError();
        v = (e == hipSuccess && cus > 0 && per_cu > 0) ? cus * per_cu : -1;
        __atomic_store_n(&cache[dev][form], v, __ATOMIC_RELAXED);
    }
    return v > 0 ? v : 0;
}

static int in_bwd_impl(const char* who, const void* g1, int ldg1, const void* g2, int ldg2, const float* r1_dz, const float* r1_w, const void* a, int lda,
                       const double* stats, double* red, void* dz, int lddz, double* dbias, int batch, int h, int w, int c, float slope, int dtype,
                       void* stream) {
    const bool r1 = r1_dz != nullptr;
    double* const keep = g_keep_dzsum;
    g_keep_dzsum = nullptr;
    double* const fscr = g_fused_scratch;
    const size_t fscr_n = g_fused_doubles;
    g_fused_scratch = nullptr;
    g_fused_doubles = 0;
    SHM_REQUIRE(!keep || dbias, SHM_E_SHAPE, "%s: the per-sample dz sums are staged only with a bias gradient", who);
    SHM_CHECK_C(c, who);
    SHM_REQUIRE((r1 || ldg1 % 4 == 0) && lda % 4 == 0 && lddz % 4 == 0 && (!g2 || ldg2 % 4 == 0), SHM_E_SHAPE, "%s: bad pitch", who);
    SHM_REQUIRE(!g2 || (h % 2 == 0 && w % 2 == 0), SHM_E_SHAPE, "%s: pooled gradient needs even h,w", who);
    SHM_REQUIRE(!(r1 && g2), SHM_E_SHAPE, "%s: the rank-1 form takes no pooled gradient", who);
    if (batch == 0 || h * w == 0) return SHM_OK;
    hipStream_t st = (hipStream_t)stream;
    // `red` is zero on entry by contract and zero again on return (no memset in front of every launch)
    InBwdArgs k{g1, g2, a, stats, red, dz, dbias, ldg1, ldg2, lda, lddz, h, w, c, 0, slope, shm_tune(SHM_TUNE_ELEM_REVERSE), r1_dz, r1_w};
    const int hw = h * w;
    k.nbatch = batch;
    InBwdArgs kr = k;
    k.interleave = kr.interleave = shm_tune(SHM_TUNE_ELEM_INTERLEAVE);
    k.nt = shm_tune(SHM_TUNE_ELEM_NT);          // the apply pass is the last reader of g1
    const int rb = shm_tune(SHM_TUNE_ELEM_REDUCE_BLOCKS);
    // bf16 activations: the reduce pass with eight channels (16 bytes) per thread
    const bool wide8 = (dtype == SHM_BF16 || dtype == SHM_BF16_GF32) && c % 8 == 0 && c >= 8 && c <= 1024 && (r1 || ldg1 % 8 == 0) && lda % 8 == 0 &&
                       (!g2 || ldg2 % 8 == 0) && 256 / (c / 8) >= 1;
    // The one-pass form (in_bwd_fused8_kernel, "elem.fused_bwd"): bf16 tensors, barrier groups of (sample, CB = min(c, 64) channels), whole slices of
    // 16384 / CB pixels, at most "elem.fused_max_slices" (256) blocks per group; with a pooled gradient: 64-channel groups and whole tiles
    {
        const int cb = c < 64 ? c : 64;
        const bool cb_ok = c >= 8 && (c < 64 ? (c & (c - 1)) == 0 : c % 64 == 0);
        const int slice = cb_ok ? 16384 / cb : 1;
        // pooled form: tiles of (256 / Wt) rows x Wt = min(w, 128) columns
        const int wt = w < 128 ? w : 128;
        const bool g2_tiles = cb == 64 && wt >= 2 && (wt & (wt - 1)) == 0 && w % wt == 0 && 256 % wt == 0 && (256 / wt) % 2 == 0 && h % (256 / wt) == 0;
        const bool base_ok = fscr && shm_tune(SHM_TUNE_ELEM_FUSED_BWD) && dtype == SHM_BF16 && !r1 && cb_ok && c <= 1024 && ldg1 % 8 == 0 && lda % 8 == 0 &&
                             lddz % 8 == 0 && (!g2 || ldg2 % 8 == 0) && batch <= 65535 && fscr_n >= fused_scratch_doubles(batch, hw, c);
        const int max_slices = shm_tune(SHM_TUNE_ELEM_FUSED_MAX_SLICES);
        const int hold = shm_tune(SHM_TUNE_ELEM_FUSED_HOLD);          // 0 automatic, 1 the round-5 kernel only (g and a held), 2 the g-held kernel only
        const int ncb = c / cb;
        const size_t red_bytes = (size_t)batch * c * 3 * sizeof(double);
        k.fold = keep ? 0 : 1;             // (the per-sample sums are wanted too -- SHM_NORM_SCALED's second term: the separate fold kernel copies them out)
        // round 6: g held, a streamed twice; slices of 32768 / CB pixels (sixteen pixel slots per thread), half the blocks per group.  Groups of up
        // to 512 blocks where twice that fits the device (the 512 x 512 x 64 maps of BASELINE configs[3]): the knob's default 256 bounds the round-5
        // kernel, this one takes 2 x its value
        // Measured (tools/probes/in_bwd_fusedg_ab.py, n = 40 / 160): it wins where a group has many blocks -- 256 x 256 x 64: 287 -> 252 us, n = 160:
        // 1072 -> 879 us -- and loses on the smaller maps, whose short groups do not cover its two extra round trips (128 x 128 x 128: 122 -> 147 us):
        // automatic dispatch takes it from 256 slices of the 8-slot kind per group on.
        const int slice16 = 2 * slice;
        const int fgv = shm_tune(SHM_TUNE_ELEM_FUSED_GVARIANT);
        if (base_ok && hold != 1 && !g2 && hw % slice16 == 0 && (hold == 2 || hw / slice >= 256) && hw / slice16 <= 2 * max_slices &&
            2 * (hw / slice16) <= fused_resident_blocks(fgv == 1 ? 3 : 2)) {
            const int bpi = hw / slice16;
            // ONE scratch layout for both kernels (the caller's buffer is "zero behind the partial rows" whichever kernel ran last): means, counters
            // and flags sit behind the rows region of the 16384 / CB-pixel slicing; this kernel's rows fill half of it
            float* const fres = (float*)(fscr + fused_row_doubles(batch, (size_t)hw * cb / 16384, c));
            unsigned* const fsync = (unsigned*)(fres + (size_t)batch * c * 2);
            unsigned* const ferr = fsync + (size_t)batch * ncb * SHM_FUSED_SYNC_WORDS;
            const dim3 gridf(bpi, ncb, batch);
            const unsigned arrivals = gridf.x + (shm_tune(SHM_TUNE_ELEM_FUSED_TEST_STALL) ? 1u : 0u);
            if (fgv == 1) hipLaunchKernelGGL((in_bwd_fusedg_kernel<8, 8, 3>), gridf, dim3(256), 0, st, k, (float*)fscr, fres, fsync, ferr, g_abort_dev, g_abort_host, arrivals);
            else hipLaunchKernelGGL((in_bwd_fusedg_kernel<2, 2, 4>), gridf, dim3(256), 0, st, k, (float*)fscr, fres, fsync, ferr, g_abort_dev, g_abort_host, arrivals);
            shm_set_last_kernel(fgv == 1 ? "in_bwd_fusedg_kernel<8, 8, 3>" : "in_bwd_fusedg_kernel<2, 2, 4>");
            SHM_LAUNCH_CHECK_CLEAR("shm_in_bwd(fused)", red, red_bytes, st);
            if (dbias && !k.fold) {
                hipLaunchKernelGGL(dbias_fold_kernel, dim3(shm_cdiv(c, 64)), dim3(256), 0, st, red + (size_t)batch * c * 2, dbias, batch, c, (double*)nullptr, keep);
                SHM_LAUNCH_CHECK_CLEAR("shm_in_bwd(fold)", red, red_bytes, st);
            }
            return SHM_OK;
        }
        if (base_ok && hold != 2 && hw % slice == 0 && hw / slice <= max_slices && (!g2 || g2_tiles) && 2 * (hw / slice) <= fused_resident_blocks(g2 ? 1 : 0)) {
            float* const fres = (float*)(fscr + fused_row_doubles(batch, hw / slice, c));
            unsigned* const fsync = (unsigned*)(fres + (size_t)batch * c * 2);
            unsigned* const ferr = fsync + (size_t)batch * ncb * SHM_FUSED_SYNC_WORDS;
            const dim3 gridf(hw / slice, ncb, batch);
            const unsigned arrivals = gridf.x + (shm_tune(SHM_TUNE_ELEM_FUSED_TEST_STALL) ? 1u : 0u);
            if (g2) hipLaunchKernelGGL((in_bwd_fused8_kernel<true>), gridf, dim3(256), 0, st, k, (float*)fscr, fres, fsync, ferr, g_abort_dev, g_abort_host, arrivals);
            else hipLaunchKernelGGL((in_bwd_fused8_kernel<false>), gridf, dim3(256), 0, st, k, (float*)fscr, fres, fsync, ferr, g_abort_dev, g_abort_host, arrivals);
            shm_set_last_kernel(g2 ? "in_bwd_fused8_kernel<true>" : "in_bwd_fused8_kernel<false>");
            SHM_LAUNCH_CHECK_CLEAR("shm_in_bwd(fused)", red, red_bytes, st);
            if (dbias && !k.fold) {       // (the two sum planes in front of the staging were not used: nothing to clear)
                hipLaunchKernelGGL(dbias_fold_kernel, dim3(shm_cdiv(c, 64)), dim3(256), 0, st, red + (size_t)batch * c * 2, dbias, batch, c, (double*)nullptr, keep);
                SHM_LAUNCH_CHECK_CLEAR("shm_in_bwd(fold)", red, red_bytes, st);
            }
            return SHM_OK;
        }
    }
    // Sample chunks ("elem.chunk_mb", round 3): the apply pass re-reads what the reduce pass read.  On tensors larger than the 256 MiB
    // Infinity Cache that second read comes from HBM again (the back-to-front / front-to-back walk only saves the turning point);
    // run as reduce(chunk), apply(chunk) over chunks whose g + a fit the cache, the second read stays on die.  0 = one chunk.
    const int esz_a = dtype == SHM_F32 ? 4 : 2, esz_g = dtype == SHM_BF16 ? 2 : 4;
    const size_t per_sample = (size_t)hw * c * (esz_a + (r1 ? 0 : esz_g)) + (g2 ? (size_t)hw / 4 * c * esz_g : 0);
    const size_t chunk_bytes = (size_t)shm_tune(SHM_TUNE_ELEM_CHUNK_MB) << 20;
    int per_chunk = batch;
    if (chunk_bytes && per_sample * batch > chunk_bytes) {
        per_chunk = (int)(chunk_bytes / per_sample);
        if (per_chunk < 1) per_chunk = 1;
    }
    for (int n0 = 0; n0 < batch; n0 += per_chunk) {
        const int nb = min(per_chunk, batch - n0);
        k.n0 = kr.n0 = n0;
        // The reduce pass ends every block with an LDS combine and 2c f64 atomics onto the 2c addresses of its sample: with the
        // streaming pass's ~4096 blocks a sample's address takes up to 256 serialized adds (n = 8, 256 x 256: 125 us for a pass whose
        // data moves in 40) and the 512-channel maps issue 1.3 M atomics per launch.  Fewer, longer blocks -- about the same bytes per
        // block in both dtypes: bf16 step 27.8 -> 27.1 ms, fp32 123.4 -> 122.9.  (The apply pass, one atomic per channel and block, is faster with
        // its 4096 blocks: same grid for both measured +0.15 / +0.6 ms.)
        k.chunk = shm_cdiv(hw, pix_chunks(hw, nb, c, shm_tune(SHM_TUNE_ELEM_APPLY_BLOCKS)));
        kr.chunk = shm_cdiv(hw, pix_chunks(hw, nb, c, rb ? rb : (dtype == SHM_F32 ? 1024 : 512)));
        const dim3 grid(shm_cdiv(hw, k.chunk), nb), gridr(shm_cdiv(hw, kr.chunk), nb);
        if (wide8) {
            if (r1) {
                hipLaunchKernelGGL((in_bwd_reduce8_kernel<float, false, true>), gridr, dim3(256), 0, st, kr);
            } else if (dtype == SHM_BF16) {
                if (g2) hipLaunchKernelGGL((in_bwd_reduce8_kernel<bf16_t, true>), gridr, dim3(256), 0, st, kr);
                else hipLaunchKernelGGL((in_bwd_reduce8_kernel<bf16_t, false>), gridr, dim3(256), 0, st, kr);
            } else {
                if (g2) hipLaunchKernelGGL((in_bwd_reduce8_kernel<float, true>), gridr, dim3(256), 0, st, kr);
                else hipLaunchKernelGGL((in_bwd_reduce8_kernel<float, false>), gridr, dim3(256), 0, st, kr);
            }
        } else if (r1) {
            SHM_DISPATCH_G(dtype, who, hipLaunchKernelGGL((in_bwd_reduce_kernel<T, TG, false, true>), gridr, dim3(256), 0, st, kr));
        } else if (g2) {
            SHM_DISPATCH_G(dtype, who, hipLaunchKernelGGL((in_bwd_reduce_kernel<T, TG, true>), gridr, dim3(256), 0, st, kr));
        } else {
            SHM_DISPATCH_G(dtype, who, hipLaunchKernelGGL((in_bwd_reduce_kernel<T, TG, false>), gridr, dim3(256), 0, st, kr));
        }
        SHM_LAUNCH_CHECK("shm_in_bwd(reduce)");
        if (r1)
            SHM_DISPATCH_G(dtype, who, hipLaunchKernelGGL((in_bwd_apply_kernel<T, TG, false, true>), grid, dim3(256), 0, st, k));
        else if (g2)
            SHM_DISPATCH_G(dtype, who, hipLaunchKernelGGL((in_bwd_apply_kernel<T, TG, true>), grid, dim3(256), 0, st, k));
        else
            SHM_DISPATCH_G(dtype, who, hipLaunchKernelGGL((in_bwd_apply_kernel<T, TG, false>), grid, dim3(256), 0, st, k));
    }
    shm_set_last_kernel(wide8 ? "in_bwd_reduce8_kernel + in_bwd_apply_kernel" : "in_bwd_reduce_kernel + in_bwd_apply_kernel");
    const size_t red_bytes = (size_t)batch * c * 3 * sizeof(double);
    SHM_LAUNCH_CHECK_CLEAR("shm_in_bwd(apply)", red, red_bytes, st);
    if (dbias) {
        hipLaunchKernelGGL(dbias_fold_kernel, dim3(shm_cdiv(c, 64)), dim3(256), 0, st, red + (size_t)batch * c * 2, dbias, batch, c, red, keep);
    } else {
        int r = shm_zero(red, (size_t)batch * c * 2 * sizeof(double), stream);
        if (r) return r;
    }
    SHM_LAUNCH_CHECK_CLEAR("shm_in_bwd(fold)", red, red_bytes, st);
    return SHM_OK;
}

extern "C" int shm_in_bwd(const void* g1, int ldg1, const void* g2, int ldg2, const void* a, int lda,
                          const double* stats, double* red, void* dz, int lddz, double* dbias, int batch,
                          int h, int w, int c, float slope, int dtype, void* stream) {
    SHM_REQUIRE(g1, SHM_E_SHAPE, "shm_in_bwd: null gradient");
    return in_bwd_impl("shm_in_bwd", g1, ldg1, g2, ldg2, nullptr, nullptr, a, lda, stats, red, dz, lddz, dbias, batch, h, w, c, slope, dtype, stream);
}

// InstanceNorm + LeakyReLU backward WITHOUT its reduce pass: the per-(sample, channel) sums were formed in the epilogues of the
// launches that wrote g1 / g2 (shm_conv2d_dgrad_gsum, shm_conv2d_fwd_gsum), so this is one pass over the tensors -- read g1 [+ g2],
// read a, write dz -- where shm_in_bwd makes two.
extern "C" int shm_in_bwd_apply(const void* g1, int ldg1, const void* g2, int ldg2, const void* a, int lda, const double* stats, const float* beta,
                                double* red, double* redp, double* dstage, void* dz, int lddz, double* dbias, int batch, int h, int w, int c,
                                float slope, int dtype, void* stream) {
    const char* who = "shm_in_bwd_apply";
    double* const keep = g_keep_dzsum;
    g_keep_dzsum = nullptr;
    SHM_REQUIRE(!keep || dbias, SHM_E_SHAPE, "%s: the per-sample dz sums are staged only with a bias gradient", who);
    SHM_REQUIRE(g1 && a && stats && red && dz, SHM_E_SHAPE, "%s: null pointer", who);
    SHM_REQUIRE((g2 != nullptr) == (redp != nullptr), SHM_E_SHAPE, "%s: the pooled gradient g2 and its sums redp come together", who);
    SHM_REQUIRE(!redp || beta, SHM_E_SHAPE, "%s: the pooled form needs beta", who);
    SHM_REQUIRE(!dbias || dstage, SHM_E_SHAPE, "%s: the bias gradient needs its staging scratch", who);
    SHM_CHECK_C(c, who);
    SHM_REQUIRE(ldg1 % 4 == 0 && lda % 4 == 0 && lddz % 4 == 0 && (!g2 || ldg2 % 4 == 0), SHM_E_SHAPE, "%s: bad pitch", who);
    SHM_REQUIRE(!g2 || (h % 2 == 0 && w % 2 == 0), SHM_E_SHAPE, "%s: pooled gradient needs even h,w", who);
    if (batch == 0 || h * w == 0) return SHM_OK;
    hipStream_t st = (hipStream_t)stream;
    InBwdArgs k{g1, g2, a, stats, nullptr, dz, dbias, ldg1, ldg2, lda, lddz, h, w, c, 0, slope, 0, nullptr, nullptr, red, redp, beta, dstage, SHM_GSUM_SLOTS,
                shm_tune(SHM_TUNE_ELEM_NT), 0, batch};
    k.interleave = shm_tune(SHM_TUNE_ELEM_INTERLEAVE);
    const int hw = h * w;
    k.chunk = shm_cdiv(hw, pix_chunks(hw, batch, c, shm_tune(SHM_TUNE_ELEM_APPLY_BLOCKS)));
    const dim3 grid(shm_cdiv(hw, k.chunk), batch);
    if (g2)
        SHM_DISPATCH_G(dtype, who, hipLaunchKernelGGL((in_bwd_apply_kernel<T, TG, true, false, true>), grid, dim3(256), 0, st, k));
    else
        SHM_DISPATCH_G(dtype, who, hipLaunchKernelGGL((in_bwd_apply_kernel<T, TG, false, false, true>), grid, dim3(256), 0, st, k));
    const size_t nred = (size_t)SHM_GSUM_SLOTS * batch * c * 2;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) {
        const size_t nclr = (nred * (redp ? 2 : 1) + 2047) / 2048;
        int nb = nclr < 64 ? (int)nclr : 64;
        if (nb < shm_cdiv(c, 64)) nb = shm_cdiv(c, 64);
        hipLaunchKernelGGL(gsum_finish_kernel, dim3(nb), dim3(256), 0, st, dstage, dbias, batch, c, red, nred, redp, redp ? nred : (size_t)0, keep);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {           // zero on return also on the error path
        (void)hipMemsetAsync(red, 0, nred * sizeof(double), st);
        if (redp) (void)hipMemsetAsync(redp, 0, nred * sizeof(double), st);
        if (dstage) (void)hipMemsetAsync(dstage, 0, (size_t)batch * c * sizeof(double), st);
        shm_set_error("%s: launch failed: %s", who, hipGetErrorString(e));
        return SHM_E_HIP;
    }
    return SHM_OK;
}

// The same backward for the block in front of the generator head, whose output gradient is the rank-1 tensor
// d_out[n, p, ch] = hdz[n * h * w + p] * hw_[ch] (shm_head_in_bwd's dz_out and the head kernel): formed on the fly, so the head
// never writes its input gradient and neither pass here reads it (three passes over the largest activation of the network).
extern "C" int shm_in_bwd_rank1(const float* hdz, const float* hw_, const void* a, int lda, const double* stats, double* red, void* dz, int lddz,
                                double* dbias, int batch, int h, int w, int c, float slope, int dtype, void* stream) {
    SHM_REQUIRE(hdz && hw_, SHM_E_SHAPE, "shm_in_bwd_rank1: null gradient");
    return in_bwd_impl("shm_in_bwd_rank1", nullptr, 0, nullptr, 0, hdz, hw_, a, lda, stats, red, dz, lddz, dbias, batch, h, w, c, slope, dtype, stream);
}

// ---------------------------------------------------------------------- LeakyReLU backward
template <typename T, typename TG>
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const TG* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy, T* __restrict__ dz, int lddz,
                                                        double* dpart, size_t npix, int c, size_t chunk, float slope) {
    PixMap pm(c);
    const size_t p0 = (size_t)blockIdx.x * chunk;
    const size_t p1 = p0 + chunk < npix ? p0 + chunk : npix;
    double v[1][4] = {};
    if (pm.active) {
        constexpr int U = sizeof(T) == 2 ? 8 : 4;
        size_t p = p0 + pm.pp;
        for (; p + (size_t)(U - 1) * pm.PP < p1; p += (size_t)U * pm.PP) {
            f32x4 g[U], x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g[u] = ld4(dy + (p + (size_t)u * pm.PP) * lddy + pm.cl * 4);
                x[u] = ld4(y + (p + (size_t)u * pm.PP) * ldy + pm.cl * 4);
            }
            float sd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d[e] = x[u][e] > 0.f ? g[u][e] : g[u][e] * slope;
                    sd[e] += d[e];
                }
                st4(dz + (p + (size_t)u * pm.PP) * lddz + pm.cl * 4, d);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[0][e] += (double)sd[e];
        }
        for (; p < p1; p += pm.PP) {
            f32x4 g = ld4(dy + p * lddy + pm.cl * 4);
            f32x4 x = ld4(y + p * ldy + pm.cl * 4);
            f32x4 d;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d[e] = x[e] > 0.f ? g[e] : g[e] * slope;
                v[0][e] += (double)d[e];
            }
            st4(dz + p * lddz + pm.cl * 4, d);
        }
    }
    if (dpart) block_reduce_atomic<1>(v, pm, dpart + (size_t)(blockIdx.x % SHM_LRELU_RED_SLOTS) * c, c, true);
}

extern "C" int shm_lrelu_bwd(const void* dy, int lddy, const void* y, int ldy, void* dz, int lddz,
                             double* dbias, double* red, size_t npix, int c, float slope, int dtype, void* stream) {
    SHM_REQUIRE(!dbias || red, SHM_E_SHAPE, "shm_lrelu_bwd: dbias needs the f64 scratch `red`");
    SHM_CHECK_C(c, "shm_lrelu_bwd");
    SHM_REQUIRE(lddy % 4 == 0 && ldy % 4 == 0 && lddz % 4 == 0, SHM_E_SHAPE, "shm_lrelu_bwd: bad pitch");
    if (npix == 0) return SHM_OK;
    int nch = pix_chunks((long)npix, 1, c, 4096);
    size_t chunk = (npix + nch - 1) / nch;
    SHM_DISPATCH_G(dtype, "shm_lrelu_bwd",
                 hipLaunchKernelGGL((lrelu_bwd_kernel<T, TG>), dim3(shm_cdiv((long)npix, (long)chunk)), dim3(256), 0, (hipStream_t)stream, (const TG*)dy, lddy,
                                    (const T*)y, ldy, (T*)dz, lddz, dbias ? red : nullptr, npix, c, chunk, slope));
    SHM_LAUNCH_CHECK("shm_lrelu_bwd");
    if (dbias) {
        hipLaunchKernelGGL(dbias_fold_kernel, dim3(shm_cdiv(c, 64)), dim3(256), 0, (hipStream_t)stream, red, dbias, SHM_LRELU_RED_SLOTS, c, (double*)nullptr, (double*)nullptr);
        SHM_LAUNCH_CHECK_CLEAR("shm_lrelu_bwd(fold)", red, (size_t)SHM_LRELU_RED_SLOTS * c * sizeof(double), (hipStream_t)stream);
    }
    return SHM_OK;
}

// -------------------------------------------------------------------------------- pooling
template <typename T>
__global__ void avgpool2_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int h, int w, int c4, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int cl = (int)(i % c4);
    size_t q = i / c4;                     // output pixel (n, oy, ox)
    int wo = w >> 1, ho = h >> 1;
    int ox = (int)(q % wo);
    size_t t = q / wo;
    int oy = (int)(t % ho);
    size_t n = t / ho;
    const T* b = x + ((n * h + 2 * oy) * w + 2 * ox) * ldx + cl * 4;
    f32x4 s = ld4(b) + ld4(b + ldx) + ld4(b + (size_t)w * ldx) + ld4(b + (size_t)(w + 1) * ldx);
    st4(y + q * ldy + cl * 4, s * 0.25f);
}

extern "C" int shm_avgpool2_fwd(const void* x, int ldx, void* y, int ldy, int batch, int h, int w, int c, int dtype, void* stream) {
    SHM_REQUIRE(c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, SHM_E_SHAPE, "shm_avgpool2_fwd: channels/pitch must be multiples of 4");
    SHM_REQUIRE(h % 2 == 0 && w % 2 == 0, SHM_E_SHAPE, "shm_avgpool2_fwd: odd size %dx%d", h, w);
    size_t total = (size_t)batch * (h / 2) * (w / 2) * (c / 4);
    if (total == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_avgpool2_fwd",
                 hipLaunchKernelGGL(avgpool2_kernel<T>, dim3(shm_cdiv((long)total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (T*)y, ldy, h, w,
                                    c / 4, total));
    SHM_LAUNCH_CHECK("shm_avgpool2_fwd");
    return SHM_OK;
}

// -------------------------------------------------------------------------- generator head
// y[p] = lrelu(sum_c x[p][c] w[c] + b); C/4 lanes per pixel (power of two <= 64).
// NORM: x is the UN-normalised activation of the last decoder block and the kernel applies its InstanceNorm on the fly
// (xh = (x - mean) * inv + beta, the expression of in_apply_kernel: identical fp32 values) -- the apply pass of that block and
// the normalised tensor do not exist.  grid.y = sample, npix = pixels per sample.
template <typename T, bool NORM>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, size_t npix, int c, float slope, const double* __restrict__ stats,
                                                       const float* __restrict__ beta) {
    const int lanes_c = c >> 2, PP = 256 / lanes_c;
    const int pp = threadIdx.x / lanes_c, cl = threadIdx.x % lanes_c;
    f32x4 wv = *(const f32x4*)(w + cl * 4);
    const float b = bias ? bias[0] : 0.f;
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, inv[4] = {1.f, 1.f, 1.f, 1.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NORM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = cl * 4 + e;
            mean[e] = (float)stats[((size_t)blockIdx.y * c + ch) * 2];
            inv[e] = (float)stats[((size_t)blockIdx.y * c + ch) * 2 + 1];
            bt[e] = beta[ch];
        }
        x += (size_t)blockIdx.y * npix * ldx;
        y += (size_t)blockIdx.y * npix;
    }
    for (size_t p = (size_t)blockIdx.x * PP + pp; p < npix; p += (size_t)gridDim.x * PP) {
        f32x4 xv = ld4(x + p * ldx + cl * 4);
        if constexpr (NORM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = (xv[e] - mean[e]) * inv[e] + bt[e];
        }
        float s = xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        for (int o = lanes_c >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (cl == 0) y[p] = shm_lrelu(s + b, slope);
    }
}

static bool pow2_le64(int v) { return v >= 1 && v <= 64 && (v & (v - 1)) == 0; }

extern "C" int shm_head_fwd(const void* x, int ldx, const float* w, const float* bias, float* y, size_t npix, int c, float slope, int dtype, void* stream) {
    SHM_REQUIRE(c % 4 == 0 && pow2_le64(c / 4) && ldx % 4 == 0, SHM_E_SHAPE, "shm_head_fwd: channels %d unsupported", c);
    if (npix == 0) return SHM_OK;
    int PP = 256 / (c / 4);
    long blocks = ((long)npix + PP - 1) / PP;
    if (blocks > 8192) blocks = 8192;
    SHM_DISPATCH(dtype, "shm_head_fwd",
                 hipLaunchKernelGGL((head_fwd_kernel<T, false>), dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, w, bias, y, npix, c, slope,
                                    (const double*)nullptr, (const float*)nullptr));
    SHM_LAUNCH_CHECK("shm_head_fwd");
    return SHM_OK;
}

template <typename T, typename TG, bool NORM>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ y, const float* __restrict__ dy,
                                                       TG* __restrict__ dx, int lddx, double* dpart, size_t npix, int c, float slope,
                                                       const double* __restrict__ stats, const float* __restrict__ beta, float* __restrict__ dz_out) {
    PixMap pm(c);
    f32x4 wv = *(const f32x4*)(w + pm.cl * 4);
    // NORM (see head_fwd_kernel): x un-normalised, grid.y = sample, npix = pixels per sample; dx is the gradient at the NORMALISED
    // activation (what shm_in_bwd takes), the weight gradient uses the normalised value
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, inv[4] = {1.f, 1.f, 1.f, 1.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NORM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = pm.cl * 4 + e;
            mean[e] = (float)stats[((size_t)blockIdx.y * c + ch) * 2];
            inv[e] = (float)stats[((size_t)blockIdx.y * c + ch) * 2 + 1];
            bt[e] = beta[ch];
        }
        x += (size_t)blockIdx.y * npix * ldx;
        y += (size_t)blockIdx.y * npix;
        dy += (size_t)blockIdx.y * npix;
        if (dx) dx += (size_t)blockIdx.y * npix * lddx;
        if (dz_out) dz_out += (size_t)blockIdx.y * npix;
    }
    auto norm = [&](f32x4 v) {
        if constexpr (NORM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean[e]) * inv[e] + bt[e];
        }
        return v;
    };
    double v[1][4] = {};
    double dbs = 0.0;
    constexpr int U = 4;                   // pixels in flight per thread; partial sums in fp32, accumulated in f64
    const size_t stride = (size_t)gridDim.x * pm.PP;
    size_t p = (size_t)blockIdx.x * pm.PP + pm.pp;
    for (; p + (U - 1) * stride < npix; p += U * stride) {
        float dz[U];
        f32x4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t q = p + u * stride;
            const float g = dy[q];
            dz[u] = y[q] > 0.f ? g : g * slope;
            xv[u] = norm(ld4(x + q * ldx + pm.cl * 4));
        }
        float sw[4] = {0.f, 0.f, 0.f, 0.f}, sb = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (dx) st4(dx + (p + u * stride) * lddx + pm.cl * 4, wv * dz[u]);
            if (dz_out && pm.cl == 0) dz_out[p + u * stride] = dz[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) sw[e] += xv[u][e] * dz[u];
            sb += dz[u];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[0][e] += (double)sw[e];
        if (pm.cl == 0) dbs += (double)sb;
    }
    for (; p < npix; p += stride) {
        const float g = dy[p];
        const float dz = y[p] > 0.f ? g : g * slope;
        const f32x4 xv = norm(ld4(x + p * ldx + pm.cl * 4));
        if (dx) st4(dx + p * lddx + pm.cl * 4, wv * dz);
        if (dz_out && pm.cl == 0) dz_out[p] = dz;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[0][e] += (double)xv[e] * (double)dz;
        if (pm.cl == 0) dbs += (double)dz;
    }
    // staged per slot (slot = block % SHM_LRELU_RED_SLOTS): [slot][c] weight-gradient sums, then [slot] bias sums
    const int slot = (int)((blockIdx.x + blockIdx.y) % SHM_LRELU_RED_SLOTS);
    double* slotw = dpart + (size_t)slot * c;
    block_reduce_atomic<1>(v, pm, slotw, c, true);
    dbs = shm_wave_sum(dbs);
    if ((threadIdx.x & 63) == 0 && dbs != 0.0) atomicAdd(dpart + (size_t)SHM_LRELU_RED_SLOTS * c + slot, dbs);
}

__global__ void head_fold_kernel(const double* __restrict__ dpart, double* __restrict__ dw_acc, double* __restrict__ db_acc, int c) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch > c) return;
    double s = 0.0;
    if (ch < c) {
        for (int i = 0; i < SHM_LRELU_RED_SLOTS; ++i) s += dpart[(size_t)i * c + ch];
        dw_acc[ch] += s;
    } else {
        for (int i = 0; i < SHM_LRELU_RED_SLOTS; ++i) s += dpart[(size_t)SHM_LRELU_RED_SLOTS * c + i];
        db_acc[0] += s;
    }
}

extern "C" int shm_head_bwd(const void* x, int ldx, const float* w, const float* y, const float* dy, void* dx,
                            int lddx, double* dw_acc, double* db_acc, double* red, size_t npix, int c, float slope, int dtype, void* stream) {
    SHM_REQUIRE(c % 4 == 0 && pow2_le64(c / 4) && ldx % 4 == 0 && lddx % 4 == 0, SHM_E_SHAPE, "shm_head_bwd: channels %d unsupported", c);
    SHM_REQUIRE(red && dw_acc && db_acc, SHM_E_SHAPE, "shm_head_bwd: null accumulator / scratch");
    if (npix == 0) return SHM_OK;
    int r = shm_zero(red, (size_t)SHM_LRELU_RED_SLOTS * (c + 1) * sizeof(double), stream);
    if (r) return r;
    int PP = 256 / (c / 4);
    long blocks = ((long)npix + (long)PP * 8 - 1) / ((long)PP * 8);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    SHM_DISPATCH_G(dtype, "shm_head_bwd",
                 hipLaunchKernelGGL((head_bwd_kernel<T, TG, false>), dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, w, y, dy, (TG*)dx, lddx, red,
                                    npix, c, slope, (const double*)nullptr, (const float*)nullptr, (float*)nullptr));
    SHM_LAUNCH_CHECK("shm_head_bwd");
    hipLaunchKernelGGL(head_fold_kernel, dim3(shm_cdiv(c + 1, 256)), dim3(256), 0, (hipStream_t)stream, (const double*)red, dw_acc, db_acc, c);
    SHM_LAUNCH_CHECK("shm_head_bwd(fold)");
    return SHM_OK;
}

// The generator head on the UN-normalised activation of the last decoder block + that block's InstanceNorm statistics: the
// block's apply pass (a read and a write of the largest activation of the network) is folded into the head's forward and backward.
extern "C" int shm_head_in_fwd(const void* a, int lda, const double* stats, const float* beta, const float* w, const float* bias, float* y, int batch,
                               int hw, int c, float slope, int dtype, void* stream) {
    SHM_REQUIRE(c % 4 == 0 && pow2_le64(c / 4) && lda % 4 == 0, SHM_E_SHAPE, "shm_head_in_fwd: channels %d unsupported", c);
    SHM_REQUIRE(a && stats && beta && w && y, SHM_E_SHAPE, "shm_head_in_fwd: null pointer");
    if (batch == 0 || hw == 0) return SHM_OK;
    int PP = 256 / (c / 4);
    long blocks = ((long)hw + PP - 1) / PP;
    const long cap = 8192 / batch > 1 ? 8192 / batch : 1;
    if (blocks > cap) blocks = cap;
    SHM_DISPATCH(dtype, "shm_head_in_fwd",
                 hipLaunchKernelGGL((head_fwd_kernel<T, true>), dim3((int)blocks, batch), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, w, bias, y,
                                    (size_t)hw, c, slope, stats, beta));
    SHM_LAUNCH_CHECK("shm_head_in_fwd");
    return SHM_OK;
}

extern "C" int shm_head_in_bwd(const void* a, int lda, const double* stats, const float* beta, const float* w, const float* y, const float* dy, void* dx,
                               int lddx, float* dz_out, double* dw_acc, double* db_acc, double* red, int batch, int hw, int c, float slope, int dtype,
                               void* stream) {
    SHM_REQUIRE(c % 4 == 0 && pow2_le64(c / 4) && lda % 4 == 0 && (!dx || lddx % 4 == 0), SHM_E_SHAPE, "shm_head_in_bwd: channels %d unsupported", c);
    SHM_REQUIRE(dx || dz_out, SHM_E_SHAPE, "shm_head_in_bwd: neither dx nor dz_out");
    SHM_REQUIRE(a && stats && beta && red && dw_acc && db_acc, SHM_E_SHAPE, "shm_head_in_bwd: null pointer");
    if (batch == 0 || hw == 0) return SHM_OK;
    int r = shm_zero(red, (size_t)SHM_LRELU_RED_SLOTS * (c + 1) * sizeof(double), stream);
    if (r) return r;
    int PP = 256 / (c / 4);
    long blocks = ((long)hw + (long)PP * 8 - 1) / ((long)PP * 8);
    const long cap = 4096 / batch > 1 ? 4096 / batch : 1;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    SHM_DISPATCH_G(dtype, "shm_head_in_bwd",
                 hipLaunchKernelGGL((head_bwd_kernel<T, TG, true>), dim3((int)blocks, batch), dim3(256), 0, (hipStream_t)stream, (const T*)a, lda, w, y, dy, (TG*)dx,
                                    lddx, red, (size_t)hw, c, slope, stats, beta, dz_out));
    SHM_LAUNCH_CHECK("shm_head_in_bwd");
    hipLaunchKernelGGL(head_fold_kernel, dim3(shm_cdiv(c + 1, 256)), dim3(256), 0, (hipStream_t)stream, (const double*)red, dw_acc, db_acc, c);
    SHM_LAUNCH_CHECK("shm_head_in_bwd(fold)");
    return SHM_OK;
}

// ------------------------------------------------------------------------ PatchGAN logits
__device__ __forceinline__ float block_sum_256(float v) {
    __shared__ float ws[4];
    v = shm_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    return ws[0] + ws[1] + ws[2] + ws[3];
}

// one block per output pixel
template <typename T>
__global__ __launch_bounds__(256) void patch_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w, float* __restrict__ y, int h, int wd, int c, float slope) {
    const int q = blockIdx.x;              // (n, i, j)
    const int j = q % wd, t = q / wd;
    const int i = t % h, n = t / h;
    const int c4 = c >> 2;
    float s = 0.f;
    for (int it = threadIdx.x; it < 9 * c4; it += 256) {
        int tap = it / c4, cl = it - tap * c4;
        int ii = i + tap / 3 - 1, jj = j + tap % 3 - 1;
        if ((unsigned)ii < (unsigned)h && (unsigned)jj < (unsigned)wd) {
            f32x4 xv = ld4(x + ((size_t)(n * h + ii) * wd + jj) * ldx + cl * 4);
            f32x4 wv = *(const f32x4*)(w + (size_t)tap * c + cl * 4);
            s += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
        }
    }
    s = block_sum_256(s);
    if (threadIdx.x == 0) y[q] = shm_lrelu(s, slope);
}

extern "C" int shm_patch_fwd(const void* x, int ldx, const float* w, float* y, int batch, int h, int wd, int c, float slope, int dtype, void* stream) {
    SHM_REQUIRE(c % 4 == 0 && ldx % 4 == 0, SHM_E_SHAPE, "shm_patch_fwd: channels must be a multiple of 4");
    int total = batch * h * wd;
    if (total == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_patch_fwd", hipLaunchKernelGGL(patch_fwd_kernel<T>, dim3(total), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, w, y, h, wd, c, slope));
    SHM_LAUNCH_CHECK("shm_patch_fwd");
    return SHM_OK;
}

__global__ void patch_dz_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dz, int n, float slope) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dz[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
}

// dx[n,i,j,c] = sum_tap dz[n, i-(kh-1), j-(kw-1)] * w[tap][c]
template <typename T>
__global__ void patch_dx_kernel(const float* __restrict__ dz, const float* __restrict__ w, T* __restrict__ dx, int lddx, int h, int wd, int c4, size_t total) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int cl = (int)(idx % c4);
    size_t q = idx / c4;
    int j = (int)(q % wd);
    size_t t = q / wd;
    int i = (int)(t % h);
    size_t n = t / h;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 9; ++tap) {
        int ii = i - (tap / 3 - 1), jj = j - (tap % 3 - 1);
        if ((unsigned)ii < (unsigned)h && (unsigned)jj < (unsigned)wd) {
            float g = dz[(n * h + ii) * wd + jj];
            s += *(const f32x4*)(w + (size_t)tap * c4 * 4 + cl * 4) * g;
        }
    }
    st4(dx + q * lddx + cl * 4, s);
}

// dw[tap][c] = sum_{n,i,j} x[n,i+kh-1,j+kw-1,c] * dz[n,i,j]; block = (tap, 64 channels), 16 pixel groups
template <typename T>
__global__ __launch_bounds__(1024) void patch_dw_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ dz, float* __restrict__ dw, int batch, int h, int wd, int c) {
    __shared__ double red[16][64];
    const int tap = blockIdx.x, ch = blockIdx.y * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int dh = tap / 3 - 1, dwv = tap % 3 - 1;
    double s = 0.0;
    if (ch < c) {
        // pixel group g takes samples g, g + 16, ...; the tap's valid output window is a rectangle, so the inner loop has no index
        // division and no branch and its loads are independent (the flat loop over pixels it replaces ran one dependent load per
        // ~1 us: 193 us for 12.6 MB)
        const int i0 = dh < 0 ? -dh : 0, i1 = dh > 0 ? h - dh : h;
        const int j0 = dwv < 0 ? -dwv : 0, j1 = dwv > 0 ? wd - dwv : wd;
        for (int n = g; n < batch; n += 16) {
            const T* xn = x + (size_t)n * h * wd * ldx + ch;
            const float* dzn = dz + (size_t)n * h * wd;
            for (int i = i0; i < i1; ++i) {
                const T* xr = xn + (size_t)((i + dh) * wd + dwv) * ldx;
                const float* dr = dzn + i * wd;
#pragma unroll 8
                for (int j = j0; j < j1; ++j) s += (double)(float)xr[(size_t)j * ldx] * (double)dr[j];
            }
        }
    }
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && ch < c) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x & 63];
        dw[(size_t)tap * c + ch] = (float)t;
    }
}

extern "C" int shm_patch_bwd(const void* x, int ldx, const float* w, const float* y, const float* dy, float* dz,
                             void* dx, int lddx, float* dw, int batch, int h, int wd, int c, float slope, int dtype, void* stream) {
    SHM_REQUIRE(c % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0, SHM_E_SHAPE, "shm_patch_bwd: channels must be a multiple of 4");
    int npx = batch * h * wd;
    if (npx == 0) return SHM_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(patch_dz_kernel, dim3(shm_cdiv(npx, 256)), dim3(256), 0, st, y, dy, dz, npx, slope);
    SHM_LAUNCH_CHECK("shm_patch_bwd(dz)");
    size_t total = (size_t)npx * (c / 4);
    SHM_DISPATCH_G(dtype, "shm_patch_bwd",
                 hipLaunchKernelGGL(patch_dx_kernel<TG>, dim3(shm_cdiv((long)total, 256)), dim3(256), 0, st, (const float*)dz, w, (TG*)dx, lddx, h, wd, c / 4, total));
    SHM_LAUNCH_CHECK("shm_patch_bwd(dx)");
    if (dw) {
        SHM_DISPATCH_G(dtype, "shm_patch_bwd",
                     hipLaunchKernelGGL(patch_dw_kernel<T>, dim3(9, shm_cdiv(c, 64)), dim3(1024), 0, st, (const T*)x, ldx, (const float*)dz, dw, batch, h, wd, c));
        SHM_LAUNCH_CHECK("shm_patch_bwd(dw)");
    }
    return SHM_OK;
}

// --------------------------------------------------------------------------------- Dense(5)
constexpr int DENSE_MAX_OUT = 8;

template <typename T>
__global__ __launch_bounds__(256) void dense_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int k, int nout) {
    const int n = blockIdx.x;
    float acc[DENSE_MAX_OUT] = {};
    const T* xr = x + (size_t)n * k;
    if (nout == 5 && (k & 3) == 0 && ((size_t)xr & (4 * sizeof(T) - 1)) == 0 && ((size_t)w & 15) == 0) {
        // the classifier's shape (Dense(5)): four inputs x five outputs per iteration = one 8/16-byte load of x and five 16-byte
        // loads of w per lane, eight of them in flight (the scalar loop below is one dependent 4-byte load chain per lane:
        // 233 us for 96 samples of 65536 inputs, where the data is 14 MB)
        const int k4 = k >> 2;
#pragma unroll 2
        for (int i4 = threadIdx.x; i4 < k4; i4 += 256) {
            float xv[4];
            if constexpr (sizeof(T) == 4) {
                const f32x4 v = *(const f32x4*)(xr + 4 * (size_t)i4);
                xv[0] = v[0], xv[1] = v[1], xv[2] = v[2], xv[3] = v[3];
            } else {
                const uint2 v = *(const uint2*)(xr + 4 * (size_t)i4);
                xv[0] = __builtin_bit_cast(float, v.x << 16), xv[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
                xv[2] = __builtin_bit_cast(float, v.y << 16), xv[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
            }
            const f32x4* wp = (const f32x4*)(w + 20 * (size_t)i4);
            float wv[20];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const f32x4 t = wp[q];
                wv[4 * q] = t[0], wv[4 * q + 1] = t[1], wv[4 * q + 2] = t[2], wv[4 * q + 3] = t[3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[j] += xv[e] * wv[5 * e + j];
        }
    } else {
        for (int i = threadIdx.x; i < k; i += 256) {
            float xv = (float)xr[i];
            for (int j = 0; j < nout; ++j) acc[j] += xv * w[(size_t)i * nout + j];
        }
    }
    for (int j = 0; j < nout; ++j) {
        float s = block_sum_256(acc[j]);
        if (threadIdx.x == 0) y[(size_t)n * nout + j] = s;
    }
}

extern "C" int shm_dense_fwd(const void* x, const float* w, float* y, int batch, int k, int nout, int dtype, void* stream) {
    SHM_REQUIRE(nout >= 1 && nout <= DENSE_MAX_OUT, SHM_E_SHAPE, "shm_dense_fwd: nout %d > %d", nout, DENSE_MAX_OUT);
    if (batch == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_dense_fwd", hipLaunchKernelGGL(dense_fwd_kernel<T>, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const T*)x, w, y, k, nout));
    SHM_LAUNCH_CHECK("shm_dense_fwd");
    return SHM_OK;
}

// thread per k: dx[n][k] += sum_j dy[n][j] w[k][j];  dw[k][j] = sum_n x[n][k] dy[n][j]
template <typename T, typename TG>
__global__ __launch_bounds__(256) void dense_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy, TG* __restrict__ dx,
                                                        float* __restrict__ dw, int batch, int k, int nout) {
    extern __shared__ float sdy[];          // [batch][nout]
    for (int i = threadIdx.x; i < batch * nout; i += 256) sdy[i] = dy[i];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k) return;
    float wv[DENSE_MAX_OUT], acc[DENSE_MAX_OUT] = {};
    for (int j = 0; j < nout; ++j) wv[j] = w[(size_t)i * nout + j];
    int n = 0;
    for (; n + 4 <= batch; n += 4) {                  // four samples per iteration: eight independent loads in flight before the stores
        float xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = (float)x[(size_t)(n + u) * k + i];
            dv[u] = (float)dx[(size_t)(n + u) * k + i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float s = 0.f;
            for (int j = 0; j < nout; ++j) {
                float g = sdy[(n + u) * nout + j];
                s += g * wv[j];
                acc[j] += xv[u] * g;
            }
            dx[(size_t)(n + u) * k + i] = (TG)(dv[u] + s);
        }
    }
    for (; n < batch; ++n) {
        float xv = (float)x[(size_t)n * k + i];
        float s = 0.f;
        for (int j = 0; j < nout; ++j) {
            float g = sdy[n * nout + j];
            s += g * wv[j];
            acc[j] += xv * g;
        }
        dx[(size_t)n * k + i] = (TG)((float)dx[(size_t)n * k + i] + s);
    }
    if (dw)
        for (int j = 0; j < nout; ++j) dw[(size_t)i * nout + j] = acc[j];
}

extern "C" int shm_dense_bwd(const void* x, const float* w, const float* dy, void* dx, float* dw, int batch, int k, int nout, int dtype, void* stream) {
    SHM_REQUIRE(nout >= 1 && nout <= DENSE_MAX_OUT, SHM_E_SHAPE, "shm_dense_bwd: nout %d > %d", nout, DENSE_MAX_OUT);
    SHM_REQUIRE((size_t)batch * nout * 4 <= 48 * 1024, SHM_E_SHAPE, "shm_dense_bwd: batch %d too large", batch);
    if (batch == 0 || k == 0) return SHM_OK;
    SHM_DISPATCH_G(dtype, "shm_dense_bwd",
                 hipLaunchKernelGGL((dense_bwd_kernel<T, TG>), dim3(shm_cdiv(k, 256)), dim3(256), (size_t)batch * nout * 4, (hipStream_t)stream, (const T*)x, w,
                                    dy, (TG*)dx, dw, batch, k, nout));
    SHM_LAUNCH_CHECK("shm_dense_bwd");
    return SHM_OK;
}

// --------------------------------------------------------------------------- dropout mask
template <typename T>
__global__ void mul_mask_kernel(const T* __restrict__ x, const float* __restrict__ m, T* __restrict__ y, size_t n4, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = ld4(x + i * 4), b = ((const f32x4*)m)[i];
    st4(y + i * 4, a * b * scale);
}

extern "C" int shm_mul_mask(const void* x, const float* mask, void* y, size_t n, float scale, int dtype, void* stream) {
    SHM_REQUIRE(n % 4 == 0, SHM_E_SHAPE, "shm_mul_mask: n must be a multiple of 4");
    if (n == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_mul_mask",
                 hipLaunchKernelGGL(mul_mask_kernel<T>, dim3(shm_cdiv((long)(n / 4), 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, mask, (T*)y, n / 4, scale));
    SHM_LAUNCH_CHECK("shm_mul_mask");
    return SHM_OK;
}

// ------------------------------------------------------ live attention branch (SHM.py:404-412, 290-293, 359)
// MaxPooling2D(pool k x k, 'same' on sizes that are multiples of k) of the one-channel mask, written as channel 0 of an
// activation tensor of pitch ld (the other channels zero): the 1 -> C convolution of attention_layer then runs on the
// ordinary tap GEMM.  k = 1 copies (attention_layer(pool=False)).
template <typename T>
__global__ void mask_pool_pack_kernel(const float* __restrict__ m, T* __restrict__ dst, int ld, int s, int k, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // output pixel (b, y, x)
    if (i >= total) return;
    const int so = s / k;
    const int x = (int)(i % so), y = (int)((i / so) % so);
    const size_t b = i / ((size_t)so * so);
    const float* src = m + (b * s + (size_t)y * k) * s + (size_t)x * k;
    float v = src[0];
    for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) v = fmaxf(v, src[(size_t)dy * s + dx]);
    T* o = dst + i * ld;
    o[0] = (T)v;
    for (int c = 1; c < ld; ++c) o[c] = (T)0.f;
}

extern "C" int shm_mask_pool_pack(const float* mask, void* dst, int lddst, int batch, int s, int k, int dtype, void* stream) {
    SHM_REQUIRE(mask && dst && k >= 1 && s % k == 0 && lddst >= 1, SHM_E_SHAPE, "shm_mask_pool_pack: bad shape (s %d, k %d)", s, k);
    const size_t total = (size_t)batch * (s / k) * (s / k);
    if (total == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_mask_pool_pack",
                 hipLaunchKernelGGL(mask_pool_pack_kernel<T>, dim3(shm_cdiv((long)total, 256)), dim3(256), 0, (hipStream_t)stream, mask, (T*)dst, lddst, s, k, total));
    SHM_LAUNCH_CHECK("shm_mask_pool_pack");
    return SHM_OK;
}

// out[i] = a[i] + b[(i0 + i) % nb]  over images of `per` elements each: the skip tensor plus the attention map of its sample
// (`down_k + attn_k`, SHM.py:290-293; `x + attn_disc`, SHM.py:359), the attention map shared by every copy of a sample in the
// batched plan (image i of the batch belongs to sample (i0 + i) % nb).
template <typename T>
__global__ void add_bcast_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, size_t per4, int nb, int i0, size_t total4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const size_t img = i / per4, r = i - img * per4;
    const size_t j = ((size_t)i0 + img) % (size_t)nb;
    st4(out + i * 4, ld4(a + i * 4) + ld4(b + (j * per4 + r) * 4));
}

extern "C" int shm_add_bcast(const void* a, const void* b, void* out, int nimg, size_t per, int nb, int i0, int dtype, void* stream) {
    SHM_REQUIRE(a && b && out && per % 4 == 0 && nb >= 1 && i0 >= 0, SHM_E_SHAPE, "shm_add_bcast: bad arguments");
    const size_t total4 = (size_t)nimg * per / 4;
    if (total4 == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_add_bcast",
                 hipLaunchKernelGGL(add_bcast_kernel<T>, dim3(shm_cdiv((long)total4, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, (T*)out,
                                    per / 4, nb, i0, total4));
    SHM_LAUNCH_CHECK("shm_add_bcast");
    return SHM_OK;
}

// dst[j] (+)= sum over the images i of src with (i0 + i) % nb == j: the gradient of the broadcast above (fp32 accumulation).
template <typename T>
__global__ void sum_groups_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t per4, int nimg, int nb, int i0, int accumulate, size_t total4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // (sample j, element r)
    if (i >= total4) return;
    const size_t j = i / per4, r = i - j * per4;
    f32x4 s = accumulate ? ld4(dst + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    int first = (int)((j + (size_t)nb - (size_t)(i0 % nb)) % (size_t)nb);
    for (int img = first; img < nimg; img += nb) s += ld4(src + ((size_t)img * per4 + r) * 4);
    st4(dst + i * 4, s);
}

extern "C" int shm_sum_groups(const void* src, void* dst, int nimg, size_t per, int nb, int i0, int accumulate, int dtype, void* stream) {
    SHM_REQUIRE(src && dst && per % 4 == 0 && nb >= 1 && i0 >= 0, SHM_E_SHAPE, "shm_sum_groups: bad arguments");
    const size_t total4 = (size_t)nb * per / 4;
    if (total4 == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_sum_groups",
                 hipLaunchKernelGGL(sum_groups_kernel<T>, dim3(shm_cdiv((long)total4, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)src, (T*)dst, per / 4, nimg,
                                    nb, i0, accumulate, total4));
    SHM_LAUNCH_CHECK("shm_sum_groups");
    return SHM_OK;
}

// ------------------------------------------------ input gradient of a first layer, summed over input channels
// The step never needs the per-channel input gradient of the two first layers, only sums over input channels:
//   generator (cyclic pass, SHM.py:576-580): d genY[b,p] = sum_k sum_{j != k, flags[j]} dX_k[b,p,j]
//   discriminator (yuv_to_rgb backward):      d Y[i,p]   = sum_{c<3} dX[i,p,c]
// and conv is linear, so summing the WEIGHTS over those input channels first turns a 64 -> 10 (or 3) channel
// dgrad -- which fills 10 (3) columns of a 64-wide MFMA tile -- into a 64 -> 1 stencil that is HBM-bound:
//   out[b,y,x] (+)= sum_k sum_{taps} sum_co dz[k*batch+b, oy, ox, co] * weff[k][tap][co]
__global__ void weff_kernel(const float* __restrict__ w, int cin, int cout, unsigned mask, float* __restrict__ weff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (tap, co)
    if (i >= 9 * cout) return;
    const int t = i / cout, co = i - t * cout;
    float s = 0.f;
    for (int j = 0; j < cin; ++j)
        if ((mask >> j) & 1u) s += w[((size_t)t * cin + j) * cout + co];
    weff[i] = s;
}

extern "C" int shm_sum_input_channels(const float* w, int cin, int cout, unsigned mask, float* weff, void* stream) {
    SHM_REQUIRE(w && weff && cin >= 1 && cin <= 32 && cout >= 1, SHM_E_SHAPE, "shm_sum_input_channels: bad arguments");
    hipLaunchKernelGGL(weff_kernel, dim3(shm_cdiv(9 * cout, 256)), dim3(256), 0, (hipStream_t)stream, w, cin, cout, mask, weff);
    SHM_LAUNCH_CHECK("shm_sum_input_channels");
    return SHM_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void dgrad_sum1_kernel(const T* __restrict__ dz, int lddz, const float* __restrict__ weff, float* __restrict__ out, int nk,
                                                         int batch, int hi, int wi, int ho, int wo, int c, int stride, int pt, int pl, int accumulate) {
    const int lanes_c = c >> 2, PP = 256 / lanes_c;
    const int pp = threadIdx.x / lanes_c, cl = threadIdx.x - pp * lanes_c;
    const size_t npx = (size_t)batch * hi * wi;
    const size_t q = (size_t)blockIdx.x * PP + pp;             // output pixel (b, y, x)
    const bool live = q < npx && pp < PP;
    const size_t qq = live ? q : 0;
    const int x = (int)(qq % wi);
    const size_t t = qq / wi;
    const int y = (int)(t % hi), b = (int)(t / hi);
    float s = 0.f;
    for (int k = 0; k < nk; ++k) {
        const T* zi = dz + (size_t)(k * batch + b) * ho * wo * lddz + cl * 4;
        const float* wk = weff + (size_t)k * 9 * c + cl * 4;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ny = y + pt - kh;
            if (ny < 0 || (stride == 2 && (ny & 1))) continue;
            const int oy = stride == 2 ? ny >> 1 : ny;
            if (oy >= ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int nx = x + pl - kw;
                if (nx < 0 || (stride == 2 && (nx & 1))) continue;
                const int ox = stride == 2 ? nx >> 1 : nx;
                if (ox >= wo) continue;
                const f32x4 g = ld4(zi + ((size_t)oy * wo + ox) * lddz);
                const f32x4 wv = *(const f32x4*)(wk + (kh * 3 + kw) * c);
                s += g[0] * wv[0] + g[1] * wv[1] + g[2] * wv[2] + g[3] * wv[3];
            }
        }
    }
    for (int o = lanes_c >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (live && cl == 0) out[q] = accumulate ? out[q] + s : s;
}

// Tiled form (round 2): the pixel-per-thread-group kernel above reads every dz pixel nine times (once per output pixel it
// contributes to: 0.6 TB/s HBM-side, 290-560 us per launch in the r02 profiles).  Here a block owns a 16 x 16 tile of OUTPUT
// pixels: phase A turns every dz pixel the tile touches into its nine per-tap dot products P[t] = sum_k sum_c dz_k[.., c] *
// weff_k[t][c] (each dz pixel read once per block; weights staged in LDS), phase B gathers out[y, x] = sum of the valid taps'
// P entries from LDS.  Same sums, different order (fp32 accumulation; the parity tests hold it to 1e-5).
template <typename T>
__global__ __launch_bounds__(256) void dgrad_sum1_tiled_kernel(const T* __restrict__ dz, int lddz, const float* __restrict__ weff, float* __restrict__ out,
                                                               int nk, int batch, int hi, int wi, int ho, int wo, int c, int stride, int pt, int pl,
                                                               int accumulate) {
    constexpr int TO = 16, RMAX = TO + 2;
    __shared__ float P[9][RMAX * RMAX];
    __shared__ __attribute__((aligned(16))) float wl[5 * 9 * 64];
    const int lanes_c = c >> 2, PP = 256 / lanes_c;
    const int pp = threadIdx.x / lanes_c, cl = threadIdx.x - pp * lanes_c;
    const int tiles_x = (wi + TO - 1) / TO, tiles_y = (hi + TO - 1) / TO;
    const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - b * (tiles_x * tiles_y);
    const int y0 = (tr / tiles_x) * TO, x0 = (tr % tiles_x) * TO;
    // dz region the tile's outputs touch: oy = (y + pt - kh) / stride for kh in 0..2
    auto fdiv = [](int a, int d) { return a >= 0 ? a / d : -((-a + d - 1) / d); };
    const int oy_lo = fdiv(y0 + pt - 2, stride), oy_hi = fdiv(y0 + TO - 1 + pt, stride);
    const int ox_lo = fdiv(x0 + pl - 2, stride), ox_hi = fdiv(x0 + TO - 1 + pl, stride);
    const int R = oy_hi - oy_lo + 1, Cn = ox_hi - ox_lo + 1;             // <= 18 each
    for (int i = threadIdx.x; i < nk * 9 * c; i += 256) wl[i] = weff[i];
    __syncthreads();
    // ---- phase A
    if (pp < PP) {
        for (int j = pp; j < R * Cn; j += PP) {
            const int r = j / Cn, q = j - r * Cn;
            const int oy = oy_lo + r, ox = ox_lo + q;
            float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if ((unsigned)oy < (unsigned)ho && (unsigned)ox < (unsigned)wo) {
                for (int k = 0; k < nk; ++k) {
                    const f32x4 g = ld4(dz + (((size_t)(k * batch + b) * ho + oy) * wo + ox) * lddz + cl * 4);
                    const float* wk = wl + k * 9 * c + cl * 4;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const f32x4 wv = *(const f32x4*)(wk + t * c);
                        s[t] += g[0] * wv[0] + g[1] * wv[1] + g[2] * wv[2] + g[3] * wv[3];
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float v = s[t];
                for (int o = lanes_c >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (cl == 0) P[t][r * RMAX + q] = v;
            }
        }
    }
    __syncthreads();
    // ---- phase B: one output pixel per thread
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int y = y0 + ty, x = x0 + tx;
    if (y < hi && x < wi) {
        float v = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ny = y + pt - kh;
            if (ny < 0 || (stride == 2 && (ny & 1))) continue;
            const int oy = stride == 2 ? ny >> 1 : ny;
            if (oy >= ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int nx = x + pl - kw;
                if (nx < 0 || (stride == 2 && (nx & 1))) continue;
                const int ox = stride == 2 ? nx >> 1 : nx;
                if (ox >= wo) continue;
                v += P[kh * 3 + kw][(oy - oy_lo) * RMAX + (ox - ox_lo)];
            }
        }
        const size_t qo = ((size_t)b * hi + y) * wi + x;
        out[qo] = accumulate ? out[qo] + v : v;
    }
}

// MFMA form of the tiled kernel's phase A (round 4).  P[pixel][tap] = sum_k sum_c dz_k[pixel][c] * weff_k[tap][c] is a GEMM with M = the dz pixels of
// the tile (<= 18 x 18), N = 9 taps (16 MFMA columns) and K = nk * c: the phase above spends 36 FMAs and 36 shuffle-adds per lane and pixel on it
// (150 us in bf16 / 215 us in fp32 for a launch that moves 335 / 671 MB: vector bound, 2.2-3.1 TB/s).  Here a wave takes 16 pixels per step;
// a lane's 16-byte global load IS its A fragment (bf16: eight channels = one v_mfma_f32_16x16x32_bf16 K block; fp32: four channels = four
// v_mfma_f32_16x16x4_f32 with the channel permutation of tapgemm_wreg_f32_kernel), the weights sit in registers as the B operand -- in bf16 split
// into a high and a low bf16 part (two MFMAs), which keeps the fp32 weights' accuracy (2^-17) -- and the accumulator's four pixels of tap l15 go
// straight into the P image.  Phase B is unchanged.  c % 32 == 0 (bf16) / c % 16 == 0 (fp32), nk * c <= 320.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <typename T, int NKC>          // NKC = nk * c / (sizeof(T) == 2 ? 32 : 16): K blocks, compile time so that the weights stay in registers
__global__ __launch_bounds__(256) void dgrad_sum1_mfma_kernel(const T* __restrict__ dz, int lddz, const float* __restrict__ weff, float* __restrict__ out,
                                                              int nk, int batch, int hi, int wi, int ho, int wo, int c, int stride, int pt, int pl,
                                                              int accumulate) {
    constexpr int TO = 16, RMAX = TO + 2, KB = sizeof(T) == 2 ? 32 : 16;
    __shared__ float P[9][RMAX * RMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int tiles_x = (wi + TO - 1) / TO, tiles_y = (hi + TO - 1) / TO;
    const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - b * (tiles_x * tiles_y);
    const int y0 = (tr / tiles_x) * TO, x0 = (tr % tiles_x) * TO;
    auto fdiv = [](int a, int d) { return a >= 0 ? a / d : -((-a + d - 1) / d); };
    const int oy_lo = fdiv(y0 + pt - 2, stride), oy_hi = fdiv(y0 + TO - 1 + pt, stride);
    const int ox_lo = fdiv(x0 + pl - 2, stride), ox_hi = fdiv(x0 + TO - 1 + pl, stride);
    const int R = oy_hi - oy_lo + 1, Cn = ox_hi - ox_lo + 1;             // <= 18 each
    const int kpc = c / KB;                                               // K blocks per dz tensor
    // ---- weights -> registers: B[k][n = tap l15]; taps 9..15 are zero columns
    f32x4 wb[NKC];                       // fp32: four channels 4 lq + e; bf16: the high parts of eight channels 8 lq ..
    [[maybe_unused]] f32x4 wlo[sizeof(T) == 2 ? NKC : 1];
#pragma unroll
    for (int j = 0; j < NKC; ++j) {
        const int k = j / kpc, cb = (j - k * kpc) * KB;
        if constexpr (sizeof(T) == 2) {
            unsigned hi4[4] = {0, 0, 0, 0}, lo4[4] = {0, 0, 0, 0};
            if (l15 < 9) {
                const float* wp = weff + ((size_t)k * 9 + l15) * c + cb + lq * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float w = wp[e];
                    const bf16_t h = (bf16_t)w;
                    const bf16_t l = (bf16_t)(w - (float)h);
                    hi4[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, h) << (16 * (e & 1));
                    lo4[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, l) << (16 * (e & 1));
                }
            }
            wb[j] = __builtin_bit_cast(f32x4, u32x4_t{hi4[0], hi4[1], hi4[2], hi4[3]});
            wlo[j] = __builtin_bit_cast(f32x4, u32x4_t{lo4[0], lo4[1], lo4[2], lo4[3]});
        } else {
            wb[j] = l15 < 9 ? *(const f32x4*)(weff + ((size_t)k * 9 + l15) * c + cb + lq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- phase A: 16 dz pixels per wave and step
    const int npx = R * Cn;
    for (int g = wave; g * 16 < npx; g += 4) {
        const int j = g * 16 + l15;
        const int r = j / Cn, q = j - r * Cn;
        const int oy = oy_lo + r, ox = ox_lo + q;
        const bool ok = j < npx && (unsigned)oy < (unsigned)ho && (unsigned)ox < (unsigned)wo;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // every lane loads (lanes without a pixel read pixel 0 of their tensor and are zeroed afterwards): a load under a per-lane
        // condition inside the unrolled loop would be a branch and a full vmcnt drain per K block
        const size_t pixoff = ok ? ((size_t)oy * wo + ox) * lddz : 0;
        u32x4_t araw[NKC];
#pragma unroll
        for (int jj = 0; jj < NKC; ++jj) {
            const int k = jj / kpc, cb = (jj - k * kpc) * KB;
            araw[jj] = *(const u32x4_t*)(dz + (size_t)(k * batch + b) * ho * wo * lddz + pixoff + cb + lq * (sizeof(T) == 2 ? 8 : 4));
        }
#pragma unroll
        for (int jj = 0; jj < NKC; ++jj) {
            u32x4_t a = araw[jj];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = ok ? a[e] : 0u;
            if constexpr (sizeof(T) == 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, wb[jj]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, wlo[jj]), acc, 0, 0, 0);
            } else {
                const f32x4 af = __builtin_bit_cast(f32x4, a);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], wb[jj][e], acc, 0, 0, 0);
            }
        }
        // accumulator register e = pixel 16 g + 4 lq + e, column l15 = tap
        if (l15 < 9) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int jp = g * 16 + 4 * lq + e;
                if (jp < npx) {
                    const int rp = jp / Cn;
                    P[l15][rp * RMAX + (jp - rp * Cn)] = acc[e];
                }
            }
        }
    }
    __syncthreads();
    // ---- phase B: one output pixel per thread (as in dgrad_sum1_tiled_kernel)
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int y = y0 + ty, x = x0 + tx;
    if (y < hi && x < wi) {
        float v = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ny = y + pt - kh;
            if (ny < 0 || (stride == 2 && (ny & 1))) continue;
            const int oy = stride == 2 ? ny >> 1 : ny;
            if (oy >= ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int nx = x + pl - kw;
                if (nx < 0 || (stride == 2 && (nx & 1))) continue;
                const int ox = stride == 2 ? nx >> 1 : nx;
                if (ox >= wo) continue;
                v += P[kh * 3 + kw][(oy - oy_lo) * RMAX + (ox - ox_lo)];
            }
        }
        const size_t qo = ((size_t)b * hi + y) * wi + x;
        out[qo] = accumulate ? out[qo] + v : v;
    }
}

extern "C" int shm_conv3x3_dgrad_sum1(const void* dz, int lddz, const float* weff, float* out, int nk, int batch, int hi, int wi, int c, int stride,
                                      int accumulate, int dtype, void* stream) {
    SHM_REQUIRE(dz && weff && out, SHM_E_SHAPE, "shm_conv3x3_dgrad_sum1: null pointer");
    SHM_REQUIRE(c % 4 == 0 && pow2_le64(c / 4) && lddz % 4 == 0, SHM_E_SHAPE, "shm_conv3x3_dgrad_sum1: channels %d unsupported", c);
    SHM_REQUIRE(stride == 1 || stride == 2, SHM_E_SHAPE, "shm_conv3x3_dgrad_sum1: stride %d not in {1,2}", stride);
    int ho, wo, pt, pl;
    shm_same_pad(hi, 3, stride, &ho, &pt);
    shm_same_pad(wi, 3, stride, &wo, &pl);
    const size_t npx = (size_t)batch * hi * wi;
    if (npx == 0 || nk == 0) return SHM_OK;
    const int PP = 256 / (c / 4);
    {                                    // MFMA phase A: the step's shapes (nk = 1 or 5 tensors of 64 channels); K blocks are a template parameter
        const int kb = dtype == SHM_F32 ? 16 : 32;
        const int nkc = c % kb == 0 ? nk * c / kb : 0;
        const int tiles = shm_cdiv(hi, 16) * shm_cdiv(wi, 16);
        const dim3 grid(batch * tiles);
#define SHM_SUM1_MFMA(T_, NKC_)                                                                                                                           \
    hipLaunchKernelGGL((dgrad_sum1_mfma_kernel<T_, NKC_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)dz, lddz, weff, out, nk, batch, hi, wi, ho, wo, c, \
                       stride, pt, pl, accumulate)
        bool done = true;
        if (dtype == SHM_F32 && nkc == 4) SHM_SUM1_MFMA(float, 4);
        else if (dtype == SHM_F32 && nkc == 20) SHM_SUM1_MFMA(float, 20);
        else if (dtype == SHM_F32 && nkc == 1) SHM_SUM1_MFMA(float, 1);
        else if (dtype == SHM_F32 && nkc == 5) SHM_SUM1_MFMA(float, 5);
        else if (dtype == SHM_BF16 && nkc == 2) SHM_SUM1_MFMA(bf16_t, 2);
        else if (dtype == SHM_BF16 && nkc == 10) SHM_SUM1_MFMA(bf16_t, 10);
        else if (dtype == SHM_BF16 && nkc == 1) SHM_SUM1_MFMA(bf16_t, 1);
        else if (dtype == SHM_BF16 && nkc == 5) SHM_SUM1_MFMA(bf16_t, 5);
        else done = false;
#undef SHM_SUM1_MFMA
        if (done) {
            SHM_LAUNCH_CHECK("shm_conv3x3_dgrad_sum1");
            return SHM_OK;
        }
    }
    if (nk * c <= 5 * 64) {              // weights fit the tiled kernel's LDS staging
        const int tiles = shm_cdiv(hi, 16) * shm_cdiv(wi, 16);
        SHM_DISPATCH(dtype, "shm_conv3x3_dgrad_sum1",
                     hipLaunchKernelGGL(dgrad_sum1_tiled_kernel<T>, dim3(batch * tiles), dim3(256), 0, (hipStream_t)stream, (const T*)dz, lddz, weff, out, nk,
                                        batch, hi, wi, ho, wo, c, stride, pt, pl, accumulate));
        SHM_LAUNCH_CHECK("shm_conv3x3_dgrad_sum1");
        return SHM_OK;
    }
    SHM_DISPATCH(dtype, "shm_conv3x3_dgrad_sum1",
                 hipLaunchKernelGGL(dgrad_sum1_kernel<T>, dim3(shm_cdiv((long)npx, PP)), dim3(256), 0, (hipStream_t)stream, (const T*)dz, lddz, weff, out, nk,
                                    batch, hi, wi, ho, wo, c, stride, pt, pl, accumulate));
    SHM_LAUNCH_CHECK("shm_conv3x3_dgrad_sum1");
    return SHM_OK;
}
