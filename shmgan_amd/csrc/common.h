// Shared helpers for libshmgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/shmgan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void shm_set_error(const char* fmt, ...);

#define SHM_REQUIRE(cond, code, ...)  \
    do {                              \
        if (!(cond)) {                \
            shm_set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

#define SHM_LAUNCH_CHECK(name)                                                  \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            shm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return SHM_E_HIP;                                                   \
        }                                                                       \
    } while (0)

static inline int shm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// TF "SAME": out = ceil(in/s); pad_before = max((out-1)*s + k - in, 0) / 2
static inline void shm_same_pad(int in, int k, int s, int* out, int* before) {
    *out = (in + s - 1) / s;
    int tot = (*out - 1) * s + k - in;
    if (tot < 0) tot = 0;
    *before = tot / 2;
}

__device__ __forceinline__ float shm_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// wave64 sum via DPP-free shuffles
__device__ __forceinline__ double shm_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float shm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float shm_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ float shm_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
    return v;
}
