// Ceiling probe for BASELINE.json's north-star block (conv3x3 64 -> 64 + bias + LeakyReLU + InstanceNorm statistics at 256 x 256, n = 40, bf16):
// what can THIS chip, at the clock it holds under THIS kind of load, do on the block's two resources taken one at a time?
//
//   ceil_mfma_bf16  the block's FLOPs as a bare v_mfma_f32_16x16x32_bf16 loop on random operands held in registers: the product kernel's launch
//                   geometry (512 blocks of eight waves = four waves per SIMD), four independent accumulator tiles per wave as in the product, no
//                   LDS, no memory traffic inside the loop.  Also stamps s_memtime / s_memrealtime around the loop: in-kernel clock =
//                   d(memtime) / d(memrealtime) x 100 MHz (MI355X guide, "DVFS give-back" item 6).
//   ceil_copy       the block's algorithmic bytes as a 16-byte-per-lane streaming copy (read n*H*W*64 bf16, write the same).
//   ceil_copy_cfg   (round 6) the same copy in a chosen shape -- threads per block, 2 / 4 / 8 loads in flight per lane, blocks, plain or non-temporal
//                   accesses: bench.py calibrates the shape once on a 2 GB buffer (the round-5 probe's fixed shape reached 4.9 TB/s where the
//                   MI355X guide's float4 copy reaches 6.29) and copies the block's bytes with the best one.
//
// bench.py's north_star_block runs both in the process that times the product kernel, right after it, and prints
// ceiling_us = max(mfma_us, copy_us) and frac_of_ceiling = ceiling_us / us.  Test infrastructure: nothing in shmgan_amd/ loads this library.
//
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/probes/ceiling_ns_block.hip -o tools/probes/libceiling_ns_block.so   (__graft_entry__.build() does)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// operands: 64 lanes x 4 fragments x 16 bytes of random bf16 per wave slot (the same 4 KiB for every wave: the loop never re-reads it)
__global__ __launch_bounds__(512, 4) void ceil_mfma_kernel(const u32x4* __restrict__ ops, float* __restrict__ out, unsigned long long* __restrict__ stamps,
                                                          int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[2], b[2];
    a[0] = __builtin_bit_cast(bf16x8, ops[lane]);
    a[1] = __builtin_bit_cast(bf16x8, ops[64 + lane]);
    b[0] = __builtin_bit_cast(bf16x8, ops[128 + lane]);
    b[1] = __builtin_bit_cast(bf16x8, ops[192 + lane]);
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {           // 8 MFMAs per iteration
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[(m + j) & 1], acc[m], 0, 0, 0);
    }
    asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    // the accumulators stay live: one value per lane out (values grow; nobody reads them)
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (lane == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

// the same FLOPs as v_mfma_f32_32x32x16_bf16 (the ping-pong kernel's shape): four accumulator tiles of 16 registers, half as many instructions
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512, 2) void ceil_mfma32_kernel(const u32x4* __restrict__ ops, float* __restrict__ out, unsigned long long* __restrict__ stamps,
                                                            int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[2], b[2];
    a[0] = __builtin_bit_cast(bf16x8, ops[lane]);
    a[1] = __builtin_bit_cast(bf16x8, ops[64 + lane]);
    b[0] = __builtin_bit_cast(bf16x8, ops[128 + lane]);
    b[1] = __builtin_bit_cast(bf16x8, ops[192 + lane]);
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {           // 8 MFMAs per iteration
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(m + j) & 1], acc[m], 0, 0, 0);
    }
    asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (lane == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

__global__ __launch_bounds__(256) void ceil_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += 4 * stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
}

template <int U, bool NT>
__global__ void ceil_copy_cfg_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < n16) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < n16) {
                if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
                else dst[i + u * stride] = v[u];
            }
    }
}

extern "C" {
int ceil_copy_cfg(const void* src, void* dst, size_t bytes, int blocks, int threads, int unroll, int nt, void* stream) {
    const u32x4* s = (const u32x4*)src;
    u32x4* d = (u32x4*)dst;
    const size_t n16 = bytes / 16;
    hipStream_t st = (hipStream_t)stream;
    if (threads != 256 && threads != 512 && threads != 1024) return -2;
    if (unroll == 8 && nt) hipLaunchKernelGGL((ceil_copy_cfg_kernel<8, true>), dim3(blocks), dim3(threads), 0, st, s, d, n16);
    else if (unroll == 8) hipLaunchKernelGGL((ceil_copy_cfg_kernel<8, false>), dim3(blocks), dim3(threads), 0, st, s, d, n16);
    else if (unroll == 4 && nt) hipLaunchKernelGGL((ceil_copy_cfg_kernel<4, true>), dim3(blocks), dim3(threads), 0, st, s, d, n16);
    else if (unroll == 4) hipLaunchKernelGGL((ceil_copy_cfg_kernel<4, false>), dim3(blocks), dim3(threads), 0, st, s, d, n16);
    else if (unroll == 2 && nt) hipLaunchKernelGGL((ceil_copy_cfg_kernel<2, true>), dim3(blocks), dim3(threads), 0, st, s, d, n16);
    else if (unroll == 2) hipLaunchKernelGGL((ceil_copy_cfg_kernel<2, false>), dim3(blocks), dim3(threads), 0, st, s, d, n16);
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// mfmas_per_wave is rounded down to a multiple of 8; returns the number of MFMAs each wave issues (or < 0 on a launch error)
int ceil_mfma_bf16(const void* ops, float* out, unsigned long long* stamps, int blocks, int mfmas_per_wave, void* stream) {
    const int iters = mfmas_per_wave / 8;
    hipLaunchKernelGGL(ceil_mfma_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const u32x4*)ops, out, stamps, iters);
    return hipGetLastError() == hipSuccess ? iters * 8 : -1;
}
// the 32x32x16 form: mfmas_per_wave counts 32x32x16 instructions (32 768 FLOP each); two waves per SIMD at `blocks` = CUs
int ceil_mfma32_bf16(const void* ops, float* out, unsigned long long* stamps, int blocks, int mfmas_per_wave, void* stream) {
    const int iters = mfmas_per_wave / 8;
    hipLaunchKernelGGL(ceil_mfma32_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const u32x4*)ops, out, stamps, iters);
    return hipGetLastError() == hipSuccess ? iters * 8 : -1;
}
int ceil_copy(const void* src, void* dst, size_t bytes, int blocks, void* stream) {
    hipLaunchKernelGGL(ceil_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, bytes / 16);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
}
