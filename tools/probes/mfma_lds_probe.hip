// Probe: ceiling of the tap-GEMM inner loop (ds_read_b128 fragments -> 32 v_mfma_f32_32x32x2_f32)
// without any global traffic, for 1..N blocks per CU and several fragment schedules.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: reads then 32 MFMAs (as the kernel); 1: next step's fragments prefetched under the MFMAs; 2: no LDS reads (register operands)
__global__ __launch_bounds__(256) void probe(float* out, int iters, int use_barrier) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 3 * 256 * 16; i += 256) smem[i] = (float)((i * 7) % 13) * 0.01f;
    __syncthreads();
    const int sw = (l31 >> 2) & 3;
    const int fo0 = l31 * 16 + ((0 + h) ^ sw) * 4, fo1 = l31 * 16 + ((2 + h) ^ sw) * 4;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 av[2][2], bv[2][2];
    auto rd = [&](int stage, f32x4 (&a)[2][2], f32x4 (&b)[2][2]) {
        const float* Ab = smem + stage * 4096 + wm * 1024;
        const float* Bb = smem + stage * 4096 + 2048 + wn * 1024;
        for (int kk = 0; kk < 2; ++kk) {
            const int fo = kk ? fo1 : fo0;
            for (int i = 0; i < 2; ++i) a[kk][i] = *(const f32x4*)(Ab + i * 512 + fo);
            for (int j = 0; j < 2; ++j) b[kk][j] = *(const f32x4*)(Bb + j * 512 + fo);
        }
    };
    auto mm = [&](f32x4 (&a)[2][2], f32x4 (&b)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i][e], b[kk][j][e], acc[i][j], 0, 0, 0);
    };
    int stage = 0;
    if (MODE == 1) rd(0, av, bv);
    for (int it = 0; it < iters; ++it) {
        if (use_barrier) __builtin_amdgcn_s_barrier();
        if (MODE == 0) {
            rd(stage, av, bv);
            mm(av, bv);
        } else if (MODE == 1) {
            f32x4 an[2][2], bn[2][2];
            const int ns = stage == 2 ? 0 : stage + 1;
            rd(ns, an, bn);
            mm(av, bv);
            for (int kk = 0; kk < 2; ++kk) for (int i = 0; i < 2; ++i) { av[kk][i] = an[kk][i]; bv[kk][i] = bn[kk][i]; }
        } else {
            if (it == 0) rd(0, av, bv);
            mm(av, bv);
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, int barrier, size_t lds) {
    const int iters = 4000;
    int grid = 256 * blocks_per_cu;
    float* out;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), lds, 0, out, 100, barrier);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), lds, 0, out, iters, barrier);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 32 * (32.0 * 32 * 2 * 2);
    printf("%-28s blocks/CU %d barrier %d lds %zu KB : %7.1f TFLOP/s\n", name, blocks_per_cu, barrier, lds / 1024, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    // LDS request controls residency: 48 KB -> 3 blocks/CU, 64 KB -> 2, 128 KB -> 1
    size_t sizes[3] = {128 * 1024, 64 * 1024, 48 * 1024};
    int bpc[3] = {1, 2, 3};
    for (int k = 0; k < 3; ++k) {
        run<2>("regs only (no LDS reads)", bpc[k], 0, sizes[k]);
        run<0>("reads then MFMAs", bpc[k], 0, sizes[k]);
        run<0>("reads then MFMAs", bpc[k], 1, sizes[k]);
        run<1>("next fragments prefetched", bpc[k], 0, sizes[k]);
        run<1>("next fragments prefetched", bpc[k], 1, sizes[k]);
    }
    return 0;
}
