// Probe: what limits a 2-read / 1-write streaming pass on gfx950?  Variants of out = f(a, g) over [n][hw][c] fp32 tensors:
//   A  grid-stride, 16 B per thread, U loads in flight          (the shape of a library elementwise kernel)
//   B  one contiguous pixel chunk per block (shm_in_bwd_apply's mapping), U pixels in flight per thread
//   C  B + per-channel constants and the InstanceNorm-backward arithmetic + per-thread f64 sums (no final reduction)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/elem_probe.hip -o tools/probes/elem_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void kA(const f32x4* __restrict__ a, const f32x4* __restrict__ g, f32x4* __restrict__ o, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { x[u] = a[i + u * stride]; y[u] = g[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) o[i + u * stride] = x[u] * 0.5f + y[u];
    }
    for (; i < n4; i += stride) o[i] = a[i] * 0.5f + g[i];
}

template <int U, bool MATH>
__global__ __launch_bounds__(256) void kB(const float* __restrict__ a, const float* __restrict__ g, float* __restrict__ o, int hw, int c, int chunk,
                                          const float* __restrict__ cst, double* __restrict__ sink) {
    const int lanes_c = c >> 2, PP = 256 / lanes_c, pp = threadIdx.x / lanes_c, cl = threadIdx.x - pp * lanes_c;
    const int n = blockIdx.y, p0 = blockIdx.x * chunk, p1 = min(hw, p0 + chunk);
    f32x4 mean = {0, 0, 0, 0}, inv = {1, 1, 1, 1}, m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0};
    if (MATH) { mean = *(const f32x4*)(cst + cl * 4); inv = *(const f32x4*)(cst + c + cl * 4); m1 = *(const f32x4*)(cst + 2 * c + cl * 4); m2 = *(const f32x4*)(cst + 3 * c + cl * 4); }
    double acc[4] = {0, 0, 0, 0};
    int p = p0 + pp;
    for (; p + (U - 1) * PP < p1; p += U * PP) {
        f32x4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t off = ((size_t)n * hw + p + u * PP) * c + cl * 4;
            x[u] = *(const f32x4*)(a + off);
            y[u] = *(const f32x4*)(g + off);
        }
        float sd[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 d;
            if (MATH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (x[u][e] - mean[e]) * inv[e];
                    const float da = inv[e] * (y[u][e] - m1[e] - xh * m2[e]);
                    d[e] = x[u][e] > 0.f ? da : da * 0.2f;
                    sd[e] += d[e];
                }
            } else {
                d = x[u] * 0.5f + y[u];
            }
            *(f32x4*)(o + ((size_t)n * hw + p + u * PP) * c + cl * 4) = d;
        }
        if (MATH)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (double)sd[e];
    }
    if (MATH && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345) sink[0] = acc[0];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 100.f;   // us per call
}

int main() {
    const int n = 40, hw = 256 * 256, c = 64;
    const size_t ne = (size_t)n * hw * c, bytes = ne * 4;
    float *a, *g, *o, *cst; double* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&g, bytes)); CK(hipMalloc(&o, bytes)); CK(hipMalloc(&cst, 4 * c * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0x3c, bytes)); CK(hipMemset(g, 0x3d, bytes)); CK(hipMemset(cst, 0, 4 * c * 4));
    auto rep = [&](const char* name, float us) { printf("%-44s %8.1f us  %6.0f GB/s\n", name, us, 3.0 * bytes / us / 1e3); fflush(stdout); };
    for (int blocks : {2048, 4096, 8192, 16384})
        for (int U : {2, 4}) {
            char nm[96]; snprintf(nm, sizeof nm, "A grid-stride U=%d blocks=%d", U, blocks);
            rep(nm, U == 2 ? timeit([&] { hipLaunchKernelGGL(kA<2>, dim3(blocks), dim3(256), 0, 0, (const f32x4*)a, (const f32x4*)g, (f32x4*)o, ne / 4); })
                           : timeit([&] { hipLaunchKernelGGL(kA<4>, dim3(blocks), dim3(256), 0, 0, (const f32x4*)a, (const f32x4*)g, (f32x4*)o, ne / 4); }));
        }
    for (int per : {26, 52, 103, 206, 412}) {            // blocks per sample (x 40 samples)
        const int chunk = (hw + per - 1) / per;
        char nm[96];
        snprintf(nm, sizeof nm, "B chunk/block U=4 blocks=%d", per * n);
        rep(nm, timeit([&] { hipLaunchKernelGGL((kB<4, false>), dim3(per, n), dim3(256), 0, 0, a, g, o, hw, c, chunk, cst, sink); }));
        snprintf(nm, sizeof nm, "B chunk/block U=8 blocks=%d", per * n);
        rep(nm, timeit([&] { hipLaunchKernelGGL((kB<8, false>), dim3(per, n), dim3(256), 0, 0, a, g, o, hw, c, chunk, cst, sink); }));
        snprintf(nm, sizeof nm, "C = B + in_bwd math + f64 sums U=4 blocks=%d", per * n);
        rep(nm, timeit([&] { hipLaunchKernelGGL((kB<4, true>), dim3(per, n), dim3(256), 0, 0, a, g, o, hw, c, chunk, cst, sink); }));
    }
    return 0;
}
