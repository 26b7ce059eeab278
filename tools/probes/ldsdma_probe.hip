// Probe: raw_buffer_load_lds (16 B) lane->LDS mapping and out-of-range behaviour on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(const float* src, unsigned nbytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 256];      // 2 KB
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -7.f;       // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned lane = threadIdx.x;
    // lanes 0..47 in range (reversed order of source), lanes 48..63 out of range
    unsigned off = lane < 48 ? (47 - lane) * 16 : 0xffffffffu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 64), 16, (int)off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<float> h(48 * 4);
    for (int i = 0; i < 48 * 4; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, (unsigned)(h.size() * 4), o);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    printf("before dest: %g %g\n", r[62], r[63]);
    for (int l = 0; l < 64; l += 1) if (l < 3 || l == 47 || l == 48 || l == 63) printf("lane %d -> lds[%d..]: %g %g %g %g\n", l, 64 + 4 * l, r[64 + 4 * l], r[65 + 4 * l], r[66 + 4 * l], r[67 + 4 * l]);
    printf("after dest: %g\n", r[64 + 256]);
    return 0;
}
