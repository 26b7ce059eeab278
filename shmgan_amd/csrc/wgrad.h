// Shared pieces of the weight-gradient kernels (conv_wgrad.hip, conv_wgrad_x3.hip): block order, the halo kernels' argument block, the
// transposed bf16 fragment read.  See conv_wgrad.hip for the kernels' descriptions.
#pragma once
#include "common.h"
#include "ablate.h"

// XCD-aware block order (speed only): the dispatcher deals consecutive workgroups round-robin over the 8 XCDs, each with its
// own L2, so the blocks that share an operand tile -- same pixels, different (ci, co) tile -- land on eight different L2s and
// every tile is fetched from HBM up to eight times (wgrad_halo_bf16_kernel: L2 hit rate 0.33, 3.3 TB/s HBM-side at 38 % MFMA
// utilisation).  Remapped, XCD j works through the contiguous range [j*total/8, (j+1)*total/8) of the x-fastest block order,
// i.e. through whole pixel splits: both operand tiles of a split are fetched once per XCD and reused from its L2.  Bijective
// for any grid (guide, "XCD swizzle must be bijective").  Whatever the real placement, results are unchanged.
struct Blk3 {
    int x, y, z;
};
__device__ __forceinline__ Blk3 xcd_block_order() {
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    const unsigned lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned q = total >> 3, r = total & 7u, xcd = lin & 7u, idx = lin >> 3;
    const unsigned nw = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    Blk3 b;
    b.x = (int)(nw % nx);
    b.y = (int)((nw / nx) % ny);
    b.z = (int)(nw / (nx * ny));
    return b;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WgradHaloArgs {
    const void* x;
    const void* x2;
    int c1, ldx, ldx2;
    const void* dy;
    int lddy;
    float* part;
    int h, w, cin_ld, cin, cout;
    int npatch, patches_per_split;
    unsigned xbytes, x2bytes, dybytes;
    // "norm" (shm_conv2d_wgrad_norm): source `ntpart` (0 = x, 1 = x2) is the UN-normalised activation a of an InstanceNorm block with
    // table nt = float [batch][4][ntc] (mean, inv, beta, ring).  SHM_NORM_EXACT (kernels <1>): shm_in_norm on its halo pixels in LDS, see
    // tapgemm_halo_kernel.  SHM_NORM_SCALED (kernels <2>): sum x_hat * dz = inv * sum a_ext * dz + (beta - mean * inv) * sum dz with
    // a_ext = a inside the image and `ring` outside -- the kernels write `ring` over the out-of-image halo entries of border patches and
    // scale the rows of their slab by inv (a block's patches lie in ONE sample: the launcher cuts the splits that way); the second
    // term is shm_conv2d_wgrad_norm_finish's.
    const float* nt;
    int ntpart, ntc;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* base) {
    // pixels [0,4) and [4,8) of this lane's k group: two transposed reads 4 rows (512 B) apart
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + 4 * 64));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}


// conv_wgrad_x3.hip ("wgrad.f32_split" = 1): the fp32 3x3 unit-stride weight gradient as six bf16 MFMA products of three-plane splits of x and dY.
// hgs as for wgrad_halo_kernel<0> with patches of rows x 16 pixels (rows = 2 or 4); grid = (cin / 64, cout / 64, splits).
// stride2: patches of 2 x 16 OUTPUT pixels of a 3x3 stride-2 layer on an even map (hgs as for wgrad_halo_kernel<0, true>, re-cut to that patch)
int shm_wgrad_x3_launch(const WgradHaloArgs& hgs, int cin, int cout, int nsplit, int rows, hipStream_t st, bool stride2 = false);
