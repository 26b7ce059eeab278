// Argument block and vector types of the tap-GEMM kernel family (conv_igemm.hip, conv_wreg16.hip): see conv_igemm.hip's header comment.
#pragma once
#include "common.h"
#include "ablate.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct TapPhase {
    int oph, opw, ntaps;
    int dh[9], dw[9], widx[9];
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_t;

struct TapGemmArgs {            // x, x2, w, y, y2 are float or bf16 tensors (kernel template parameter T)
    const void* x;    // A source 1 [batch, hi, wi, c1]   pitch ldx (elements)
    const void* x2;   // A source 2 [batch, hi, wi, K-c1] pitch ldx2 (or unused, c1 == K)
    int c1, ldx, ldx2;
    const void* w;      // [taps][nout][K]
    const float* bias;  // [nout] or null
    void* y;            // channels [0,n1)
    void* y2;           // channels [n1,nout)
    int n1, ldy, ldy2;
    int hi, wi, K;      // A tensor dims
    int hg, wg;         // output grid of one phase
    int ho, wo, nout;   // full output dims
    int is, os;         // A stride, output stride
    int M;              // batch*hg*wg
    unsigned xbytes, x2bytes, wbytes;   // buffer-descriptor extents (bytes)
    unsigned ybytes, y2bytes;           // output extents, 0 when an output is larger than 4 GiB (kernels with buffer stores are then not eligible)
    double* stats;      // optional [slot][batch][nout][2] (sum, sum of squares) of the stored outputs
    int hw;             // pixels per sample (stats only; hw % 64 == 0)
    int stats_slots;    // slot copies: wave tile t of a sample adds into slot t % stats_slots
    unsigned stats_stride;   // batch * nout * 2
    float slope;
    // "gsum" (input-gradient launches, shm_conv2d_dgrad_gsum / shm_conv2d_fwd_gsum): the InstanceNorm backward of the block whose
    // OUTPUT gradient this launch writes needs, per (sample, channel), sum(g) and sum(g * x_hat) over the pixels -- a full
    // read of g and of the activation if done as a pass of its own.  The epilogue has g in registers: it adds (sum v,
    // sum v * aux) of the values as stored, aux = the block's stored activation at the same pixel and channel, into
    // gred[part] = f64 [gslots][batch][channels of the part][2].  Part 0 = output channels [0, n1) (y), part 1 = [n1, nout) (y2).
    const void* gaux[2];
    int ldgaux[2];
    double* gred[2];
    int gslots, gbatch;
    // "norm" (shm_conv2d_in_fwd_norm): source `ntpart` (0 = x, 1 = x2) is the UN-normalised activation of an InstanceNorm block,
    // nt = float [batch][4][ntc] its table (mean, inv, beta, ring).  ntmode SHM_NORM_EXACT: the kernel applies shm_in_norm to that part
    // of the operand tile in LDS.  SHM_NORM_SCALED: the normalisation is in the operands -- w holds one weight copy per sample,
    // [batch][taps][nout][K] with the part's input channels scaled by inv (wimg = bytes per copy), bias one row per sample
    // [batch][nout] = bias + sum w * (beta - mean * inv) (bias_img = nout) -- and the kernel only writes `ring` over the out-of-image
    // entries of the tile (the raw value that normalises to 0: zero padding of the NORMALISED tensor)
    const float* nt;
    int ntpart, ntc, ntmode;
    unsigned ntbytes, wimg;
    int bias_img;
    TapPhase ph[4];
};

// conv_wreg16.hip: the bf16 weights-in-registers kernel of the K <= 64, unit-stride 3x3 layers (the north star's 64 -> 64 block); the caller
// (launch_tapgemm_t's SHM_TG_WREG case) has checked eligibility.  np8 = batch * (hi / 8) * (wi / 16) patches, ncu = compute units.
int shm_wreg16_launch(const TapGemmArgs& a, int np8, int ncu, hipStream_t st, const char* who);

// conv_fwd_x3.hip ("conv.f32_split"): fp32 unit-stride 3x3 layers of more than 64 output channels as six bf16 MFMA products of exact three-plane
// splits; the caller (launch_tapgemm_t) has chosen a static-tap halo variant and decided whether the gsum sums are fused
int shm_x3_fwd_eligible(const TapGemmArgs& a);
int shm_x3_fwd_launch(const TapGemmArgs& a, int batch, bool gs_fused, hipStream_t st, const char* who);

// conv_pingpong.hip: the K = 64 layers as a one-block-per-CU ping-pong kernel (two wave groups alternating between the MFMA segment and the
// load / epilogue / store segment); shm_pp_eligible checks the shape, the caller that no gsum / norm form is wanted.
int shm_pp_eligible(const TapGemmArgs& a);
int shm_pp_launch(const TapGemmArgs& a, int batch, int ncu, hipStream_t st, const char* who);
