/*
 * libshmgan_hip.so -- C ABI of the MI355X (gfx950) kernels behind SHMGAN's
 * generator + discriminator train_step.
 *
 * The reference (Atif-Anwer/SHMGAN, /root/reference/ShmGANwithSSpecSeg.py = "SHM.py")
 * has no FFI: every op below replaces a TensorFlow/Keras call made from Python.  Each
 * entry point cites the reference call site it stands in for.  See INTEGRATION.md for
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - tensors are NHWC device pointers (the caller owns every buffer, the library never
 *    allocates device memory and keeps no pointer after return);
 *  - `void*` tensors are ACTIVATION-typed: float32 when the call's `dtype` is SHM_F32, bfloat16
 *    when it is SHM_BF16 (BASELINE configs 4-5: bf16 operands on v_mfma_f32_32x32x16_bf16, fp32
 *    accumulation and fp32 arithmetic inside every kernel).  `float*` / `double*` arguments keep
 *    their type in both modes: master weights, biases, IN beta, statistics, weight gradients,
 *    image-space tensors and losses are always fp32 / f64;
 *  - "ld*" arguments are channel pitches in ELEMENTS (>= the channel count).  MFMA operands are
 *    staged as 64-byte rows, so contraction channel counts and concat splits are multiples of
 *    16 (fp32) or 32 (bf16), pitches multiples of 4 (fp32) or 8 (bf16); the 3- and 10-channel
 *    images are stored in a 16-float / 32-bf16 pitch (zero padded);
 *  - "f64 scratch" arguments are small double accumulators; where a comment says "zero on entry" the caller
 *    zeroes them ONCE (at allocation) and every call leaves them zero again -- also when it returns an error
 *    (the entry point clears the scratch itself if a launch after the one that filled it fails); otherwise
 *    the call zeroes them;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *  - return value: SHM_OK or a negative error; shm_last_error() gives the text.
 */
#ifndef SHMGAN_HIP_H
#define SHMGAN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHM_OK 0
#define SHM_E_SHAPE (-1)
#define SHM_E_DTYPE (-2)
#define SHM_E_WORKSPACE (-3)
#define SHM_E_HIP (-4)

#define SHM_F32 0
#define SHM_BF16 1
/* bf16 activations and MFMA operands, but fp32 for the tensors marked [G] below: the gradient
 * signal on its way from an input-gradient product into the next InstanceNorm/LeakyReLU backward.
 * An option for callers that want that signal unrounded; measured on the whole step it changes no
 * per-tensor gradient cosine beyond the 4th digit and costs 1.7 % (DESIGN.md, bf16 path), so the
 * host side defaults to plain SHM_BF16.  Accepted by the functions that have a [G] argument. */
#define SHM_BF16_GF32 2

int shm_version(void);
const char* shm_last_error(void);
/* Symbol (as rocprofv3 prints it, without the "void " prefix and the argument list) of the MFMA kernel
 * the calling thread's last convolution entry point dispatched to: the tile/variant choice depends on
 * shape and dtype, and bench.py's per-kernel roofline keys its HIP-event timings by it. */
const char* shm_last_kernel(void);

/* ---- dispatch tuning ---------------------------------------------------------------
 * The convolution entry points choose among several MFMA kernel variants by shape and dtype.  These
 * process-wide integer knobs override that choice (parity tests force every variant; tools sweep them):
 *   "tapgemm.variant"           0 automatic (default), or one of SHM_TG_*: a forced variant the shape is not
 *                               eligible for makes the conv call return SHM_E_SHAPE (it never falls back silently)
 *   "tapgemm.halo_min_blocks"   fp32: from this many 128-channel halo blocks on the 128-wide halo block is taken without comparing
 *                               the fill of its last round with the 64-wide block's (default 1024)
 *   "tapgemm.small_grid_blocks" grids with fewer 128x128 tiles take the 64x128 tile (default 1024)
 *   "tapgemm.phase4_min_blocks" stride-2 transposed 3x3 products with at least this many fused (16x16 input pixels x 64 channels) blocks
 *                               take the four-phases-in-one-block kernel (default 256)
 *   "tapgemm.wreg16"            bf16 weights-in-registers layers (one source, 32 / 64 input channels): 2 (default) = the one-block-per-CU ping-pong
 *                               kernel (tapgemm_pp_bf16_kernel, v_mfma_f32_32x32x16_bf16) where the shape allows (64 input channels, map of whole
 *                               8 x 32-pixel patches) and the eight-wave form elsewhere, 1 = the eight-wave form with 16-column wave tiles
 *                               (tapgemm_wreg16_bf16_kernel, v_mfma_f32_16x16x32_bf16, four waves per SIMD), 0 = four-wave form with 32-column tiles
 *   "tapgemm.flat_epilogue"     1 = treat every output as larger than 4 GiB: element stores through 64-bit addresses, no buffer-store kernels (tests), default 0
 *   "wgrad.variant"             0 automatic, 1 generic kernels only, 2 halo kernels without thin-input packing, 3 no stride-2 halo form
 *   "wgrad.blocks"              split-K block target, 0 automatic (1024 fp32 / 256 bf16)
 *   "wgrad.bf16_rows"           bf16 halo weight gradient: pixel rows per LDS stage, 0 automatic (4 when the map height allows), 2, 4
 *   "wgrad.f32_split"           fp32 3x3 unit-stride weight gradient (the halo kernel's shapes, plain and SHM_NORM_EXACT sources): 1 = six v_mfma_f32_32x32x16_bf16
 *                               products of the exact three-plane bf16 splits of x and dY with fp32 accumulation (wgrad_halo_x3_kernel; rel-L2 ~2.5e-7
 *                               on random-sign operands; NOT a bit-faithful fp32 dot product -- see "conv.f32_split" below), 0 (default) = exact-fp32 MFMA.
 *                               Opt-in: bench.py --dtype f32x3
 *   "conv.f32_split"            fp32 3x3 unit-stride forward / input gradient: 1 = six bf16 MFMA products of exact three-plane splits where
 *                               the shape fits (tapgemm_halo_x3_kernel), 0 (default) = exact-fp32 MFMA.  Opt-in
 *   "elem.fused_bwd"            bf16 shm_in_bwd: 1 (default) = the one-pass form where shm_in_bwd_fused_scratch was given and the shape fits, 0 = two passes
 *   "wgrad.bf16_wide"           bf16 weight gradient, the eight-wave 64 ci x 128 co block (cout >= 128): 0 automatic (= 2), 1 never, 2 at stride 2 only,
 *                               3 at unit stride only, 4 both
 *   "stats.fusion"              1 InstanceNorm statistics in the conv epilogue (default), 0 separate pass
 *   "elem.fused_max_slices"     the one-pass form: most slices (= blocks that must be resident together) per barrier group, default 256, at most 512; the launcher also
 *                               requires twice the group's blocks to fit the device (shm_set_abort_words below)
 *   "elem.fused_hold"           the one-pass form's kernel: 0 automatic (in_bwd_fusedg_kernel -- the gradient held in registers, the activation streamed
 *                               twice, slices of 32768 / min(c, 64) pixels, up to 2 x "elem.fused_max_slices" blocks per group -- on maps of at least
 *                               256 x 16384 / min(c, 64) pixels without a pooled gradient, where it is measured faster; in_bwd_fused8_kernel otherwise),
 *                               1 in_bwd_fused8_kernel only, 2 in_bwd_fusedg_kernel wherever the map has whole slices
 *   "elem.fused_gvariant"       in_bwd_fusedg_kernel's register-budget form: 0 <2, 2, 4> (four blocks per CU; default), 1 <8, 8, 3>
 *   "elem.fused_test_stall"     tests only: 1 = the one-pass form's barriers wait for one block more than the grid has, i.e. every barrier
 *                               times out (~1 s per resident generation of blocks) and the abort words are set; default 0
 *   "elem.reverse"              1 InstanceNorm apply / backward-reduce passes walk the tensor back to front (default: the tail the
 *                               producer just wrote is still in the Infinity Cache), 0 front to back
 *   "elem.reduce_blocks"        block target of the InstanceNorm-backward reduce pass, 0 automatic (1024 fp32 / 512 bf16: every block ends
 *                               in f64 atomics on its sample's 2c sums)
 *   "elem.nt_loads"             1 the InstanceNorm-backward apply pass reads its gradient tensor (dead after that read) with non-temporal
 *                               loads, 0 plain loads (default; round-3 A/B in DESIGN.md section 8)
 *   "elem.chunk_mb"             shm_in_bwd runs its reduce and apply passes per chunk of samples whose tensors fit this many MiB, so that
 *                               the apply pass re-reads them from the Infinity Cache; 0 = the whole batch at once (default)
 *   "elem.interleave"           bf16 activations: 1 (default) the InstanceNorm-backward passes deal a sample's pixel tiles round-robin over its
 *                               blocks, 0 one contiguous chunk per block (bf16 step 26.2 -> 26.0 ms; float32 always takes the chunk form)
 *   "elem.stream_blocks"        block target of the passes without a per-block prologue or reduction (InstanceNorm apply and its pooling
 *                               forms), default 32768: short blocks keep the addresses in flight a narrow band of the tensors
 *   "elem.apply_blocks"         block target of the InstanceNorm-backward apply pass, default 4096 (every block ends in an LDS reduction and
 *                               one f64 atomic per channel for the bias gradient: 5.2 TB/s at 4-8k blocks, 4.3 at 32k; without a bias
 *                               gradient the pass keeps gaining up to 32k: tools/probes/bwd_blocks.py)
 * value < 0 restores the knob's default; key "reset" restores all.  Initial values may be given in the
 * environment (SHM_TAPGEMM_VARIANT, SHM_TAPGEMM_HALO_MIN, SHM_TAPGEMM_SMALLM, SHM_TAPGEMM_PHASE4_MIN, SHM_WGRAD_VARIANT, SHM_WGRAD_BF16_ROWS,
 * SHM_WGRAD_BLOCKS, SHM_STATS_FUSION, SHM_ELEM_REVERSE, SHM_ELEM_REDUCE_BLOCKS, SHM_ELEM_NT, SHM_ELEM_CHUNK_MB, SHM_ELEM_INTERLEAVE, SHM_ELEM_STREAM_BLOCKS, SHM_ELEM_APPLY_BLOCKS), read once.  Knobs change scheduling only, never results beyond the
 * summation order of a tile shape. */
#define SHM_TG_AUTO 0
#define SHM_TG_HALO128 1
#define SHM_TG_HALO64 2
#define SHM_TG_DMA_128x128 3
#define SHM_TG_DMA_64x128 4
#define SHM_TG_DMA_128x64 5
#define SHM_TG_DMA_256x64 6
#define SHM_TG_DMA_256x128 7
#define SHM_TG_HALO128_PH8 8
#define SHM_TG_DMA_128x128_BK32 9
#define SHM_TG_DMA_128x128_NST4 10
#define SHM_TG_HALO128_ST 12            /* halo kernels with the nine taps unrolled: static fragment addresses, conflict-free swizzle */
#define SHM_TG_HALO64_ST 13
#define SHM_TG_PHASE4 14               /* 3x3 stride-2 transposed products (Conv2DTranspose forward, stride-2 input gradient): four phases fused in one block */
#define SHM_TG_DMA_64x64 15             /* DMA tap GEMM, 64 x 64 tile (four waves of 32 x 32): grids too small to fill the chip with larger tiles */
#define SHM_TG_HALO128_ST_W4 16         /* static-tap halo block of four waves, wave tile 128 pixels x 64 channels */
#define SHM_TG_WREG 11                 /* bf16, 3x3 s1, <= 64 input channels: weights in registers, persistent blocks */
int shm_set_tuning(const char* key, int value);
int shm_get_tuning(const char* key, int* value);

/* ---- weight layout ---------------------------------------------------------------
 * [ntaps][rows][cols] -> [ntaps][cols][rows_pad] (zero padded): HWIO -> K-contiguous
 * [tap][Cout][Cin] for the implicit-GEMM B operand. */
int shm_transpose_taps(const float* w, void* wt, int ntaps, int rows, int cols, int rows_pad,
                       int dtype, void* stream);
/* `count` (<= 48) such transposes in one launch: the host arrays hold one entry per layer (weights of a whole model after an
 * optimizer step). */
int shm_transpose_taps_multi(int count, const void* const* w, void* const* wt, const int* ntaps, const int* rows,
                             const int* cols, const int* rows_pad, int dtype, void* stream);
/* dst[i] = (dtype) src[i]: operand copies of weights that are already K-contiguous as stored. */
int shm_cast_f32(const float* src, void* dst, size_t n, int dtype, void* stream);

/* ---- convolutions (implicit GEMM on v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16) ------
 * Keras Conv2D(k in {1,3}, strides in {1,2}, padding='same') + bias + LeakyReLU(slope)
 * (SHM.py:244-245, 254-326, 365-369, 387).  y is [G]-typed under SHM_BF16_GF32 (the call is then the
 * input-gradient of a Conv2DTranspose).  Input = channel concat of x (c1 channels,
 * pitch ldx) and optional x2 (cin-c1 channels, pitch ldx2): Concatenate() SHM.py:299,306,
 * 313,320 is never materialised.  wk = [k*k][cout][cin] (shm_transpose_taps of HWIO, in the call's dtype),
 * cin and c1 multiples of 16 (fp32) / 32 (bf16).  bias may be NULL; slope 1.0f = no activation, 0.0f = ReLU.
 * Compact 3-channel images (the discriminator's input, SHM.py:353): x may hold ONE 16-byte chunk per pixel -- ldx = 4 (fp32: r, g, b, 0) or 8
 * (bf16: r, g, b, five zeros) -- under a weight copy of cin = 16 / 32 whose channels >= 3 are zero; k = 3, stride 2 and an output width that is a
 * multiple of 16 then run the streaming kernels of conv_rgb.hip (forward and shm_conv2d_wgrad alike), other shapes the generic ones. */
int shm_conv2d_fwd(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* wk,
                   const float* bias, void* y, int ldy, int batch, int hi, int wi, int cin,
                   int cout, int ksize, int stride, float slope, int dtype, void* stream);

/* The same convolution fused with the InstanceNormalization statistics of its output
 * (Conv2D -> LeakyReLU -> InstanceNormalization, SHM.py:244-245): the epilogue accumulates
 * sum / sum of squares per (sample, channel); on return (stream order) stats holds
 * (mean, rsqrt(var + eps)) exactly as shm_in_stats would leave it.  stats = f64 [batch*cout*2];
 * scratch = f64 [SHM_STATS_SLOTS*batch*cout*2] or NULL: slot copies of the running sums, so that the
 * hw/64 atomic adds per (sample, channel) do not queue on one address.  The scratch must be ZERO on entry
 * (zero it once when allocating) and is zero again on return: no memset per call. */
#define SHM_STATS_SLOTS 16
int shm_conv2d_in_fwd(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* wk,
                      const float* bias, void* y, int ldy, int batch, int hi, int wi, int cin,
                      int cout, int ksize, int stride, float slope, double* stats, double* scratch,
                      float eps, int dtype, void* stream);

/* Input-gradient of the same conv.  dy [batch,ho,wo,cout] (pitch lddy), w = HWIO
 * [k*k][cin][cout] as stored (it is already K-contiguous for this product; in bf16 mode the shm_cast_f32 copy),
 * cout a multiple of 16 (fp32) / 32 (bf16).
 * dx channels [0,n1) go to dx (pitch lddx), [n1,cin) to dx2 (pitch lddx2): the split of a
 * concat gradient.  Pass dx2=NULL, n1=cin for a single destination.  dx, dx2 are [G] tensors. */
int shm_conv2d_dgrad(const void* dy, int lddy, const void* w, void* dx, void* dx2, int n1,
                     int lddx, int lddx2, int batch, int hi, int wi, int cin, int cout,
                     int ksize, int stride, int dtype, void* stream);

/* Keras Conv2DTranspose(k=3, strides=2, 'same') + bias + LeakyReLU (SHM.py:298,305,312,319).
 * x [batch,hi,wi,cin]; w = Keras layout [3][3][cout][cin] as stored; y [batch,2hi,2wi,cout]. */
int shm_conv2d_transpose_fwd(const void* x, int ldx, const void* w, const float* bias, void* y,
                             int ldy, int batch, int hi, int wi, int cin, int cout, float slope,
                             int dtype, void* stream);

/* Weight gradient: dw[t][ci][co] (+)= sum_pixels x[pix*stride + tap][ci] * dy[pix][co]
 * (HWIO).  For Conv2DTranspose pass x = dz of the transposed conv (2H res), dy = its input
 * (H res), stride 2: the result is the Keras [3][3][cout][cin] layout.  cin_ld = number of
 * input channels to read (multiple of 4 fp32 / 8 bf16, pad channels must be zero), cin = rows stored; dw is
 * always fp32.  workspace: split-K partial slabs (fp32), at least shm_conv2d_wgrad_workspace() bytes; the
 * slabs are summed in a fixed order (deterministic, no float atomics). */
size_t shm_conv2d_wgrad_workspace(int batch, int ho, int wo, int cin, int cout, int ksize);
/* The two phases of shm_conv2d_wgrad as separate calls (bench.py times the MFMA kernel on its own):
 * _partial writes *nsplit_out slabs [ksize*ksize][cin][cout] to workspace; _reduce sums them into dw. */
int shm_conv2d_wgrad_partial(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* dy,
                             int lddy, int batch, int hi, int wi, int cin, int cin_ld, int cout,
                             int ksize, int stride, void* workspace, size_t ws_bytes, int dtype,
                             int* nsplit_out, void* stream);
int shm_conv2d_wgrad_reduce(const void* workspace, float* dw, size_t n, int nsplit, int accumulate,
                            void* stream);
int shm_conv2d_wgrad(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* dy,
                     int lddy, float* dw, int batch, int hi, int wi, int cin, int cin_ld,
                     int cout, int ksize, int stride, int accumulate, void* workspace,
                     size_t ws_bytes, int dtype, void* stream);

/* ---- The fused block: InstanceNormalization applied by the CONSUMER (SHM.py:244-245 "Conv -> LeakyReLU -> IN", north-star
 * "IN apply folded into the consumer's load") ------------------------------------------------------------------------
 * A Conv -> LeakyReLU -> InstanceNorm block stores its un-normalised activation a and its statistics; instead of a pass that reads a
 * and writes the normalised tensor (shm_in_apply: 2x the activation's bytes), the convolution and the weight gradient that consume
 * the block's output read a itself and apply (a - mean) * inv + beta to their operand tile in LDS -- the kernels that stage the A
 * operand as a halo image (unit-stride 3x3 layers on maps that are multiples of 16; *_norm_supported says whether a shape runs on
 * one).  Out-of-image taps stay zero (zero padding of the NORMALISED tensor) and the arithmetic is shm_in_apply's, so results
 * are bit-identical to shm_in_apply followed by the plain entry point.
 *   nt = float [batch][4][c]: per sample the planes mean[c], inv[c], beta[c], ring[c] = mean - beta / inv (the raw value whose
 *   normalised image is 0) of the producing block (c = its channel count = the channel count of the source it describes), written
 *   by shm_conv2d_in_fwd_norm(nt_out, beta_out) or shm_in_norm_table.
 *   norm_mode SHM_NORM_EXACT / SHM_NORM_SCALED (below); with SHM_NORM_SCALED `wk` and `bias` are shm_conv2d_norm_prepare's wk_n, bias_n.
 *   nt_x / nt_x2: table of source x / x2, or NULL = that source is used as stored; at most one of the two. */
#define SHM_NORM_EXACT 0       /* the kernel applies (a - mean) * inv + beta to its operand tile in LDS: bit-identical to shm_in_apply + plain call */
#define SHM_NORM_SCALED 1      /* the normalisation is in the operands: per-sample weights w * inv and bias rows (shm_conv2d_norm_prepare);
                                  the kernels only write `ring` over out-of-image taps.  Same result to rounding, no per-element work in
                                  the MFMA kernels */
int shm_in_norm_table(const double* stats, const float* beta, float* nt, int batch, int c, void* stream);
/* SHM_NORM_SCALED operands: wk_n [batch][taps][cout][cin] (activation dtype) = wk with the channels [part_lo, part_lo + c) of the
 * folded source scaled by that sample's inv, bias_n [batch][cout] = bias + sum wk * (beta - mean * inv) over those channels.  nt is the
 * folded source's table, c its channel count. */
int shm_conv2d_norm_prepare(const void* wk, const float* bias, const float* nt, int c, int part_lo, void* wk_n, float* bias_n,
                            int batch, int cin, int cout, int ksize, int dtype, void* stream);
/* shm_conv2d_in_fwd with (a) sources normalised on the fly and (b) optionally this block's own table as a by-product of the
 * statistics finalisation (nt_out [batch][4][cout] with beta_out [cout]; NULL = not wanted).  With nt_x == nt_x2 == NULL it is
 * shm_conv2d_in_fwd.  SHM_E_SHAPE if the kernel chosen for the shape cannot normalise in LDS (never a silent fallback). */
int shm_conv2d_in_fwd_norm(const void* x, const void* x2, int c1, int ldx, int ldx2, const float* nt_x, const float* nt_x2,
                           int norm_mode, const void* wk, const float* bias, void* y, int ldy, int batch, int hi, int wi, int cin,
                           int cout, int ksize, int stride, float slope, double* stats, double* scratch, float eps, float* nt_out,
                           const float* beta_out, int dtype, void* stream);
/* 1 if shm_conv2d_in_fwd_norm would take a normalised source `norm_part` (0 = x, 1 = x2; c1 = channels of x when there are two
 * sources, else 0) for this shape, batch and the current tuning knobs; 0 otherwise.  Launches nothing. */
int shm_conv2d_norm_supported(int batch, int hi, int wi, int cin, int c1, int cout, int ksize, int stride, int norm_part, int dtype);
/* shm_conv2d_wgrad on sources normalised on the fly, and its query.  SHM_NORM_SCALED: the kernels compute
 * inv * sum a_ext * dz (a_ext = a inside the image, `ring` outside) per sample -- the workspace must then hold
 * shm_conv2d_wgrad_norm_workspace() bytes (splits on sample boundaries) -- and the caller completes the gradient with
 * shm_conv2d_wgrad_norm_finish: dw[tap][part_lo + k][co] += sum_n (beta[k] - mean_n[k] * inv_n[k]) * dzsum[n][co], dzsum = float64
 * [batch][cout] per-sample channel sums of dz = what shm_in_bwd_keep_dz_sums(dst) makes the NEXT shm_in_bwd / shm_in_bwd_apply /
 * shm_in_bwd_rank1 call of the thread copy out of its bias-gradient staging (one-shot; that call must take a bias gradient). */
int shm_conv2d_wgrad_norm(const void* x, const void* x2, int c1, int ldx, int ldx2, const float* nt_x, const float* nt_x2,
                          int norm_mode, const void* dy, int lddy, float* dw, int batch, int hi, int wi, int cin, int cin_ld, int cout,
                          int ksize, int stride, int accumulate, void* workspace, size_t ws_bytes, int dtype, void* stream);
int shm_conv2d_wgrad_partial_norm(const void* x, const void* x2, int c1, int ldx, int ldx2, const float* nt_x, const float* nt_x2,
                                  int norm_mode, const void* dy, int lddy, int batch, int hi, int wi, int cin, int cin_ld, int cout,
                                  int ksize, int stride, void* workspace, size_t ws_bytes, int dtype, int* nsplit_out, void* stream);
int shm_conv2d_wgrad_norm_supported(int batch, int hi, int wi, int cin, int cin_ld, int c1, int cout, int ksize, int stride,
                                    int norm_part, int dtype);
size_t shm_conv2d_wgrad_norm_workspace(int batch, int hi, int wi, int cin, int cout, int ksize, int dtype);
int shm_conv2d_wgrad_norm_finish(float* dw, const float* nt, const double* dzsum, int batch, int c, int part_lo, int cin, int cout,
                                 int ksize, void* stream);
int shm_in_bwd_keep_dz_sums(double* dst);
/* One-pass bf16 form of shm_in_bwd (round 5, in_bwd_fused8_kernel: a block keeps its slice of g1 / g2 / a in registers between the reduce and
 * the apply phase, the blocks of a sample meet at a per-sample barrier; 3 tensor passes over HBM instead of 5).  The NEXT shm_in_bwd call of this
 * thread may use `scratch` = f64 [n_doubles], n_doubles >= SHM_IN_BWD_FUSED_DOUBLES(batch, h * w, c), zero on entry and zero again on return
 * (outside the per-block partial rows at its front, which every launch rewrites in full and which may hold anything).
 * One-shot (NULL disarms).  Taken for dtype SHM_BF16, c in {8, 16, 32} or a multiple of 64 up to 1024, h * w a multiple of the
 * 16384 / min(c, 64) pixel slice and at most 256 slices per map ("elem.fused_max_slices"; with a pooled gradient g2: c a multiple of 64 and
 * whole tiles of (256 / Wt) rows x Wt = min(w, 128) columns), tuning
 * "elem.fused_bwd" = 1 (default); every other call runs the two passes.  No float atomics: the sums are added in block order (bitwise
 * reproducible). */
#define SHM_IN_BWD_FUSED_CB(c) ((c) < 64 ? (c) : 64)
#define SHM_IN_BWD_FUSED_DOUBLES(batch, hw, c)                                                                                      \
    (((size_t)(batch) * ((size_t)(hw) * SHM_IN_BWD_FUSED_CB(c) / 16384) * 3 * (size_t)(c) + 1) / 2 + (size_t)(batch) * (size_t)(c) + \
     (size_t)(batch) * ((size_t)(c) / SHM_IN_BWD_FUSED_CB(c)) * 288 + 1)
int shm_in_bwd_fused_scratch(double* scratch, size_t n_doubles);
/* The one-pass form's barrier needs every block of a group resident at once.  The launcher takes it only when TWICE the group's blocks fit the
 * current device (its CU count x hipOccupancyMaxActiveBlocksPerMultiprocessor of the kernel, queried once: a partitioned or smaller part falls
 * back to the two passes), and a barrier that still waits ~1 s gives up instead of hanging: the launch then completes with wrong means, sets the
 * last u32 of `scratch` and the caller's ABORT WORDS:
 *   dev_word   u32 in device memory, OR-ed to non-zero.  shm_adam_clip reads it ON THE DEVICE and applies nothing while it is set: gradients
 *              built on unfinished sums never reach the weights, however far the host has run ahead of the stream;
 *   host_word  u32 in mapped (pinned) host memory, set to 1: the host sees it without synchronising.
 * Both stay set until the caller clears them; NULLs disarm.  Persistent per calling thread (configuration, like the tuning table). */
int shm_set_abort_words(unsigned* dev_word, unsigned* host_word);
/* Measurement hook (bench.py, north_star_block.ceiling): while dev2 (two u64 in device memory) is set on the calling thread, one wave of the
 * middle block of every tapgemm_pp_bf16_kernel launch writes dev2[0] = shader-clock ticks (s_memtime) and dev2[1] = 100 MHz ticks (s_memrealtime)
 * of its patch loop: the clock the kernel actually held = dev2[0] / dev2[1] x 0.1 GHz.  NULL disarms (default). */
int shm_set_clock_probe(unsigned long long* dev2);

/* Opt-in fp32 arithmetic from bf16 MFMAs for the 3x3 unit-stride forward / input-gradient layers: tuning "conv.f32_split" = 1 (round 5,
 * csrc/conv_fwd_x3.hip; the weight gradient's twin is "wgrad.f32_split").  Every fp32 operand is split EXACTLY into three bf16 planes and the six
 * plane products with i + j <= 2 accumulate in fp32: rel-L2 ~4e-7 against float64 on random-sign operands, the exact-fp32 MFMA's own figure.
 * Accuracy limits (round 6, tests/test_x3_gpu.py): the bf16 MFMA truncates what falls below its alignment window when it adds its products to the
 * accumulator, ~0.3-0.5 ulp per accumulation step and always towards zero -- on operands of ONE sign the error is a one-sided shrink that grows with
 * the accumulation chain: -5.3e-6 relative on the step's longest weight-gradient reduction (K = 2.6 M, ~320 steps per split-K slab; exact-fp32 MFMA:
 * 2.4e-7) and -1.1e-5 on the deepest forward product (K = 4608), i.e. outside the 1e-5 single-op bound there; operands below 2^-110 lose the third
 * plane (7e-5 at 2^-120).  The step's own operands (random-sign weights and gradients, O(1) activations) stay at the exact kernels' figures.  shm_conv2d_fwd / shm_conv2d_in_fwd(_norm, SHM_NORM_EXACT) / shm_conv2d_dgrad and their _gsum forms then run
 * tapgemm_halo_x3_kernel where the launcher would have taken a static-tap halo or weights-in-registers kernel: fp32 tensors, K % 32 == 0, outputs
 * below 4 GiB, more than 64 output channels or a map height that is a multiple of 32.  Nothing else changes: no extra arguments, no workspace. */
/* pooled = AveragePooling2D(2)(InstanceNorm apply(a)) WITHOUT writing the normalised tensor: the encoder level's skip consumers
 * normalise a on the fly, only the pool's consumer needs a tensor.  Same bits as shm_in_apply_pool's `pooled`. */
int shm_in_pool(const void* a, int lda, const double* stats, const float* beta, void* pooled, int ldp, int batch, int h, int w,
                int c, int dtype, void* stream);

/* ---- InstanceNormalization (tfa, axis=-1, eps, gamma==1, constant beta) -----------
 * SHM.py:245...:388; op chain Generator_summary.txt:9-36.
 * stats = f64 [batch*c*2]; on return (stream order) stats[(n*c+ch)*2] = mean over H*W,
 * stats[(n*c+ch)*2+1] = rsqrt(biased variance + eps). */
int shm_in_stats(const void* a, int lda, double* stats, int batch, int hw, int c, float eps,
                 int dtype, void* stream);
/* out = (a - mean) * inv + beta[c]  (out may alias a). */
int shm_in_apply(const void* a, int lda, const double* stats, const float* beta, void* out,
                 int ldo, int batch, int hw, int c, int dtype, void* stream);
/* out = InstanceNorm apply as above AND pooled = AveragePooling2D(2)(out) in the same pass (the second block of an
 * encoder level feeds both the skip connection and the pool, SHM.py:246-247): bit-identical to shm_in_apply followed by
 * shm_avgpool2_fwd, without the pooling pass's read of the normalised tensor.  h, w even; pooled [batch, h/2, w/2, c], pitch ldp. */
int shm_in_apply_pool(const void* a, int lda, const double* stats, const float* beta, void* out, int ldo,
                      void* pooled, int ldp, int batch, int h, int w, int c, int dtype, void* stream);
/* Backward of LeakyReLU -> IN given the gradient at the IN output:
 *   d_out = g1 + 0.25 * g2[h/2][w/2]   (g2 = gradient of AveragePooling2D(2,2), may be NULL)
 *   dz = lrelu'(a) * inv * (d_out - mean(d_out) - xhat * mean(d_out * xhat))
 * red = f64 scratch [batch*c*3], ZERO on entry and zero again on return; dbias = f64 accumulator [c]
 * (NOT zeroed, may be NULL).
 * g1, g2 are [G] tensors; a and dz are activation-typed. */
int shm_in_bwd(const void* g1, int ldg1, const void* g2, int ldg2, const void* a, int lda,
               const double* stats, double* red, void* dz, int lddz, double* dbias, int batch,
               int h, int w, int c, float slope, int dtype, void* stream);

/* ---- the fused block's backward: InstanceNorm sums in the producing epilogue ("gsum") -------------------------------------
 * The IN backward needs, per (sample, channel), sum(d_out) and sum(d_out * xhat) before it can write dz: shm_in_bwd collects
 * them in a reduce pass of its own over d_out and the stored activation a.  But d_out is itself the OUTPUT of an input-gradient
 * product -- the dgrad of the next layer (SHM.py:244-245 block order), or the stride-2 product that is Conv2DTranspose's input
 * gradient -- so that product's epilogue, which holds d_out in registers, adds the sums itself:
 *
 *   shm_conv2d_dgrad_gsum = shm_conv2d_dgrad + for each output part (dx: channels [0,n1), dx2: the rest) an optional pair
 *       (aux, red): red[(slot, n, ch)][0] += sum_pixels g, red[...][1] += sum_pixels g * aux, g = the value as stored.
 *       aux = activation-typed tensor [batch, hi, wi] (pitch ldaux) aligned with the part: the consumer block's stored
 *       activation a (so that sum g*xhat = inv * (sum g*a - mean * sum g)), or -- when dx is the gradient of an
 *       AveragePooling2D output -- the pooled NORMALISED tensor, i.e. this layer's own input (sum g*avgpool(xhat) =
 *       sum g*pooled - beta * sum g).
 *   shm_conv2d_fwd_gsum = shm_conv2d_fwd + the same for its single output (the stride-2 forward form is the input gradient
 *       of Conv2DTranspose).
 *   red = f64 [SHM_GSUM_SLOTS][batch][channels of the part][2], ZERO on entry; the slot copies cut the per-address atomic
 *       chains.  Whether the sums were taken in the epilogue or by a follow-up reduce pass (kernels without a gsum epilogue:
 *       the fused four-phase kernel, odd alignments, tiny maps) is the library's business: red is complete on return.
 *   shm_in_bwd_apply = shm_in_bwd without its reduce pass: d_out = g1 + 0.25 * unpool(g2); red = sums of g1 against a,
 *       redp = sums of g2 against the pooled normalised tensor (NULL iff g2 is NULL; needs beta [c]); dstage = f64 [batch*c]
 *       staging of the bias gradient (required with dbias).  red / redp / dstage are zero again on return. */
#define SHM_GSUM_SLOTS 8
int shm_conv2d_dgrad_gsum(const void* dy, int lddy, const void* w, void* dx, void* dx2, int n1, int lddx, int lddx2, int batch,
                          int hi, int wi, int cin, int cout, int ksize, int stride, const void* aux, int ldaux, double* red,
                          const void* aux2, int ldaux2, double* red2, int dtype, void* stream);
int shm_conv2d_fwd_gsum(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* wk, const float* bias, void* y,
                        int ldy, int batch, int hi, int wi, int cin, int cout, int ksize, int stride, float slope,
                        const void* aux, int ldaux, double* red, int dtype, void* stream);
int shm_in_bwd_apply(const void* g1, int ldg1, const void* g2, int ldg2, const void* a, int lda, const double* stats,
                     const float* beta, double* red, double* redp, double* dstage, void* dz, int lddz, double* dbias,
                     int batch, int h, int w, int c, float slope, int dtype, void* stream);

/* Input gradient of a FIRST layer when only its sum over a set of input channels is needed (the step never
 * uses more: d genY sums the cyclic inputs' view channels, SHM.py:576-580; yuv_to_rgb's backward sums r,g,b).
 * Conv is linear, so the weights are summed first and the 64 -> 10 / 3 channel dgrad becomes a 64 -> 1 stencil:
 *   shm_sum_input_channels: weff[t][co] = sum_{j : mask bit j} w[t][j][co]   (w = HWIO [9][cin][cout], cin <= 32)
 *   shm_conv3x3_dgrad_sum1: out[b,y,x] (+)= sum_k sum_taps sum_co dz[k*batch+b, oy, ox, co] * weff[k][tap][co]
 * dz [nk*batch, ho, wo, c] activation-typed (the layer's pre-activation gradient), weff f32 [nk][9][c],
 * out f32 [batch, hi, wi]; stride 1 or 2 with TF SAME padding. */
int shm_sum_input_channels(const float* w, int cin, int cout, unsigned mask, float* weff, void* stream);
int shm_conv3x3_dgrad_sum1(const void* dz, int lddz, const float* weff, float* out, int nk, int batch,
                           int hi, int wi, int c, int stride, int accumulate, int dtype, void* stream);

/* LeakyReLU backward for blocks without IN (Conv2DTranspose, SHM.py:298): dz = dy*lrelu'(y); dy is [G].
 * dbias = f64 accumulator [c] (NOT zeroed, may be NULL); red = f64 scratch [SHM_LRELU_RED_SLOTS*c], zero on
 * entry and on return (needed when dbias is given: the per-channel sums are staged over slots, not on c
 * addresses). */
#define SHM_LRELU_RED_SLOTS 64
int shm_lrelu_bwd(const void* dy, int lddy, const void* y, int ldy, void* dz, int lddz,
                  double* dbias, double* red, size_t npix, int c, float slope, int dtype, void* stream);

/* AveragePooling2D(2,2,'same') on even sizes (SHM.py:249,258,267,276). */
int shm_avgpool2_fwd(const void* x, int ldx, void* y, int ldy, int batch, int h, int w, int c,
                     int dtype, void* stream);

/* f64 accumulator -> f32 (dst = or += src). */
int shm_cvt_f64_f32(const double* src, float* dst, size_t n, int accumulate, void* stream);
int shm_zero(void* p, size_t bytes, void* stream);

/* ---- 1-output-channel layers ------------------------------------------------------
 * Generator head Conv2D(1, k=1) + LeakyReLU (SHM.py:326). */
int shm_head_fwd(const void* x, int ldx, const float* w, const float* bias, float* y, size_t npix,
                 int c, float slope, int dtype, void* stream);
/* dz = dy*lrelu'(y); dx [G] = dz (x) w; dw_acc[c] += sum x*dz; db_acc[0] += sum dz (f64, not zeroed);
 * red = f64 scratch [SHM_LRELU_RED_SLOTS*(c+1)] (slot staging of the two sums). */
int shm_head_bwd(const void* x, int ldx, const float* w, const float* y, const float* dy, void* dx,
                 int lddx, double* dw_acc, double* db_acc, double* red, size_t npix, int c, float slope,
                 int dtype, void* stream);
/* The same two on the UN-normalised activation a [batch, hw, c] of the block in front of the head + its InstanceNorm statistics
 * (shm_conv2d_in_fwd's `stats`) and beta: the head applies (a - mean) * inv + beta on the fly, so that block needs no
 * shm_in_apply and its normalised tensor is never written.  dx [G] (may be NULL) is the gradient at the NORMALISED activation
 * (shm_in_bwd's g1); dz_out [batch*hw] fp32 (may be NULL) = dy * lrelu'(y), the scalar factor of that gradient (dx = dz_out (x) w). */
int shm_head_in_fwd(const void* a, int lda, const double* stats, const float* beta, const float* w, const float* bias,
                    float* y, int batch, int hw, int c, float slope, int dtype, void* stream);
int shm_head_in_bwd(const void* a, int lda, const double* stats, const float* beta, const float* w, const float* y,
                    const float* dy, void* dx, int lddx, float* dz_out, double* dw_acc, double* db_acc, double* red,
                    int batch, int hw, int c, float slope, int dtype, void* stream);
/* shm_in_bwd for the block in front of the head: its output gradient is the rank-1 tensor hdz[n*h*w + p] * hw_[ch] (hdz =
 * shm_head_in_bwd's dz_out [batch*h*w] with dx = NULL, hw_ = the head kernel [c]), formed on the fly: the head writes no input
 * gradient and neither pass of the backward reads one.  Otherwise as shm_in_bwd (red, dz, dbias, slope). */
int shm_in_bwd_rank1(const float* hdz, const float* hw_, const void* a, int lda, const double* stats, double* red, void* dz,
                     int lddz, double* dbias, int batch, int h, int w, int c, float slope, int dtype, void* stream);
/* PatchGAN logits Conv2D(1, k=3, no bias) + LeakyReLU (SHM.py:365-369). x [batch,h,w,c]. */
int shm_patch_fwd(const void* x, int ldx, const float* w, float* y, int batch, int h, int wd, int c,
                  float slope, int dtype, void* stream);
/* dz = dy*lrelu'(y) (written to dz [batch,h,w]); dx [G] = transposed conv of dz (overwritten);
 * dw[9*c] = sum x*dz (overwritten; may be NULL). */
int shm_patch_bwd(const void* x, int ldx, const float* w, const float* y, const float* dy, float* dz,
                  void* dx, int lddx, float* dw, int batch, int h, int wd, int c, float slope,
                  int dtype, void* stream);
/* Flatten + Dense(5, no bias) (SHM.py:371-375): logits[n][j] = sum_k x[n][k] * w[k][j]. */
int shm_dense_fwd(const void* x, const float* w, float* y, int batch, int k, int nout, int dtype,
                  void* stream);
/* dx[n][k] += sum_j dy[n][j]*w[k][j] (accumulates onto dx, a [G] tensor); dw[k][j] = sum_n x[n][k]*dy[n][j]
 * (dw may be NULL). */
int shm_dense_bwd(const void* x, const float* w, const float* dy, void* dx, float* dw, int batch,
                  int k, int nout, int dtype, void* stream);
/* Dropout(0.2) as keep-mask multiply (SHM.py:363): y = x * mask * scale. */
int shm_mul_mask(const void* x, const float* mask, void* y, size_t n, float scale, int dtype,
                 void* stream);

/* ---- colour / standardisation / input assembly (SHM.py:480-531, 544-553, 576-624) -- */
/* tf.image.rgb_to_yuv + custom_per_image_standardization (SHM.py:1271-1309), per sample.
 * acc = f64 scratch [batch*2]; scale_out[batch] receives max(std, 1/256). */
int shm_rgb2yuv_std(const float* rgb, float* yuv, double* acc, float* scale_out, int batch,
                    size_t npix, void* stream);
/* averageCbCr (SHM.py:505): out[b,p,0:2] = mean_k yuv_k[b,p,1:3]. */
int shm_avg_cbcr(const float* y0, const float* y1, const float* y2, const float* y3,
                 const float* y4, float* out, size_t n_pix_total, void* stream);
/* 10-channel generator inputs in a pitch of ldo elements, zero padded (SHM.py:517-531, 576-594).
 * mode 0: G(1) input  -> out [batch,...]: view k = flags[k] ? 0 : Y_k ; one-hot tail = ED.
 * mode 1: cyclic inputs -> out [5*batch,...] (k-major): view j = (j==k) ? 0 : (flags[j] ? genY : Y_j);
 * one-hot k.  yuv_k are the standardised [batch,S,S,3] tensors, genY [batch,S,S,1]. */
int shm_build_gen_input(const float* y0, const float* y1, const float* y2, const float* y3,
                        const float* y4, const float* gen_y, int flags_mask, int mode, void* out,
                        int ldo, int batch, size_t npix, int dtype, void* stream);
/* Gradient of the cyclic inputs back into genY (G o G chain): dgenY[b,p] += sum over k, j!=k with
 * flags[j] of dcyc[k*batch+b, p, j]. */
int shm_cyc_input_bwd(const void* dcyc, int ld, int flags_mask, float* dgen_y, int batch,
                      size_t npix, int dtype, void* stream);
/* tf.image.yuv_to_rgb of concat([Y, avgCbCr]) (SHM.py:544-553, 613-624) for nimg = reps*batch
 * images (image i uses cbcr[i % batch]); writes rgb [nimg,S,S,3] and, if dpad != NULL, the padded
 * discriminator input of pitch ldp (rgb + optional GaussianNoise `noise` [nimg,S,S,3], SHM.py:352). */
int shm_yuv2rgb(const float* ych, const float* cbcr, const float* noise, float* rgb, void* dpad,
                int ldp, int nimg, int batch, size_t npix, int dtype, void* stream);
/* Pack raw rgb [nimg,S,S,3] (+ noise) into the padded discriminator input of pitch ldp. */
int shm_pack_rgb16(const float* rgb, const float* noise, void* dpad, int ldp, size_t npix_total,
                   int dtype, void* stream);
/* dY[i,p] (+)= sum_c d_pad[i,p,c], c<3 (yuv_to_rgb backward: every channel has dRGB/dY = 1). */
int shm_rgb16_to_dy(const void* d16, int ld, float* dy, size_t npix_total, int accumulate, int dtype,
                    void* stream);

/* Step-level random draws (GaussianNoise(0.1) SHM.py:352, Dropout(0.2) SHM.py:363): counter-based Philox-4x32-10, reproducible
 * from (seed, stream_id) whatever the launch geometry.  out[n] ~ N(0, stddev^2);  mask[n] = 1 with probability 1 - rate, else 0
 * (the keep mask shm_mul_mask multiplies by, scaled 1/(1-rate)). */
int shm_randn(float* out, size_t n, float stddev, unsigned long long seed, unsigned stream_id, void* stream);
int shm_keep_mask(float* out, size_t n, float rate, unsigned long long seed, unsigned stream_id, void* stream);

/* ---- losses (SHM.py:669-844) ------------------------------------------------------
 * Discriminator-head losses and their gradients.  Sample order in the D batch:
 * [D1: B][D3: 5B, k-major][D2: B][D4: 5B, k-major].  rf [12B, np], cls [12B,5].
 * Outputs: loss[16] (f64 sums over the batch of: D1_RF, D3_RF, D2_RF, D4_RF, D1_cls, D3_cls,
 * D4_cls), drf_D/dcls_D = gradient of mean_b(total_D+total_Class), drf_G [6B,np] = gradient of
 * mean_b((D1_RF + D3_RF)/6).
 * xent_mode selects the class-logit gradient of tf.nn.softmax_cross_entropy_with_logits (SHM.py:695-713):
 *   SHM_XENT_TF_FUSED (the reference AS EXECUTED): TF's fused kernel returns backprop = softmax - labels and the registered
 *     gradient is grad_loss * backprop, which is the derivative only for labels that sum to 1.  The D1 term's label row is
 *     [0,0,0,0,TARGET_LABELS] (SHM.py:477, 533, 688, 702; TARGET_LABELS ~ U(0.8,1.2), SHM.py:986), so the executed gradient
 *     is softmax - T*onehot;
 *   SHM_XENT_INTENDED: the true derivative sum(labels)*softmax - labels = T*(softmax - onehot).
 * The loss values are the same in both modes; the modes coincide when target == 1 and for every one-hot row (D3, D4). */
#define SHM_XENT_TF_FUSED 0
#define SHM_XENT_INTENDED 1
int shm_dhead_losses(const float* rf, const float* cls, double* loss, float* drf_d, float* dcls_d,
                     float* drf_g, int batch, int np, float target, int xent_mode, void* stream);
/* Image-space generator losses + gradient wrt the generated Y planes.
 * gen_rgb [B,S,S,3], cyc_rgb [5B,S,S,3] (k-major), cyc_y [5B,S,S,1], cbcr [B,S,S,2],
 * orig_k [B,S,S,3] raw rgb, ds_k [B,S,S,3] standardised yuv.
 * loss (f64 [32]): sums over the batch, see shmgan_amd/losses.py for the slot map.
 * dgen_y [B,S,S,1], dcyc_y [5B,S,S,1] are overwritten with d mean_b(10*L1 + 10*ssim + 10*NST).
 * ws: f64/f32 scratch of shm_image_losses_workspace() bytes. */
size_t shm_image_losses_workspace(int batch, int s);
int shm_image_losses(const float* gen_rgb, const float* cyc_rgb, const float* cyc_y,
                     const float* cbcr, const float* const* orig, const float* const* ds,
                     int flags_mask, float style_factor, double* loss, float* dgen_y, float* dcyc_y,
                     void* ws, size_t ws_bytes, int batch, int s, void* stream);

/* ---- SpecSeg mask network, inference only (SpecSeg.py:27-98; SpecSeg.predict at SHM.py:492) --
 * Its Conv2D(3x3, relu) layers are shm_conv2d_fwd with slope 0.  The rest: */
/* dst[p, 0:nc] = src[p, c0:c0+nc], dst[p, nc:lddst] = 0 (the Y plane into a 16-float pitch). */
int shm_pack_channels(const float* src, int ldsrc, int c0, int nc, float* dst, int lddst, size_t npix,
                      void* stream);
/* Keras BatchNormalization(axis=-1) in inference mode (SpecSeg.py:37,43,49,55,60):
 * out = (a - moving_mean) * gamma / sqrt(moving_var + eps) + beta. */
int shm_bn_apply(const float* a, int lda, const float* gamma, const float* beta, const float* mean,
                 const float* var, float eps, float* out, int ldo, size_t npix, int c, void* stream);
/* MaxPooling2D((2,2)) on even sizes (SpecSeg.py:38,44,50,56). */
int shm_maxpool2_fwd(const float* x, int ldx, float* y, int ldy, int batch, int h, int w, int c,
                     void* stream);
/* Conv2DTranspose(k=2, strides=2, 'same') + bias (SpecSeg.py:63,69,75,81).  w = Keras layout
 * [2][2][cout][cin] as stored; y [batch,2hi,2wi,cout]; slope 1.0f = linear. */
int shm_conv2d_transpose2x2_fwd(const void* x, int ldx, const void* w, const float* bias, void* y,
                                int ldy, int batch, int hi, int wi, int cin, int cout, float slope,
                                int dtype, void* stream);
/* Conv2D(1, (1,1), activation='sigmoid') (SpecSeg.py:88). */
int shm_head_sigmoid_fwd(const float* x, int ldx, const float* w, const float* bias, float* y,
                         size_t npix, int c, void* stream);
/* Spec_loss terms (SHM.py:792-796, logged only): loss[k] (f64 [5], zeroed by the call) = sum over
 * samples, pixels and the 3 yuv channels of (mask * cyc_k_yuv - mask * ds_k)^2, where
 * cyc_k_yuv = concat(cyc_y[k*batch + b], cbcr[b]); mask [batch,S,S,1].  ds = host array of 5
 * device pointers.  reduce_mean = loss / (batch*npix*3). */
int shm_spec_loss(const float* cyc_y, const float* cbcr, const float* const* ds, const float* mask,
                  double* loss, int batch, size_t npix, void* stream);

/* ---- live attention branch: attention_layer SHM.py:404-412, its use at SHM.py:248-275, 290-293, 358-359 ------------
 * The reference evaluates attention_layer on a constant zero mask at model-build time (SURVEY finding 3), so the executed
 * graph adds zeros; these entry points serve the "as-intended" mode (attention="live"): the SpecSeg mask of the step goes
 * through MaxPooling2D -> Conv2D(1->C, 3x3, leaky_relu) -> Conv2D(C->C, 3x3, leaky_relu) (the two convolutions are
 * shm_conv2d_fwd / _dgrad / _wgrad + shm_lrelu_bwd) and is added to the skip tensor. */
/* MaxPooling2D(k x k) of mask [batch,s,s,1] (fp32) into channel 0 of dst [batch,s/k,s/k,lddst] (activation-typed, other
 * channels zeroed); k = 1 copies (attention_layer(pool=False), SHM.py:248). */
int shm_mask_pool_pack(const float* mask, void* dst, int lddst, int batch, int s, int k, int dtype, void* stream);
/* out[i] = a[i] + b[(i0 + i) % nb] for nimg images of `per` elements (per % 4 == 0): skip + attention map of the image's
 * sample (SHM.py:290-293, 359); in the batched plan image i0 + i of the batch is a copy of sample (i0 + i) % nb. */
int shm_add_bcast(const void* a, const void* b, void* out, int nimg, size_t per, int nb, int i0, int dtype, void* stream);
/* Its gradient wrt b: dst[j] (+)= sum of src[i] over the images with (i0 + i) % nb == j. */
int shm_sum_groups(const void* src, void* dst, int nimg, size_t per, int nb, int i0, int accumulate, int dtype, void* stream);

/* ---- input pipeline (datasetLoader.py:47-60: image_dataset_from_directory -> /255 -> flip_up_down) ----
 * tf.image.resize(bilinear, half-pixel centres, no antialias) of one decoded uint8 image [hin,win,c] to
 * float32 [ho,wo,c], times `scale` (1/255), optionally flipped top-to-bottom. */
int shm_resize_bilinear_u8(const unsigned char* src, int hin, int win, int c, float* dst, int ho, int wo,
                           float scale, int flip_ud, void* stream);

/* ---- optimizer (SHM.py:169-175, 859-872) ------------------------------------------
 * tf.clip_by_value(g,-1,1) + Keras adam_v2.Adam: m += (g-m)(1-b1); v += (g^2-v)(1-b2);
 * w -= alpha * m / (sqrt(v) + eps); alpha = lr_t*sqrt(1-b2^t)/(1-b1^t) computed by the caller.
 * gscale multiplies g before the clip (1/world_size for data parallel).  With shm_set_abort_words armed on this thread the
 * kernel checks the device word first and leaves w, m, v untouched while it is non-zero. */
int shm_adam_clip(float* w, float* m, float* v, const float* g, size_t n, float alpha, float beta1,
                  float beta2, float eps, float gscale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SHMGAN_HIP_H */
