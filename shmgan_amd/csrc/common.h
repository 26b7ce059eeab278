// Shared helpers for libshmgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/shmgan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16_t;            // activation element type of the bf16 path (SHM_BF16)

void shm_set_error(const char* fmt, ...);
void shm_set_last_kernel(const char* fmt, ...);        // symbol of the MFMA kernel a convolution entry point chose

// conv_rgb.hip: the 3-channel stride-2 first layer on the compact image layout (one 16-byte chunk per pixel).  1 = launched, 0 = not that
// shape (the caller goes on to its generic kernels), < 0 = SHM_E_*.
int shm_rgb_s2_fwd_launch(const void* x, int ldx, const void* wk, int K, const float* bias, void* y, int ldy, int batch, int hi, int wi, int cout, float slope,
                          double* stats, int stats_slots, unsigned stats_stride, size_t xbytes, size_t ybytes, int dtype, hipStream_t st);
int shm_rgb_s2_wgrad_launch(const void* x, int ldx, const void* dy, int lddy, float* part, size_t ws_bytes, int batch, int hi, int wi, int cin, int cout,
                            size_t xbytes, size_t dybytes, int dtype, int* nsplit_out, hipStream_t st);

// Dispatch knobs behind shm_set_tuning()/shm_get_tuning() (include/shmgan_hip.h lists the keys).  Process-wide
// atomics read at every launch; the initial value comes from the environment variable named in the table of
// norm_elem.hip (so tools/ablate_conv.py keeps working), -1/0 = the built-in choice.
enum ShmTune {
    SHM_TUNE_TAPGEMM_VARIANT = 0,     // SHM_TG_* below; 0 = automatic
    SHM_TUNE_TAPGEMM_HALO_MIN,        // fp32: 128-wide halo blocks from which the 128-wide block is taken unconditionally
    SHM_TUNE_TAPGEMM_SMALL_GRID,      // grids below this many 128x128 tiles take the 64x128 tile
    SHM_TUNE_TAPGEMM_PHASE4_MIN,      // four-phase (stride-2 transposed) launches with at least this many fused blocks take tapgemm_phase4_kernel
    SHM_TUNE_WGRAD_VARIANT,           // 0 = automatic, 1 = generic kernel only, 2 = halo kernel but no thin-input packing, 3 = no stride-2 halo form
    SHM_TUNE_WGRAD_BLOCKS,            // split-K target (blocks), 0 = automatic
    SHM_TUNE_WGRAD_BF16_ROWS,         // wgrad_halo_bf16_kernel: pixel rows per stage, 0 = automatic (4 when the map allows), 2 or 4
    SHM_TUNE_STATS_FUSION,            // 1 = InstanceNorm statistics in the conv epilogue (default), 0 = separate pass
    SHM_TUNE_ELEM_REVERSE,            // 1 = in_apply / in_bwd_reduce walk the tensor back to front (Infinity-Cache reuse), 0 = front to back
    SHM_TUNE_ELEM_REDUCE_BLOCKS,      // block target of the InstanceNorm-backward reduce pass
    SHM_TUNE_ELEM_NT,                 // 1 = the InstanceNorm-backward apply pass reads its (dead afterwards) gradient tensor with non-temporal loads
    SHM_TUNE_ELEM_CHUNK_MB,           // shm_in_bwd: reduce + apply per chunk of samples whose tensors fit this many MiB (0 = the whole batch at once)
    SHM_TUNE_ELEM_INTERLEAVE,         // InstanceNorm-backward apply pass: a sample's blocks take pixel tiles round-robin instead of one contiguous chunk each
    SHM_TUNE_ELEM_STREAM_BLOCKS,      // block target of the streaming elementwise passes (InstanceNorm apply / backward apply, pooling forms)
    SHM_TUNE_ELEM_APPLY_BLOCKS,       // block target of the InstanceNorm-backward apply pass
    SHM_TUNE_TAPGEMM_WREG16,          // bf16 weights-in-registers layers: 2 = ping-pong kernel where eligible (conv_pingpong.hip), 1 = the eight-wave form with 16-column wave tiles (v_mfma_f32_16x16x32_bf16), 0 = the four-wave form
    SHM_TUNE_WGRAD_BF16_WIDE,         // bf16 weight gradient, eight-wave 64 x 128 block: 0 = automatic (stride 2 only), 1 never, 2 stride 2, 3 unit stride, 4 both
    SHM_TUNE_WGRAD_F32_SPLIT,         // fp32 3x3 unit-stride weight gradient: 1 = six bf16 MFMA products of exact three-plane splits (conv_wgrad_x3.hip), 0 = exact-fp32 MFMA (default)
    SHM_TUNE_TAPGEMM_FLAT_EPILOGUE,   // 1 = treat the outputs as larger than 4 GiB (tests: the 64-bit-address epilogues and the kernels that do not need buffer stores)
    SHM_TUNE_ELEM_FUSED_BWD,          // 1 = bf16 InstanceNorm backward in one pass where eligible and shm_in_bwd_fused_scratch was given (in_bwd_fused8_kernel), 0 = two passes
    SHM_TUNE_ELEM_FUSED_MAX_SLICES,   // in_bwd_fused8_kernel: most slices (= blocks) per sample; a sample's blocks must be resident together (1024 fit an idle chip)
    SHM_TUNE_CONV_F32_SPLIT,          // fp32 3x3 unit-stride forward / input-gradient layers (> 64 output channels): 1 = six bf16 MFMA products of exact three-plane splits (conv_fwd_x3.hip), 0 = exact-fp32 MFMA (default)
    SHM_TUNE_ELEM_FUSED_TEST_STALL,   // tests only: 1 = in_bwd_fused8_kernel's barriers wait for one block more than the grid has (the timeout path)
    SHM_TUNE_ELEM_FUSED_HOLD,         // one-pass bf16 InstanceNorm backward: 0 automatic (g held, a streamed twice where eligible; else g and a held), 1 = g and a held only (round 5), 2 = g held only
    SHM_TUNE_ELEM_FUSED_GVARIANT,     // in_bwd_fusedg_kernel's register-budget form: 0 = <2, 2, 4> (four blocks per CU, two transient loads per batch; default), 1 = <8, 8, 3>
    SHM_TUNE_COUNT
};
int shm_tune(int id);
unsigned long long* shm_clock_probe();       // this thread's shm_set_clock_probe buffer (norm_elem.hip), or null
const unsigned* shm_abort_dev_word();       // this thread's shm_set_abort_words device word (norm_elem.hip), or null

// Barrier of the LDS-DMA pipelines.  A stage is refilled by DMA instructions issued AFTER the barrier that follows its last use, so a
// wave must not enter that barrier with fragment reads of the stage still queued: the MFMAs that consume them are register-only
// instructions which hipcc is free to sink below the barrier (it does, with the nine taps unrolled), taking the implicit
// s_waitcnt lgkmcnt with them -- and with the LDS pipe saturated (bf16) a queued ds_read can then be overtaken by the DMA write
// of another wave (tools/probes/conv_repeat_probe.py: one 16-byte weight chunk stale in 1 of 30 launches).  Hence the explicit wait; a
// __syncthreads() would also drain vmcnt, i.e. the DMA pipeline.
#define SHM_LDS_BARRIER()                                       \
    do {                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_s_barrier();                           \
    } while (0)

// LDS-DMA the compiler does not see (round 5).  hipcc cannot tell an LDS access from the destination of an LDS-DMA in flight when the access has
// no memory operand it can reason about -- ds_read_b64_tr_b16 through its builtin, any inline-asm access -- and waits vmcnt(0) in front of
// it: in the bf16 weight-gradient kernels that is the first transposed read of every stage, i.e. the stage pipeline ("DMA two stages ahead,
// counted s_waitcnt vmcnt") drained to the DMA issued a moment before (MFMA busy 0.34).  Issued as inline asm the DMA is invisible to that
// pass and the kernels' own counted waits are the only ones.  rs: the four descriptor words (shm_rsrc_words); lds_addr: wave-uniform LDS
// byte address of the 1 KiB destination (lane l lands at + 16 l); voff: per-lane byte offset into the buffer, out of range = zeros.
typedef unsigned shm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ shm_u32x4 shm_rsrc_words(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    return shm_u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
// m0 is written behind the compiler's back and hipcc refuses "m0" as a clobber ("reserved register ... may not be preserved", -Winline-asm): the
// asm saves and restores it, so a compiler-issued m0 consumer in the same kernel (an LDS-DMA builtin whose m0 hipcc set earlier, movrel, sendmsg)
// still finds its value (round-5 advisor).  Two scalar moves per DMA, in the shadow of the MFMAs.
__device__ __forceinline__ void shm_dma16(const shm_u32x4 rs, const unsigned lds_addr, const unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rs)
                 : "memory");
}
__device__ __forceinline__ unsigned shm_lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

#define SHM_TG_COUNT 17           // SHM_TG_* of include/shmgan_hip.h

// 4-channel vector access in either element type; arithmetic is always fp32.
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
    const uint2 u = *(const uint2*)p;
    f32x4 r;
    r[0] = __uint_as_float(u.x << 16);
    r[1] = __uint_as_float(u.x & 0xffff0000u);
    r[2] = __uint_as_float(u.y << 16);
    r[3] = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
// the same for data that is dead after this read (a gradient signal consumed by its only reader): non-temporal, so that the
// stream does not displace tensors the next kernels re-read from L2 / Infinity Cache
__device__ __forceinline__ f32x4 ld4nt(const float* p) { return __builtin_nontemporal_load((const f32x4*)p); }
__device__ __forceinline__ f32x4 ld4nt(const bf16_t* p) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t u = __builtin_nontemporal_load((const u32x2_t*)p);
    f32x4 r;
    r[0] = __uint_as_float(u[0] << 16);
    r[1] = __uint_as_float(u[0] & 0xffff0000u);
    r[2] = __uint_as_float(u[1] << 16);
    r[3] = __uint_as_float(u[1] & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t o;
    o[0] = (bf16_t)v[0];
    o[1] = (bf16_t)v[1];
    o[2] = (bf16_t)v[2];
    o[3] = (bf16_t)v[3];
    *(bf16x4_t*)p = o;
}
// value as it will read back from a tensor of type T
__device__ __forceinline__ float rnd_as(const float*, float v) { return v; }
__device__ __forceinline__ float rnd_as(const bf16_t*, float v) { return (float)(bf16_t)v; }

// run `...` with T = float or bf16_t according to an SHM_F32 / SHM_BF16 dtype argument
#define SHM_DISPATCH(dtype, who, ...)                                          \
    do {                                                                       \
        if ((dtype) == SHM_BF16) {                                             \
            using T = bf16_t;                                                  \
            __VA_ARGS__;                                                       \
        } else if ((dtype) == SHM_F32) {                                       \
            using T = float;                                                   \
            __VA_ARGS__;                                                       \
        } else {                                                               \
            shm_set_error("%s: dtype %d not in {SHM_F32, SHM_BF16}", who, (int)(dtype)); \
            return SHM_E_DTYPE;                                                \
        }                                                                      \
    } while (0)
// same with a second type TG for gradient-signal tensors: fp32 under SHM_BF16_GF32
#define SHM_DISPATCH_G(dtype, who, ...)                                        \
    do {                                                                       \
        if ((dtype) == SHM_BF16) {                                             \
            using T = bf16_t;                                                  \
            using TG = bf16_t;                                                 \
            __VA_ARGS__;                                                       \
        } else if ((dtype) == SHM_BF16_GF32) {                                 \
            using T = bf16_t;                                                  \
            using TG = float;                                                  \
            __VA_ARGS__;                                                       \
        } else if ((dtype) == SHM_F32) {                                       \
            using T = float;                                                   \
            using TG = float;                                                  \
            __VA_ARGS__;                                                       \
        } else {                                                               \
            shm_set_error("%s: dtype %d not in {SHM_F32, SHM_BF16, SHM_BF16_GF32}", who, (int)(dtype)); \
            return SHM_E_DTYPE;                                                \
        }                                                                      \
    } while (0)

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

// The same for an entry point whose f64 scratch is "zero on entry, zero on return": if a launch after the one that
// filled the scratch fails, clear it (best effort) so that a transient error does not poison every later call.
#define SHM_LAUNCH_CHECK_CLEAR(name, ptr, bytes, st)                            \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            (void)hipMemsetAsync((ptr), 0, (bytes), (st));                      \
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
// LeakyReLU for 0 <= slope <= 1 as max(u, u * slope) in TWO instructions, both visible to the compiler: v_mul_f32 + v_maximum3_f32 (the IEEE-754-2019
// `maximum`, a gfx950 instruction: NaN-propagating, so hipcc does not quiet a possible signalling NaN in u first -- the third instruction that fmaxf costs on
// an MFMA result).  Same bits as shm_lrelu for every finite u (u and u * slope never differ in sign); a NaN stays a NaN, so a diverged activation still shows
// in the statistics and the losses; u = -inf at slope 0 gives NaN, as shm_lrelu does (-inf * 0).  Rounds 3-4 spelled the max as inline asm (v_med3 / v_max_f32):
// opaque to the scheduler and to hipcc's hazard recogniser -- the round-4 store-data fault below was such an asm write, and an asm READ of an MFMA result gets
// no wait states (advisor findings, round 4).  No inline-asm arithmetic is left in the kernels.
// HAZARD (round 4): a 16- or 12-byte store reads its data VGPRs after it has issued, and a VALU write into one of them must stay two wait states
// away.  hipcc spaces that out itself, except after a BUFFER store whose soffset is an SGPR (its table exempts those; the MI355X does not): there the
// next tile's LeakyReLU landed in a register of the tile just stored, and ~4e-4 of conv3x3s2_rgb_fwd_kernel's values of that register went out
// wrong, only under load.  Keep the offsets of wide buffer stores in the immediate field; tools/check_isa_hazards.py scans the built library for
// the pattern (tests/test_abi.py runs it).
__device__ __forceinline__ float shm_lrelu_max(float u, float slope) { return __builtin_elementwise_maximum(u, u * slope); }

// InstanceNormalization apply, (x - mean) * inv + beta, in ONE spelling for the stand-alone pass (shm_in_apply) and for the
// consumers that normalise their operand tile in LDS ("fused block", shm_conv2d_in_fwd_norm / shm_conv2d_wgrad_norm): a
// subtraction and one fused multiply-add, so that both give the same bits.
__device__ __forceinline__ float shm_in_norm(float x, float mean, float inv, float beta) { return __builtin_fmaf(x - mean, inv, beta); }
// A block's normalisation table for such consumers: float [batch][4][c] = per sample the planes mean, inv, beta[c] and
// ring = mean - beta / inv, the RAW value whose normalised image is 0 (what an out-of-image tap has to read in SHM_NORM_SCALED mode).
#define SHM_NT_PLANES 4
#define SHM_NT_MAXC 256           // normalised channels a folding consumer keeps in LDS

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
