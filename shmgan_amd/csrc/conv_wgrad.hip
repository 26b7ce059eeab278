// Convolution weight gradient on v_mfma_f32_32x32x2_f32 (exact fp32) and, for the bf16 path, on
// v_mfma_f32_32x32x16_bf16 with transposed LDS reads (wgrad_bf16_kernel) -- gfx950.
//
//   dW[tap][ci][co] = sum_{pixels p} X[src(p, tap)][ci] * dY[p][co]
//
// GEMM view per tap: M = ci, N = co, K = pixels.  Both operands are pixel-major in HBM
// (channels contiguous), which is exactly the [k][m] / [k][n] LDS image the f32 MFMA's
// one-float-per-lane operands want: 32 lanes read 32 consecutive floats (conflict-free
// ds_read_b32), the two lane halves read two consecutive pixels.
//
// One block = a 64(ci) x 64(co) tile for ALL taps over one slice of the pixels (split-K):
// the dY tile is staged once per 16 pixels and shared by the 9 taps; the 9 shifted X tiles
// are fetched through L1/L2.  Wave w owns the 32x32 sub-tile (w>>1, w&1) of every tap
// (9 accumulators).  Partial slabs go to a workspace and are summed in a fixed order by
// wgrad_reduce_kernel (deterministic, no float atomics).
#include "wgrad.h"

#include <stdlib.h>

struct WgradArgs {            // x, x2, dy: float (wgrad_kernel) or bf16 (wgrad_bf16_kernel) tensors
    const void* x;
    const void* x2;
    int c1, ldx, ldx2;
    const void* dy;
    int lddy;
    float* part;
    int hi, wi, ho, wo;
    int cin_ld, cin, cout;
    int is, ntaps;
    int dh[9], dw[9];
    int M, pix_per_split;
    unsigned xbytes, x2bytes, dybytes;
};

// Stage = 8 pixels.  Thread -> (half, k, c4): pixel slot k (0..7), 4-channel lane c4 (0..15), and
// the taps {half, half+2, ...}.  Two LDS stages + two register sets (loads two stages ahead),
// raw buffer loads with out-of-range offsets for padding / tails (no divergent load branches).
// STRADDLE: the 64-channel tile may contain channels of both concat sources (c1 % 64 != 0; only the
// small-filter test configurations): every X load is then issued against both descriptors with one
// of them masked out of range, so the descriptor stays wave-uniform (no waterfall loop).
template <int NT, bool STRADDLE>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr int BKP = 8;
    constexpr int NTL = (NT + 1) / 2;          // tap loads per thread
    __shared__ __attribute__((aligned(16))) float Xs[2][NT][BKP][64];
    __shared__ __attribute__((aligned(16))) float Ds[2][BKP][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int mi = wave >> 1, ni = wave & 1;
    // XCD-aware block order (round 3, as in the bf16 kernels): the blocks of one pixel split -- same operand tiles, different
    // (ci, co) tile -- are dealt to ONE XCD's L2 instead of eight (r02 PMC: 969 MB HBM-side per launch, L2 hit rate 0.38)
    const Blk3 blk = xcd_block_order();
    const int ci0 = blk.x * 64, co0 = blk.y * 64;
    const int p_begin = blk.z * a.pix_per_split;
    const int p_end = min(a.M, p_begin + a.pix_per_split);
    const int nstages = (p_end - p_begin + BKP - 1) / BKP;

    const int half = __builtin_amdgcn_readfirstlane(tid >> 7);      // wave-uniform: waves 0,1 / 2,3
    const int k = (tid >> 4) & 7, c4 = tid & 15;
    const int c = ci0 + c4 * 4;
    const bool second = STRADDLE ? (c >= a.c1) : (ci0 >= a.c1);
    const bool xvalid = c < a.cin_ld;
    const int ld = second ? a.ldx2 : a.ldx;
    const int cc = second ? c - a.c1 : c;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000);
    // tap table of this wave pair, hoisted out of the loop (wave-uniform -> SGPRs)
    constexpr int KS = NT == 9 ? 3 : 1;
    int tofb[NTL];            // byte offset of tap j relative to the centre pixel
    unsigned tbit[NTL];       // its bit in the 9-bit validity mask (0: tap not owned by this wave)
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
        const int t = half + 2 * j;
        const bool tv = t < NT;
        const int tt = tv ? t : 0;
        tofb[j] = (a.dh[tt] * a.wi + a.dw[tt]) * ld * 4;
        tbit[j] = tv ? (1u << tt) : 0u;
    }
    int rdh[KS], cdw[KS];     // the KS distinct row / column displacements (tap = kh*KS + kw)
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        rdh[i] = a.dh[i * KS];
        cdw[i] = a.dw[i];
    }
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    const int co = co0 + c4 * 4;
    const bool dvalid = (co < a.cout) && half == 0;

    // running pixel coordinate of this thread's slot (advances by BKP per stage)
    int p = p_begin + k;
    int ow, oh, n;
    {
        const int pp = p < a.M ? p : 0;
        ow = pp % a.wo;
        const int t2 = pp / a.wo;
        oh = t2 % a.ho;
        n = t2 / a.ho;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto gload = [&](f32x4 (&rx)[NTL], f32x4& rd) {
        const bool ok = p < p_end;
        const int ihb = oh * a.is, iwb = ow * a.is;
        const unsigned base = (unsigned)(((n * a.hi + ihb) * a.wi + iwb) * ld + cc) * 4u;
        // 9-bit tap validity mask of this pixel: bit kh*KS+kw = row kh inside AND column kw inside
        unsigned rm = 0, cm = 0;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            rm |= ((unsigned)(ihb + rdh[i]) < (unsigned)a.hi ? 1u : 0u) << i;
            cm |= ((unsigned)(iwb + cdw[i]) < (unsigned)a.wi ? 1u : 0u) << i;
        }
        unsigned m9 = 0;
#pragma unroll
        for (int i = 0; i < KS; ++i) m9 |= (rm & (1u << i)) ? (cm << (i * KS)) : 0u;
        if (!(ok && xvalid)) m9 = 0;
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            unsigned off = (m9 & tbit[j]) ? base + (unsigned)tofb[j] : 0xffffffffu;
            if constexpr (abl::sameline) off = (m9 & tbit[j]) ? (unsigned)(c4 * 16 + (off & 0x300u)) : 0xffffffffu;        // timing only
            if (STRADDLE) {
                u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)(second ? 0xffffffffu : off), 0, 0);
                u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)(second ? off : 0xffffffffu), 0, 0);
                rx[j] = __builtin_bit_cast(f32x4, v1 | v2);
            } else {
                u32x4 v1 = second ? __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)off, 0, 0)
                                  : __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)off, 0, 0);
                rx[j] = __builtin_bit_cast(f32x4, v1);
            }
        }
        const unsigned offd = (ok && dvalid) ? (unsigned)(p * a.lddy + co) * 4u : 0xffffffffu;
        rd = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)offd, 0, 0));
        // advance to the next stage
        p += BKP;
        ow += BKP;
        if (ow >= a.wo) {                      // at most one wrap when wo >= BKP (every real layer)
            do {
                ow -= a.wo;
                if (++oh == a.ho) {
                    oh = 0;
                    ++n;
                }
            } while (ow >= a.wo);
        }
    };
    auto sstore = [&](int buf, const f32x4 (&rx)[NTL], const f32x4& rd) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int t = half + 2 * j;
            if (t < NT) *(f32x4*)(&Xs[buf][t][k][c4 * 4]) = rx[j];
        }
        if (half == 0) *(f32x4*)(&Ds[buf][k][c4 * 4]) = rd;
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < BKP / 2; ++kk) {
            const int kr = 2 * kk + h;
            const float bv = Ds[buf][kr][ni * 32 + l31];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float av = Xs[buf][t][kr][mi * 32 + l31];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    };

    if (nstages > 0) {
        f32x4 rx0[NTL], rx1[NTL], rd0, rd1;
        gload(rx0, rd0);
        if (nstages > 1) gload(rx1, rd1);
        sstore(0, rx0, rd0);
        __syncthreads();
        int s = 0;
#define WG_BAR()                                  \
    do {                                          \
        if constexpr (!abl::nobar) __syncthreads(); \
    } while (0)
#define WG_GLOAD(a_, b_)                            \
    do {                                            \
        if constexpr (!abl::noload) gload(a_, b_);    \
    } while (0)
#define WG_SSTORE(i_, a_, b_)                            \
    do {                                                 \
        if constexpr (!abl::nostore) sstore(i_, a_, b_);   \
    } while (0)
        for (; s + 3 < nstages; s += 2) {
            WG_GLOAD(rx0, rd0);
            compute(0);
            WG_SSTORE(1, rx1, rd1);
            WG_BAR();
            WG_GLOAD(rx1, rd1);
            compute(1);
            WG_SSTORE(0, rx0, rd0);
            WG_BAR();
        }
        const int left = nstages - s;
        if (left >= 3) gload(rx0, rd0);
        compute(0);
        if (left >= 2) {
            sstore(1, rx1, rd1);
            __syncthreads();
            compute(1);
            if (left >= 3) {
                sstore(0, rx0, rd0);
                __syncthreads();
                compute(0);
            }
        }
    }

    // partial slab [split][tap][cin][cout]
    float* out = a.part + (size_t)blk.z * NT * a.cin * a.cout;
    const int con = co0 + ni * 32 + l31;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ci < a.cin && con < a.cout) out[((size_t)t * a.cin + ci) * a.cout + con] = acc[t][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 operands, fp32 accumulation: v_mfma_f32_32x32x16_bf16 contracts 16 pixels per instruction.
// Both operands are pixel-major ([pixel][channel], channels contiguous) but the MFMA wants, per lane,
// 8 consecutive PIXELS of one channel: the LDS image keeps the HBM layout (128-byte rows of 64
// channels) and the fragments are fetched with ds_read_b64_tr_b16 (hardware transpose: a 16-lane group
// reads a 4-pixel x 16-channel block column-major).  Rows whose index has bit 1 set hold their two
// 64-byte halves swapped, which makes the four rows x two channel blocks a 32-lane half reads hit 32
// distinct 8-byte bank pairs.  Stage = 16 pixels; same work split, split-K slabs and (register
// staged, two-stages-ahead) pipeline as wgrad_kernel.
template <int NT, bool STRADDLE>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradArgs a) {
    constexpr int BKP = 16;
    constexpr int NTL = (NT + 1) / 2;          // tap loads per thread
    __shared__ __attribute__((aligned(16))) unsigned short Xs[2][NT][BKP][64];
    __shared__ __attribute__((aligned(16))) unsigned short Ds[2][BKP][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int mi = wave >> 1, ni = wave & 1;
    const Blk3 blk = xcd_block_order();
    const int ci0 = blk.x * 64, co0 = blk.y * 64;
    const int p_begin = blk.z * a.pix_per_split;
    const int p_end = min(a.M, p_begin + a.pix_per_split);
    const int nstages = (p_end - p_begin + BKP - 1) / BKP;

    const int half = __builtin_amdgcn_readfirstlane(tid >> 7);      // wave-uniform: waves 0,1 / 2,3
    const int k = (tid >> 3) & 15, c8 = tid & 7;                    // pixel slot, 8-channel (16-byte) lane
    const int c = ci0 + c8 * 8;
    const bool second = STRADDLE ? (c >= a.c1) : (ci0 >= a.c1);
    const bool xvalid = c < a.cin_ld;
    const int ld = second ? a.ldx2 : a.ldx;
    const int cc = second ? c - a.c1 : c;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000);
    constexpr int KS = NT == 9 ? 3 : 1;
    int tofb[NTL];
    unsigned tbit[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
        const int t = half + 2 * j;
        const bool tv = t < NT;
        const int tt = tv ? t : 0;
        tofb[j] = (a.dh[tt] * a.wi + a.dw[tt]) * ld * 2;
        tbit[j] = tv ? (1u << tt) : 0u;
    }
    int rdh[KS], cdw[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        rdh[i] = a.dh[i * KS];
        cdw[i] = a.dw[i];
    }
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    const int co = co0 + c8 * 8;
    const bool dvalid = (co < a.cout) && half == 0;

    int p = p_begin + k;
    int ow, oh, n;
    {
        const int pp = p < a.M ? p : 0;
        ow = pp % a.wo;
        const int t2 = pp / a.wo;
        oh = t2 % a.ho;
        n = t2 / a.ho;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto gload = [&](u32x4 (&rx)[NTL], u32x4& rd) {
        const bool ok = p < p_end;
        const int ihb = oh * a.is, iwb = ow * a.is;
        const unsigned base = (unsigned)(((n * a.hi + ihb) * a.wi + iwb) * ld + cc) * 2u;
        unsigned rm = 0, cm = 0;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            rm |= ((unsigned)(ihb + rdh[i]) < (unsigned)a.hi ? 1u : 0u) << i;
            cm |= ((unsigned)(iwb + cdw[i]) < (unsigned)a.wi ? 1u : 0u) << i;
        }
        unsigned m9 = 0;
#pragma unroll
        for (int i = 0; i < KS; ++i) m9 |= (rm & (1u << i)) ? (cm << (i * KS)) : 0u;
        if (!(ok && xvalid)) m9 = 0;
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const unsigned off = (m9 & tbit[j]) ? base + (unsigned)tofb[j] : 0xffffffffu;
            if (STRADDLE) {
                u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)(second ? 0xffffffffu : off), 0, 0);
                u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)(second ? off : 0xffffffffu), 0, 0);
                rx[j] = v1 | v2;
            } else {
                rx[j] = second ? __builtin_amdgcn_raw_buffer_load_b128(rs2, (int)off, 0, 0)
                               : __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)off, 0, 0);
            }
        }
        const unsigned offd = (ok && dvalid) ? (unsigned)(p * a.lddy + co) * 2u : 0xffffffffu;
        rd = __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)offd, 0, 0);
        p += BKP;
        ow += BKP;
        if (ow >= a.wo) {
            do {
                ow -= a.wo;
                if (++oh == a.ho) {
                    oh = 0;
                    ++n;
                }
            } while (ow >= a.wo);
        }
    };
    const int scol = (c8 ^ (((k >> 1) & 1) << 2)) * 8;              // swizzled 16-byte chunk of this thread's row
    auto sstore = [&](int buf, const u32x4 (&rx)[NTL], const u32x4& rd) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int t = half + 2 * j;
            if (t < NT) *(u32x4*)(&Xs[buf][t][k][scol]) = rx[j];
        }
        if (half == 0) *(u32x4*)(&Ds[buf][k][scol]) = rd;
    };
    // transposed-read addressing: lane (group g = lane>>4, i = lane&15) supplies row 8h + (i>>2) (+4 for the
    // second read) and the 4 channels [32*tile + 16*(g&1) + 4*(i&3), +4)
    const int frow = 8 * h + ((lane & 15) >> 2);
    const int fsw = ((frow >> 1) & 1) << 5;
    const int fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int fa = frow * 64 + ((mi * 32 + fcol) ^ fsw);
    const int fb = frow * 64 + ((ni * 32 + fcol) ^ fsw);
    auto compute = [&](int buf) {
        const bf16x8 bv = tr_frag(&Ds[buf][0][0] + fb);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 av = tr_frag(&Xs[buf][t][0][0] + fa);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
        }
    };

    if (nstages > 0) {
        u32x4 rx0[NTL], rx1[NTL], rd0, rd1;
        gload(rx0, rd0);
        if (nstages > 1) gload(rx1, rd1);
        sstore(0, rx0, rd0);
        __syncthreads();
        int s = 0;
        for (; s + 3 < nstages; s += 2) {
            gload(rx0, rd0);
            compute(0);
            sstore(1, rx1, rd1);
            __syncthreads();
            gload(rx1, rd1);
            compute(1);
            sstore(0, rx0, rd0);
            __syncthreads();
        }
        const int left = nstages - s;
        if (left >= 3) gload(rx0, rd0);
        compute(0);
        if (left >= 2) {
            sstore(1, rx1, rd1);
            __syncthreads();
            compute(1);
            if (left >= 3) {
                sstore(0, rx0, rd0);
                __syncthreads();
                compute(0);
            }
        }
    }

    float* out = a.part + (size_t)blk.z * NT * a.cin * a.cout;
    const int con = co0 + ni * 32 + l31;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ci < a.cin && con < a.cout) out[((size_t)t * a.cin + ci) * a.cout + con] = acc[t][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// 3x3 / stride-1 weight gradient with an LDS halo patch.
//
// A stage is a patch of 2 x 16 output pixels of one image.  Its 4 x 18 input halo (64 channels)
// and its 2 x 16 dY pixels are brought in ONCE by LDS-DMA (buffer_load ... lds, out-of-image
// pixels read as zeros through the descriptor range check); the nine taps are then nine shifted
// views of the same LDS image, i.e. only an immediate offset on the ds_read_b32 that feeds each
// MFMA.  Compared with fetching nine shifted tiles through L1 this moves 3x fewer bytes and
// needs ~5x fewer address instructions per MFMA.  Three stages, DMA two stages ahead, counted
// s_waitcnt vmcnt, raw s_barrier.  Work split: block = (64 ci, 64 co, slice of patches); wave w
// owns the 32x32 sub-tile (w>>1, w&1) of all nine taps; partial slabs as in wgrad_kernel.
// A 16-byte global load the COMPILER does not see as a vector-memory operation (inline asm, drained on the spot).  The table
// registers of the NM kernels are re-read when a block moves on to the next image, i.e. under a branch: as plain loads hipcc has
// to assume them outstanding at every later use and puts s_waitcnt vmcnt(0) in front of each normalisation -- which also waits
// for the LDS-DMA of the stage just issued, in the middle of the MFMA stream (measured: +6-10 % on the kernel).
__device__ __forceinline__ f32x4 load16_drained(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// NM: a wave normalises the halo items it DMA'd itself, one stage ahead of their use.  A lane's four channels are the same for
// every item and patch (no swizzle in this image), so their (mean, inv, beta) live in registers and are re-read when the image
// changes (at most a few times per block).
// S2 (round 3; IS = conv stride 2): the stride-2 3x3 layers (SAME padding of an even map: no pad before, one row / column after).  A stage is a patch of
// 2 x 8 OUTPUT pixels and its 5 x 17 input halo -- 22 + 4 DMA items, the same 26 KiB stage, per-wave DMA counts and waits as the
// unit-stride form, with half the MFMAs per barrier; tap (kh, kw) of output pixel (qr, qc) is halo pixel (2 qr + kh, 2 qc + kw), still an
// immediate offset on the ds_read_b32 (a lane reads one float of a 128-byte run whatever the pixel stride: no bank conflicts).
// Against wgrad_kernel<9> (nine shifted tiles through registers, a barrier per 8 pixels): 5.3 input pixels fetched per output pixel
// instead of 9, no VGPR staging, no ds_write, a barrier per 16 pixels.
template <int NM = 0, bool S2 = false>
__global__ __launch_bounds__(256, 2) void wgrad_halo_kernel(const WgradHaloArgs a) {
    static_assert(!S2 || NM == 0, "norm: unit-stride form");
    constexpr int IS = S2 ? 2 : 1;
    constexpr int PW = IS == 1 ? 16 : 8, HC = IS * PW + 3 - IS, HR = 2 * IS + 3 - IS;    // patch 2 x 16, halo 4 x 18 | 2 x 8, 5 x 17
    constexpr int PAD = IS == 1 ? 1 : 0;
    constexpr int NHP = HR * HC, NPX = 2 * PW;          // 72 halo pixels, 32 output pixels | 85, 16
    constexpr int NXI = (NHP + 3) / 4, NDI = NPX / 4;   // DMA items (4 pixel rows each): 18 + 8 | 22 + 4
    constexpr int NXJ = (NXI + 3) / 4;                  // X items per wave, at most
    static_assert(NXI + NDI == 26, "26 items per stage: waves 0, 1 issue seven, waves 2, 3 six (wait_older)");
    constexpr int STAGE = (NXI + NDI) * 256;            // floats per stage
    constexpr int NST = 3;
    __shared__ __attribute__((aligned(1024))) float smem[NST * STAGE];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int mi = wave >> 1, ni = wave & 1;
    // XCD-aware block order (round 3): the (ci, co) tiles of one patch slice share their x and dY tiles; dealt round-robin
    // over the eight XCDs every tile was fetched into eight L2s (r02 PMC: 1141 MB HBM-side per launch against ~530 MB of
    // operands, L2 hit rate 0.36); remapped, a slice's tiles run on one XCD
    const Blk3 blk = xcd_block_order();
    const int ci0 = blk.x * 64, co0 = blk.y * 64;
    const int pid0 = blk.z * a.patches_per_split;
    const int pid1 = min(a.npatch, pid0 + a.patches_per_split);
    const int nstages = pid1 - pid0;

    // DMA lane mapping: one instruction = 4 pixel rows x 64 channels; lane -> (pixel l>>4, c4 = l&15)
    const int dpx = lane >> 4, c4 = lane & 15;
    const bool second = ci0 >= a.c1;
    const int ldX = second ? a.ldx2 : a.ldx;
    const int cX = ci0 + c4 * 4;
    const bool xvalid = cX < a.cin_ld;
    const int ccX = second ? cX - a.c1 : cX;
    const int coD = co0 + c4 * 4;
    const bool dvalid = coD < a.cout;
    const __amdgpu_buffer_rsrc_t rsx = second ? __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000)
                                              : __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    // items 0..17: halo rows [4i,4i+4); items 18..25: dY rows.  Wave w takes items w, w+4, ...
    // per-lane constants of the X items (j = 0..4): halo coordinates of this lane's pixel
    [[maybe_unused]] int hr[NXJ], hc[NXJ];                 // (NM: norm_x)
#pragma unroll
    for (int j = 0; j < NXJ; ++j) {
        const int hp = 4 * (wave + 4 * j) + dpx;
        hr[j] = hp / HC;
        hc[j] = hp - hr[j] * HC;
    }

    // running patch coordinate (block-uniform), in OUTPUT pixels (ho x wo = h / IS x w / IS)
    const int ho = a.h / IS, wo = a.w / IS;
    int n, pr, pc;
    {
        // Patches are numbered DOWN the columns of an image (round 4; until then along the rows): a block walks a contiguous range, and the
        // halos of vertically adjacent patches share two of their four rows (half the halo) where horizontally adjacent ones share two of
        // eighteen columns -- walking down, the shared rows were fetched one stage ago (L2 / L1 hits), walking along, 16 stages and a few
        // hundred KiB per resident block ago, i.e. from beyond L2 (r03: 1291 MB HBM-side per launch for 805 MB of operands, L2 hit 0.19)
        const int ppc = ho / 2, ppi = ppc * (wo / PW);
        const int p = pid0 < a.npatch ? pid0 : 0;
        n = p / ppi;
        const int r = p - n * ppi;
        pc = (r / ppc) * PW;
        pr = (r % ppc) * 2;
    }
    // DMA addressing, one v_add and one masked select per instruction: a lane's byte offset in item j is a per-lane constant plus the
    // patch origin, and whether its halo pixel lies outside the image depends only on which edges of the image the patch touches
    // (block-uniform, four bits) and on which edges of the halo the lane's pixel sits (per-lane constant, four bits per item; a fifth
    // marks lanes with nothing to fetch -- channel tail, tail of the last halo item -- and is always asked for).  (Until round 3 every
    // stage recomputed coordinates, range tests and exec-masked selects per item: ~450 instructions between the barrier and the
    // stage's first MFMA.)  The LDS destination is item * 1 KiB for both kinds of item (the dY rows follow the halo); descriptor and
    // origin are scalar selects.
    unsigned off0[7], bma = 0, bmb = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int item = wave + 4 * j;
        unsigned bits;
        if (item < NXI) {
            const int hp = 4 * item + dpx;
            const int r = hp / HC, c = hp - r * HC;
            off0[j] = (unsigned)((r * a.w + c) * ldX + ccX) * 4u;
            bits = !(xvalid && hp < NHP) ? 16u : ((PAD && r == 0) ? 1u : 0u) | (r == HR - 1 ? 2u : 0u) | ((PAD && c == 0) ? 4u : 0u) | (c == HC - 1 ? 8u : 0u);
        } else {
            const int q = 4 * (item - NXI) + dpx;
            off0[j] = (unsigned)(((q / PW) * (a.w / IS) + q % PW) * a.lddy + coD) * 4u;
            bits = (dvalid && item < NXI + NDI) ? 0u : 16u;
        }
        if (j < 4)
            bma |= bits << (8 * j);
        else
            bmb |= bits << (8 * (j - 4));
    }
    auto dma = [&](int stage) {
        float* sx = smem + stage * STAGE;
        const int org = (n * a.h + IS * pr - PAD) * a.w + (IS * pc - PAD);       // pixel index of halo (0,0)
        const unsigned edges = 16u | ((PAD && pr == 0) ? 1u : 0u) | (pr + 2 == ho ? 2u : 0u) | ((PAD && pc == 0) ? 4u : 0u) | (pc + PW == wo ? 8u : 0u);
        const unsigned xb = (unsigned)(org * ldX) * 4u, db = (unsigned)(((n * ho + pr) * wo + pc) * a.lddy) * 4u;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int item = wave + 4 * j;
            if (j < 6 || wave < 2) {
                const bool isx = item < NXI;                   // wave-uniform
                const unsigned out = (j < 4 ? bma : bmb) & (edges << (8 * (j & 3)));
                const unsigned off = out ? 0xffffffffu : off0[j] + (isx ? xb : db);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(isx ? rsx : rsd, (lds_ptr)(sx + item * 256), 16, (int)off, 0, 0, 0);
            }
        }
        pr += 2;
        if (pr == ho) {
            pr = 0;
            pc += PW;
            if (pc == wo) {
                pc = 0;
                ++n;
            }
        }
    };

    // NM: coordinates of the next stage to normalise (they run one stage behind dma()'s), the lane's table entries and their image
    const bool nm_on = NM && a.nt != nullptr && (int)second == a.ntpart;        // block-uniform
    [[maybe_unused]] int n2 = n, pr2 = pr, pc2 = pc, nimg = -1;
    [[maybe_unused]] f32x4 nmean = {0.f, 0.f, 0.f, 0.f}, ninv = nmean, nbeta = nmean;
    // One straight-line piece per stage (interior patches: no per-lane tests).
    const int n_blk = n;                                    // NM = 2: the sample of this block's patches
    [[maybe_unused]] auto norm_x = [&](int stage) {
        if (n2 != nimg) {                                   // block-uniform
            nimg = n2;
            if (xvalid) {
                const float* t = a.nt + (size_t)n2 * SHM_NT_PLANES * a.ntc + ccX;
                if constexpr (NM == 2) {
                    nbeta = load16_drained(t + 3 * a.ntc);              // ring
                } else {
                    nmean = load16_drained(t);
                    ninv = load16_drained(t + a.ntc);
                    nbeta = load16_drained(t + 2 * a.ntc);
                }
            }
        }
        float* sx = smem + stage * STAGE + lane * 4;
        if constexpr (NM == 2) {
            // SHM_NORM_SCALED: `ring` over the out-of-image halo entries of a border patch; nothing to do inside the image
            if (!(pr2 > 0 && pr2 + 2 < a.h && pc2 > 0 && pc2 + PW < a.w)) {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int item = wave + 4 * j;
                    if (item < 18) {
                        const int iy = pr2 - 1 + hr[j], ix = pc2 - 1 + hc[j];
                        if (xvalid && !((unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w)) *(f32x4*)(sx + item * 256) = nbeta;
                    }
                }
            }
        } else
        // interior patch (the whole 4 x 18 halo inside the image) of a full 64-channel tile: every lane of every item normalises, no
        // per-lane tests -- block-uniform, 7 of 8 patches of a 256 x 256 map
        if (pr2 > 0 && pr2 + 2 < a.h && pc2 > 0 && pc2 + PW < a.w && ci0 + 64 <= a.cin_ld) {
            f32x4 x[5];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = *(const f32x4*)(sx + (wave + 4 * j) * 256);
            if (wave < 2) x[4] = *(const f32x4*)(sx + (wave + 16) * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[j][e] = shm_in_norm(x[j][e], nmean[e], ninv[e], nbeta[e]);
                *(f32x4*)(sx + (wave + 4 * j) * 256) = x[j];
            }
            if (wave < 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[4][e] = shm_in_norm(x[4][e], nmean[e], ninv[e], nbeta[e]);
                *(f32x4*)(sx + (wave + 16) * 256) = x[4];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int item = wave + 4 * j;
                if (item < 18) {
                    const int iy = pr2 - 1 + hr[j], ix = pc2 - 1 + hc[j];
                    if (xvalid && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w) {
                        f32x4 x = *(const f32x4*)(sx + item * 256);
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = shm_in_norm(x[e], nmean[e], ninv[e], nbeta[e]);
                        *(f32x4*)(sx + item * 256) = x;
                    }
                }
            }
        }
        pr2 += 2;
        if (pr2 == a.h) {
            pr2 = 0;
            pc2 += PW;
            if (pc2 == a.w) {
                pc2 = 0;
                ++n2;
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int xl = hh * 64 * IS + mi * 32 + l31;  // + ((IS*qr+kh)*HC + IS*qc + kw)*64, qc even part
    const int dl = hh * 64 + ni * 32 + l31;       // + 2*kk*64
    auto compute = [&](int stage) {
        const float* X = smem + stage * STAGE + xl;
        const float* D = smem + stage * STAGE + NXI * 256 + dl;
#pragma unroll
        for (int kk = 0; kk < NPX / 2; ++kk) {
            const int qr = kk / (PW / 2), qc = 2 * (kk % (PW / 2));
            const float bv = D[kk * 128];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float av = X[((IS * qr + t / 3) * HC + IS * qc + t % 3) * 64];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    };

    // wait until this wave's DMA items of every stage but the youngest one in flight have landed
    auto wait_older = [&](bool younger_in_flight) {
        if (younger_in_flight) {
            if (wave < 2)
                asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    if (nstages > 0) {
        dma(0);
        if (nstages > 1) dma(1);
        if constexpr (NM)
            if (nm_on) {
                wait_older(nstages > 1);
                norm_x(0);
            }
        int cur = 0, nxt2 = 2;
        for (int s = 0; s < nstages; ++s) {
            wait_older(s + 1 < nstages);
            SHM_LDS_BARRIER();
            asm volatile("" ::: "memory");
            if (s + 2 < nstages) dma(nxt2);
            compute(cur);
            asm volatile("" ::: "memory");
            cur = (cur == NST - 1) ? 0 : cur + 1;
            nxt2 = (nxt2 == NST - 1) ? 0 : nxt2 + 1;
            // NM: stage s + 1 (issued before the stage in flight): every wave normalises its own items of it behind this stage's last
            // MFMA (issued, not finished: they and the partner block's keep the matrix pipe busy); the barrier of step s + 1 publishes
            // them.  The MFMA loop itself stays the plain kernel's: with the normalisation inside it (one piece at K step 8, or an
            // item per K step) the loop falls into basic blocks -- 47-63 s_waitcnt instead of 20, +5-7 % on the kernel even for
            // blocks that normalise nothing.  Timing-only builds: with the normalisation removed and the wait kept the kernel is as
            // fast as the plain one (+0.2-0.5 %); the pass itself costs 3-6 % -- its read / fma / write chain (~500 cycles per
            // 9216-cycle stage) is serial in every wave at the same time, and the two blocks of a CU run in lockstep.
            if constexpr (NM)
                if (nm_on && s + 1 < nstages) {
                    wait_older(s + 2 < nstages);
                    norm_x(cur);
                    asm volatile("" ::: "memory");
                }
        }
    }

    // NM = 2: the slab's rows times inv of the block's sample (row r of a lane: channel ci0 + 32 mi + (r & 3) + 8 (r >> 2) + 4 hh),
    // applied on the way out (scaling the accumulators in place made hipcc spill 100 registers)
    float sc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 1.f;
    if constexpr (NM == 2)
        if (nm_on) {
            const float* iv = a.nt + ((size_t)n_blk * SHM_NT_PLANES + 1) * a.ntc + (ci0 - (second ? a.c1 : 0)) + mi * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 s4 = *(const f32x4*)(iv + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) sc[4 * g + e] = s4[e];
            }
        }
    float* out = a.part + (size_t)blk.z * 9 * a.cin * a.cout;
    const int con = co0 + ni * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (ci < a.cin && con < a.cout) out[((size_t)t * a.cin + ci) * a.cout + con] = NM == 2 ? acc[t][r] * sc[r] : acc[t][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Thin-input variant of wgrad_halo_kernel (fp32, 3x3, 9*cin <= 96: the generator's 10-channel and the
// discriminator's 3-channel first layers).  The general kernels give every tap its own 64-row ci tile, of which
// 10 (3) rows are real (22 / 6 TFLOP/s, 2.6 ms per step).  Here the (tap, ci) pairs are PACKED into the MFMA rows --
// row r holds (tap r / cin, ci r % cin) -- which only changes the per-lane offset of the ds_read_b32 into the same
// LDS halo image (16 floats = one 64-byte row per halo pixel).  IS = conv stride: the halo of a 2 x 16 output patch
// is 4 x 18 input pixels at stride 1 (SAME pad 1 before) and 5 x 33 at stride 2 (pad 0 before).  Wave w takes column
// tile w & 1 and patch row w >> 1; the two patch rows write separate split-K slabs.
template <int NRT, int IS>
__global__ __launch_bounds__(256, 2) void wgrad_halo_thin_kernel(const WgradHaloArgs a) {
    constexpr int PW = 16, XP = 16;
    constexpr int HR = 2 * IS + 3 - IS, HC = PW * IS + 3 - IS, PAD = IS == 1 ? 1 : 0;
    constexpr int NHP = HR * HC, NPX = 2 * PW;
    constexpr int NXI = (NHP * XP * 4 + 1023) / 1024;    // DMA items (1 KiB = 16 halo pixels) for the halo; 8 more for dY
    constexpr int XF = NXI * 256;                        // halo region padded to whole items
    constexpr int STAGE = XF + NPX * 64;                 // floats
    constexpr int NST = 3;
    constexpr int NIT = NXI + 8, CHI = (NIT + 3) / 4, CLO = NIT / 4, NHI = NIT % 4;   // items per wave: CHI for waves < NHI
    __shared__ __attribute__((aligned(1024))) float smem[NST * STAGE];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int ni = wave & 1, qr = wave >> 1;
    const int co0 = blockIdx.y * 64;
    const int pid0 = blockIdx.z * a.patches_per_split;
    const int pid1 = min(a.npatch, pid0 + a.patches_per_split);
    const int nstages = pid1 - pid0;
    const int ho = a.h / IS, wo = a.w / IS;

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    // X items: lane -> (halo pixel 16 i + (l >> 2), 4-float chunk l & 3); D items: lane -> (pixel 4 j + (l >> 4), chunk l & 15)
    const int xpx = lane >> 2, xch = lane & 3;
    const bool xcv = xch * 4 < a.cin_ld;
    const int dpx = lane >> 4, dch = lane & 15;
    const int coD = co0 + dch * 4;
    const bool dvalid = coD < a.cout;

    int n, pr, pc;                                       // patch origin in OUTPUT pixels
    {
        const int ppr = wo / PW, ppi = (ho / 2) * ppr;
        const int p = pid0 < a.npatch ? pid0 : 0;
        n = p / ppi;
        const int r = p - n * ppi;
        pr = (r / ppr) * 2;
        pc = (r % ppr) * PW;
    }
    auto dma = [&](int stage) {
        float* sx = smem + stage * STAGE;
        float* sd = sx + XF;
        const int y0 = IS * pr - PAD, x0 = IS * pc - PAD;       // input pixel of halo (0,0)
#pragma unroll
        for (int j = 0; j < CHI; ++j) {
            const int item = wave + 4 * j;
            if (item < NXI) {
                const int hp = 16 * item + xpx;
                const int hr = hp / HC, hc = hp - hr * HC;
                const int iy = y0 + hr, ix = x0 + hc;
                const bool v = xcv && hp < NHP && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
                const unsigned off = v ? (unsigned)(((n * a.h + iy) * a.w + ix) * a.ldx + xch * 4) * 4u : 0xffffffffu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sx + item * 256), 16, (int)off, 0, 0, 0);
            } else if (item < NIT) {
                const int q = 4 * (item - NXI) + dpx;
                const int oy = pr + (q >> 4), ox = pc + (q & 15);
                const unsigned off = dvalid ? (unsigned)(((n * ho + oy) * wo + ox) * a.lddy + coD) * 4u : 0xffffffffu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (lds_ptr)(sd + (item - NXI) * 256), 16, (int)off, 0, 0, 0);
            }
        }
        pc += PW;
        if (pc == wo) {
            pc = 0;
            pr += 2;
            if (pr == ho) {
                pr = 0;
                ++n;
            }
        }
    };

    f32x16 acc[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // packed rows: lane's row of row-tile rt is (tap, ci) = divmod(rt*32 + l31, cin); rows >= 9*cin read a valid
    // address and are never stored
    const int rows = 9 * a.cin;
    int xoff[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        const int idx = rt * 32 + l31;
        const int tap = idx < rows ? idx / a.cin : 0, ci = idx < rows ? idx - tap * a.cin : 0;
        xoff[rt] = ((tap / 3) * HC + tap % 3) * XP + ci;
    }
    const int xb = (IS * qr * HC + IS * hh) * XP;         // + IS*2*kk*XP
    const int db = (qr * PW + hh) * 64 + ni * 32 + l31;   // + 2*kk*64
    auto compute = [&](int stage) {
        const float* X = smem + stage * STAGE + xb;
        const float* D = smem + stage * STAGE + XF + db;
#pragma unroll
        for (int kk = 0; kk < PW / 2; ++kk) {
            const float bv = D[kk * 128];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const float av = X[kk * 2 * IS * XP + xoff[rt]];
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rt], 0, 0, 0);
            }
        }
    };

    if (nstages > 0) {
        dma(0);
        if (nstages > 1) dma(1);
        int cur = 0, nxt2 = 2;
        for (int s = 0; s < nstages; ++s) {
            if (s + 1 < nstages) {                 // one younger stage in flight: CHI or CLO DMA instructions of this wave
                if (NHI != 0 && wave < NHI)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CHI) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CLO) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            SHM_LDS_BARRIER();
            asm volatile("" ::: "memory");
            if (s + 2 < nstages) dma(nxt2);
            compute(cur);
            asm volatile("" ::: "memory");
            cur = (cur == NST - 1) ? 0 : cur + 1;
            nxt2 = (nxt2 == NST - 1) ? 0 : nxt2 + 1;
        }
    }

    float* out = a.part + ((size_t)blockIdx.z * 2 + qr) * 9 * a.cin * a.cout;
    const int con = co0 + ni * 32 + l31;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int idx = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;       // = tap*cin + ci
            if (idx < rows && con < a.cout) out[(size_t)idx * a.cout + con] = acc[rt][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 version of the halo-patch weight gradient (3x3, stride 1): the LDS image of wgrad_halo_kernel in
// the row format of wgrad_bf16_kernel.  Stage = 2 x 16 output pixels: a 4 x 20 halo image (18 columns
// used; the pitch of 20 keeps bit 1 of the row index independent of the tap's row offset, so one
// swizzled address per kw serves all taps through immediates) of 128-byte rows (64 channels) and 32 dY
// rows, brought in by LDS-DMA with the half-swap swizzle applied on the source side; two K steps of 16
// pixels, nine v_mfma_f32_32x32x16_bf16 each, operands via ds_read_b64_tr_b16.
// R = pixel rows per stage (2 or 4).  The x fragment of tap row kh at K step (pixel row) q is the fragment of tap row 0 at
// q + kh, so a stage of R rows needs (R + 2) x 3 fragment reads for 9 R MFMAs (hipcc keeps the shared ones in registers):
// R = 4 reads 22 fragments per 36 MFMAs where two R = 2 stages read 28, with half the barriers and 3/4 of the halo bytes.
// NM: as in wgrad_halo_kernel (a lane's eight channels are the same for every item and patch: 24 table registers).
template <int R, int NM = 0>
__global__ __launch_bounds__(256, 2) void wgrad_halo_bf16_kernel(const WgradHaloArgs a) {
    constexpr int PW = 16, HP = 20;                     // patch R x 16; halo R + 2 rows, LDS pitch 20 (18 valid)
    constexpr int NHR = (R + 2) * HP, NPX = R * PW;     // R = 2: 80 halo rows, 32 dY rows (14 KiB); R = 4: 120 + 64 (23 KiB)
    constexpr int STAGE = (NHR + NPX) * 64;             // bf16 elements per stage
    constexpr int NST = 3;
    constexpr int NXI = NHR / 8, NDI = NPX / 8;         // DMA items (8 rows of 128 B each): 10 + 4 / 15 + 8
    constexpr int NIT = NXI + NDI, NJ = (NIT + 3) / 4;  // items per wave: waves below NIT % 4 (or all) take NJ, the others NJ - 1
    constexpr int NXJ = (NXI + 3) / 4;                  // halo items per wave (at most)
    extern __shared__ __attribute__((aligned(1024))) unsigned short smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int mi = wave >> 1, ni = wave & 1;
    const Blk3 blk = xcd_block_order();
    const int ci0 = blk.x * 64, co0 = blk.y * 64;
    const int pid0 = blk.z * a.patches_per_split;
    const int pid1 = min(a.npatch, pid0 + a.patches_per_split);
    const int nstages = pid1 - pid0;

    // DMA lane mapping: lane -> (row l>>3 of the item, 16-byte chunk l&7); LDS chunk j of row r holds source
    // chunk j ^ (4 * bit1(r)).  Items are 8 rows, so bit1(r) = bit1(l>>3).
    const int drow = lane >> 3;
    const int sch = (lane & 7) ^ (((drow >> 1) & 1) << 2);
    const bool second = ci0 >= a.c1;
    const int ldX = second ? a.ldx2 : a.ldx;
    const int cX = ci0 + sch * 8;
    const bool xvalid = cX < a.cin_ld;
    const int ccX = second ? cX - a.c1 : cX;
    const int coD = co0 + sch * 8;
    const bool dvalid = coD < a.cout;
    // descriptors as words: the DMA is issued as inline asm (common.h, shm_dma16)
    const shm_u32x4 rsx = second ? shm_rsrc_words(a.x2, a.x2bytes) : shm_rsrc_words(a.x, a.xbytes);
    const shm_u32x4 rsd = shm_rsrc_words(a.dy, a.dybytes);
    // items 0..9: halo rows [8i, 8i+8); items 10..13: dY rows.  Wave w takes items w, w+4, w+8, w+12.
    [[maybe_unused]] int hr[NXJ], hc[NXJ];                 // (NM: norm_x)
#pragma unroll
    for (int j = 0; j < NXJ; ++j) {
        const int hp = 8 * (wave + 4 * j) + drow;
        hr[j] = hp / HP;
        hc[j] = hp - hr[j] * HP;
    }

    int n, pr, pc;
    {
        const int ppc = a.h / R, ppi = ppc * (a.w / PW);           // patches numbered down the columns of an image, see wgrad_halo_kernel
        const int p = pid0 < a.npatch ? pid0 : 0;
        n = p / ppi;
        const int r = p - n * ppi;
        pc = (r / ppc) * PW;
        pr = (r % ppc) * R;
    }
    // DMA addressing as in wgrad_halo_kernel: per-lane constant offset + patch origin, edge bits (five per item, one register)
    static_assert(NJ <= 6, "five mask bits per item in one register");
    unsigned off0[NJ], bm = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int item = wave + 4 * j;
        unsigned bits;
        if (item < NXI) {
            const int hp = 8 * item + drow;
            const int r_ = hp / HP, c_ = hp - r_ * HP;
            off0[j] = (unsigned)((r_ * a.w + c_) * ldX + ccX) * 2u;
            bits = !(xvalid && c_ < PW + 2) ? 16u : (r_ == 0 ? 1u : 0u) | (r_ == R + 1 ? 2u : 0u) | (c_ == 0 ? 4u : 0u) | (c_ == PW + 1 ? 8u : 0u);
        } else {
            const int q = 8 * (item - NXI) + drow;
            off0[j] = (unsigned)(((q >> 4) * a.w + (q & 15)) * a.lddy + coD) * 2u;
            bits = (dvalid && item < NIT) ? 0u : 16u;
        }
        bm |= bits << (5 * j);
    }
    auto dma = [&](int stage) {
        unsigned short* sx = smem + stage * STAGE;
        const int org = (n * a.h + pr - 1) * a.w + (pc - 1);       // pixel index of halo (0,0)
        const unsigned edges = 16u | (pr == 0 ? 1u : 0u) | (pr + R == a.h ? 2u : 0u) | (pc == 0 ? 4u : 0u) | (pc + PW == a.w ? 8u : 0u);
        const unsigned xb = (unsigned)(org * ldX) * 2u, db = (unsigned)(((n * a.h + pr) * a.w + pc) * a.lddy) * 2u;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int item = wave + 4 * j;
            if (j < NJ - 1 || item < NIT) {
                const bool isx = item < NXI;                   // wave-uniform
                const unsigned off = (bm & (edges << (5 * j))) ? 0xffffffffu : off0[j] + (isx ? xb : db);
                shm_dma16(isx ? rsx : rsd, shm_lds_addr(sx + item * 512), off);
            }
        }
        pr += R;
        if (pr == a.h) {
            pr = 0;
            pc += PW;
            if (pc == a.w) {
                pc = 0;
                ++n;
            }
        }
    };

    const bool nm_on = NM && a.nt != nullptr && (int)second == a.ntpart;        // block-uniform
    [[maybe_unused]] int n2 = n, pr2 = pr, pc2 = pc, nimg = -1;
    [[maybe_unused]] f32x4 nmean[2] = {}, ninv[2] = {}, nbeta[2] = {};
    const int n_blk = n;                                    // NM = 2: the sample of this block's patches
    [[maybe_unused]] auto norm_x = [&](int stage) {
        if (n2 != nimg) {                                   // block-uniform
            nimg = n2;
            if (xvalid) {
                const float* t = a.nt + (size_t)n2 * SHM_NT_PLANES * a.ntc + ccX;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    if constexpr (NM == 2) {
                        nbeta[hf] = load16_drained(t + 3 * a.ntc + 4 * hf);        // ring
                    } else {
                        nmean[hf] = load16_drained(t + 4 * hf);
                        ninv[hf] = load16_drained(t + a.ntc + 4 * hf);
                        nbeta[hf] = load16_drained(t + 2 * a.ntc + 4 * hf);
                    }
                }
            }
        }
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        unsigned short* sx = smem + stage * STAGE + lane * 8;
        if constexpr (NM == 2) {
            // SHM_NORM_SCALED: `ring` over the out-of-image halo entries of a border patch (the two dummy columns of the pitch stay zero)
            if (!(pr2 > 0 && pr2 + R < a.h && pc2 > 0 && pc2 + PW < a.w)) {
                u32x4_t rg;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        rg[2 * hf + e] = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)nbeta[hf][2 * e]) |
                                         ((unsigned)__builtin_bit_cast(unsigned short, (bf16_t)nbeta[hf][2 * e + 1]) << 16);
#pragma unroll
                for (int j = 0; j < NXJ; ++j) {
                    const int item = wave + 4 * j;
                    if (item < NXI) {
                        const int iy = pr2 - 1 + hr[j], ix = pc2 - 1 + hc[j];
                        if (xvalid && hc[j] < PW + 2 && !((unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w)) *(u32x4_t*)(sx + item * 512) = rg;
                    }
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            const int item = wave + 4 * j;
            if (item < NXI) {
                const int iy = pr2 - 1 + hr[j], ix = pc2 - 1 + hc[j];
                if (xvalid && hc[j] < PW + 2 && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w) {
                    u32x4_t x = *(const u32x4_t*)(sx + item * 512);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const unsigned u = x[2 * hf + e];
                            const bf16_t lo = (bf16_t)shm_in_norm(__uint_as_float(u << 16), nmean[hf][2 * e], ninv[hf][2 * e], nbeta[hf][2 * e]);
                            const bf16_t hi = (bf16_t)shm_in_norm(__uint_as_float(u & 0xffff0000u), nmean[hf][2 * e + 1], ninv[hf][2 * e + 1], nbeta[hf][2 * e + 1]);
                            x[2 * hf + e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
                        }
                    *(u32x4_t*)(sx + item * 512) = x;
                }
            }
        }
        }
        pr2 += R;
        if (pr2 == a.h) {
            pr2 = 0;
            pc2 += PW;
            if (pc2 == a.w) {
                pc2 = 0;
                ++n2;
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read addresses (elements): lane supplies row 8hh + (i>>2) [+4 for the second read] and channels
    // [32*tile + 16*(g&1) + 4*(i&3), +4); tap (kh,kw) and K step qr enter as immediates, except that kw shifts
    // the row and with it bit 1 of the row index -> one address per kw
    const int fq = 8 * hh + ((lane & 15) >> 2);
    const int fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    int fa[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int row = fq + kw;
        fa[kw] = row * 64 + ((mi * 32 + fcol) ^ (((row >> 1) & 1) << 5));
    }
    const int fb = fq * 64 + ((ni * 32 + fcol) ^ (((fq >> 1) & 1) << 5));
    auto compute = [&](int stage) {
        const unsigned short* X = smem + stage * STAGE;
        const unsigned short* D = X + NHR * 64;
#pragma unroll
        for (int qr = 0; qr < R; ++qr) {
            const bf16x8 bv = tr_frag(D + fb + qr * PW * 64);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bf16x8 av = tr_frag(X + fa[t % 3] + (qr + t / 3) * HP * 64);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
            }
        }
    };

    // wait until this wave's DMA items of every stage but the youngest one in flight (NJ or NJ - 1 instructions) have landed
    auto wait_older = [&](bool younger_in_flight) {
        if (younger_in_flight) {
            if (NIT % 4 == 0 || wave < NIT % 4)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    if (nstages > 0) {
        dma(0);
        if (nstages > 1) dma(1);
        if constexpr (NM)
            if (nm_on) {
                wait_older(nstages > 1);
                norm_x(0);
            }
        int cur = 0, nxt2 = 2;
        for (int s = 0; s < nstages; ++s) {
            wait_older(s + 1 < nstages);
            SHM_LDS_BARRIER();
            asm volatile("" ::: "memory");
            if (s + 2 < nstages) dma(nxt2);
            compute(cur);
            asm volatile("" ::: "memory");
            cur = (cur == NST - 1) ? 0 : cur + 1;
            nxt2 = (nxt2 == NST - 1) ? 0 : nxt2 + 1;
            // NM: see wgrad_halo_kernel
            if constexpr (NM)
                if (nm_on && s + 1 < nstages) {
                    wait_older(s + 2 < nstages);
                    norm_x(cur);
                    asm volatile("" ::: "memory");
                }
        }
    }

    // NM = 2: the slab's rows times inv of the block's sample (see wgrad_halo_kernel)
    float sc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 1.f;
    if constexpr (NM == 2)
        if (nm_on) {
            const float* iv = a.nt + ((size_t)n_blk * SHM_NT_PLANES + 1) * a.ntc + (ci0 - (second ? a.c1 : 0)) + mi * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 s4 = *(const f32x4*)(iv + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) sc[4 * g + e] = s4[e];
            }
        }
    float* out = a.part + (size_t)blk.z * 9 * a.cin * a.cout;
    const int con = co0 + ni * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (ci < a.cin && con < a.cout) out[((size_t)t * a.cin + ci) * a.cout + con] = NM == 2 ? acc[t][r] * sc[r] : acc[t][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Round 4: the bf16 halo weight gradient as an EIGHT-wave block that owns 64 input channels x 128 output channels -- two
// 64 x 64 tiles that share one x halo image -- for unit stride (S2 = false: stages of 4 x 16 output pixels, the LDS image of
// wgrad_halo_bf16_kernel<4>) and for stride 2 (S2 = true: the 3x3 / stride-2 convolutions and, with the roles of x and dY swapped,
// Conv2DTranspose; stages of 2 x 16 output pixels).
//
// Why: with the split-K target the two-stream step wants (256 blocks: every slab is 9 * cin * cout floats written and read again,
// 45-90 % of the operand bytes on the deep layers) the four-wave kernel runs ONE block = one wave per SIMD on a CU, and a wave's
// DMA issue, fragment reads and MFMAs are then serial (MFMA busy 0.34 against 0.56 with two blocks per CU).  Here a CU holds two
// waves per SIMD at the same number of slabs, and the two co tiles share the x halo -- the larger part of a stage (15 of 23 KiB):
// 31 DMA items per 72 wave-MFMAs instead of 46, and x is fetched from L2 / HBM once for 128 output channels.
//
// Stride 2 (SAME padding of an even map: nothing before the first row / column, one after the last): output pixel (qr, qc), tap
// (kh, kw) reads input pixel (2 qr + kh, 2 qc + kw).  A K step is 16 consecutive output pixels of one row, i.e. input columns
// 2 k + kw: the halo image therefore keeps the EVEN and ODD input columns of a halo row as two runs of consecutive LDS rows
// ([17 even | 3 unused | 16 odd] = 36 rows of 128 B per halo row; the DMA source address is per lane, so the order of the LDS rows is
// free).  Tap column kw = 0 / 1 / 2 is then run (even, k) / (odd, k) / (even, k + 1): sixteen consecutive rows, exactly the access of
// the unit-stride image (same half-swap swizzle on bit 1 of the row index, no bank conflicts), and because every offset between taps
// and K steps is a multiple of four rows, (even, k) and (odd, k) share one swizzled address register.  5 halo rows x 36 = 180 rows
// (23 items) + 2 x 32 dY rows (8 items): the same 31 items and 31 KiB per stage as the unit-stride form, with 18 MFMAs per wave.
// Against wgrad_bf16_kernel<9> (nine shifted tiles through registers and ds_write, a barrier per 16 pixels, MFMA busy 0.18): 5.2
// input pixels fetched per output pixel instead of 9, no VGPR staging, a barrier per 32 pixels.
// MODE 0: unit stride, stages of 4 x 16 pixels.  MODE 1: stride 2, stages of 2 x 16 output pixels (5 x 33 halo).  MODE 2: stride 2 on maps
// whose output width is only a multiple of 8 (the discriminator's last layer, 16 x 16 -> 8 x 8): stages of 4 x 8 output pixels, 9 x 17 halo
// stored as [9 even | 3 unused | 8 odd] = 20 LDS rows per halo row -- the same 180 rows; a K step is two output rows of eight pixels, so the
// lane half hh of a fragment sits two halo rows (40 LDS rows) further down instead of eight plane entries further on.
template <int MODE>
__global__ __launch_bounds__(512, 2) void wgrad_halo8_bf16_kernel(const WgradHaloArgs a) {
    constexpr bool S2 = MODE != 0;
    constexpr int R = MODE == 1 ? 2 : 4;                // output rows per stage
    constexpr int PW = MODE == 2 ? 8 : 16;
    constexpr int HP = MODE == 2 ? 12 : 20;             // S1: LDS pitch of a halo row (18 valid); S2: pitch of the even run
    constexpr int HRP = MODE == 1 ? 36 : 20;            // LDS rows per halo row
    constexpr int NHROW = S2 ? 2 * R + 1 : R + 2;       // halo rows: 6 | 5 | 9
    constexpr int NHR = NHROW * HRP;                    // 120 | 180 | 180
    constexpr int KSTEPS = R * PW / 16;                 // K steps (16 output pixels) per stage
    constexpr int NXI = (NHR + 7) / 8, NDT = R * PW / 8;       // x items 23 | 15, dY items per co tile 4 | 8
    constexpr int NIT = NXI + 2 * NDT;                  // 31 | 31
    static_assert(NIT == 31, "31 items per stage: waves 0-6 issue four, wave 7 three");
    constexpr int NJ = 4;
    constexpr int XROWS = NXI * 8, DROWS = NDT * 8;     // LDS rows of the x region, of one dY tile
    constexpr int STAGE = (XROWS + 2 * DROWS) * 64;     // bf16 elements: 248 rows of 128 B
    constexpr int NST = 3;
    extern __shared__ __attribute__((aligned(1024))) unsigned short smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int cot = wave >> 2, mi = (wave >> 1) & 1, ni = wave & 1;       // co tile of the pair, 32 x 32 sub-tile
    const Blk3 blk = xcd_block_order();
    const int ci0 = blk.x * 64, co0 = blk.y * 128;
    const int pid0 = blk.z * a.patches_per_split;
    const int pid1 = min(a.npatch, pid0 + a.patches_per_split);
    const int nstages = pid1 - pid0;
    const int ho = S2 ? a.h / 2 : a.h, wo = S2 ? a.w / 2 : a.w;

    // DMA lane mapping: lane -> (row l >> 3 of the item, 16-byte chunk l & 7); LDS chunk j of row r holds source chunk j ^ (4 * bit1(r))
    const int drow = lane >> 3;
    const int sch = (lane & 7) ^ (((drow >> 1) & 1) << 2);
    const bool second = ci0 >= a.c1;
    const int ldX = second ? a.ldx2 : a.ldx;
    const int cX = ci0 + sch * 8;
    const bool xvalid = cX < a.cin_ld;
    const int ccX = second ? cX - a.c1 : cX;
    // descriptors as words: the DMA is issued as inline asm (common.h, shm_dma16)
    const shm_u32x4 rsx = second ? shm_rsrc_words(a.x2, a.x2bytes) : shm_rsrc_words(a.x, a.xbytes);
    const shm_u32x4 rsd = shm_rsrc_words(a.dy, a.dybytes);

    int n, pr, pc;                                      // patch origin in OUTPUT pixels
    {
        const int ppc = ho / R, ppi = ppc * (wo / PW);             // patches numbered down the columns of an image, see wgrad_halo_kernel
        const int p = pid0 < a.npatch ? pid0 : 0;
        n = p / ppi;
        const int r = p - n * ppi;
        pc = (r / ppc) * PW;
        pr = (r % ppc) * R;
    }
    // per-lane constants of this wave's items (item = wave + 8 j): byte offset inside the halo / patch, and five mask bits -- which
    // edges of the halo the lane's pixel sits on (1 top, 2 bottom, 4 left, 8 right) and 16 for lanes with nothing to fetch
    unsigned off0[NJ], bm = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int item = wave + 8 * j;
        unsigned bits;
        if (item < NXI) {
            const int row = 8 * item + drow;
            int r_, c_;
            bool ok;
            if constexpr (S2) {
                r_ = row / HRP;
                const int t = row - r_ * HRP;
                c_ = t < HP ? 2 * t : 2 * (t - HP) + 1;
                ok = row < NHR && (t < HP ? t <= PW : true);
                bits = (r_ == NHROW - 1 ? 2u : 0u) | (c_ == 2 * PW ? 8u : 0u);
            } else {
                r_ = row / HP;
                c_ = row - r_ * HP;
                ok = c_ < PW + 2;
                bits = (r_ == 0 ? 1u : 0u) | (r_ == R + 1 ? 2u : 0u) | (c_ == 0 ? 4u : 0u) | (c_ == PW + 1 ? 8u : 0u);
            }
            off0[j] = (unsigned)((r_ * a.w + c_) * ldX + ccX) * 2u;
            if (!(xvalid && ok)) bits = 16u;
        } else {
            const int d = item - NXI, tile = d / NDT;
            const int q = 8 * (d - tile * NDT) + drow;
            const int coD = co0 + 64 * tile + sch * 8;
            off0[j] = (unsigned)(((q / PW) * wo + (q % PW)) * a.lddy + coD) * 2u;
            bits = (coD < a.cout && item < NIT) ? 0u : 16u;
        }
        bm |= bits << (5 * j);
    }
    auto dma = [&](int stage) {
        unsigned short* sx = smem + stage * STAGE;
        const int org = S2 ? (n * a.h + 2 * pr) * a.w + 2 * pc : (n * a.h + pr - 1) * a.w + (pc - 1);       // input pixel of halo (0, 0)
        const unsigned edges = 16u | ((!S2 && pr == 0) ? 1u : 0u) | (pr + R == ho ? 2u : 0u) | ((!S2 && pc == 0) ? 4u : 0u) | (pc + PW == wo ? 8u : 0u);
        const unsigned xb = (unsigned)(org * ldX) * 2u, db = (unsigned)(((n * ho + pr) * wo + pc) * a.lddy) * 2u;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int item = wave + 8 * j;
            if (j < NJ - 1 || item < NIT) {
                const bool isx = item < NXI;                   // wave-uniform
                const unsigned off = (bm & (edges << (5 * j))) ? 0xffffffffu : off0[j] + (isx ? xb : db);
                shm_dma16(isx ? rsx : rsd, shm_lds_addr(sx + item * 512), off);
            }
        }
        pr += R;
        if (pr == ho) {
            pr = 0;
            pc += PW;
            if (pc == wo) {
                pc = 0;
                ++n;
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read addresses (elements), as in wgrad_halo_bf16_kernel: the lane supplies row fq [+4 for the second read] and four
    // channels of its 16-channel block.  S1: one address per kw (kw shifts the row, and with it bit 1 of the row index); S2: kw = 0 and
    // kw = 1 are 20 rows apart (same bit 1: an immediate), kw = 2 is one row on
    const int fq = 8 * hh + ((lane & 15) >> 2);                          // pixel of the K step this lane supplies (dY rows are in pixel order)
    const int fqx = (MODE == 2 ? 2 * HRP * hh : 8 * hh) + ((lane & 15) >> 2);     // ... and its row in the x image (every term but the last is 0 mod 4)
    const int fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    int fa[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int row = fqx + (S2 ? (kw == 2 ? 1 : 0) : kw);
        fa[kw] = row * 64 + ((mi * 32 + fcol) ^ (((row >> 1) & 1) << 5)) + ((S2 && kw == 1) ? HP * 64 : 0);
    }
    const int fb = fq * 64 + ((ni * 32 + fcol) ^ (((fq >> 1) & 1) << 5)) + (XROWS + cot * DROWS) * 64;
    auto compute = [&](int stage) {
        const unsigned short* X = smem + stage * STAGE;
#pragma unroll
        for (int qr = 0; qr < KSTEPS; ++qr) {
            const bf16x8 bv = tr_frag(X + fb + qr * 16 * 64);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hrow = MODE == 0 ? qr + t / 3 : MODE == 1 ? 2 * qr + t / 3 : 4 * qr + t / 3;
                const bf16x8 av = tr_frag(X + fa[t % 3] + hrow * HRP * 64);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
            }
        }
    };

    // wait until this wave's DMA items of every stage but the youngest one in flight have landed (four items per stage; wave 7: three)
    auto wait_older = [&](bool younger_in_flight) {
        if (younger_in_flight) {
            if (wave < 7)
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    if (nstages > 0) {
        dma(0);
        if (nstages > 1) dma(1);
        int cur = 0, nxt2 = 2;
        for (int s = 0; s < nstages; ++s) {
            wait_older(s + 1 < nstages);
            SHM_LDS_BARRIER();
            asm volatile("" ::: "memory");
            if (s + 2 < nstages) dma(nxt2);
            compute(cur);
            asm volatile("" ::: "memory");
            cur = (cur == NST - 1) ? 0 : cur + 1;
            nxt2 = (nxt2 == NST - 1) ? 0 : nxt2 + 1;
        }
    }

    float* out = a.part + (size_t)blk.z * 9 * a.cin * a.cout;
    const int con = co0 + 64 * cot + ni * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (ci < a.cin && con < a.cout) out[((size_t)t * a.cin + ci) * a.cout + con] = acc[t][r];
        }
    }
}

// dw[i] (+)= sum_k part[k][i], summed in a fixed order (4 interleaved chains, then 0+1+2+3).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, size_t n, int nsplit, int accumulate) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + e;
    float s = 0.f;
    if (i < n)
        for (int k = g; k < nsplit; k += 4) s += part[(size_t)k * n + i];
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        dw[i] = accumulate ? dw[i] + t : t;
    }
}

// 16 bytes per lane and SIXTEEN interleaved chains per element group (slab k goes to chain k % 16, four loads in flight per
// chain, chains combined as a fixed tree): a reduce is a chain of dependent slab loads -- with four chains and up to 1024 slabs
// every launch took 13 (bf16) to 26 us (fp32) whatever its size, 49 launches per step.  Deterministic (fixed order); the order
// differs from wgrad_reduce_kernel's, which keeps serving element counts that are not a multiple of four.  n % 4 == 0.
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ part, float* __restrict__ dw, size_t n4, int nsplit, int accumulate) {
    __shared__ f32x4 red[16][16];
    const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + e;               // group of four elements
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        const f32x4* p = (const f32x4*)part + i;
        int k = g;
        for (; k + 48 < nsplit; k += 64) {
            const f32x4 a = p[(size_t)k * n4], b = p[(size_t)(k + 16) * n4], c = p[(size_t)(k + 32) * n4], d = p[(size_t)(k + 48) * n4];
            s += a;
            s += b;
            s += c;
            s += d;
        }
        for (; k < nsplit; k += 16) s += p[(size_t)k * n4];
    }
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
        f32x4 t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = red[2 * q][e] + red[2 * q + 1][e];
        const f32x4 r = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        f32x4* o = (f32x4*)dw + i;
        *o = accumulate ? *o + r : r;
    }
}

static int wgrad_splits(int batch, int ho, int wo, int cin, int cout, int esz = 4) {
    // two 256-thread blocks fit per CU (LDS): two rounds of 512 blocks keep every CU busy and
    // the split-K slab traffic (ns * 9*cin*cout floats written + read) small
    long M = (long)batch * ho * wo;
    int tiles = shm_cdiv(cin, 64) * shm_cdiv(cout, 64);
    // (bf16: the MFMA kernel is ~6x faster, so the slab traffic of the split weighs more and the launches, which run on the
    // second stream beside the input-gradient chain, should leave that chain room: 256 blocks measured best with the four-row
    // stages -- step 25.8 ms at 512, 24.9 at 256, 25.3 at 384, 26.1 at 192, 29.7 at 128; fp32: 1024 (122.3 ms; 512: 123.3, 2048: 122.9))
    const int target_tuned = shm_tune(SHM_TUNE_WGRAD_BLOCKS);
    const int target = target_tuned ? target_tuned : (esz == 2 ? 256 : 1024);
    int want = shm_cdiv(target, tiles);
    long maxs = (M + 255) / 256;                 // at least 256 pixels per split
    if (want > maxs) want = (int)maxs;
    if (want < 1) want = 1;
    return want;
}

extern "C" size_t shm_conv2d_wgrad_workspace(int batch, int ho, int wo, int cin, int cout, int ksize) {
    int ns = wgrad_splits(batch, ho, wo, cin, cout);
    if (9 * cin <= 96) ns *= 2;                          // wgrad_halo_thin_kernel writes two slabs per split
    if (9 * cin <= 32 && ns < 1024) ns = 1024;           // conv3x3s2_rgb_wgrad_kernel (conv_rgb.hip): one slab per block, streaming -- blocks are what it needs
    if (shm_tune(SHM_TUNE_WGRAD_BLOCKS)) ns *= 2;        // wgrad_halo8_bf16_kernel: half as many (ci, co) tiles, twice the splits for a given block target
    return (size_t)ns * ksize * ksize * cin * cout * sizeof(float);
}

// norm request of shm_conv2d_wgrad_norm around its launch (WgradHaloArgs::nt); query = shm_conv2d_wgrad_norm_supported's dry run
struct WNormReq {
    const float* nt;
    int part, c, mode;
    bool query, query_ok;
};

// SHM_NORM_SCALED: a block's patches must lie in one sample -- the largest divisor of the patches per image that does not exceed
// the split the automatic choice would take
static int wgrad_norm_aligned_pps(int pps, int ppi) {
    int d = pps > ppi ? ppi : pps;
    if (d < 1) d = 1;
    while (ppi % d) --d;
    return d;
}
static thread_local WNormReq g_wnorm = {};

// Phase 1 of shm_conv2d_wgrad: the MFMA kernel; *nsplit_out receives the number of partial slabs written.
extern "C" int shm_conv2d_wgrad_partial(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* dy,
                                        int lddy, int batch, int hi, int wi, int cin, int cin_ld, int cout,
                                        int ksize, int stride, void* workspace, size_t ws_bytes, int dtype,
                                        int* nsplit_out, void* stream) {
    SHM_REQUIRE(dtype == SHM_F32 || dtype == SHM_BF16, SHM_E_DTYPE, "shm_conv2d_wgrad: dtype %d not in {SHM_F32, SHM_BF16}", dtype);
    const int esz = dtype == SHM_BF16 ? 2 : 4, vec = 16 / esz;      // 16-byte loads: 4 floats / 8 bf16
    SHM_REQUIRE(ksize == 1 || ksize == 3, SHM_E_SHAPE, "shm_conv2d_wgrad: ksize %d not in {1,3}", ksize);
    SHM_REQUIRE(stride == 1 || stride == 2, SHM_E_SHAPE, "shm_conv2d_wgrad: stride %d not in {1,2}", stride);
    SHM_REQUIRE(x && dy && workspace, SHM_E_SHAPE, "shm_conv2d_wgrad: null pointer");
    SHM_REQUIRE(cin_ld % vec == 0 && cin_ld >= cin && cout % vec == 0, SHM_E_SHAPE,
                "shm_conv2d_wgrad: cin_ld %d / cout %d must be multiples of %d", cin_ld, cout, vec);
    SHM_REQUIRE(ldx % vec == 0 && lddy % vec == 0 && (!x2 || (ldx2 % vec == 0 && c1 % vec == 0)), SHM_E_SHAPE,
                "shm_conv2d_wgrad: pitches must be multiples of %d", vec);
    int ho, wo, pt, pl;
    shm_same_pad(hi, ksize, stride, &ho, &pt);
    shm_same_pad(wi, ksize, stride, &wo, &pl);
    SHM_REQUIRE((size_t)batch * hi * wi < (1u << 31), SHM_E_SHAPE, "shm_conv2d_wgrad: pixel count overflows int32");
    WgradArgs a{};
    a.x = x;
    a.x2 = x2;
    a.c1 = x2 ? c1 : cin_ld;
    a.ldx = ldx;
    a.ldx2 = ldx2;
    a.dy = dy;
    a.lddy = lddy;
    a.part = (float*)workspace;
    a.hi = hi;
    a.wi = wi;
    a.ho = ho;
    a.wo = wo;
    a.cin_ld = cin_ld;
    a.cin = cin;
    a.cout = cout;
    a.is = stride;
    a.ntaps = ksize * ksize;
    for (int kh = 0; kh < ksize; ++kh)
        for (int kw = 0; kw < ksize; ++kw) {
            a.dh[kh * ksize + kw] = kh - pt;
            a.dw[kh * ksize + kw] = kw - pl;
        }
    a.M = batch * ho * wo;
    int ns = wgrad_splits(batch, ho, wo, cin, cout, esz);
    size_t need = (size_t)ns * a.ntaps * cin * cout * sizeof(float);
    SHM_REQUIRE(ws_bytes >= need, SHM_E_WORKSPACE, "shm_conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    {
        const size_t lim = 0xfffffff0ull;
        size_t xb = (size_t)batch * hi * wi * ldx * esz, x2b = x2 ? (size_t)batch * hi * wi * ldx2 * esz : 0;
        size_t db = (size_t)batch * ho * wo * lddy * esz;
        SHM_REQUIRE(xb < lim && x2b < lim && db < lim, SHM_E_SHAPE, "shm_conv2d_wgrad: operand larger than 4 GiB (32-bit buffer offsets)");
        a.xbytes = (unsigned)xb;
        a.x2bytes = (unsigned)x2b;
        a.dybytes = (unsigned)db;
    }
    int pps = shm_cdiv(a.M, ns);
    pps = (pps + 15) / 16 * 16;
    ns = shm_cdiv(a.M, pps);
    a.pix_per_split = pps;
    hipStream_t st = (hipStream_t)stream;
    // the 3-channel stride-2 first layer on the compact image layout (conv_rgb.hip); wgrad.variant 1 keeps the generic kernel
    if (ksize == 3 && stride == 2 && !x2 && ldx * esz == 16 && !g_wnorm.nt && !g_wnorm.query && shm_tune(SHM_TUNE_WGRAD_VARIANT) != 1) {
        const int r = shm_rgb_s2_wgrad_launch(x, ldx, dy, lddy, (float*)workspace, ws_bytes, batch, hi, wi, cin, cout, a.xbytes, a.dybytes, dtype, nsplit_out, st);
        if (r < 0) return r;
        if (r == 1) return SHM_OK;
    }
    const bool straddle = x2 && (c1 % 64 != 0);
    const int wv = shm_tune(SHM_TUNE_WGRAD_VARIANT);       // 0 automatic, 1 generic kernels only, 2 no thin-input packing, 3 no stride-2 halo form
    const int no_halo = wv == 1;
    const bool halo_ok = ksize == 3 && stride == 1 && wi % 16 == 0 && hi % 2 == 0 && !straddle && !no_halo;
    const int no_thin = wv == 2;
    // thin first layers: (tap, ci) pairs packed into the MFMA rows; patches of 2 x 16 OUTPUT pixels
    const bool thin_ok = !no_thin && !no_halo && ksize == 3 && !x2 && 9 * cin <= 96 && ldx == 16 && cin_ld <= 16 && wo % 16 == 0 && ho % 2 == 0 &&
                         hi % stride == 0 && wi % stride == 0;
    // stride 2 (wv == 3: not this form): SAME padding of an even map puts nothing before the first row / column
    const bool halo2_ok = ksize == 3 && stride == 2 && pt == 0 && pl == 0 && hi % 2 == 0 && wi % 2 == 0 && wo % 8 == 0 && ho % 2 == 0 && !straddle &&
                          !no_halo && wv != 3 && !(thin_ok && dtype == SHM_F32);
    // norm: the halo-image kernels normalise their x halo in LDS (a block's 64 input channels lie in one source: no straddle)
    const bool want_nm = g_wnorm.nt != nullptr;
    if (want_nm || g_wnorm.query) {
        const int pc = x2 ? (g_wnorm.part ? cin_ld - c1 : c1) : cin_ld;
        const bool ok = want_nm && halo_ok && (dtype == SHM_BF16 || !thin_ok) && g_wnorm.c == pc && (g_wnorm.part == 0 || x2 != nullptr) && pc % vec == 0;
        if (g_wnorm.query) {
            g_wnorm.query_ok = ok;
            return SHM_OK;
        }
        SHM_REQUIRE(ok, SHM_E_SHAPE,
                    "shm_conv2d_wgrad_norm: the kernel this shape runs on cannot normalise its source in LDS (unit-stride 3x3, map width a multiple of "
                    "16, concat split a multiple of 64; ask shm_conv2d_wgrad_norm_supported) -- use shm_in_apply");
    }
    // bf16, eight-wave block over 64 ci x 128 co (round 4): unit stride on maps whose height is a multiple of four, stride 2 on even maps
    // whose output width is a multiple of 16; "wgrad.bf16_wide" = 1 keeps the four-wave kernels
    const bool w8_s1 = halo_ok && hi % 4 == 0 && shm_tune(SHM_TUNE_WGRAD_BF16_ROWS) != 2;
    const bool w8_s2_any = ksize == 3 && stride == 2 && pt == 0 && pl == 0 && hi % 2 == 0 && wi % 2 == 0 && !straddle && !no_halo && wv != 3;
    const bool w8_s2_16 = w8_s2_any && wo % 16 == 0 && ho % 2 == 0, w8_s2_8 = w8_s2_any && !w8_s2_16 && wo % 8 == 0 && ho % 4 == 0;
    const bool w8_s2 = w8_s2_16 || w8_s2_8;
    // "wgrad.bf16_wide": 0 automatic = stride 2 only, 1 never, 2 stride 2 only, 3 unit stride only, 4 both.  Unit stride is NOT the
    // automatic choice although the eight-wave block is 9-27 % faster than wgrad_halo_bf16_kernel<4> launch for launch (tools/bench_wgrad_bf16.py):
    // in the two-stream step the weight gradients run on the second stream beside the input-gradient chain, which is the critical path;
    // a block that fills a CU (eight waves at 210 VGPRs, 93 KiB of LDS) for the whole launch keeps that chain's kernels off the CU, the
    // four-wave kernel at one block per CU leaves them half of it (bf16 step, S=256 B=8 / S=512 B=4 / B=32: both 25.5 / 48.9 / 89.6 ms, stride 2
    // only 25.0 / 47.8 / 88.5, neither 25.1 / 48.6 / 89.8, unit stride only 25.8 / 50.6 / 91.6).  At stride 2 it replaces wgrad_bf16_kernel<9>, which
    // needs 30-43 % more time on every layer and is no lighter on a CU.
    const int wide = shm_tune(SHM_TUNE_WGRAD_BF16_WIDE);
    const bool wide_s1 = wide == 3 || wide == 4, wide_s2 = wide == 0 || wide == 2 || wide == 4;
    // (64 output channels -- the discriminator's 3-channel first layer -- stay on wgrad_bf16_kernel<9>: with the pair's second co tile empty
    // the eight-wave block measured 296 us against 271)
    const bool w8_take_s1 = w8_s1 && wide_s1 && cout >= 128, w8_take_s2 = w8_s2 && wide_s2 && cout >= 128 && !w8_take_s1;
    if (dtype == SHM_BF16 && (w8_take_s1 || w8_take_s2) && !want_nm) {
        WgradHaloArgs hgs{};
        hgs.x = x;
        hgs.x2 = x2;
        hgs.c1 = a.c1;
        hgs.ldx = ldx;
        hgs.ldx2 = ldx2;
        hgs.dy = dy;
        hgs.lddy = lddy;
        hgs.part = (float*)workspace;
        hgs.h = hi;
        hgs.w = wi;
        hgs.cin_ld = cin_ld;
        hgs.cin = cin;
        hgs.cout = cout;
        const int w8_mode = w8_take_s1 ? 0 : w8_s2_16 ? 1 : 2;
        const int rows = w8_mode == 1 ? 2 : 4, pw = w8_mode == 2 ? 8 : 16;       // output rows / columns per stage
        hgs.npatch = batch * (ho / rows) * (wo / pw);
        // one block per CU (93 KiB of LDS): the block target counts 64 x 128 tiles, i.e. twice the splits of the four-wave kernel's choice
        int nsh;
        {
            const int tiles8 = shm_cdiv(cin, 64) * shm_cdiv(cout, 128);
            const int tuned = shm_tune(SHM_TUNE_WGRAD_BLOCKS);
            nsh = shm_cdiv(tuned ? tuned : 256, tiles8);
            const long maxs = ((long)a.M + 255) / 256;
            if (nsh > maxs) nsh = (int)maxs;
            if (nsh < 1) nsh = 1;
        }
        if (nsh > hgs.npatch) nsh = hgs.npatch;
        hgs.patches_per_split = shm_cdiv(hgs.npatch, nsh);
        nsh = shm_cdiv(hgs.npatch, hgs.patches_per_split);
        SHM_REQUIRE(ws_bytes >= (size_t)nsh * 9 * cin * cout * sizeof(float), SHM_E_WORKSPACE, "shm_conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes,
                    (size_t)nsh * 9 * cin * cout * sizeof(float));
        hgs.xbytes = a.xbytes;
        hgs.x2bytes = a.x2bytes;
        hgs.dybytes = a.dybytes;
        ns = nsh;
        constexpr unsigned kLds8 = 3u * 248u * 128u;    // 93 KiB
        static const hipError_t attr_a = hipFuncSetAttribute((const void*)wgrad_halo8_bf16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds8);
        static const hipError_t attr_b = hipFuncSetAttribute((const void*)wgrad_halo8_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds8);
        static const hipError_t attr_c = hipFuncSetAttribute((const void*)wgrad_halo8_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds8);
        SHM_REQUIRE(attr_a == hipSuccess && attr_b == hipSuccess && attr_c == hipSuccess, SHM_E_HIP, "shm_conv2d_wgrad: cannot reserve 93 KiB of LDS: %s",
                    hipGetErrorString(attr_a != hipSuccess ? attr_a : attr_b != hipSuccess ? attr_b : attr_c));
        const dim3 grid8(shm_cdiv(cin, 64), shm_cdiv(cout, 128), nsh);
        if (w8_mode == 0)
            hipLaunchKernelGGL((wgrad_halo8_bf16_kernel<0>), grid8, dim3(512), kLds8, st, hgs);
        else if (w8_mode == 1)
            hipLaunchKernelGGL((wgrad_halo8_bf16_kernel<1>), grid8, dim3(512), kLds8, st, hgs);
        else
            hipLaunchKernelGGL((wgrad_halo8_bf16_kernel<2>), grid8, dim3(512), kLds8, st, hgs);
        shm_set_last_kernel("wgrad_halo8_bf16_kernel<%d>", w8_mode);
    } else if (dtype == SHM_BF16 && halo_ok) {
        WgradHaloArgs hgs{};
        hgs.x = x;
        hgs.x2 = x2;
        hgs.c1 = a.c1;
        hgs.ldx = ldx;
        hgs.ldx2 = ldx2;
        hgs.dy = dy;
        hgs.lddy = lddy;
        hgs.part = (float*)workspace;
        hgs.h = hi;
        hgs.w = wi;
        hgs.cin_ld = cin_ld;
        hgs.cin = cin;
        hgs.cout = cout;
        const int rows = (hi % 4 == 0 && shm_tune(SHM_TUNE_WGRAD_BF16_ROWS) != 2) ? 4 : 2;      // pixel rows per stage
        hgs.npatch = batch * (hi / rows) * (wi / 16);
        int nsh = ns < hgs.npatch ? ns : hgs.npatch;
        hgs.patches_per_split = shm_cdiv(hgs.npatch, nsh);
        if (want_nm && g_wnorm.mode) hgs.patches_per_split = wgrad_norm_aligned_pps(hgs.patches_per_split, (hi / rows) * (wi / 16));
        nsh = shm_cdiv(hgs.npatch, hgs.patches_per_split);
        SHM_REQUIRE(ws_bytes >= (size_t)nsh * 9 * cin * cout * sizeof(float), SHM_E_WORKSPACE,
                    "shm_conv2d_wgrad: workspace %zu < %zu bytes (SHM_NORM_SCALED: shm_conv2d_wgrad_norm_workspace)", ws_bytes, (size_t)nsh * 9 * cin * cout * sizeof(float));
        hgs.xbytes = a.xbytes;
        hgs.x2bytes = a.x2bytes;
        hgs.dybytes = a.dybytes;
        hgs.nt = g_wnorm.nt;
        hgs.ntpart = g_wnorm.part;
        hgs.ntc = g_wnorm.c;
        ns = nsh;
        const int nmode = want_nm ? 1 + g_wnorm.mode : 0;
        const dim3 gridb(shm_cdiv(cin, 64), shm_cdiv(cout, 64), nsh);
        if (rows == 4) {
            constexpr unsigned kLds = 3u * (6 * 20 + 4 * 16) * 128u;      // 69 KiB
            static const hipError_t attr = hipFuncSetAttribute((const void*)wgrad_halo_bf16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
            static const hipError_t attr1 = hipFuncSetAttribute((const void*)wgrad_halo_bf16_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
            static const hipError_t attr2 = hipFuncSetAttribute((const void*)wgrad_halo_bf16_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
            SHM_REQUIRE(attr == hipSuccess && attr1 == hipSuccess && attr2 == hipSuccess, SHM_E_HIP, "shm_conv2d_wgrad: cannot reserve 69 KiB of LDS: %s",
                        hipGetErrorString(attr != hipSuccess ? attr : attr1 != hipSuccess ? attr1 : attr2));
            if (nmode == 2)
                hipLaunchKernelGGL((wgrad_halo_bf16_kernel<4, 2>), gridb, dim3(256), kLds, st, hgs);
            else if (nmode == 1)
                hipLaunchKernelGGL((wgrad_halo_bf16_kernel<4, 1>), gridb, dim3(256), kLds, st, hgs);
            else
                hipLaunchKernelGGL((wgrad_halo_bf16_kernel<4>), gridb, dim3(256), kLds, st, hgs);
        } else if (nmode == 2) {
            hipLaunchKernelGGL((wgrad_halo_bf16_kernel<2, 2>), gridb, dim3(256), 3u * (4 * 20 + 2 * 16) * 128u, st, hgs);
        } else if (nmode == 1) {
            hipLaunchKernelGGL((wgrad_halo_bf16_kernel<2, 1>), gridb, dim3(256), 3u * (4 * 20 + 2 * 16) * 128u, st, hgs);
        } else {
            hipLaunchKernelGGL((wgrad_halo_bf16_kernel<2>), gridb, dim3(256), 3u * (4 * 20 + 2 * 16) * 128u, st, hgs);
        }
        if (nmode)
            shm_set_last_kernel("wgrad_halo_bf16_kernel<%d, %d>", rows, nmode);
        else
            shm_set_last_kernel("wgrad_halo_bf16_kernel<%d>", rows);
    } else if (dtype == SHM_BF16) {
        dim3 grid(shm_cdiv(cin, 64), shm_cdiv(cout, 64), ns);
        if (ksize == 3) {
            if (straddle)
                hipLaunchKernelGGL((wgrad_bf16_kernel<9, true>), grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((wgrad_bf16_kernel<9, false>), grid, dim3(256), 0, st, a);
        } else {
            if (straddle)
                hipLaunchKernelGGL((wgrad_bf16_kernel<1, true>), grid, dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((wgrad_bf16_kernel<1, false>), grid, dim3(256), 0, st, a);
        }
        shm_set_last_kernel("wgrad_bf16_kernel<%d, %s>", ksize * ksize, straddle ? "true" : "false");
    } else if (dtype == SHM_F32 && thin_ok && stride == 2) {
        // first layer of the discriminator (3 channels, stride 2): patches over the OUTPUT map
        WgradHaloArgs hgs{};
        hgs.x = x;
        hgs.ldx = ldx;
        hgs.dy = dy;
        hgs.lddy = lddy;
        hgs.part = (float*)workspace;
        hgs.h = hi;
        hgs.w = wi;
        hgs.cin_ld = cin_ld;
        hgs.cin = cin;
        hgs.cout = cout;
        hgs.npatch = batch * (ho / 2) * (wo / 16);
        int nsh = ns < hgs.npatch ? ns : hgs.npatch;
        hgs.patches_per_split = shm_cdiv(hgs.npatch, nsh);
        nsh = shm_cdiv(hgs.npatch, hgs.patches_per_split);
        hgs.xbytes = a.xbytes;
        hgs.dybytes = a.dybytes;
        SHM_REQUIRE(ws_bytes >= (size_t)2 * nsh * 9 * cin * cout * sizeof(float), SHM_E_WORKSPACE, "shm_conv2d_wgrad: workspace too small");
        ns = 2 * nsh;
        if (9 * cin <= 32) {
            hipLaunchKernelGGL((wgrad_halo_thin_kernel<1, 2>), dim3(1, shm_cdiv(cout, 64), nsh), dim3(256), 0, st, hgs);
            shm_set_last_kernel("wgrad_halo_thin_kernel<1, 2>");
        } else {
            hipLaunchKernelGGL((wgrad_halo_thin_kernel<3, 2>), dim3(1, shm_cdiv(cout, 64), nsh), dim3(256), 0, st, hgs);
            shm_set_last_kernel("wgrad_halo_thin_kernel<3, 2>");
        }
    } else if (dtype == SHM_F32 && halo2_ok) {
        // stride-2 3x3 layers: patches of 2 x 8 OUTPUT pixels with a 5 x 17 input halo
        WgradHaloArgs hgs{};
        hgs.x = x;
        hgs.x2 = x2;
        hgs.c1 = a.c1;
        hgs.ldx = ldx;
        hgs.ldx2 = ldx2;
        hgs.dy = dy;
        hgs.lddy = lddy;
        hgs.part = (float*)workspace;
        hgs.h = hi;
        hgs.w = wi;
        hgs.cin_ld = cin_ld;
        hgs.cin = cin;
        hgs.cout = cout;
        // "wgrad.f32_split" (opt-in, round 6): the six-bf16-product form on patches of 2 x 16 output pixels (conv_wgrad_x3.hip, S2)
        const bool x3s2 = shm_tune(SHM_TUNE_WGRAD_F32_SPLIT) == 1 && wo % 16 == 0 && !want_nm;
        hgs.npatch = batch * (ho / 2) * (wo / (x3s2 ? 16 : 8));
        int nsh = ns < hgs.npatch ? ns : hgs.npatch;
        hgs.patches_per_split = shm_cdiv(hgs.npatch, nsh);
        nsh = shm_cdiv(hgs.npatch, hgs.patches_per_split);
        hgs.xbytes = a.xbytes;
        hgs.x2bytes = a.x2bytes;
        hgs.dybytes = a.dybytes;
        ns = nsh;
        if (x3s2) {
            const int rc = shm_wgrad_x3_launch(hgs, cin, cout, nsh, 2, st, true);
            if (rc != SHM_OK) return rc;
        } else {
            hipLaunchKernelGGL((wgrad_halo_kernel<0, true>), dim3(shm_cdiv(cin, 64), shm_cdiv(cout, 64), nsh), dim3(256), 0, st, hgs);
            shm_set_last_kernel("wgrad_halo_kernel<0, true>");
        }
    } else if (halo_ok) {
        const bool thin = thin_ok;
        WgradHaloArgs hgs{};
        hgs.x = x;
        hgs.x2 = x2;
        hgs.c1 = a.c1;
        hgs.ldx = ldx;
        hgs.ldx2 = ldx2;
        hgs.dy = dy;
        hgs.lddy = lddy;
        hgs.part = (float*)workspace;
        hgs.h = hi;
        hgs.w = wi;
        hgs.cin_ld = cin_ld;
        hgs.cin = cin;
        hgs.cout = cout;
        hgs.npatch = batch * (hi / 2) * (wi / 16);
        int nsh = ns < hgs.npatch ? ns : hgs.npatch;
        hgs.patches_per_split = shm_cdiv(hgs.npatch, nsh);
        if (want_nm && g_wnorm.mode) hgs.patches_per_split = wgrad_norm_aligned_pps(hgs.patches_per_split, (hi / 2) * (wi / 16));
        nsh = shm_cdiv(hgs.npatch, hgs.patches_per_split);
        SHM_REQUIRE(ws_bytes >= (size_t)nsh * 9 * cin * cout * sizeof(float), SHM_E_WORKSPACE,
                    "shm_conv2d_wgrad: workspace %zu < %zu bytes (SHM_NORM_SCALED: shm_conv2d_wgrad_norm_workspace)", ws_bytes, (size_t)nsh * 9 * cin * cout * sizeof(float));
        hgs.xbytes = a.xbytes;
        hgs.x2bytes = a.x2bytes;
        hgs.dybytes = a.dybytes;
        ns = nsh;
        if (thin) {                                    // two slabs per block (one per patch row)
            SHM_REQUIRE(ws_bytes >= (size_t)2 * nsh * 9 * cin * cout * sizeof(float), SHM_E_WORKSPACE, "shm_conv2d_wgrad: workspace too small");
            ns = 2 * nsh;
            hipLaunchKernelGGL((wgrad_halo_thin_kernel<3, 1>), dim3(1, shm_cdiv(cout, 64), nsh), dim3(256), 0, st, hgs);
            shm_set_last_kernel("wgrad_halo_thin_kernel<3, 1>");
        } else {
        dim3 gridh(shm_cdiv(cin, 64), shm_cdiv(cout, 64), nsh);
        hgs.nt = g_wnorm.nt;
        hgs.ntpart = g_wnorm.part;
        hgs.ntc = g_wnorm.c;
        if ((!want_nm || g_wnorm.mode == 0) && shm_tune(SHM_TUNE_WGRAD_F32_SPLIT) == 1) {        // "wgrad.f32_split": conv_wgrad_x3.hip (plain and SHM_NORM_EXACT sources)
            // stages of four pixel rows where the map allows ("wgrad.bf16_rows" = 2 keeps two): the patches and the split are re-cut for them
            // (the normalising form keeps two rows: with four its 24 table values spill)
            const int rows = (hi % 4 == 0 && shm_tune(SHM_TUNE_WGRAD_BF16_ROWS) != 2 && !want_nm) ? 4 : 2;
            if (rows == 4) {
                hgs.npatch = batch * (hi / 4) * (wi / 16);
                int n4 = ns < hgs.npatch ? ns : hgs.npatch;
                hgs.patches_per_split = shm_cdiv(hgs.npatch, n4);
                n4 = shm_cdiv(hgs.npatch, hgs.patches_per_split);
                nsh = n4;             // (never more slabs than the two-row cut: the workspace check above holds)
                ns = n4;
            }
            const int rc = shm_wgrad_x3_launch(hgs, cin, cout, nsh, rows, st);
            if (rc != SHM_OK) return rc;
        } else {
        if (want_nm && g_wnorm.mode)
            hipLaunchKernelGGL(wgrad_halo_kernel<2>, gridh, dim3(256), 0, st, hgs);
        else if (want_nm)
            hipLaunchKernelGGL(wgrad_halo_kernel<1>, gridh, dim3(256), 0, st, hgs);
        else
            hipLaunchKernelGGL(wgrad_halo_kernel<0>, gridh, dim3(256), 0, st, hgs);
        shm_set_last_kernel(want_nm ? (g_wnorm.mode ? "wgrad_halo_kernel<2>" : "wgrad_halo_kernel<1>") : "wgrad_halo_kernel");
        }
        }
    } else {
    dim3 grid(shm_cdiv(cin, 64), shm_cdiv(cout, 64), ns);
    if (ksize == 3) {
        if (straddle)
            hipLaunchKernelGGL((wgrad_kernel<9, true>), grid, dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((wgrad_kernel<9, false>), grid, dim3(256), 0, st, a);
    } else {
        if (straddle)
            hipLaunchKernelGGL((wgrad_kernel<1, true>), grid, dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((wgrad_kernel<1, false>), grid, dim3(256), 0, st, a);
    }
    shm_set_last_kernel("wgrad_kernel<%d, %s>", ksize * ksize, straddle ? "true" : "false");
    }
    SHM_LAUNCH_CHECK("shm_conv2d_wgrad");
    if (nsplit_out) *nsplit_out = ns;
    return SHM_OK;
}

// Phase 2: dw[i] (+)= sum over the nsplit slabs, in a fixed order.
extern "C" int shm_conv2d_wgrad_reduce(const void* workspace, float* dw, size_t n, int nsplit, int accumulate, void* stream) {
    SHM_REQUIRE(workspace && dw && nsplit >= 1, SHM_E_SHAPE, "shm_conv2d_wgrad_reduce: bad arguments");
    if (n == 0) return SHM_OK;
    if (n % 4 == 0 && ((size_t)workspace & 15) == 0 && ((size_t)dw & 15) == 0)
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(shm_cdiv((long)(n / 4), 16)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dw, n / 4,
                           nsplit, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(shm_cdiv((long)n, 64)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dw, n, nsplit,
                           accumulate);
    SHM_LAUNCH_CHECK("shm_conv2d_wgrad(reduce)");
    return SHM_OK;
}

// Would shm_conv2d_wgrad_norm run on a kernel that normalises its source in LDS?  A dry run of the variant choice; nothing is launched.
extern "C" int shm_conv2d_wgrad_norm_supported(int batch, int hi, int wi, int cin, int cin_ld, int c1, int cout, int ksize, int stride, int norm_part, int dtype) {
    if (dtype != SHM_F32 && dtype != SHM_BF16) return 0;
    if (norm_part != 0 && norm_part != 1) return 0;
    const bool two = c1 > 0 && c1 < cin_ld;
    if (norm_part == 1 && !two) return 0;
    static const __attribute__((aligned(256))) char dummy[256] = {};
    const size_t ws = shm_conv2d_wgrad_workspace(batch, hi, wi, cin, cout, ksize) * 2;
    g_wnorm.nt = (const float*)dummy;
    g_wnorm.part = norm_part;
    g_wnorm.c = two ? (norm_part ? cin_ld - c1 : c1) : cin_ld;
    g_wnorm.query = true;
    g_wnorm.query_ok = false;
    int ns = 0;
    const int r = shm_conv2d_wgrad_partial(dummy, two ? dummy : nullptr, two ? c1 : 0, two ? c1 : cin_ld, two ? cin_ld - c1 : 0, dummy, cout, batch, hi, wi, cin,
                                           cin_ld, cout, ksize, stride, (void*)dummy, ws, dtype, &ns, nullptr);
    const bool ok = r == SHM_OK && g_wnorm.query_ok;
    g_wnorm = WNormReq{};
    return ok ? 1 : 0;
}

// Workspace of shm_conv2d_wgrad_norm(SHM_NORM_SCALED): the splits are cut on sample boundaries, which can take more slabs than
// shm_conv2d_wgrad_workspace allows for.
extern "C" size_t shm_conv2d_wgrad_norm_workspace(int batch, int hi, int wi, int cin, int cout, int ksize, int dtype) {
    const int esz = dtype == SHM_BF16 ? 2 : 4;
    const int ns = wgrad_splits(batch, hi, wi, cin, cout, esz);
    const int rows = dtype == SHM_BF16 ? ((hi % 4 == 0 && shm_tune(SHM_TUNE_WGRAD_BF16_ROWS) != 2) ? 4 : 2) : 2;
    if (hi % rows || wi % 16) return shm_conv2d_wgrad_workspace(batch, hi, wi, cin, cout, ksize);
    const int ppi = (hi / rows) * (wi / 16), npatch = batch * ppi;
    const int nsh = ns < npatch ? ns : npatch;
    const int pps = wgrad_norm_aligned_pps(shm_cdiv(npatch, nsh), ppi);
    const size_t aligned = (size_t)shm_cdiv(npatch, pps) * ksize * ksize * cin * cout * sizeof(float);
    const size_t plain = shm_conv2d_wgrad_workspace(batch, hi, wi, cin, cout, ksize);
    return aligned > plain ? aligned : plain;
}

// SHM_NORM_SCALED, the second term of the weight gradient: dw[tap][part_lo + k][co] += sum_n (beta[k] - mean_n[k] * inv_n[k]) * dzsum[n][co]
// for every tap (with `ring` in the out-of-image taps the sum over the pixels does not depend on the tap).  dzsum = per-sample channel
// sums of dz (shm_in_bwd_keep_dz_sums).
__global__ __launch_bounds__(256) void wgrad_norm_finish_kernel(float* __restrict__ dw, const float* __restrict__ nt, const double* __restrict__ dzsum, int batch,
                                                                int c, int part_lo, int cin, int cout, int ntaps) {
    // a thread owns one (k, co) pair and walks the samples four at a time (independent loads in flight: as a chain of `batch`
    // round trips the kernel took 60-80 us)
    const int co = blockIdx.x * 64 + (threadIdx.x & 63), k = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (co >= cout || k >= c) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const size_t ts = (size_t)SHM_NT_PLANES * c;
    int n = 0;
    for (; n + 3 < batch; n += 4) {
        const float* t = nt + (size_t)n * ts + k;
        const float m0 = t[0], i0 = t[c], b0 = t[2 * c], m1 = t[ts], i1 = t[ts + c], b1 = t[ts + 2 * c];
        const float m2 = t[2 * ts], i2 = t[2 * ts + c], b2 = t[2 * ts + 2 * c], m3 = t[3 * ts], i3 = t[3 * ts + c], b3 = t[3 * ts + 2 * c];
        const double d0 = dzsum[(size_t)n * cout + co], d1 = dzsum[(size_t)(n + 1) * cout + co], d2 = dzsum[(size_t)(n + 2) * cout + co],
                     d3 = dzsum[(size_t)(n + 3) * cout + co];
        s0 += ((double)b0 - (double)m0 * (double)i0) * d0;
        s1 += ((double)b1 - (double)m1 * (double)i1) * d1;
        s2 += ((double)b2 - (double)m2 * (double)i2) * d2;
        s3 += ((double)b3 - (double)m3 * (double)i3) * d3;
    }
    for (; n < batch; ++n) {
        const float* t = nt + (size_t)n * ts + k;
        s0 += ((double)t[2 * c] - (double)t[0] * (double)t[c]) * dzsum[(size_t)n * cout + co];
    }
    const float sf = (float)((s0 + s1) + (s2 + s3));
    for (int tap = 0; tap < ntaps; ++tap) dw[((size_t)tap * cin + part_lo + k) * cout + co] += sf;
}

extern "C" int shm_conv2d_wgrad_norm_finish(float* dw, const float* nt, const double* dzsum, int batch, int c, int part_lo, int cin, int cout, int ksize,
                                            void* stream) {
    SHM_REQUIRE(dw && nt && dzsum, SHM_E_SHAPE, "shm_conv2d_wgrad_norm_finish: null pointer");
    SHM_REQUIRE(c > 0 && part_lo >= 0 && part_lo + c <= cin, SHM_E_SHAPE, "shm_conv2d_wgrad_norm_finish: part [%d, %d) outside %d channels", part_lo,
                part_lo + c, cin);
    if (batch == 0 || cout == 0) return SHM_OK;
    hipLaunchKernelGGL(wgrad_norm_finish_kernel, dim3(shm_cdiv(cout, 64), shm_cdiv(c, 4)), dim3(256), 0, (hipStream_t)stream, dw, nt, dzsum, batch, c, part_lo,
                       cin, cout, ksize * ksize);
    SHM_LAUNCH_CHECK("shm_conv2d_wgrad_norm_finish");
    return SHM_OK;
}

extern "C" int shm_conv2d_wgrad_partial_norm(const void* x, const void* x2, int c1, int ldx, int ldx2, const float* nt_x, const float* nt_x2, int norm_mode,
                                             const void* dy, int lddy, int batch, int hi, int wi, int cin, int cin_ld, int cout, int ksize, int stride,
                                             void* workspace, size_t ws_bytes, int dtype, int* nsplit_out, void* stream) {
    SHM_REQUIRE(!(nt_x && nt_x2), SHM_E_SHAPE, "shm_conv2d_wgrad_norm: at most one source can be normalised on the fly");
    SHM_REQUIRE(!nt_x2 || x2, SHM_E_SHAPE, "shm_conv2d_wgrad_norm: nt_x2 without a second source");
    SHM_REQUIRE(norm_mode == SHM_NORM_EXACT || norm_mode == SHM_NORM_SCALED, SHM_E_SHAPE, "shm_conv2d_wgrad_norm: norm_mode %d", norm_mode);
    if (nt_x || nt_x2) {
        g_wnorm.nt = nt_x ? nt_x : nt_x2;
        g_wnorm.part = nt_x ? 0 : 1;
        g_wnorm.c = x2 ? (nt_x ? c1 : cin_ld - c1) : cin_ld;
        g_wnorm.mode = norm_mode;
    }
    const int r = shm_conv2d_wgrad_partial(x, x2, c1, ldx, ldx2, dy, lddy, batch, hi, wi, cin, cin_ld, cout, ksize, stride, workspace, ws_bytes, dtype,
                                           nsplit_out, stream);
    g_wnorm = WNormReq{};
    return r;
}

// shm_conv2d_wgrad on a source that is the UN-normalised activation of an InstanceNorm block (nt_x / nt_x2: that block's table, at
// most one of the two)
extern "C" int shm_conv2d_wgrad_norm(const void* x, const void* x2, int c1, int ldx, int ldx2, const float* nt_x, const float* nt_x2, int norm_mode,
                                     const void* dy, int lddy, float* dw, int batch, int hi, int wi, int cin, int cin_ld, int cout, int ksize, int stride,
                                     int accumulate, void* workspace, size_t ws_bytes, int dtype, void* stream) {
    SHM_REQUIRE(!(nt_x && nt_x2), SHM_E_SHAPE, "shm_conv2d_wgrad_norm: at most one source can be normalised on the fly");
    SHM_REQUIRE(!nt_x2 || x2, SHM_E_SHAPE, "shm_conv2d_wgrad_norm: nt_x2 without a second source");
    SHM_REQUIRE(norm_mode == SHM_NORM_EXACT || norm_mode == SHM_NORM_SCALED, SHM_E_SHAPE, "shm_conv2d_wgrad_norm: norm_mode %d", norm_mode);
    if (nt_x || nt_x2) {
        g_wnorm.nt = nt_x ? nt_x : nt_x2;
        g_wnorm.part = nt_x ? 0 : 1;
        g_wnorm.c = x2 ? (nt_x ? c1 : cin_ld - c1) : cin_ld;
        g_wnorm.mode = norm_mode;
    }
    const int r = shm_conv2d_wgrad(x, x2, c1, ldx, ldx2, dy, lddy, dw, batch, hi, wi, cin, cin_ld, cout, ksize, stride, accumulate, workspace, ws_bytes, dtype,
                                   stream);
    g_wnorm = WNormReq{};
    return r;
}

extern "C" int shm_conv2d_wgrad(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* dy,
                                int lddy, float* dw, int batch, int hi, int wi, int cin, int cin_ld,
                                int cout, int ksize, int stride, int accumulate, void* workspace,
                                size_t ws_bytes, int dtype, void* stream) {
    SHM_REQUIRE(dw, SHM_E_SHAPE, "shm_conv2d_wgrad: null pointer");
    int ns = 0;
    int r = shm_conv2d_wgrad_partial(x, x2, c1, ldx, ldx2, dy, lddy, batch, hi, wi, cin, cin_ld, cout, ksize, stride, workspace, ws_bytes, dtype,
                                     &ns, stream);
    if (r) return r;
    return shm_conv2d_wgrad_reduce(workspace, dw, (size_t)ksize * ksize * cin * cout, ns, accumulate, stream);
}
