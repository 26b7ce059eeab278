// Implicit-GEMM convolution family on v_mfma_f32_32x32x2_f32 (gfx950, exact fp32).
//
// One kernel ("tap GEMM") serves Conv2D forward (k=1/3, stride 1/2), its input-gradient
// (stride 1: flipped taps; stride 2: four output phases) and Conv2DTranspose forward
// (= input-gradient of the stride-2 conv):
//
//     out[pix(m), n] = act( bias[n] + sum_{tap} sum_{k<K} A[src(m, tap), k] * B[tap][n][k] )
//
//   M = batch * grid_h * grid_w output positions of one phase, N = output channels,
//   K = channels of the A tensor (multiple of 16).  A is NHWC (optionally the channel
//   concat of two tensors), B is [tap][N][K] (K contiguous), so both operands are staged
//   as rows of 16 consecutive floats: 16-byte global loads, ds_write_b128 into LDS rows
//   padded to 20 floats (conflict-free ds_read_b128 per 16-lane group), and each lane
//   feeds four consecutive MFMAs from one 16-byte LDS read (the k order inside an 8-wide
//   group is permuted identically for A and B, which a dot product does not see).
//
// Block = 256 threads = 4 waves of 64x64 outputs each (2x2 MFMA tiles), block tile 128x128
// (2x2 waves) or 256x64 (4x1 waves, for Cout <= 64), BK = 16, two LDS stages and two register
// sets: global loads run two K-steps ahead of the MFMAs that consume them.
#include "common.h"

#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct TapPhase {
    int oph, opw, ntaps;
    int dh[9], dw[9], widx[9];
};

struct TapGemmArgs {
    const float* x;   // A source 1 [batch, hi, wi, c1]   pitch ldx
    const float* x2;  // A source 2 [batch, hi, wi, K-c1] pitch ldx2 (or unused, c1 == K)
    int c1, ldx, ldx2;
    const float* w;     // [taps][nout][K]
    const float* bias;  // [nout] or null
    float* y;           // channels [0,n1)
    float* y2;          // channels [n1,nout)
    int n1, ldy, ldy2;
    int hi, wi, K;      // A tensor dims
    int hg, wg;         // output grid of one phase
    int ho, wo, nout;   // full output dims
    int is, os;         // A stride, output stride
    int M;              // batch*hg*wg
    unsigned xbytes, x2bytes, wbytes;   // buffer-descriptor extents (bytes)
    double* stats;      // optional [batch][nout][2] (sum, sum of squares) of the stored outputs
    int hw;             // pixels per sample (stats only; hw % 64 == 0)
    float slope;
    TapPhase ph[4];
};

// BM x BN block tile, WGM x WGN waves (4 waves), every wave owns a 64x64 sub-tile (2x2 MFMA tiles).
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 3) void tapgemm_kernel(const TapGemmArgs a) {
    static_assert(WGM * WGN == 4 && BM / WGM == 64 && BN / WGN == 64, "wave tile must be 64x64");
    constexpr int LDK = 20;          // 16 + 4 pad floats per LDS row
    constexpr int AR = BM / 64;      // A rows staged per thread
    constexpr int BR = BN / 64;      // B rows staged per thread
    constexpr int TM = 2, TN = 2;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];

    const TapPhase& P = a.ph[blockIdx.z];
    const int tid = threadIdx.x, quad = tid & 3, lrow = tid >> 2;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    int pixbase[AR], ih0[AR], iw0[AR];
    bool mval[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        int m = m0 + lrow + 64 * j;
        mval[j] = m < a.M;
        int mm = mval[j] ? m : 0;
        int ow = mm % a.wg, t = mm / a.wg;
        int oh = t % a.hg, n = t / a.hg;
        ih0[j] = oh * a.is;
        iw0[j] = ow * a.is;
        pixbase[j] = (n * a.hi + ih0[j]) * a.wi + iw0[j];
    }
    const int ntaps = P.ntaps;
    const int ksteps = ntaps * (a.K >> 4);

    // K-step order: (32-channel group, tap, 16-channel half).  The two halves of a 128-byte line
    // are read in consecutive steps (second one hits L1), and the 9 shifted reads of a group reuse
    // the same input rows.  Everything advances incrementally; the tap-table entries of the NEXT
    // load are fetched one step ahead so their scalar-load latency sits behind an MFMA block.
    const int nch = a.K >> 4;
    int ld_g = 0, ld_tap = 0, ld_sub = 0, ld_c0 = 0;    // position of the next gload
    int t_dh = P.dh[0], t_dw = P.dw[0], t_wi = P.widx[0];
    auto advance = [&]() {
        const int nsub = (nch - ld_g) >= 2 ? 2 : 1;
        if (++ld_sub == nsub) {
            ld_sub = 0;
            if (++ld_tap == ntaps) {
                ld_tap = 0;
                ld_g += 2;
            }
            t_dh = P.dh[ld_tap];
            t_dw = P.dw[ld_tap];
            t_wi = P.widx[ld_tap];
        }
        ld_c0 = (ld_g + ld_sub) << 4;
    };
    // Operands are fetched with raw buffer loads: a lane whose tap falls outside the image (SAME
    // padding), whose row is past M or whose weight row is past N gets byte offset 0xffffffff,
    // which the hardware range check turns into zeros.  No divergent branch around a load, so
    // hipcc can wait with counted s_waitcnt vmcnt(N) and the loads really stay two steps ahead.
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
    unsigned wrow[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        int nn = n0 + lrow + 64 * j;
        wrow[j] = nn < a.nout ? (unsigned)(nn * a.K + quad * 4) * 4u : 0xffffffffu;
    }
    auto gload = [&](f32x4 (&ra)[AR], f32x4 (&rb)[BR]) {
        const int c0 = ld_c0;
        const bool second = c0 >= a.c1;
        const int ld = second ? a.ldx2 : a.ldx;
        const int cc = (second ? c0 - a.c1 : c0) + quad * 4;
        const int dh = t_dh, dw = t_dw;
        const int doff = dh * a.wi + dw;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            int ih = ih0[j] + dh, iw = iw0[j] + dw;
            bool ok = mval[j] && (unsigned)ih < (unsigned)a.hi && (unsigned)iw < (unsigned)a.wi;
            unsigned off = ok ? (unsigned)((pixbase[j] + doff) * ld + cc) * 4u : 0xffffffffu;
#ifdef SHM_ABL_SAMELINE
            off = ok ? (unsigned)(quad * 16 + (off & 0x40u)) : 0xffffffffu;      // timing only: every lane hits one line
#endif
#ifdef SHM_ABL_NOADDR
            off = (unsigned)(pixbase[j] * ld + quad * 4) * 4u;                  // timing only: no per-step address work
#endif
            u32x4 v = second ? __builtin_amdgcn_raw_buffer_load_b128(rsx2, (int)off, 0, 0)
                             : __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)off, 0, 0);
            ra[j] = __builtin_bit_cast(f32x4, v);
        }
        const unsigned wbase = (unsigned)((t_wi * a.nout) * a.K + c0) * 4u;
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            unsigned off = wrow[j] == 0xffffffffu ? 0xffffffffu : wrow[j] + wbase;
            rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)off, 0, 0));
        }
        advance();
    };
    auto sstore = [&](int buf, const f32x4 (&ra)[AR], const f32x4 (&rb)[BR]) {
#pragma unroll
        for (int j = 0; j < AR; ++j) *(f32x4*)(&As[buf][(lrow + 64 * j) * LDK + quad * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < BR; ++j) *(f32x4*)(&Bs[buf][(lrow + 64 * j) * LDK + quad * 4]) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const float* Ab = &As[buf][(wm * 64 + l31) * LDK + h * 4];
        const float* Bb = &Bs[buf][(wn * 64 + l31) * LDK + h * 4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *(const f32x4*)(Ab + i * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *(const f32x4*)(Bb + j * 32 * LDK + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
        }
    };

    // Two LDS stages, two register sets: the global loads of step s+2 are issued while step s is
    // computed and are written to LDS at the end of step s+1 (two MFMA phases of flight time).
    f32x4 ra0[AR], rb0[BR], ra1[AR], rb1[BR];
    gload(ra0, rb0);
    if (ksteps > 1) gload(ra1, rb1);
    sstore(0, ra0, rb0);
    __syncthreads();
    int s = 0;
#ifdef SHM_ABL_NOBAR
#define SHM_BAR()
#else
#define SHM_BAR() __syncthreads()
#endif
#ifdef SHM_ABL_NOLOAD
#define SHM_GLOAD(a_, b_)
#else
#define SHM_GLOAD(a_, b_) gload(a_, b_)
#endif
#ifdef SHM_ABL_NOSTORE
#define SHM_SSTORE(i_, a_, b_)
#else
#define SHM_SSTORE(i_, a_, b_) sstore(i_, a_, b_)
#endif
#ifdef SHM_SCHED_PIN
#define SHM_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define SHM_PIN()
#endif
    for (; s + 3 < ksteps; s += 2) {      // steady state: steps s, s+1 computed, s+2, s+3 fetched
        SHM_GLOAD(ra0, rb0);
        compute(0);
        SHM_PIN();
        SHM_SSTORE(1, ra1, rb1);
        SHM_BAR();
        SHM_GLOAD(ra1, rb1);
        compute(1);
        SHM_PIN();
        SHM_SSTORE(0, ra0, rb0);
        SHM_BAR();
    }
    // tail: 1..3 steps left; buf0 holds step s, (ra1, rb1) hold step s+1 if it exists
    const int left = ksteps - s;
    if (left >= 3) gload(ra0, rb0);
    compute(0);
    if (left >= 2) {
        sstore(1, ra1, rb1);
        __syncthreads();
        compute(1);
        if (left >= 3) {
            sstore(0, ra0, rb0);
            __syncthreads();
            compute(0);
        }
    }

    // epilogue: bias + LeakyReLU + store.  C/D map: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
    const bool direct = (a.os == 1);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + wm * 64 + i * 32 + row;
            if (m >= a.M) continue;
            size_t opix;
            if (direct) {
                opix = (size_t)m;
            } else {
                int ow = m % a.wg, t = m / a.wg;
                int oh = t % a.hg, n = t / a.hg;
                opix = ((size_t)n * a.ho + (oh * a.os + P.oph)) * a.wo + (ow * a.os + P.opw);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (n < a.nout) {
                    float v = acc[i][j][r];
                    if (a.bias) v += a.bias[n];
                    v = shm_lrelu(v, a.slope);
                    if (n < a.n1)
                        a.y[opix * a.ldy + n] = v;
                    else
                        a.y2[opix * a.ldy2 + (n - a.n1)] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA variant: operands go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds), no VGPR
// staging and no ds_write.  One wave-instruction fills 16 LDS rows of 64 bytes (lane l -> byte
// 16*l of the destination), so rows are unpadded; bank conflicts of the ds_read_b128 fragment
// reads are removed by an XOR swizzle applied on the SOURCE side: LDS chunk q of row r holds
// channel chunk q ^ ((r >> 2) & 3).  Out-of-image taps / tail rows use byte offset 0xffffffff:
// the descriptor's range check makes the DMA write zeros (tools/ldsdma_probe.hip).
// Three LDS stages; the DMA of step s+2 is issued right after the barrier of step s, waits are
// counted (s_waitcnt vmcnt(N)), barriers are raw s_barrier (a __syncthreads would drain vmcnt).
template <int BM, int BN, int WGM, int WGN, int NST, int BK>
__global__ __launch_bounds__(64 * WGM * WGN) void tapgemm_dma_kernel(const TapGemmArgs a) {
    static_assert(BK == 16 || BK == 32, "K step of 16 (64-byte LDS rows) or 32 (128-byte rows)");
    constexpr int NW = WGM * WGN;                    // waves per block (4 or 8)
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int RPI = 256 / BK;                    // rows per DMA instruction (1 KiB)
    constexpr int CPR = BK / 4;                      // 16-byte chunks per row
    constexpr int SWS = BK == 16 ? 2 : 1, SWM = CPR - 1;      // swizzle: chunk ^= (row >> SWS) & SWM
    constexpr int NKK = BK / 8;                      // 8-wide k groups per step
    constexpr int NA = BM / (RPI * NW), NB = BN / (RPI * NW);   // DMA instructions per wave and stage
    static_assert(NA >= 1 && NB >= 1 && BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "whole DMA instructions per wave");
    constexpr int NLD = NA + NB;
    constexpr int STAGE = (BM + BN) * BK;            // floats
    static_assert(NST >= 2 && NST <= 4, "NST stages: DMA NST-1 steps ahead");
    __shared__ __attribute__((aligned(1024))) float smem[NST * STAGE];

    const TapPhase& P = a.ph[blockIdx.z];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    // (an XCD-aware tile order -- contiguous M ranges per XCD, N tiles innermost -- was measured
    // 1 % slower: the kernel is not L2/HBM bound)
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // DMA lane mapping: instruction j of this wave covers rows wave*(BM/4)+16j .. +15
    const int drow = lane / CPR, dq = lane % CPR;
    // Per row: byte offset of the centre pixel in each source, and a bitmask of the taps that fall
    // inside the image (bit t of okm) -- the per-step address work is one add and one select.
    unsigned rowb1[NA], rowb2[NA], okm[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = wave * (BM / NW) + RPI * j + drow;
        const int m = m0 + row;
        const bool mv = m < a.M;
        const int mm = mv ? m : 0;
        const int ow = mm % a.wg, t = mm / a.wg;
        const int oh = t % a.hg, n = t / a.hg;
        const int ih0 = oh * a.is, iw0 = ow * a.is;
        const int pixbase = (n * a.hi + ih0) * a.wi + iw0;
        const int acoff = (dq ^ ((row >> SWS) & SWM)) * 4;   // swizzled channel offset inside the K step
        rowb1[j] = (unsigned)(pixbase * a.ldx + acoff) * 4u;
        rowb2[j] = (unsigned)(pixbase * a.ldx2 + acoff) * 4u;
        unsigned mk = 0;
        for (int tp = 0; tp < P.ntaps; ++tp) {
            const int ih = ih0 + P.dh[tp], iw = iw0 + P.dw[tp];
            mk |= (mv && (unsigned)ih < (unsigned)a.hi && (unsigned)iw < (unsigned)a.wi) ? (1u << tp) : 0u;
        }
        okm[j] = mk;
    }
    unsigned wrow[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = wave * (BN / NW) + RPI * j + drow;
        const int nn = n0 + row;
        wrow[j] = nn < a.nout ? (unsigned)(nn * a.K + (dq ^ ((row >> SWS) & SWM)) * 4) * 4u : 0xffffffffu;
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);

    const int ntaps = P.ntaps;
    const int nch = a.K / BK;
    const int ksteps = ntaps * nch;
    // The tap table lives in two VGPRs (lane t holds tap t) and is read with v_readlane: a scalar
    // memory load inside the K loop would share lgkmcnt with the ds_reads and force every fragment
    // wait to lgkmcnt(0) (SMEM returns out of order).
    const int tl = lane < ntaps ? lane : 0;
    const int tapoff_v = P.dh[tl] * a.wi + P.dw[tl];      // pixel displacement of tap `lane`
    const int tapw_v = P.widx[tl];                        // its weight slice
    int ld_g = 0, ld_tap = 0, ld_sub = 0, ld_c0 = 0;
    auto advance = [&]() {
        if (BK == 32) {                          // (chunk, tap): a step already covers a whole 128-B line
            if (++ld_tap == ntaps) {
                ld_tap = 0;
                ld_c0 += 32;
            }
            return;
        }
        const int nsub = (nch - ld_g) >= 2 ? 2 : 1;
        if (++ld_sub == nsub) {
            ld_sub = 0;
            if (++ld_tap == ntaps) {
                ld_tap = 0;
                ld_g += 2;
            }
        }
        ld_c0 = (ld_g + ld_sub) << 4;
    };
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto dma = [&](int stage) {
        float* sa = smem + stage * STAGE + wave * (BM / NW) * BK;
        float* sb = smem + stage * STAGE + BM * BK + wave * (BN / NW) * BK;
        const int c0 = ld_c0;
        const bool second = c0 >= a.c1;
        const int ld = second ? a.ldx2 : a.ldx;
        const int cc = second ? c0 - a.c1 : c0;
        const int t_off = __builtin_amdgcn_readlane(tapoff_v, ld_tap);
        const int t_wi = __builtin_amdgcn_readlane(tapw_v, ld_tap);
        const unsigned stepb = (unsigned)(t_off * ld + cc) * 4u;                     // wave-uniform
        const unsigned tbit = 1u << ld_tap;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            unsigned off = (okm[j] & tbit) ? (second ? rowb2[j] : rowb1[j]) + stepb : 0xffffffffu;
#ifdef SHM_ABL_FIXADDR
            off = rowb1[j];                       // timing only: constant address, no per-step work
#endif
#ifdef SHM_ABL_SAMELINE
            off = (okm[j] & tbit) ? (unsigned)(dq * 16 + (off & 0x40u)) : 0xffffffffu;      // timing only
#endif
            if (second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx2, (lds_ptr)(sa + j * 256), 16, (int)off, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sa + j * 256), 16, (int)off, 0, 0, 0);
        }
        const unsigned wbase = (unsigned)((t_wi * a.nout) * a.K + c0) * 4u;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            unsigned off = wrow[j] == 0xffffffffu ? 0xffffffffu : wrow[j] + wbase;
#if defined(SHM_ABL_FIXADDR) || defined(SHM_ABL_SAMELINE)
            off = wrow[j] == 0xffffffffu ? 0xffffffffu : (unsigned)(dq * 16);
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * 256), 16, (int)off, 0, 0, 0);
        }
#ifndef SHM_ABL_FIXADDR
        advance();
#endif
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: row = tile row (l31 + 32*i), logical chunk 2*kk+h, physical chunk ^ swizzle(row)
    const int sw = (l31 >> SWS) & SWM;
    int fo[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) fo[kk] = l31 * BK + ((2 * kk + h) ^ sw) * 4;       // floats
    auto compute = [&](int stage) {
        const float* Ab = smem + stage * STAGE + wm * WTM * BK;
        const float* Bb = smem + stage * STAGE + BM * BK + wn * WTN * BK;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            f32x4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *(const f32x4*)(Ab + i * 32 * BK + fo[kk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *(const f32x4*)(Bb + j * 32 * BK + fo[kk]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
        }
    };

    constexpr int AHEAD = NST - 1;                  // stages in flight beyond the one being computed
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
        if (t < ksteps) dma(t);
    int cur = 0, nxt = AHEAD % NST;
    for (int s = 0; s < ksteps; ++s) {
        // stage s must have landed: everything but the DMAs of the (up to AHEAD-1) stages issued after it
        const int younger = min(AHEAD - 1, ksteps - 1 - s);
        if (younger >= 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
        else if (younger == 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // all waves: stage s landed, compute(s-1) finished
        asm volatile("" ::: "memory");
#ifndef SHM_ABL_NODMA
        if (s + AHEAD < ksteps) dma(nxt);      // overwrites the buffer compute(s-1) was reading
#endif
        compute(cur);
        asm volatile("" ::: "memory");
        cur = (cur == NST - 1) ? 0 : cur + 1;
        nxt = (nxt == NST - 1) ? 0 : nxt + 1;
    }

    const bool direct = (a.os == 1);
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + wm * WTM + i * 32 + row;
            if (m >= a.M) continue;
            size_t opix;
            if (direct) {
                opix = (size_t)m;
            } else {
                int ow = m % a.wg, t = m / a.wg;
                int oh = t % a.hg, n = t / a.hg;
                opix = ((size_t)n * a.ho + (oh * a.os + P.oph)) * a.wo + (ow * a.os + P.opw);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (n < a.nout) {
                    float v = acc[i][j][r];
                    if (a.bias) v += a.bias[n];
                    v = shm_lrelu(v, a.slope);
                    s1[j] += v;
                    s2[j] += v * v;
                    if (n < a.n1)
                        a.y[opix * a.ldy + n] = v;
                    else
                        a.y2[opix * a.ldy2 + (n - a.n1)] = v;
                }
            }
        }
    }
    // InstanceNorm statistics of the tile just written: the 64 rows of a wave belong to one sample
    // (hw % 64 == 0), so one f64 atomic per (wave, column, moment).
    if (a.stats) {
        const int mw = m0 + wm * WTM;
        if (mw < a.M) {
            const int img = mw / a.hw;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
                float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (h == 0 && n < a.nout) {
                    double* dst = a.stats + ((size_t)img * a.nout + n) * 2;
                    atomicAdd(dst, (double)t1);
                    atomicAdd(dst + 1, (double)t2);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// 3x3 / stride-1 tap GEMM with an LDS halo for the A operand (forward conv and its dgrad).
//
// Block = 16 x 16 output pixels of one image (M = 256) x 128 output channels, 8 waves of 64x64.
// Per 16-channel chunk the 18 x 18 input halo is DMA'd into LDS ONCE (double buffered, fetched
// while the previous chunk's nine taps are computed); the nine taps read it through nine shifted
// fragment addresses.  Only the weight slice (128 rows x 64 B) is streamed per tap (3 stages, DMA
// two taps ahead).  Per tap a wave issues 1 DMA instruction instead of 4, and the A operand moves
// 6.4x fewer bytes.  Same LDS row format as tapgemm_dma_kernel: 64-byte rows, chunk ^= (row>>2)&3
// applied on the DMA source side; halo pixels outside the image use offset 0xffffffff (zeros).
__global__ __launch_bounds__(512) void tapgemm_halo_kernel(const TapGemmArgs a) {
    constexpr int BN = 128, WGN = 2;                  // 8 waves: 4 (M) x 2 (N)
    constexpr int HC = 18, NHR = 384;                 // halo 18 x 18 = 324 rows, padded to 24 DMA items
    constexpr int ASTG = NHR * 16, BSTG = BN * 16;    // floats per stage
    constexpr int NA = 3, NB = 1;                     // DMA instructions per wave: A per chunk, B per tap
    __shared__ __attribute__((aligned(1024))) float smem[2 * ASTG + 3 * BSTG];
    float* const sA = smem;
    float* const sB = smem + 2 * ASTG;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const TapPhase& P = a.ph[0];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    // block -> (image, patch)
    const int ppr = a.wi >> 4, ppi = (a.hi >> 4) * ppr;
    const int img = blockIdx.x / ppi, prem = blockIdx.x - img * ppi;
    const int y0 = (prem / ppr) << 4, x0 = (prem % ppr) << 4;
    const int n0 = blockIdx.y * BN;

    // ---- DMA lane constants.  A item it (0..23) = halo rows [16 it, 16 it + 16); wave w owns items w, w+8, w+16
    const int drow = lane >> 2, dq = lane & 3;
    unsigned arow1[NA], arow2[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int hrow = 16 * (wave + 8 * j) + drow;
        const int hr = hrow / HC, hc = hrow - hr * HC;
        const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
        const bool v = hrow < HC * HC && (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
        const int pix = (img * a.hi + iy) * a.wi + ix;
        const int coff = (dq ^ ((hrow >> 2) & 3)) * 4;
        arow1[j] = v ? (unsigned)(pix * a.ldx + coff) * 4u : 0xffffffffu;
        arow2[j] = v ? (unsigned)(pix * a.ldx2 + coff) * 4u : 0xffffffffu;
    }
    unsigned wrow;
    {
        const int row = wave * 16 + drow;                  // B rows [16 wave, 16 wave + 16)
        const int nn = n0 + row;
        wrow = nn < a.nout ? (unsigned)(nn * a.K + (dq ^ ((row >> 2) & 3)) * 4) * 4u : 0xffffffffu;
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);

    const int nch = a.K >> 4;
    const int ksteps = 9 * nch;
    // tap table in VGPR lanes: halo row shift (dh*18 + dw) and weight slice of tap `lane`
    const int tl = lane < 9 ? lane : 0;
    const int tapsh_v = P.dh[tl] * HC + P.dw[tl];
    const int tapw_v = P.widx[tl];

    auto dma_a = [&](int chunk) {                  // halo of 16-channel chunk `chunk` into A stage chunk & 1
        const int c0 = chunk << 4;
        const bool second = c0 >= a.c1;
        const unsigned cb = (unsigned)(second ? c0 - a.c1 : c0) * 4u;
        float* dst = sA + (chunk & 1) * ASTG + wave * 256;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const unsigned r = second ? arow2[j] : arow1[j];
            const unsigned off = r == 0xffffffffu ? r : r + cb;
            if (second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx2, (lds_ptr)(dst + j * 8 * 256), 16, (int)off, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst + j * 8 * 256), 16, (int)off, 0, 0, 0);
        }
    };
    int ld_tap = 0, ld_chunk = 0, ld_stage = 0;    // position of the next weight DMA
    auto dma_b = [&]() {
        const int t_wi = __builtin_amdgcn_readlane(tapw_v, ld_tap);
        const unsigned wbase = (unsigned)((t_wi * a.nout) * a.K + (ld_chunk << 4)) * 4u;
        const unsigned off = wrow == 0xffffffffu ? wrow : wrow + wbase;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sB + ld_stage * BSTG + wave * 256), 16, (int)off, 0, 0, 0);
        if (++ld_tap == 9) {
            ld_tap = 0;
            ++ld_chunk;
        }
        ld_stage = ld_stage == 2 ? 0 : ld_stage + 1;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing.  A: lane -> patch pixel (4 wm + 2 i + (l31 >> 4), l31 & 15), halo row of the
    // centre tap; B: as in tapgemm_dma_kernel
    int hb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) hb[i] = (4 * wm + 2 * i + (l31 >> 4) + 1) * HC + (l31 & 15) + 1;
    const int swb = (l31 >> 2) & 3;
    const int fb0 = l31 * 16 + ((0 + h) ^ swb) * 4, fb1 = l31 * 16 + ((2 + h) ^ swb) * 4;

    auto compute = [&](int chunk, int tap, int bstage) {
        const float* Ab = sA + (chunk & 1) * ASTG;
        const float* Bb = sB + bstage * BSTG + wn * 64 * 16;
        const int sh = __builtin_amdgcn_readlane(tapsh_v, tap);
        int fa[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hrow = hb[i] + sh;
            const int sw = (hrow >> 2) & 3;
            fa[i][0] = hrow * 16 + ((0 + h) ^ sw) * 4;
            fa[i][1] = hrow * 16 + ((2 + h) ^ sw) * 4;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *(const f32x4*)(Ab + fa[i][kk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = *(const f32x4*)(Bb + j * 512 + (kk ? fb1 : fb0));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
        }
    };

    // ---- pipeline.  DMA issue order per wave: A(0); B(0); B(1); then at step s: [A(chunk+1) if tap == 0]; B(s+2).
    dma_a(0);
    dma_b();
    if (ksteps > 1) dma_b();
    int tap = 0, chunk = 0, bst = 0;
    for (int s = 0; s < ksteps; ++s) {
        // B(s) (and, in order before it, the halo of this chunk) must have landed.  Issued after B(s):
        // B(s+1), preceded by the A items of step s-1 if that step opened a chunk.
        if (s + 1 < ksteps) {
            if (tap == 1 && chunk + 1 < nch)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tap == 0 && chunk + 1 < nch) dma_a(chunk + 1);      // other A stage: last read in the previous chunk
        if (s + 2 < ksteps) dma_b();
        compute(chunk, tap, bst);
        asm volatile("" ::: "memory");
        bst = bst == 2 ? 0 : bst + 1;
        if (++tap == 9) {
            tap = 0;
            ++chunk;
        }
    }

    // ---- epilogue: bias + LeakyReLU + store (+ InstanceNorm statistics)
    float s1[2], s2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int py = 4 * wm + 2 * i + (row >> 4), px = row & 15;
            const size_t opix = ((size_t)img * a.hi + (y0 + py)) * a.wi + (x0 + px);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (n < a.nout) {
                    float v = acc[i][j][r];
                    if (a.bias) v += a.bias[n];
                    v = shm_lrelu(v, a.slope);
                    s1[j] += v;
                    s2[j] += v * v;
                    if (n < a.n1)
                        a.y[opix * a.ldy + n] = v;
                    else
                        a.y2[opix * a.ldy2 + (n - a.n1)] = v;
                }
            }
        }
    }
    if (a.stats) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
            float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (h == 0 && n < a.nout) {
                double* dst = a.stats + ((size_t)img * a.nout + n) * 2;
                atomicAdd(dst, (double)t1);
                atomicAdd(dst + 1, (double)t2);
            }
        }
    }
}

static thread_local double* g_conv_stats = nullptr;     // set by shm_conv2d_in_fwd around its conv launch
static thread_local int g_conv_hw = 0;

static int launch_tapgemm(TapGemmArgs& a, int batch, int nphase, hipStream_t st, const char* who) {
    a.stats = g_conv_stats;
    a.hw = g_conv_hw;
    SHM_REQUIRE(a.K % 16 == 0 && a.K > 0, SHM_E_SHAPE, "%s: contraction channels %d must be a multiple of 16", who, a.K);
    SHM_REQUIRE(a.c1 % 16 == 0, SHM_E_SHAPE, "%s: concat split %d must be a multiple of 16", who, a.c1);
    SHM_REQUIRE(a.ldx % 4 == 0 && (a.x2 == nullptr || a.ldx2 % 4 == 0), SHM_E_SHAPE, "%s: input pitch must be a multiple of 4", who);
    SHM_REQUIRE((size_t)batch * a.hi * a.wi < (1u << 31) && (size_t)batch * a.ho * a.wo < (1u << 31), SHM_E_SHAPE, "%s: pixel count overflows int32", who);
    a.M = batch * a.hg * a.wg;
    if (a.M == 0 || a.nout == 0) return SHM_OK;
    {
        const size_t lim = 0xfffffff0ull;
        size_t xb = (size_t)batch * a.hi * a.wi * a.ldx * 4, x2b = a.x2 ? (size_t)batch * a.hi * a.wi * a.ldx2 * 4 : 0;
        size_t wb = 0;
        for (int p = 0; p < nphase; ++p)
            for (int t = 0; t < a.ph[p].ntaps; ++t) {
                size_t e = (size_t)(a.ph[p].widx[t] + 1) * a.nout * a.K * 4;
                if (e > wb) wb = e;
            }
        SHM_REQUIRE(xb < lim && x2b < lim && wb < lim, SHM_E_SHAPE, "%s: operand larger than 4 GiB (32-bit buffer offsets)", who);
        a.xbytes = (unsigned)xb;
        a.x2bytes = (unsigned)x2b;
        a.wbytes = (unsigned)wb;
    }
    static const int use_dma = getenv("SHM_TAPGEMM_REG") ? 0 : 1;
    static const int dma_small = getenv("SHM_TAPGEMM_SMALL") ? atoi(getenv("SHM_TAPGEMM_SMALL")) : 1;
    static const int dma_big = getenv("SHM_TAPGEMM_BIG") ? atoi(getenv("SHM_TAPGEMM_BIG")) : 0;
    static const int use_halo = getenv("SHM_TAPGEMM_NOHALO") ? 0 : 1;
    if (use_dma && use_halo && nphase == 1 && a.is == 1 && a.os == 1 && a.ph[0].ntaps == 9 && a.nout > 64 &&
        a.hi % 16 == 0 && a.wi % 16 == 0 && a.hg == a.hi && a.wg == a.wi) {
        bool unit = true;                      // every tap within the 1-pixel halo
        for (int t = 0; t < 9; ++t) unit = unit && a.ph[0].dh[t] >= -1 && a.ph[0].dh[t] <= 1 && a.ph[0].dw[t] >= -1 && a.ph[0].dw[t] <= 1;
        // 2 blocks of 8 waves per CU = 512 slots: below ~4 rounds the coarser (256-row) tiles lose more to
        // grid quantization than the halo reuse gains (measured: 32x32 maps 113 vs 133 TFLOP/s)
        static const int halo_min = getenv("SHM_TAPGEMM_HALO_MIN") ? atoi(getenv("SHM_TAPGEMM_HALO_MIN")) : 1024;
        const long nblk = (long)batch * (a.hi / 16) * (a.wi / 16) * shm_cdiv(a.nout, 128);
        if (unit && nblk >= halo_min) {
            dim3 grid(batch * (a.hi / 16) * (a.wi / 16), shm_cdiv(a.nout, 128), 1);
            hipLaunchKernelGGL(tapgemm_halo_kernel, grid, dim3(512), 0, st, a);
            SHM_LAUNCH_CHECK(who);
            return SHM_OK;
        }
    }
    if (use_dma) {
        auto grid1d = [&](int bm, int bn) { return dim3(shm_cdiv(a.M, bm), shm_cdiv(a.nout, bn), nphase); };
        static const int bk32 = getenv("SHM_TAPGEMM_BK32") ? atoi(getenv("SHM_TAPGEMM_BK32")) : 0;
        if (a.nout > 64 && bk32 && a.K % 32 == 0 && (a.x2 == nullptr || a.c1 % 32 == 0)) {
            if (bk32 == 2)
                hipLaunchKernelGGL((tapgemm_dma_kernel<128, 128, 2, 2, 4, 16>), grid1d(128, 128), dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((tapgemm_dma_kernel<128, 128, 2, 2, 2, 32>), grid1d(128, 128), dim3(256), 0, st, a);
        } else if (a.nout > 64 && dma_big && a.M >= 256 * 512) {
            hipLaunchKernelGGL((tapgemm_dma_kernel<256, 128, 4, 2, 3, 16>), grid1d(256, 128), dim3(512), 0, st, a);
        } else if (a.nout > 64) {
            hipLaunchKernelGGL((tapgemm_dma_kernel<128, 128, 2, 2, 3, 16>), grid1d(128, 128), dim3(256), 0, st, a);
        } else if (dma_small == 0) {
            hipLaunchKernelGGL((tapgemm_dma_kernel<256, 64, 4, 1, 2, 16>), grid1d(256, 64), dim3(256), 0, st, a);
        } else {
            hipLaunchKernelGGL((tapgemm_dma_kernel<128, 64, 2, 2, 3, 16>), grid1d(128, 64), dim3(256), 0, st, a);
        }
        SHM_LAUNCH_CHECK(who);
        return SHM_OK;
    }
    if (a.nout > 64) {
        dim3 grid(shm_cdiv(a.M, 128), shm_cdiv(a.nout, 128), nphase);
        hipLaunchKernelGGL((tapgemm_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, a);
    } else {
        dim3 grid(shm_cdiv(a.M, 256), 1, nphase);
        hipLaunchKernelGGL((tapgemm_kernel<256, 64, 4, 1>), grid, dim3(256), 0, st, a);
    }
    SHM_LAUNCH_CHECK(who);
    return SHM_OK;
}

// ------------------------------------------------------------------------------------
// [ntaps][rows][cols] -> [ntaps][cols][rows_pad]
__global__ void transpose_taps_kernel(const float* __restrict__ w, float* __restrict__ wt, int rows, int cols, int rows_pad) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const float* src = w + (size_t)t * rows * cols;
    float* dst = wt + (size_t)t * cols * rows_pad;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < rows_pad) dst[(size_t)c * rows_pad + r] = tile[threadIdx.x][i];
    }
}

extern "C" int shm_transpose_taps(const float* w, float* wt, int ntaps, int rows, int cols, int rows_pad, void* stream) {
    SHM_REQUIRE(rows_pad >= rows && ntaps > 0 && rows > 0 && cols > 0, SHM_E_SHAPE, "shm_transpose_taps: bad shape");
    dim3 grid(shm_cdiv(cols, 32), shm_cdiv(rows_pad, 32), ntaps);
    hipLaunchKernelGGL(transpose_taps_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, w, wt, rows, cols, rows_pad);
    SHM_LAUNCH_CHECK("shm_transpose_taps");
    return SHM_OK;
}

// ------------------------------------------------------------------------------------
extern "C" int shm_conv2d_fwd(const float* x, const float* x2, int c1, int ldx, int ldx2, const float* wk,
                              const float* bias, float* y, int ldy, int batch, int hi, int wi, int cin,
                              int cout, int ksize, int stride, float slope, void* stream) {
    SHM_REQUIRE(ksize == 1 || ksize == 3, SHM_E_SHAPE, "shm_conv2d_fwd: ksize %d not in {1,3}", ksize);
    SHM_REQUIRE(stride == 1 || stride == 2, SHM_E_SHAPE, "shm_conv2d_fwd: stride %d not in {1,2}", stride);
    SHM_REQUIRE(x && wk && y, SHM_E_SHAPE, "shm_conv2d_fwd: null pointer");
    TapGemmArgs a{};
    a.x = x;
    a.x2 = x2;
    a.c1 = x2 ? c1 : cin;
    a.ldx = ldx;
    a.ldx2 = ldx2;
    a.w = wk;
    a.bias = bias;
    a.y = y;
    a.y2 = nullptr;
    a.n1 = cout;
    a.ldy = ldy;
    a.ldy2 = 0;
    a.hi = hi;
    a.wi = wi;
    a.K = cin;
    int ho, wo, pt, pl;
    shm_same_pad(hi, ksize, stride, &ho, &pt);
    shm_same_pad(wi, ksize, stride, &wo, &pl);
    a.hg = a.ho = ho;
    a.wg = a.wo = wo;
    a.nout = cout;
    a.is = stride;
    a.os = 1;
    a.slope = slope;
    TapPhase& P = a.ph[0];
    P.oph = P.opw = 0;
    P.ntaps = ksize * ksize;
    for (int kh = 0; kh < ksize; ++kh)
        for (int kw = 0; kw < ksize; ++kw) {
            int t = kh * ksize + kw;
            P.dh[t] = kh - pt;
            P.dw[t] = kw - pl;
            P.widx[t] = t;
        }
    return launch_tapgemm(a, batch, 1, (hipStream_t)stream, "shm_conv2d_fwd");
}

int shm_in_finalize_internal(double* stats, int total, int hw, double eps, hipStream_t st);

extern "C" int shm_conv2d_in_fwd(const float* x, const float* x2, int c1, int ldx, int ldx2, const float* wk,
                                 const float* bias, float* y, int ldy, int batch, int hi, int wi, int cin,
                                 int cout, int ksize, int stride, float slope, double* stats, float eps,
                                 void* stream) {
    SHM_REQUIRE(stats, SHM_E_SHAPE, "shm_conv2d_in_fwd: null stats");
    int ho, wo, pt;
    shm_same_pad(hi, ksize, stride, &ho, &pt);
    shm_same_pad(wi, ksize, stride, &wo, &pt);
    const int hw = ho * wo;
    static const int fuse = getenv("SHM_NO_STATS_FUSION") ? 0 : 1;
    if (!fuse || hw % 64 != 0 || getenv("SHM_TAPGEMM_REG")) {       // tiny maps: separate statistics pass
        int r = shm_conv2d_fwd(x, x2, c1, ldx, ldx2, wk, bias, y, ldy, batch, hi, wi, cin, cout, ksize, stride, slope, stream);
        if (r) return r;
        return shm_in_stats(y, ldy, stats, batch, hw, cout, eps, stream);
    }
    int r = shm_zero(stats, (size_t)batch * cout * 2 * sizeof(double), stream);
    if (r) return r;
    g_conv_stats = stats;
    g_conv_hw = hw;
    r = shm_conv2d_fwd(x, x2, c1, ldx, ldx2, wk, bias, y, ldy, batch, hi, wi, cin, cout, ksize, stride, slope, stream);
    g_conv_stats = nullptr;
    if (r) return r;
    return shm_in_finalize_internal(stats, batch * cout, hw, (double)eps, (hipStream_t)stream);
}

// Transposed stride-2 product shared by Conv2DTranspose forward and the stride-2 dgrad:
//   out[2a+ph] = sum_{k : k = ph+pt (mod 2)} A[a + (ph+pt-k)/2] * B[k]
static void fill_s2_phases(TapGemmArgs& a, int pt, int pl) {
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            TapPhase& P = a.ph[ph * 2 + pw];
            P.oph = ph;
            P.opw = pw;
            int nt = 0;
            for (int kh = 0; kh < 3; ++kh) {
                if (((ph + pt - kh) & 1) != 0) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    if (((pw + pl - kw) & 1) != 0) continue;
                    P.dh[nt] = (ph + pt - kh) / 2;   // exact
                    P.dw[nt] = (pw + pl - kw) / 2;
                    P.widx[nt] = kh * 3 + kw;
                    ++nt;
                }
            }
            P.ntaps = nt;
        }
}

extern "C" int shm_conv2d_dgrad(const float* dy, int lddy, const float* w, float* dx, float* dx2, int n1,
                                int lddx, int lddx2, int batch, int hi, int wi, int cin, int cout,
                                int ksize, int stride, void* stream) {
    SHM_REQUIRE(ksize == 1 || ksize == 3, SHM_E_SHAPE, "shm_conv2d_dgrad: ksize %d not in {1,3}", ksize);
    SHM_REQUIRE(stride == 1 || (stride == 2 && ksize == 3), SHM_E_SHAPE, "shm_conv2d_dgrad: stride %d unsupported", stride);
    SHM_REQUIRE(dy && w && dx, SHM_E_SHAPE, "shm_conv2d_dgrad: null pointer");
    int ho, wo, pt, pl;
    shm_same_pad(hi, ksize, stride, &ho, &pt);
    shm_same_pad(wi, ksize, stride, &wo, &pl);
    TapGemmArgs a{};
    a.x = dy;
    a.x2 = nullptr;
    a.c1 = cout;
    a.ldx = lddy;
    a.w = w;            // HWIO [t][cin][cout] == [t][N=cin][K=cout]
    a.bias = nullptr;
    a.y = dx;
    a.y2 = dx2;
    a.n1 = dx2 ? n1 : cin;
    a.ldy = lddx;
    a.ldy2 = lddx2;
    a.hi = ho;
    a.wi = wo;
    a.K = cout;
    a.ho = hi;
    a.wo = wi;
    a.nout = cin;
    a.is = 1;
    a.slope = 1.f;
    if (stride == 1) {
        a.hg = hi;
        a.wg = wi;
        a.os = 1;
        TapPhase& P = a.ph[0];
        P.oph = P.opw = 0;
        P.ntaps = ksize * ksize;
        for (int kh = 0; kh < ksize; ++kh)
            for (int kw = 0; kw < ksize; ++kw) {
                int t = kh * ksize + kw;
                P.dh[t] = pt - kh;   // dx[p] = sum_t dy[p - (k - pad)] * W_t^T
                P.dw[t] = pl - kw;
                P.widx[t] = t;
            }
        return launch_tapgemm(a, batch, 1, (hipStream_t)stream, "shm_conv2d_dgrad");
    }
    SHM_REQUIRE(hi % 2 == 0 && wi % 2 == 0, SHM_E_SHAPE, "shm_conv2d_dgrad: stride 2 needs even input size");
    a.hg = hi / 2;
    a.wg = wi / 2;
    a.os = 2;
    fill_s2_phases(a, pt, pl);
    return launch_tapgemm(a, batch, 4, (hipStream_t)stream, "shm_conv2d_dgrad");
}

extern "C" int shm_conv2d_transpose_fwd(const float* x, int ldx, const float* w, const float* bias, float* y,
                                        int ldy, int batch, int hi, int wi, int cin, int cout, float slope,
                                        void* stream) {
    SHM_REQUIRE(x && w && y, SHM_E_SHAPE, "shm_conv2d_transpose_fwd: null pointer");
    // the stride-2 SAME conv that maps [2hi,2wi] back to [hi,wi] has pad_before = 0
    int ho2, wo2, pt, pl;
    shm_same_pad(2 * hi, 3, 2, &ho2, &pt);
    shm_same_pad(2 * wi, 3, 2, &wo2, &pl);
    TapGemmArgs a{};
    a.x = x;
    a.x2 = nullptr;
    a.c1 = cin;
    a.ldx = ldx;
    a.w = w;            // Keras [t][cout][cin] == [t][N=cout][K=cin]
    a.bias = bias;
    a.y = y;
    a.y2 = nullptr;
    a.n1 = cout;
    a.ldy = ldy;
    a.hi = hi;
    a.wi = wi;
    a.K = cin;
    a.hg = hi;
    a.wg = wi;
    a.ho = 2 * hi;
    a.wo = 2 * wi;
    a.nout = cout;
    a.is = 1;
    a.os = 2;
    a.slope = slope;
    fill_s2_phases(a, pt, pl);
    return launch_tapgemm(a, batch, 4, (hipStream_t)stream, "shm_conv2d_transpose_fwd");
}

// Keras Conv2DTranspose(k=2, strides=2) (SpecSeg.py:63,69,75,81): non-overlapping, every output
// phase (ph, pw) is a 1x1 product with its own tap: y[2a+ph, 2b+pw] = bias + x[a, b] . w[ph][pw].
extern "C" int shm_conv2d_transpose2x2_fwd(const float* x, int ldx, const float* w, const float* bias, float* y,
                                           int ldy, int batch, int hi, int wi, int cin, int cout, float slope,
                                           void* stream) {
    SHM_REQUIRE(x && w && y, SHM_E_SHAPE, "shm_conv2d_transpose2x2_fwd: null pointer");
    TapGemmArgs a{};
    a.x = x;
    a.x2 = nullptr;
    a.c1 = cin;
    a.ldx = ldx;
    a.w = w;            // Keras [2][2][cout][cin] == [t][N=cout][K=cin]
    a.bias = bias;
    a.y = y;
    a.y2 = nullptr;
    a.n1 = cout;
    a.ldy = ldy;
    a.hi = hi;
    a.wi = wi;
    a.K = cin;
    a.hg = hi;
    a.wg = wi;
    a.ho = 2 * hi;
    a.wo = 2 * wi;
    a.nout = cout;
    a.is = 1;
    a.os = 2;
    a.slope = slope;
    for (int p = 0; p < 4; ++p) {
        TapPhase& P = a.ph[p];
        P.oph = p >> 1;
        P.opw = p & 1;
        P.ntaps = 1;
        P.dh[0] = P.dw[0] = 0;
        P.widx[0] = p;
    }
    return launch_tapgemm(a, batch, 4, (hipStream_t)stream, "shm_conv2d_transpose2x2_fwd");
}
