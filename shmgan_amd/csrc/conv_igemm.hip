// Implicit-GEMM convolution family on v_mfma_f32_32x32x2_f32 (exact fp32) and, for the bf16 path,
// v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate) -- gfx950.
//
// One kernel ("tap GEMM") serves Conv2D forward (k=1/3, stride 1/2), its input-gradient
// (stride 1: flipped taps; stride 2: four output phases) and Conv2DTranspose forward
// (= input-gradient of the stride-2 conv):
//
//     out[pix(m), n] = act( bias[n] + sum_{tap} sum_{k<K} A[src(m, tap), k] * B[tap][n][k] )
//
//   M = batch * grid_h * grid_w output positions of one phase, N = output channels,
//   K = channels of the A tensor (multiple of 16 fp32 / 32 bf16).  A is NHWC (optionally the
//   channel concat of two tensors), B is [tap][N][K] (K contiguous), so both operands are staged
//   as 64-byte rows (16 floats or 32 bf16) and each lane feeds four consecutive f32 MFMAs (or one
//   bf16 MFMA) from one 16-byte LDS read (the k order inside a group is permuted identically for
//   A and B, which a dot product does not see).
//
// Block = 4 waves of 64x64 outputs each (2x2 MFMA tiles), block tile 128x128 (Cout > 64) or
// 128x64; operands go HBM/L2 -> LDS by DMA three stages deep (tapgemm_dma_kernel), or through an
// 18x18 LDS halo for unit-stride 3x3 layers (tapgemm_halo_kernel).
#include "tapgemm.h"

#include <stdlib.h>
#include <type_traits>

// which part of a split output a channel belongs to, its channel index inside the part and the part's channel count
__device__ __forceinline__ int gsum_part(const TapGemmArgs& a, int n, int& nl, int& pc) {
    const int p = n < a.n1 ? 0 : 1;
    nl = p ? n - a.n1 : n;
    pc = p ? a.nout - a.n1 : a.n1;
    return p;
}

// value of the activation-typed tensor `aux` (float or bf16) as float
template <typename T>
__device__ __forceinline__ float gsum_aux(const void* aux, size_t idx) {
    return (float)((const T*)aux)[idx];
}

// LDS-staged (bf16) epilogues: a lane holds eight consecutive channels of one pixel as stored (v) and loads the same eight of
// aux (16 bytes); per-lane partial sums over the rows the lane visits
__device__ __forceinline__ void gsum_wide_accum(const u32x4& v, const u32x4& av, float (&t1)[8], float (&t2)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v0 = __uint_as_float(v[e] << 16), v1 = __uint_as_float(v[e] & 0xffff0000u);
        const float a0 = __uint_as_float(av[e] << 16), a1 = __uint_as_float(av[e] & 0xffff0000u);
        t1[2 * e] += v0;
        t1[2 * e + 1] += v1;
        t2[2 * e] += v0 * a0;
        t2[2 * e + 1] += v1 * a1;
    }
}

// CW = 4 (a wave tile of 32 channels: lane = 4 rr + ch): reduce-scatter of the sixteen per-lane sums over the sixteen lanes rr that
// share a channel group -- fifteen shuffles instead of 64, no values carried across patches -- after which lane (rr, ch) holds the
// wave's total of ONE (moment, channel) pair: moment rr >> 3, channel 8 ch + (rr & 7).  One 64-lane atomic instruction per call.
__device__ __forceinline__ float gsum_scatter16(float (&t1)[8], float (&t2)[8], int lane) {
    float v[16];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] = t1[e];
        v[8 + e] = t2[e];
    }
    // step s (lane bit 5, 4, 3, 2): keep the half of the remaining values selected by that bit, add the partner's copy of them
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int half = 8 >> s, bit = 32 >> s;
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int e = 0; e < half; ++e) {
            const float keep = up ? v[half + e] : v[e];
            const float send = up ? v[e] : v[half + e];
            v[e] = keep + __shfl_xor(send, bit, 64);
        }
    }
    return v[0];          // value index = lane >> 2 (step s fixes index bit 3 - s from lane bit 5 - s)
}

// ... combined over the lanes that hold the same channels (lane % CW equal) and added to dst[(channel) * 2 + {0, 1}]
template <int CW>
__device__ __forceinline__ void gsum_wide_flush(float (&t1)[8], float (&t2)[8], int lane, double* dst) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = CW; o < 64; o <<= 1) {
            t1[e] += __shfl_xor(t1[e], o, 64);
            t2[e] += __shfl_xor(t2[e], o, 64);
        }
    }
    if (lane < CW && dst) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            atomicAdd(dst + 2 * e, (double)t1[e]);
            atomicAdd(dst + 2 * e + 1, (double)t2[e]);
        }
    }
}

// One 16-byte fragment per operand tile: four f32 MFMAs (K = 2 each) or one bf16 MFMA (K = 16).
template <typename T, int TM, int TN>
__device__ __forceinline__ void tap_mfma(const f32x4 (&av)[TM], const f32x4 (&bv)[TN], f32x16 (&acc)[TM][TN]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv[j]),
                                                                   acc[i][j], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA variant: operands go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds), no VGPR
// staging and no ds_write.  One wave-instruction fills 16 LDS rows of 64 bytes (lane l -> byte
// 16*l of the destination), so rows are unpadded; bank conflicts of the ds_read_b128 fragment
// reads are removed by an XOR swizzle applied on the SOURCE side: LDS chunk q of row r holds
// channel chunk q ^ ((r >> 2) & 3).  Out-of-image taps / tail rows use byte offset 0xffffffff:
// the descriptor's range check makes the DMA write zeros (tools/probes/ldsdma_probe.hip).
// Three LDS stages; the DMA of step s+2 is issued right after the barrier of step s, waits are
// counted (s_waitcnt vmcnt(N)), barriers are raw s_barrier (a __syncthreads would drain vmcnt).
// T = float or bf16_t.  BK counts 4-byte words per LDS row (16 -> 64-byte rows); a K step covers
// BKE = BK*4/sizeof(T) channels.
template <typename T, typename TO, int BM, int BN, int WGM, int WGN, int NST, int BK>
// (eight-wave blocks with element-store epilogues: two blocks per CU fit in LDS, i.e. four waves per SIMD -- the second launch bound
// keeps them at 128 VGPRs, where hipcc left to itself lands between 121 and 155 depending on the epilogue code around the loop)
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN == 8 && sizeof(TO) == 4) ? 4 : 1) void tapgemm_dma_kernel(const TapGemmArgs a) {
    static_assert(BK == 16 || BK == 32, "K step of 16 words (64-byte LDS rows) or 32 (128-byte rows)");
    constexpr int ESZ = sizeof(T);
    constexpr int BKE = BK * 4 / ESZ;                // channels per K step
    constexpr int CHE = 16 / ESZ;                    // channels per 16-byte chunk
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
    // 1 % slower in fp32 (round 1) and 0.6 % slower on the whole bf16 step (round 2): the kernel is not L2/HBM bound)
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
        const int acoff = (dq ^ ((row >> SWS) & SWM)) * CHE;   // swizzled channel offset inside the K step
        rowb1[j] = (unsigned)(pixbase * a.ldx + acoff) * (unsigned)ESZ;
        rowb2[j] = (unsigned)(pixbase * a.ldx2 + acoff) * (unsigned)ESZ;
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
        wrow[j] = nn < a.nout ? (unsigned)(nn * a.K + (dq ^ ((row >> SWS) & SWM)) * CHE) * (unsigned)ESZ : 0xffffffffu;
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);

    const int ntaps = P.ntaps;
    const int nch = a.K / BKE;
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
                ld_c0 += BKE;
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
        ld_c0 = (ld_g + ld_sub) * BKE;
    };
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // the two pixel pitches as opaque scalars: hipcc otherwise re-reads the selected one from the kernel arguments in every K step -- a
    // scalar memory load whose s_waitcnt lgkmcnt(0) also drains the wave's ds_reads (found in the ISA of the 256 x 128 tile)
    int ldx_s = a.ldx, ldx2_s = a.ldx2;
    asm volatile("" : "+s"(ldx_s), "+s"(ldx2_s));
    auto dma = [&](int stage) {
        float* sa = smem + stage * STAGE + wave * (BM / NW) * BK;
        float* sb = smem + stage * STAGE + BM * BK + wave * (BN / NW) * BK;
        const int c0 = ld_c0;
        const bool second = c0 >= a.c1;
        const int ld = second ? ldx2_s : ldx_s;
        const int cc = second ? c0 - a.c1 : c0;
        const int t_off = __builtin_amdgcn_readlane(tapoff_v, ld_tap);
        const int t_wi = __builtin_amdgcn_readlane(tapw_v, ld_tap);
        const unsigned stepb = (unsigned)(t_off * ld + cc) * (unsigned)ESZ;          // wave-uniform
        const unsigned tbit = 1u << ld_tap;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            unsigned off = (okm[j] & tbit) ? (second ? rowb2[j] : rowb1[j]) + stepb : 0xffffffffu;
            if constexpr (abl::fixaddr) off = rowb1[j];                       // timing only: constant address, no per-step work
            if constexpr (abl::sameline) off = (okm[j] & tbit) ? (unsigned)(dq * 16 + (off & 0x40u)) : 0xffffffffu;      // timing only
            if (second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx2, (lds_ptr)(sa + j * 256), 16, (int)off, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sa + j * 256), 16, (int)off, 0, 0, 0);
        }
        const unsigned wbase = (unsigned)((t_wi * a.nout) * a.K + c0) * (unsigned)ESZ;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            unsigned off = wrow[j] == 0xffffffffu ? 0xffffffffu : wrow[j] + wbase;
            if constexpr (abl::fixaddr || abl::sameline) off = wrow[j] == 0xffffffffu ? 0xffffffffu : (unsigned)(dq * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * 256), 16, (int)off, 0, 0, 0);
        }
        if constexpr (!abl::fixaddr) advance();
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
            tap_mfma<T, TM, TN>(av, bv, acc);
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
        SHM_LDS_BARRIER();          // all waves: stage s landed, compute(s-1) finished
        asm volatile("" ::: "memory");
        if constexpr (!abl::nodma)
            if (s + AHEAD < ksteps) dma(nxt);      // overwrites the buffer compute(s-1) was reading
        compute(cur);
        asm volatile("" ::: "memory");
        cur = (cur == NST - 1) ? 0 : cur + 1;
        nxt = (nxt == NST - 1) ? 0 : nxt + 1;
    }

    const bool direct = (a.os == 1);
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
    // the bias of the lane's columns, once: read inside the store loops it is re-fetched per element (the stores may alias it for
    // all hipcc knows) and every fetch waits with vmcnt(0), i.e. for the stores of the element before as well -- the epilogue
    // became a chain of store round trips
    float bj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + l31;
        bj[j] = (a.bias && n < a.nout) ? a.bias[n] : 0.f;
        // consume the value here, in straight-line code: first used inside the exec-masked element blocks below, hipcc's waitcnt
        // pass keeps the load "pending" along the skipped paths and puts s_waitcnt vmcnt(0) -- a drain of the stores -- in
        // front of every element
        asm volatile("" : "+v"(bj[j]));
    }
    // bf16 outputs (round 2): as in the halo kernels the wave's tile goes through LDS (free once every wave is past its last
    // fragment read) and leaves as 16-byte stores -- the accumulator layout gives a lane one 2-byte element per row, i.e.
    // TM*TN*16 two-byte store instructions per wave.  Works for the strided (four-phase) outputs too: a pixel's channels are
    // contiguous whatever the pixel stride.
    constexpr bool kWide = sizeof(TO) == 2 && WTM * WTN * 2 * NW <= NST * STAGE * 4;
    const bool wide = kWide && (a.nout % 8 == 0) && (a.n1 % 8 == 0) && (a.ldy % 8 == 0) && (((size_t)a.y & 15) == 0) &&
                      (a.y2 == nullptr || ((a.ldy2 % 8 == 0) && (((size_t)a.y2 & 15) == 0)));
    if constexpr (kWide) if (wide) {
        constexpr int CW = WTN / 8;                      // 16-byte chunks per tile row
        constexpr int RPW = 64 / CW;                     // tile rows per store instruction
        __syncthreads();
        unsigned short* tile = (unsigned short*)smem + wave * (WTM * WTN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const bool mv = m0 + wm * WTM + row < a.M;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WTN + j * 32 + l31;
                    float v = acc[i][j][r] + bj[j];
                    const TO vo = (TO)shm_lrelu(v, a.slope);
                    v = (mv && n < a.nout) ? (float)vo : 0.f;          // statistics of the value as stored
                    s1[j] += v;
                    s2[j] = __builtin_fmaf(v, v, s2[j]);          // (an explicit fma: left to hipcc, one instantiation contracts and another does not)
                    const int col = j * 32 + l31;
                    tile[row * WTN + ((((col >> 3) ^ (row & (CW - 1))) << 3) | (col & 7))] = __builtin_bit_cast(unsigned short, vo);
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // same-wave LDS hand-off
        const int rr = lane / CW, ch = lane % CW;
        const int n = n0 + wn * WTN + ch * 8;
        int gnl, gpc;
        const int gp = gsum_part(a, n, gnl, gpc);
        const bool gs = a.gred[gp] != nullptr && n < a.nout;         // per lane: its eight channels lie in one part
        const unsigned short* gaux = (const unsigned short*)a.gaux[gp] + gnl;
        float t1[8], t2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t1[e] = t2[e] = 0.f;
#pragma unroll
        for (int it = 0; it < WTM / RPW; ++it) {
            const int row = it * RPW + rr;
            const u32x4 v = *(const u32x4*)(tile + row * WTN + ((ch ^ (row & (CW - 1))) << 3));
            const int m = m0 + wm * WTM + row;
            if (m < a.M && n < a.nout) {
                size_t opix;
                if (direct) {
                    opix = (size_t)m;
                } else {
                    const int ow = m % a.wg, t = m / a.wg;
                    const int oh = t % a.hg, ni = t / a.hg;
                    opix = ((size_t)ni * a.ho + (oh * a.os + P.oph)) * a.wo + (ow * a.os + P.opw);
                }
                if (n < a.n1)
                    *(u32x4*)((unsigned short*)a.y + opix * a.ldy + n) = v;
                else
                    *(u32x4*)((unsigned short*)a.y2 + opix * a.ldy2 + (n - a.n1)) = v;
                if (gs) gsum_wide_accum(v, *(const u32x4*)(gaux + opix * a.ldgaux[gp]), t1, t2);
            }
        }
        if (a.gred[0] || a.gred[1]) {                      // wave-uniform
            const int mw = m0 + wm * WTM;
            const int img = mw / a.hw;
            const int slot = ((mw - img * a.hw) / WTM) % a.gslots;
            double* dst = (gs && mw < a.M) ? a.gred[gp] + ((size_t)slot * a.gbatch * gpc + (size_t)img * gpc + gnl) * 2 : nullptr;
            gsum_wide_flush<CW>(t1, t2, lane, dst);
        }
    }
    // narrow path, gsum.  The 32 columns of a (wave, j) group lie in one output part (n1 % 32 == 0, checked by the launcher), so
    // "this group takes sums", its aux tensor and pitch are scalars: the sixteen aux loads of a 32 x 32 tile are issued back to
    // back in front of the tile's stores (a per-element conditional load made hipcc wait for every load AND the store before it).
    const bool gs_any = a.gred[0] != nullptr || a.gred[1] != nullptr;
    auto out_pix = [&](int m) -> size_t {
        if (direct) return (size_t)m;
        const int ow = m % a.wg, t = m / a.wg;
        const int oh = t % a.hg, n = t / a.hg;
        return ((size_t)n * a.ho + (oh * a.os + P.oph)) * a.wo + (ow * a.os + P.opw);
    };
    // Element stores (and the gsum aux loads) without per-element address arithmetic: the rows of a lane's 32 x 32 accumulator tile are
    // GEMM rows mb + 8 g + 4 h + e (g = r >> 2, e = r & 3, mb a multiple of 32), so when the phase grid is a multiple of 8 pixels wide
    // the output pixel of a row is a SCALAR -- (n, oh, ow) of row mb + 8 g, four scalar decompositions per tile -- plus e and 4 h pixel
    // steps: one per-lane address register for the whole wave tile, everything else in the instruction's scalar offset.  (Per
    // element it was a 64-bit address from two integer divisions: ~40 VALU instructions, 64 elements per lane.)  Needs outputs below
    // 4 GiB (scalar descriptors) and every 32-column group inside one output part.
    const bool fastep = !abl::nostore && a.ybytes != 0 && (a.y2 == nullptr || (a.y2bytes != 0 && a.n1 % 32 == 0)) && (direct || a.wg % 8 == 0) &&
                        a.M % 8 == 0;
    if (!wide && fastep) {
        unsigned sp[TM][4];                // scalar: output pixel of GEMM row mb + 8 g of tile i (one decomposition per tile, then steps of 8)
        bool sv[TM][4];                    // ... and whether that row group exists (M % 8 == 0)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = __builtin_amdgcn_readfirstlane(m0 + wm * WTM + i * 32);
            int ow = mb % a.wg, t = mb / a.wg;
            int oh = t % a.hg, ni = t / a.hg;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sv[i][g] = mb + 8 * g < a.M;
                sp[i][g] = direct ? (unsigned)(mb + 8 * g) : (unsigned)((ni * a.ho + (oh * a.os + P.oph)) * a.wo + (ow * a.os + P.opw));
                ow += 8;
                if (ow >= a.wg) {          // wg % 8 == 0: a step of 8 ends exactly on the row end
                    ow = 0;
                    if (++oh == a.hg) {
                        oh = 0;
                        ++ni;
                    }
                }
            }
        }
        auto elem_stores = [&](auto gsx) {
            constexpr bool GSX = decltype(gsx)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = __builtin_amdgcn_readfirstlane(n0 + wn * WTN + j * 32);
            const int gp = nb < a.n1 ? 0 : 1;
            const bool on = GSX && a.gred[gp] != nullptr && nb < a.nout;
            const int n = nb + l31;
            const int nl = n - (gp ? a.n1 : 0);
            const int pc = gp ? a.nout - a.n1 : a.n1;
            const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(gp ? a.y2 : a.y, 0, gp ? a.y2bytes : a.ybytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a.gaux[gp], 0, on ? 0xfffffff0u : 0u, 0x00020000);
            const unsigned ldyb = (unsigned)(gp ? a.ldy2 : a.ldy) * (unsigned)sizeof(TO), ldab = (unsigned)a.ldgaux[gp] * (unsigned)sizeof(T);
            const unsigned lanepix = (unsigned)(4 * h * a.os);
            const unsigned yo = lanepix * ldyb + (unsigned)(n < a.nout ? nl : 0) * (unsigned)sizeof(TO);
            const unsigned ao = lanepix * ldab + (unsigned)(n < a.nout ? nl : 0) * (unsigned)sizeof(T);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                [[maybe_unused]] float q[GSX ? 16 : 1];
#pragma unroll
                for (int r = 0; r < (GSX ? 16 : 1); ++r) q[r] = 0.f;
                if constexpr (GSX) if (on) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (sv[i][g]) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned so = (sp[i][g] + (unsigned)(e * a.os)) * ldab;
                                if constexpr (sizeof(T) == 4)
                                    q[4 * g + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, ao, so, 0));
                                else
                                    q[4 * g + e] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsa, ao, so, 0) << 16);
                            }
                        }
                }
                if (n < a.nout) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (sv[i][g]) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = 4 * g + e;
                                float v = acc[i][j][r] + bj[j];
                                const TO vo = (TO)shm_lrelu(v, a.slope);
                                v = (float)vo;                       // statistics of the value as stored
                                s1[j] += v;
                                if constexpr (GSX)
                                    s2[j] += v * q[r];
                                else
                                    s2[j] = __builtin_fmaf(v, v, s2[j]);
                                const unsigned so = (sp[i][g] + (unsigned)(e * a.os)) * ldyb;
                                if constexpr (sizeof(TO) == 4)
                                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vo), rsy, yo, so, 0);
                                else
                                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, vo), rsy, yo, so, 0);
                            }
                        }
                }
            }
            const int mw = m0 + wm * WTM;
            if (on && mw < a.M) {
                const int img = mw / a.hw;
                const int slot = ((mw - img * a.hw) / WTM) % a.gslots;
                const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
                const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
                if (h == 0 && n < a.nout) {
                    double* dst = a.gred[gp] + ((size_t)slot * a.gbatch * pc + (size_t)img * pc + nl) * 2;
                    atomicAdd(dst, (double)t1);
                    atomicAdd(dst + 1, (double)t2);
                }
            }
        }
        };
        if (gs_any)
            elem_stores(std::true_type{});
        else
            elem_stores(std::false_type{});
    }
    if (!wide && gs_any && !fastep) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = __builtin_amdgcn_readfirstlane(n0 + wn * WTN + j * 32);
            const int gp = nb < a.n1 ? 0 : 1;
            const bool on = a.gred[gp] != nullptr && nb < a.nout;
            const int n = nb + l31;
            const int nl = n - (gp ? a.n1 : 0);
            const int pc = gp ? a.nout - a.n1 : a.n1;
            const T* gaux = (const T*)a.gaux[gp] + (n < a.nout ? nl : 0);
            const size_t ldg = (size_t)a.ldgaux[gp];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float q[16];
                if (on) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        q[r] = m < a.M ? (float)gaux[out_pix(m) * ldg] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) q[r] = 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m >= a.M || n >= a.nout) continue;
                    const size_t opix = out_pix(m);
                    float v = acc[i][j][r] + bj[j];
                    const TO vo = (TO)shm_lrelu(v, a.slope);
                    v = (float)vo;
                    s1[j] += v;
                    s2[j] += v * q[r];
                    if (n < a.n1)
                        ((TO*)a.y)[opix * a.ldy + n] = vo;
                    else
                        ((TO*)a.y2)[opix * a.ldy2 + (n - a.n1)] = vo;
                }
            }
            const int mw = m0 + wm * WTM;
            if (on && mw < a.M) {
                const int img = mw / a.hw;
                const int slot = ((mw - img * a.hw) / WTM) % a.gslots;
                const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
                const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
                if (h == 0 && n < a.nout) {
                    double* dst = a.gred[gp] + ((size_t)slot * a.gbatch * pc + (size_t)img * pc + nl) * 2;
                    atomicAdd(dst, (double)t1);
                    atomicAdd(dst + 1, (double)t2);
                }
            }
        }
    }
    if (!wide && !gs_any && !fastep) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + wm * WTM + i * 32 + row;
            if (m >= a.M) continue;
            const size_t opix = out_pix(m);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (n < a.nout) {
                    float v = acc[i][j][r] + bj[j];
                    const TO vo = (TO)shm_lrelu(v, a.slope);
                    v = (float)vo;                       // statistics of the value as stored
                    s1[j] += v;
                    s2[j] = __builtin_fmaf(v, v, s2[j]);          // (an explicit fma: left to hipcc, one instantiation contracts and another does not)
                    if (n < a.n1)
                        ((TO*)a.y)[opix * a.ldy + n] = vo;
                    else
                        ((TO*)a.y2)[opix * a.ldy2 + (n - a.n1)] = vo;
                }
            }
        }
    }
    }
    // InstanceNorm statistics of the tile just written: the 64 rows of a wave belong to one sample
    // (hw % 64 == 0), so one f64 atomic per (wave, column, moment).
    // InstanceNorm statistics of the tile just written: the 64 rows of a wave belong to one sample
    // (hw % 64 == 0), so one f64 atomic per (wave, column, moment).  (Combining the row-waves of a block
    // through LDS first was measured: the two extra block barriers cost more than the atomics they save,
    // -3.5 % fp32 / -12 % bf16 on this kernel.)
    if (a.stats) {
        const int mw = m0 + wm * WTM;
        if (mw < a.M) {
            const int img = mw / a.hw;
            const int slot = ((mw - img * a.hw) / WTM) % a.stats_slots;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
                float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (h == 0 && n < a.nout) {
                    double* dst = a.stats + (size_t)slot * a.stats_stride + ((size_t)img * a.nout + n) * 2;
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
// PH = patch height (16 or 8 pixel rows of 16): M = PH*16 rows, PH/4 row-waves.  PH = 8 halves the A stages
// (3 four-wave blocks per CU instead of 2 eight-wave ones: smaller barrier groups) at 11 % more halo traffic.
// ST (round 2): the nine taps of a chunk are unrolled, which makes every fragment address a patch- and chunk-independent
// register (one per (tap, tile); the second k group is an XOR, the B stage an immediate) -- no address arithmetic between the
// barrier and the first ds_read of a K step -- and lets the halo use the conflict-free swizzle ((R >> 1) + R / 18) & 3 that
// cost 3 % when its arithmetic sat on that path.
// TM = 32-row MFMA tiles per wave along M (2: wave tile 64 pixels x 64 channels; 4, static taps only: 128 x 64 -- half the waves,
// six fragment reads per eight MFMAs instead of four per four, twice the MFMAs per barrier: the bf16 form, whose K step is 8x shorter).
// GS: the gsum epilogue (input-gradient launches, see TapGemmArgs) -- an instantiation of its own, so that the forward kernels
// carry none of its code or registers.
// NM: "norm" -- one source is the UN-normalised activation of an InstanceNorm block (TapGemmArgs::nt): every wave applies
// shm_in_norm to the halo items it DMA'd itself, in LDS, once they have landed and before the barrier that opens the chunk -- the
// stand-alone normalisation pass (a read and a write of the whole activation) is gone, for 3 ds_read_b128 + 4 fma + 1 ds_write_b128 per
// 1 KiB item and 2304 (fp32) MFMAs.  Out-of-image halo pixels were DMA'd as zeros and are left alone: zero padding of the
// NORMALISED tensor, as the layer defines it.  The (mean, inv, beta) planes of the block's image sit in LDS (3 x ntc floats).
// NM = 0: none; 1: SHM_NORM_EXACT (above); 2: SHM_NORM_SCALED (TapGemmArgs: per-sample weights and bias rows, `ring` over the out-of-image entries).
template <typename T, typename TO, int BN, int PH = 16, bool ST = false, int TM = 2, bool GS = false, int NM = 0>
__global__ __launch_bounds__(BN * PH / (2 * TM), ST ? BN * PH / (256 * TM) : 1) void tapgemm_halo_kernel(const TapGemmArgs a) {
    static_assert(TM == 2 || (TM == 4 && ST), "four M tiles per wave: static-tap form only");
    static_assert(!NM || (ST && TM == 2 && !GS), "norm: static-tap forward form");
    constexpr int ESZ = sizeof(T), CHE = 16 / ESZ, BKE = 64 / ESZ;      // channels per 16-byte chunk / per 64-byte row
    constexpr int WGM = PH / (2 * TM), WGN = BN / 64, NW = WGM * WGN;   // waves: PH / (2 TM) (M) x (BN/64) (N)
    constexpr int HC = 18, NIT = PH == 16 ? 24 : 12;  // halo (PH+2) x 18 rows, padded to NIT DMA items of 16 rows
    constexpr int NHR = NIT * 16;
    constexpr int ASTG = NHR * 16, BSTG = BN * 16;    // floats per stage
    constexpr int NA = NIT / NW, NB = (BN / 16) / NW; // DMA instructions per wave: A per chunk, B per tap
    static_assert(NIT % NW == 0 && (BN / 16) % NW == 0, "DMA items divide over the waves");
    __shared__ __attribute__((aligned(1024))) float smem[2 * ASTG + 3 * BSTG];
    __shared__ __attribute__((aligned(1024))) float snt[NM ? SHM_NT_PLANES * SHM_NT_MAXC : 4];
    float* const sA = smem;
    float* const sB = smem + 2 * ASTG;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const TapPhase& P = a.ph[0];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    // block -> (image, patch)
    const int ppr = a.wi >> 4, ppi = (a.hi / PH) * ppr;
    const int img = blockIdx.x / ppi, prem = blockIdx.x - img * ppi;
    const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
    const int n0 = blockIdx.y * BN;
    // NM = 2: only a patch on the image border reads the table (its `ring` plane) -- an interior block skips the table, the barrier that
    // publishes it and every norm_a (in bf16 the extra DMA round trip in the prologue is 10-25 % of a block's life)
    [[maybe_unused]] const bool nm_table = NM == 1 || (NM == 2 && (y0 == 0 || y0 + PH == a.hi || x0 == 0 || x0 + 16 == a.wi));      // block-uniform

    // ---- DMA lane constants.  A item it (0..23) = halo rows [16 it, 16 it + 16); wave w owns items w, w+NW, ...
    const int drow = lane >> 2, dq = lane & 3;
    unsigned arow1[NA], arow2[NA];
    // NM: c = first channel (within its 64-byte row) of the lane's 16 bytes of item j: c for an image pixel, -1 - c for a halo pixel
    // outside the image, INT_MIN for the unused tail rows of the last item
    [[maybe_unused]] int nmv[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int hrow = 16 * (wave + NW * j) + drow;
        const int hr = hrow / HC, hc = hrow - hr * HC;
        const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
        const bool v = hrow < (PH + 2) * HC && (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
        const int pix = (img * a.hi + iy) * a.wi + ix;
        const int coff = (dq ^ (ST ? ((hrow >> 1) + hr) & 3 : (hrow >> 2) & 3)) * CHE;
        arow1[j] = v ? (unsigned)(pix * a.ldx + coff) * (unsigned)ESZ : 0xffffffffu;
        arow2[j] = v ? (unsigned)(pix * a.ldx2 + coff) * (unsigned)ESZ : 0xffffffffu;
        nmv[j] = v ? coff : hrow < (PH + 2) * HC ? -1 - coff : (int)0x80000000;
    }
    unsigned wrow[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = (wave + NW * j) * 16 + drow;       // B item wave + NW j = rows [16 item, 16 item + 16)
        const int nn = n0 + row;
        wrow[j] = nn < a.nout ? (unsigned)(nn * a.K + (dq ^ ((row >> 2) & 3)) * CHE) * (unsigned)ESZ : 0xffffffffu;
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);

    const int nch = a.K / BKE;
    const int ksteps = 9 * nch;
    // tap table in VGPR lanes: halo row shift (dh*18 + dw) and weight slice of tap `lane`
    const int tl = lane < 9 ? lane : 0;
    const int tapsh_v = P.dh[tl] * HC + P.dw[tl];
    const int tapw_v = P.widx[tl];

    auto dma_a = [&](int chunk) {                  // halo of 64-byte channel chunk `chunk` into A stage chunk & 1
        const int c0 = chunk * BKE;
        const bool second = c0 >= a.c1;
        const unsigned cb = (unsigned)(second ? c0 - a.c1 : c0) * (unsigned)ESZ;
        float* dst = sA + (chunk & 1) * ASTG + wave * 256;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const unsigned r = second ? arow2[j] : arow1[j];
            const unsigned off = r == 0xffffffffu ? r : r + cb;
            if (second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx2, (lds_ptr)(dst + j * NW * 256), 16, (int)off, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst + j * NW * 256), 16, (int)off, 0, 0, 0);
        }
    };
    int ld_tap = 0, ld_chunk = 0, ld_stage = 0;    // position of the next weight DMA
    auto dma_b = [&]() {
        const int t_wi = __builtin_amdgcn_readlane(tapw_v, ld_tap);
        const unsigned wbase = (unsigned)((t_wi * a.nout) * a.K + ld_chunk * BKE) * (unsigned)ESZ + (NM == 2 ? (unsigned)img * a.wimg : 0u);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const unsigned off = wrow[j] == 0xffffffffu ? wrow[j] : wrow[j] + wbase;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sB + ld_stage * BSTG + (wave + NW * j) * 256), 16, (int)off, 0, 0, 0);
        }
        if (++ld_tap == 9) {
            ld_tap = 0;
            ++ld_chunk;
        }
        ld_stage = ld_stage == 2 ? 0 : ld_stage + 1;
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing.  A: lane -> patch pixel (4 wm + 2 i + (l31 >> 4), l31 & 15), halo row of the
    // centre tap; B: as in tapgemm_dma_kernel
    // Swizzle (R >> 2) & 3 on the halo row index R: because halo rows start at arbitrary offsets, a third of the
    // fragment reads see a 2-way bank conflict (SQ_LDS_BANK_CONFLICT).  The conflict-free function for this access
    // pattern is ((R >> 1) + R / 18) & 3 (exhaustive check over taps and lane groups); it was measured 3 % SLOWER in
    // both dtypes -- its per-tap address work sits on the barrier -> first ds_read critical path, the conflicts do not.
    int hb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) hb[i] = (2 * TM * wm + 2 * i + (l31 >> 4) + 1) * HC + (l31 & 15) + 1;
    const int swb = (l31 >> 2) & 3;
    const int fb0 = l31 * 16 + ((0 + h) ^ swb) * 4, fb1 = l31 * 16 + ((2 + h) ^ swb) * 4;

    [[maybe_unused]] f32x4 abl_frag = {0.f, 0.f, 0.f, 0.f};
    if constexpr (abl::nolds) abl_frag = *(const f32x4*)(sA + lane * 4);
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
            if constexpr (abl::nolds) {
                // timing only: fragments from registers (one read per block), MFMAs + barriers + DMA unchanged
                for (int i = 0; i < 2; ++i) av[i] = abl_frag;
                for (int j = 0; j < 2; ++j) bv[j] = abl_frag;
                asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(bv[0]), "+v"(bv[1]));
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) av[i] = *(const f32x4*)(Ab + fa[i][kk]);
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[j] = *(const f32x4*)(Bb + j * 512 + (kk ? fb1 : fb0));
            }
            if constexpr (TM == 2) tap_mfma<T, 2, 2>(av, bv, acc);
        }
    };

    // NM: normalise this wave's own items of the A stage of `chunk` in place (they have landed: the caller waited)
    // (the plane pitch as an opaque scalar: re-read from the kernel arguments inside the tap loop it is a scalar memory load whose
    // s_waitcnt lgkmcnt(0) drains the ds_reads)
    [[maybe_unused]] int ntc_s = NM ? a.ntc : 0;
    if constexpr (NM != 0) asm volatile("" : "+s"(ntc_s));
    [[maybe_unused]] auto norm_a = [&](int chunk) {
        const int c0 = chunk * BKE;
        const bool second = c0 >= a.c1;
        if ((int)second != a.ntpart) return;                      // block-uniform: this chunk's source is used as stored
        const float* tb0 = snt + (second ? c0 - a.c1 : c0);
        float* dst = sA + (chunk & 1) * ASTG + wave * 256 + lane * 4;
        if constexpr (NM == 2) {
            // SHM_NORM_SCALED: only a patch on the image border has anything to do -- its out-of-image halo entries (DMA'd as zeros) get
            // `ring`; (w * inv) * ring + w * (beta - mean * inv) = 0, the tap's contribution under zero padding of the normalised tensor
            if (!nm_table) return;          // block-uniform
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int m = nmv[j];
                if (m < 0 && m != (int)0x80000000) {
                    const float* tb = tb0 + (-1 - m) + 3 * ntc_s;
                    float* p = dst + j * NW * 256;
                    if constexpr (ESZ == 4) {
                        *(f32x4*)p = *(const f32x4*)tb;
                    } else {
                        const f32x4 r0 = *(const f32x4*)tb, r1 = *(const f32x4*)(tb + 4);
                        u32x4 x;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            x[e] = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)r0[2 * e]) |
                                   ((unsigned)__builtin_bit_cast(unsigned short, (bf16_t)r0[2 * e + 1]) << 16);
                            x[2 + e] = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)r1[2 * e]) |
                                       ((unsigned)__builtin_bit_cast(unsigned short, (bf16_t)r1[2 * e + 1]) << 16);
                        }
                        *(u32x4*)p = x;
                    }
                }
            }
            return;
        } else {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if (nmv[j] >= 0) {
                const float* tb = tb0 + nmv[j];
                float* p = dst + j * NW * 256;
                if constexpr (ESZ == 4) {
                    f32x4 x = *(const f32x4*)p;
                    const f32x4 mean = *(const f32x4*)tb, inv = *(const f32x4*)(tb + ntc_s), beta = *(const f32x4*)(tb + 2 * ntc_s);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = shm_in_norm(x[e], mean[e], inv[e], beta[e]);
                    *(f32x4*)p = x;
                } else {
                    u32x4 x = *(const u32x4*)p;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const f32x4 mean = *(const f32x4*)(tb + 4 * hf), inv = *(const f32x4*)(tb + ntc_s + 4 * hf),
                                    beta = *(const f32x4*)(tb + 2 * ntc_s + 4 * hf);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const unsigned u = x[2 * hf + e];
                            const bf16_t lo = (bf16_t)shm_in_norm(__uint_as_float(u << 16), mean[2 * e], inv[2 * e], beta[2 * e]);
                            const bf16_t hi = (bf16_t)shm_in_norm(__uint_as_float(u & 0xffff0000u), mean[2 * e + 1], inv[2 * e + 1], beta[2 * e + 1]);
                            x[2 * hf + e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
                        }
                    }
                    *(u32x4*)p = x;
                }
            }
        }
        }
    };

    // ---- pipeline.  DMA issue order per wave: [NM: table piece]; A(0); B(0); B(1); then at step s: [A(chunk+1) if tap == 0]; B(s+2).
    if constexpr (NM) if (nm_table) {
        // the (mean, inv, beta, ring) planes of this block's image, 4 x ntc floats, in 1 KiB pieces (reads past the table give zeros)
        const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void*)a.nt, 0, a.ntbytes, 0x00020000);
        if (wave < SHM_NT_PLANES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsn, (lds_ptr)(snt + wave * 256), 16,
                                                     (int)((unsigned)img * 16u * (unsigned)a.ntc + (unsigned)wave * 1024u + (unsigned)lane * 16u), 0, 0, 0);
    }
    dma_a(0);
    dma_b();
    if (ksteps > 1) dma_b();
    if constexpr (NM) if (nm_table) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");        // table piece and A(0) of this wave
        SHM_LDS_BARRIER();                                                    // ... the table pieces of every wave
        asm volatile("" ::: "memory");
        norm_a(0);
    }
    if constexpr (ST) {
        // fragment addresses of the nine taps (floats, relative to the A stage): registers for the whole block
        // (tile i sits 2 i patch rows = 36 i halo rows further on: (R >> 1) + R / 18 grows by 20 i, the swizzle does not change, and
        // the tile offset 2304 i bytes leaves bit 5 alone -- one register per tap, tiles and k groups as immediates / one XOR)
        int fs[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hrow = hb[0] + P.dh[t] * HC + P.dw[t];
            fs[t] = hrow * 16 + ((h ^ (((hrow >> 1) + hrow / HC) & 3)) << 2);        // k group 1: this address ^ 8
        }
        typedef const __attribute__((address_space(3))) f32x4* lds_f4;
        const unsigned sA_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)sA;
        int tw[9];                                     // weight slice of tap t (scalars)
#pragma unroll
        for (int t = 0; t < 9; ++t) tw[t] = __builtin_amdgcn_readlane(tapw_v, t);
        auto dma_b_at = [&](int t_wi, int chunk2, int stage2) {
            const unsigned wbase = (unsigned)((t_wi * a.nout) * a.K + chunk2 * BKE) * (unsigned)ESZ + (NM == 2 ? (unsigned)img * a.wimg : 0u);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const unsigned off = wrow[j] == 0xffffffffu ? wrow[j] : wrow[j] + wbase;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sB + stage2 * BSTG + (wave + NW * j) * 256), 16, (int)off, 0, 0, 0);
            }
        };
        for (int chunk = 0; chunk < nch; ++chunk) {
            // LDS byte address of the A stage: the k-group-1 address is formed as (stage + offset) ^ 32 inside the chunk loop,
            // so that the compiler keeps 18 address registers, not 36 (the stage base is a multiple of 64 bytes)
            const unsigned Ab = sA_lds + (unsigned)((chunk & 1) * ASTG * 4);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // B(s) (and, in order before it, the halo of this chunk) must have landed; issued after B(s): B(s+1),
                // preceded by the A items of step s-1 if that step opened a chunk
                if (tap == 8 && chunk + 1 == nch)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (tap == 1 && chunk + 1 < nch)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
                SHM_LDS_BARRIER();
                asm volatile("" ::: "memory");
                if constexpr (!abl::nodma) {
                    if (tap == 0 && chunk + 1 < nch) dma_a(chunk + 1);
                    // the weight slice of step s + 2: tap, chunk carry and stage are compile-time here (9 % 3 == 0) -- the running
                    // (tap, chunk, stage) state of dma_b() cost ~25 scalar / vector instructions per tap, against eight bf16 MFMAs
                    if (tap < 7 || chunk + 1 < nch) dma_b_at(tw[(tap + 2) % 9], chunk + (tap + 2) / 9, (tap + 2) % 3);
                }
                const float* Bb = sB + (tap % 3) * BSTG + wn * 64 * 16;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    f32x4 av[TM], bv[2];
                    if constexpr (abl::nolds) {
                        for (int i = 0; i < TM; ++i) av[i] = abl_frag;
                        for (int j = 0; j < 2; ++j) bv[j] = abl_frag;
                        asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(bv[0]), "+v"(bv[1]));
                    } else {
                        const lds_f4 ap = (lds_f4)(size_t)((Ab + (unsigned)(fs[tap] << 2)) ^ (unsigned)(kk << 5));
#pragma unroll
                        for (int i = 0; i < TM; ++i) av[i] = ap[i * (2 * HC * 4)];               // 36 halo rows of 64 bytes per tile
#pragma unroll
                        for (int j = 0; j < 2; ++j) bv[j] = *(const f32x4*)(Bb + j * 512 + (kk ? fb1 : fb0));
                    }
                    if constexpr (!abl::nomfma)
                        tap_mfma<T, TM, 2>(av, bv, acc);
                    else
                        asm volatile("" :: "v"(av[0]), "v"(av[1]), "v"(bv[0]), "v"(bv[1]));
                }
                asm volatile("" ::: "memory");
                // NM: A(chunk + 1) was issued at tap 0 in front of B(2), which this step's wait covered: the wave's own items have
                // landed; the other waves read them after the barriers of taps 3..8 and of the next chunk's tap 0
                if constexpr (NM)
                    if (tap == 2 && chunk + 1 < nch && nm_table) {
                        norm_a(chunk + 1);
                        asm volatile("" ::: "memory");
                    }
            }
        }
    } else {
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
        SHM_LDS_BARRIER();
        asm volatile("" ::: "memory");
        if constexpr (!abl::nodma) {
            if (tap == 0 && chunk + 1 < nch) dma_a(chunk + 1);      // other A stage: last read in the previous chunk
            if (s + 2 < ksteps) dma_b();
        }
        if constexpr (!abl::nomfma) compute(chunk, tap, bst);
        asm volatile("" ::: "memory");
        bst = bst == 2 ? 0 : bst + 1;
        if (++tap == 9) {
            tap = 0;
            ++chunk;
        }
    }
    }

    // ---- epilogue: bias + LeakyReLU + store (+ InstanceNorm statistics)
    float s1[2], s2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) s1[j] = s2[j] = 0.f;
    // the bias of the lane's columns, once (see tapgemm_dma_kernel: a per-element fetch serialises the stores)
    float bj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        bj[j] = (a.bias && n < a.nout) ? a.bias[(NM == 2 ? (size_t)img * a.bias_img : (size_t)0) + n] : 0.f;
        asm volatile("" : "+v"(bj[j]));           // waited for here, once (see tapgemm_dma_kernel)
    }
    // bf16 outputs: the MFMA accumulator layout gives each lane one 2-byte element per row, i.e. 64 two-byte
    // store instructions per wave -- measured 29 % of a 64-channel 256x256 layer.  Stage the wave's 64 x 64 tile
    // through LDS (free once every wave is past its last fragment read) and write 16 bytes per lane instead:
    // 8 store instructions per wave, each covering 8 pixel rows of 128 contiguous bytes.
    constexpr bool kWide = sizeof(TO) == 2;
    const bool wide = kWide && (a.nout % 8 == 0) && (a.n1 % 8 == 0) && (a.ldy % 8 == 0) && (((size_t)a.y & 15) == 0) &&
                      (a.y2 == nullptr || ((a.ldy2 % 8 == 0) && (((size_t)a.y2 & 15) == 0)));
    if constexpr (kWide) if (wide) {
        __syncthreads();
        unsigned short* tile = (unsigned short*)smem + wave * (TM * 32 * 64);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + wn * 64 + j * 32 + l31;
                    float v = acc[i][j][r] + bj[j];
                    const TO vo = (TO)shm_lrelu(v, a.slope);
                    v = n < a.nout ? (float)vo : 0.f;
                    s1[j] += v;
                    s2[j] = __builtin_fmaf(v, v, s2[j]);          // (an explicit fma: left to hipcc, one instantiation contracts and another does not)
                    // 16-byte chunk c of row `row` lives at chunk c ^ (row & 7): conflict-free 16-byte reads below
                    const int col = j * 32 + l31;
                    tile[row * 64 + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7))] = __builtin_bit_cast(unsigned short, vo);
                }
            }
        }
        // same-wave LDS hand-off: the ds ops of one wave complete in order; keep the compiler from moving the
        // (differently typed) reads above the writes
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int rr = lane >> 3, ch = lane & 7;
        const int n = n0 + wn * 64 + ch * 8;
        // gsum: the wave's 64 columns lie in one output part (n1 % 64 == 0, checked by the launcher): part, pitch and descriptor are
        // scalars, aux is read with 32-bit offsets (the part is below 4 GiB)
        const int gp = __builtin_amdgcn_readfirstlane(n0 + wn * 64) < a.n1 ? 0 : 1;
        const int gpc = gp ? a.nout - a.n1 : a.n1, gnl = n - (gp ? a.n1 : 0);
        const bool gon = GS && a.gred[gp] != nullptr;
        const bool gs = gon && n < a.nout;
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a.gaux[gp], 0, gon ? 0xfffffff0u : 0u, 0x00020000);
        const unsigned ldab = (unsigned)a.ldgaux[gp] * 2u;
        float t1[8], t2[8];
        u32x4 gav[4 * TM];
        if constexpr (GS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) t1[e] = t2[e] = 0.f;
            // every aux row of the lane first (the accumulators are dead by now: 32 registers are free), then the stores -- left to
            // itself hipcc also hoists the LDS reads and the store addresses of all eight rows and spills 200 registers
#pragma unroll
            for (int it = 0; it < 4 * TM; ++it) {
                const int row = it * 8 + rr;
                const int i = row >> 5, r32 = row & 31;
                const int py = 2 * TM * wm + 2 * i + (r32 >> 4), px = r32 & 15;
                const unsigned opix = (unsigned)((img * a.hi + (y0 + py)) * a.wi + (x0 + px));
                gav[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, opix * ldab + (unsigned)(n < a.nout ? gnl : 0) * 2u, 0, 0));
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int it = 0; it < 4 * TM; ++it) {
            const int row = it * 8 + rr;
            const u32x4 v = *(const u32x4*)(tile + row * 64 + ((ch ^ (row & 7)) << 3));
            const int i = row >> 5, r32 = row & 31;
            const int py = 2 * TM * wm + 2 * i + (r32 >> 4), px = r32 & 15;
            const size_t opix = ((size_t)img * a.hi + (y0 + py)) * a.wi + (x0 + px);
            if (!abl::nostore && n < a.nout) {
                if (n < a.n1)
                    *(u32x4*)((unsigned short*)a.y + opix * a.ldy + n) = v;
                else
                    *(u32x4*)((unsigned short*)a.y2 + opix * a.ldy2 + (n - a.n1)) = v;
            }
            if constexpr (GS) {
                gsum_wide_accum(v, gav[it], t1, t2);
                asm volatile("" ::: "memory");             // one row at a time
            }
        }
        if constexpr (GS) {
            if (gon) {                                     // wave-uniform
                const int slot = (prem * WGM + wm) % a.gslots;
                double* dst = gs ? a.gred[gp] + ((size_t)slot * a.gbatch * gpc + (size_t)img * gpc + gnl) * 2 : nullptr;
                gsum_wide_flush<8>(t1, t2, lane, dst);
            }
        }
    }
    // narrow path, gsum.  The 32 columns of a (wave, j) group lie in one output part (n1 % 32 == 0, checked by the launcher), so
    // "this group takes sums", its aux tensor and pitch are scalars: the sixteen aux loads of a 32 x 32 tile are issued back to
    // back in front of the tile's stores (a per-element conditional load made hipcc wait for every load AND the store before it:
    // 64 serialized round trips per wave tile).
    // (bf16 outputs take their sums in the LDS-staged path above: the launcher only fuses when that path's alignment conditions hold)
    const bool gs_any = GS && sizeof(TO) == 4 && (a.gred[0] != nullptr || a.gred[1] != nullptr);
    if constexpr (GS && sizeof(TO) == 4) if (!wide && gs_any) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = __builtin_amdgcn_readfirstlane(n0 + wn * 64 + j * 32);
            const int gp = nb < a.n1 ? 0 : 1;
            const bool on = a.gred[gp] != nullptr && nb < a.nout;
            const int n = nb + l31;
            const int nl = n - (gp ? a.n1 : 0);
            const int pc = gp ? a.nout - a.n1 : a.n1;
            // aux through a scalar descriptor and 32-bit offsets (the part is below 4 GiB); zero-length when the group takes no sums
            const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a.gaux[gp], 0, on ? 0xfffffff0u : 0u, 0x00020000);
            const unsigned ldab = (unsigned)a.ldgaux[gp] * (unsigned)sizeof(T), nlb = (unsigned)(n < a.nout ? nl : 0) * (unsigned)sizeof(T);
            // ... and so is the group's output part (the launcher fuses the sums only when the outputs are below 4 GiB).  A lane's
            // address is ONE register per 32 x 32 tile -- its pixel of accumulator row 0 -- plus a scalar offset per row (row r of a lane
            // is pixel (r >> 3, 8 ((r >> 2) & 1) + (r & 3)) of the tile's two patch rows): no address arithmetic and no address
            // registers in the element loops (with 64-bit element addresses hipcc kept a pixel index per row and spilled 37 of them
            // to scratch around the sixteen loads)
            const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(gp ? a.y2 : a.y, 0, gp ? a.y2bytes : a.ybytes, 0x00020000);
            const unsigned ldyb = (unsigned)(gp ? a.ldy2 : a.ldy) * (unsigned)sizeof(TO), nyb = (unsigned)(n < a.nout ? nl : 0) * (unsigned)sizeof(TO);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned pix0 = (unsigned)((img * a.hi + (y0 + 2 * TM * wm + 2 * i)) * a.wi + x0 + 4 * h);
                const unsigned ao = pix0 * ldab + nlb, yo = pix0 * ldyb + nyb;
                float q[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned cr = (unsigned)((r >> 3) * a.wi + 8 * ((r >> 2) & 1) + (r & 3));         // scalar
                    if constexpr (sizeof(T) == 4)
                        q[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, ao, cr * ldab, 0));
                    else
                        q[r] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsa, ao, cr * ldab, 0) << 16);
                }
                if (n < a.nout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned cr = (unsigned)((r >> 3) * a.wi + 8 * ((r >> 2) & 1) + (r & 3));
                        float v = acc[i][j][r] + bj[j];
                        const TO vo = (TO)shm_lrelu(v, a.slope);
                        v = (float)vo;
                        s1[j] += v;
                        s2[j] += v * q[r];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vo), rsy, yo, cr * ldyb, 0);
                    }
                }
            }
            if (on) {
                const int slot = (prem * WGM + wm) % a.gslots;
                const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
                const float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
                if (h == 0 && n < a.nout) {
                    double* dst = a.gred[gp] + ((size_t)slot * a.gbatch * pc + (size_t)img * pc + nl) * 2;
                    atomicAdd(dst, (double)t1);
                    atomicAdd(dst + 1, (double)t2);
                }
            }
        }
    }
    // plain element stores: as in the gsum path above, through a scalar descriptor with one address register per 32 x 32 tile and a
    // scalar offset per row when the outputs are below 4 GiB and a 32-column group lies in one output part
    const bool ybuf = !abl::nostore && a.ybytes != 0 && (a.y2 == nullptr || (a.y2bytes != 0 && a.n1 % 32 == 0));
    if (!wide && !gs_any && ybuf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = __builtin_amdgcn_readfirstlane(n0 + wn * 64 + j * 32);
            const int gp = nb < a.n1 ? 0 : 1;
            const int n = nb + l31;
            const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(gp ? a.y2 : a.y, 0, gp ? a.y2bytes : a.ybytes, 0x00020000);
            const unsigned ldyb = (unsigned)(gp ? a.ldy2 : a.ldy) * (unsigned)sizeof(TO);
            const unsigned nyb = (unsigned)(n < a.nout ? n - (gp ? a.n1 : 0) : 0) * (unsigned)sizeof(TO);
            if (n < a.nout) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const unsigned yo = (unsigned)((img * a.hi + (y0 + 2 * TM * wm + 2 * i)) * a.wi + x0 + 4 * h) * ldyb + nyb;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned cr = (unsigned)((r >> 3) * a.wi + 8 * ((r >> 2) & 1) + (r & 3));         // scalar
                        float v = acc[i][j][r] + bj[j];
                        const TO vo = (TO)shm_lrelu(v, a.slope);
                        v = (float)vo;
                        s1[j] += v;
                        s2[j] = __builtin_fmaf(v, v, s2[j]);
                        if constexpr (sizeof(TO) == 4)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vo), rsy, yo, cr * ldyb, 0);
                        else
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, vo), rsy, yo, cr * ldyb, 0);
                    }
                }
            }
        }
    }
    if (!wide && !gs_any && !ybuf) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int py = 2 * TM * wm + 2 * i + (row >> 4), px = row & 15;
            const size_t opix = ((size_t)img * a.hi + (y0 + py)) * a.wi + (x0 + px);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (n < a.nout) {
                    float v = acc[i][j][r] + bj[j];
                    const TO vo = (TO)shm_lrelu(v, a.slope);
                    v = (float)vo;
                    s1[j] += v;
                    s2[j] = __builtin_fmaf(v, v, s2[j]);          // (an explicit fma: left to hipcc, one instantiation contracts and another does not)
                    if (!abl::nostore || v == 123.456f)         // (timing-only build: keep the value live, store nothing)
                    {
                        if (n < a.n1)
                            ((TO*)a.y)[opix * a.ldy + n] = vo;
                        else
                            ((TO*)a.y2)[opix * a.ldy2 + (n - a.n1)] = vo;
                    }
                }
            }
        }
    }
    }
    if (a.stats) {
        const int slot = (prem * WGM + wm) % a.stats_slots;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float t1 = s1[j] + __shfl_xor(s1[j], 32, 64);
            float t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (h == 0 && n < a.nout) {
                double* dst = a.stats + (size_t)slot * a.stats_stride + ((size_t)img * a.nout + n) * 2;
                atomicAdd(dst, (double)t1);
                atomicAdd(dst + 1, (double)t2);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 3x3 / stride-1 tap GEMM for K <= 64 input channels with the WEIGHTS IN REGISTERS (persistent blocks).
//
// The 64-channel 256 x 256 layers are the HBM-side layers of the bf16 step (3 FLOP per byte and tap): in
// tapgemm_halo_kernel a block lives for 18 K-steps between a 2-3 us halo prologue and its store epilogue, with a
// barrier and a weight DMA per tap.  Here the whole weight tensor of a 32-column slice -- 9 taps x K <= 64
// channels = 144 VGPRs per lane -- is loaded ONCE per block and kept in registers; a block (4 waves: 2 (M) x 2 (N),
// wave tile 64 pixels x 32 channels) then walks a contiguous range of 8 x 16-pixel patches:
//   * the 10 x 18 halo of patch p+1 is DMA'd into the other LDS buffer right after the barrier that opens patch p,
//     i.e. it lands under the 72 MFMAs and the epilogue of patch p;
//   * ONE barrier per patch (halo landed for all waves = everybody is done reading the other buffer), no weight
//     traffic, no per-tap synchronisation: the nine taps are nine shifted fragment addresses into the halo;
//   * epilogue as in the halo kernel (LeakyReLU, bf16 rounding, LDS-staged 16-byte stores; the bias is the accumulators'
//     initial value); the InstanceNorm
//     sums are kept in registers (f64) across the patches of one image and flushed with one atomic per column when the
//     image changes: ~40x fewer atomics.
// Two blocks per CU (64 KB of LDS, 256 VGPRs each): they run out of step, so one block's epilogue (VALU, stores)
// overlaps the other's MFMAs on the same SIMDs.  (An explicit ping-pong -- one 8-wave block whose two halves swap MFMA and
// epilogue roles at every barrier -- was built and measured 30 % SLOWER: a wave's MFMA chain waits on its own ds_reads, and
// with the partner pinned to the epilogue nobody fills those bubbles.)  LDS rows are 64 bytes as in the other kernels (DMA
// source-side swizzle, 0xffffffff offsets -> zeros for halo pixels outside the image); the chunk swizzle is
// ((R >> 1) + R / 18) & 3 on the halo row R, which makes every 16-lane group of the fragment reads hit 16 distinct 16-byte
// bank units for all nine taps (brute-force check: tools/probes/halo_swizzle_check.py).  Its address arithmetic is patch independent
// here, so unlike in tapgemm_halo_kernel it costs nothing per tap.
// GS: the gsum epilogue (input-gradient launches, see TapGemmArgs) for bf16 outputs.
// NM: "norm" (see tapgemm_halo_kernel / tapgemm_wreg_f32_kernel).
template <typename TO, int NCH, bool GS = false, int NM = 0>
__global__ __launch_bounds__(256, 2) void tapgemm_wreg_kernel(const TapGemmArgs a, const int npatch) {
    typedef bf16_t T;
    static_assert(!GS || sizeof(TO) == 2, "the gsum epilogue of this kernel is the LDS-staged bf16 one");
    static_assert(!NM || !GS, "norm: forward form");
    constexpr int PH = 8, HC = 18, NIT = 12;            // halo (PH + 2) x 18 = 180 rows, padded to 12 DMA items of 16 rows
    constexpr int ASTG = NIT * 256;                     // floats per 32-channel chunk
    constexpr int ABUF = NCH * ASTG;                    // floats per halo buffer
    static_assert(NIT * NCH % 4 == 0, "DMA items divide over the four waves");
    __shared__ __attribute__((aligned(1024))) float smem[2 * ABUF + 4 * 1024 + (NM ? 4 * 256 : 0)];       // + 4 KB store staging per wave (+ NM: 1 KB table)
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const TapPhase& P = a.ph[0];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.y * 64;
    const int ppr = a.wi >> 4, ppi = (a.hi / PH) * ppr;

    // contiguous patch range of this block
    const int per = (npatch + gridDim.x - 1) / gridDim.x;
    const int q0 = blockIdx.x * per, q1 = min(npatch, q0 + per);
    if (q0 >= q1) return;

    // ---- weights -> registers: lane (l31, h) holds W[tap][n][c*32 + kk*16 + 8h .. +7] for its column n
    const int ncol = n0 + wn * 32 + l31;
    bf16x8 bw[9][NCH][2];
    float bias;
    // (NM, SHM_NORM_SCALED: the weight copy and the bias row of image `img`, see tapgemm_wreg_f32_kernel)
    auto load_w = [&](int img) {
        const bf16_t* wp = (const bf16_t*)a.w + (NM == 2 ? (size_t)img * (a.wimg >> 1) : (size_t)0);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 v = {};
                    if (ncol < a.nout) v = *(const bf16x8*)(wp + ((size_t)P.widx[t] * a.nout + ncol) * a.K + c * 32 + kk * 16 + h * 8);
                    bw[t][c][kk] = v;
                }
        bias = (a.bias && ncol < a.nout) ? a.bias[(NM == 2 ? (size_t)img * a.bias_img : (size_t)0) + ncol] : 0.f;
    };
    load_w(NM == 2 ? q0 / ppi : 0);

    // ---- halo DMA: item it (0 .. NIT*NCH-1) = chunk it / NIT, halo rows [16 (it % NIT), +16); wave w owns items w, w+4, ...
    // NIT / 4 = 3 items per wave and chunk: item j of chunk c covers halo rows 16 (wave + 4 j) + drow, so the lane keeps
    // three halo row numbers and derives the rest per patch (registers are what this kernel is short of).
    const int drow = lane >> 2, dq = lane & 3;
    static_assert(NIT == 12, "three DMA items per wave and chunk");
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const unsigned pixb = (unsigned)a.ldx * 2u;
    float* const tbl = smem + 2 * ABUF + 4 * 1024 + wave * 256;     // NM: this wave's copy of the planes of the image of the halo in flight
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void*)a.nt, 0, NM ? a.ntbytes : 0u, 0x00020000);
    // (the instantiations that are out of registers -- gsum, SHM_NORM_SCALED with two chunks: neither runs in the default bf16 step --
    // keep recomputing the halo coordinates per patch from the lane id: four more live registers would be four more spills)
    constexpr bool kDmaConst = !(NCH == 2 && (GS || NM == 2));
    [[maybe_unused]] unsigned doff[3], dbm = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int hrow = 16 * (wave + 4 * j) + drow;
        const int hr = hrow / HC, hc = hrow - hr * HC;
        doff[j] = (unsigned)(hr * a.wi + hc) * pixb + (unsigned)((dq ^ (((hrow >> 1) + hr) & 3)) << 4);
        dbm |= (hrow >= (PH + 2) * HC ? 16u : (hr == 0 ? 1u : 0u) | (hr == PH + 1 ? 2u : 0u) | (hc == 0 ? 4u : 0u) | (hc == HC - 1 ? 8u : 0u)) << (5 * j);
    }
    auto dma = [&](int q, int buf) {
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
        float* dst = smem + buf * ABUF + wave * 256;
        // 4 x ntc <= 256 floats (checked by the launcher); the previous table was last read a patch ago.  NM = 2: only a patch on the
        // image border reads it (the `ring` plane)
        if constexpr (NM)
            if (NM == 1 || y0 == 0 || y0 + PH == a.hi || x0 == 0 || x0 + 16 == a.wi)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsn, (lds_ptr)tbl, 16, (int)((unsigned)img * 16u * (unsigned)a.ntc + (unsigned)lane * 16u), 0, 0, 0);
        if constexpr (!kDmaConst) {
            int dr = drow;
            asm volatile("" : "+v"(dr));        // recompute the halo coordinates per patch: hoisted, they are spilled
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int hrow = 16 * (wave + 4 * j) + dr;
                const int hr = hrow / HC, hc = hrow - hr * HC;
                const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
                const bool v = hrow < (PH + 2) * HC && (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
                const unsigned off = v ? (unsigned)((img * a.hi + iy) * a.wi + ix) * pixb + (unsigned)((dq ^ (((hrow >> 1) + hr) & 3)) << 4) : 0xffffffffu;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst + c * ASTG + j * 4 * 256), 16,
                                                             (int)(v ? off + 64u * c : 0xffffffffu), 0, 0, 0);
            }
            return;
        }
        // per-lane constants (byte offset of the lane's pixel inside the halo incl. the source-side swizzle, five edge bits per item)
        // + the patch's origin and edge bits: see tapgemm_wreg_f32_kernel
        const unsigned edges = 16u | (y0 == 0 ? 1u : 0u) | (y0 + PH == a.hi ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + 16 == a.wi ? 8u : 0u);
        const unsigned baseb = (unsigned)((img * a.hi + y0 - 1) * a.wi + x0 - 1) * pixb;           // halo (0, 0); may wrap below zero
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool out = (dbm & (edges << (5 * j))) != 0;
            const unsigned off = doff[j] + baseb;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst + c * ASTG + j * 4 * 256), 16,
                                                         (int)(out ? 0xffffffffu : off + 64u * c), 0, 0, 0);
        }
    };

    // NM: normalise this wave's items of halo(q) in buffer buf (landed: the caller waited)
    [[maybe_unused]] auto norm_a = [&](int q, int buf) {
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
        if (NM == 2 && !(y0 == 0 || y0 + PH == a.hi || x0 == 0 || x0 + 16 == a.wi)) return;       // block-uniform: no out-of-image halo entry
        float* dst = smem + buf * ABUF + wave * 256 + lane * 4;
        int dr = drow;
        asm volatile("" : "+v"(dr));        // as in dma(): nothing of this is kept across the MFMA loop
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int hrow = 16 * (wave + 4 * j) + dr;
            const int hr = hrow / HC, hc = hrow - hr * HC;
            const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
            const bool inside = (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
            if constexpr (NM == 2) {                // SHM_NORM_SCALED: `ring` over the out-of-image entries (see tapgemm_halo_kernel)
                if (hrow < (PH + 2) * HC && !inside) {
                    const int g8 = (dq ^ (((hrow >> 1) + hr) & 3)) << 3;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const float* tb = tbl + 3 * a.ntc + c * 32 + g8;
                        const f32x4 r0 = *(const f32x4*)tb, r1 = *(const f32x4*)(tb + 4);
                        u32x4 x;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            x[e] = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)r0[2 * e]) |
                                   ((unsigned)__builtin_bit_cast(unsigned short, (bf16_t)r0[2 * e + 1]) << 16);
                            x[2 + e] = (unsigned)__builtin_bit_cast(unsigned short, (bf16_t)r1[2 * e]) |
                                       ((unsigned)__builtin_bit_cast(unsigned short, (bf16_t)r1[2 * e + 1]) << 16);
                        }
                        *(u32x4*)(dst + c * ASTG + j * 4 * 256) = x;
                    }
                }
            } else if (hrow < (PH + 2) * HC && inside) {
                const int g8 = (dq ^ (((hrow >> 1) + hr) & 3)) << 3;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float* tb = tbl + c * 32 + g8;
                    float* p = dst + c * ASTG + j * 4 * 256;
                    u32x4 x = *(const u32x4*)p;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const f32x4 mean = *(const f32x4*)(tb + 4 * hf), inv = *(const f32x4*)(tb + a.ntc + 4 * hf),
                                    beta = *(const f32x4*)(tb + 2 * a.ntc + 4 * hf);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const unsigned u = x[2 * hf + e];
                            const bf16_t lo = (bf16_t)shm_in_norm(__uint_as_float(u << 16), mean[2 * e], inv[2 * e], beta[2 * e]);
                            const bf16_t hi = (bf16_t)shm_in_norm(__uint_as_float(u & 0xffff0000u), mean[2 * e + 1], inv[2 * e + 1], beta[2 * e + 1]);
                            x[2 * hf + e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
                        }
                    }
                    *(u32x4*)p = x;
                }
            }
        }
    };

    // ---- fragment addressing (patch independent): byte address of the centre tap's halo row for the two 32-pixel tiles
    const int hb0 = (4 * wm + (l31 >> 4) + 1) * HC + (l31 & 15) + 1;      // second tile: + 2 * HC
    int tsh[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tsh[t] = P.dh[t] * HC + P.dw[t];

    // statistics carried over the patches of one image (fp32 per lane: at most a few thousand bf16-rounded terms; the
    // cross-block sums are f64 atomics)
    float S1 = 0.f, S2 = 0.f;
    int simg = q0 / ppi;
    auto flush = [&](int img) {
        const float t1 = S1 + __shfl_xor(S1, 32, 64), t2 = S2 + __shfl_xor(S2, 32, 64);
        if (h == 0 && ncol < a.nout) {
            double* dst = a.stats + (size_t)(blockIdx.x % a.stats_slots) * a.stats_stride + ((size_t)img * a.nout + ncol) * 2;
            atomicAdd(dst, (double)t1);
            atomicAdd(dst + 1, (double)t2);
        }
        S1 = S2 = 0.f;
    };

    unsigned short* const tile = (unsigned short*)(smem + 2 * ABUF) + wave * 2048;      // 64 rows x 32 bf16
    // bf16 outputs leave through LDS-staged 16-byte stores: the launcher guarantees Cout % 64 == 0 (every wave owns 32 valid
    // columns: no conditionals in the epilogue, which cost this kernel VGPRs it does not have), 16-byte aligned pitches and
    // bases.  fp32 outputs (SHM_BF16_GF32) use element stores.
    constexpr bool kWide = sizeof(TO) == 2;
    // outputs through buffer stores: one 32-bit offset register per store instead of a 64-bit address
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy2 = __builtin_amdgcn_make_buffer_rsrc(a.y2, 0, a.y2bytes, 0x00020000);

    dma(q0, 0);
    // (Starting the block in the odd HW wave slot of its SIMDs half a patch late, to put the two blocks of a CU in anti-phase,
    // was measured with delays of 1300-5800 clocks: no effect.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (NM) norm_a(q0, 0);
    auto patch = [&](const int q) {
        const int buf = (q - q0) & 1;
        SHM_LDS_BARRIER();                   // halo(q) landed for every wave (each waited for its own part at the end
        asm volatile("" ::: "memory");                  // of the previous patch); everyone is done with the other buffer
        if constexpr (!abl::nodma)
            if (q + 1 < q1) dma(q + 1, buf ^ 1);

        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = bias;
        const float* Ab = smem + buf * ABUF;
        // gsum form: the nine fragment addresses are formed per patch -- kept across patches (hipcc hoists them) they no longer fit
        // beside the epilogue's sums and were spilled INSIDE the MFMA loop (27 scratch reloads per patch)
        int hbq = hb0;
        if constexpr (GS) asm volatile("" : "+v"(hbq));
        if constexpr (abl::wreg_prio) __builtin_amdgcn_s_setprio(1);
        // (An explicit software pipeline -- fragment reads pinned two or three steps ahead of their MFMAs with sched_barrier --
        // was measured: no gain on the forward, 15 % slower input gradients.  With two waves per SIMD the partner's MFMAs cover
        // a wave's LDS latency; hipcc's just-in-time reads keep the VGPR count at the 256 limit.)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // the swizzle is invariant under a shift by two halo lines (36 rows: (R >> 1) + R / 18 grows by 20), so the second
            // 32-pixel tile reads at a constant offset from the first: one address register per (tap, kk), the tile and the
            // channel chunk go into the instruction's offset field
            int fa[2];
            {
                const int hrow = hbq + tsh[t];
                fa[0] = hrow * 16 + ((h ^ (((hrow >> 1) + hrow / HC) & 3)) << 2);      // floats; the kk = 1 group is this address ^ 8
                fa[1] = fa[0] + 2 * HC * 16;
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x4 av;
                        if constexpr (abl::nolds) {
                            av = __builtin_bit_cast(f32x4, bw[t][c][kk]);         // timing only: no fragment reads
                            asm volatile("" : "+v"(av));
                        } else {
                            av = *(const f32x4*)(Ab + c * ASTG + (fa[i] ^ (kk << 3)));
                        }
                        if constexpr (abl::nomfma)
                            asm volatile("" ::"v"(av), "v"(bw[t][c][kk]));          // timing only: fragment reads without the MFMAs
                        else
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), bw[t][c][kk], acc[i], 0, 0, 0);
                    }
        }

        if constexpr (abl::wreg_prio) __builtin_amdgcn_s_setprio(0);
        // ---- epilogue of patch q
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
        if (a.stats && img != simg) {
            flush(simg);
            simg = img;
        }
        float s1 = 0.f, s2 = 0.f;
        if constexpr (abl::noepi) asm volatile("" ::"v"(acc[0]), "v"(acc[1]));       // timing only: no epilogue at all
        if constexpr (kWide && !abl::noepi) {
            // the wave's 64 x 32 tile through LDS (64-byte rows; a 16-lane group of the 16-byte reads below covers four
            // whole rows = all 64 banks once)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float u = acc[i][r];
                    const bf16_t vo = (bf16_t)shm_lrelu_max(u, a.slope);      // LeakyReLU for 0 <= slope <= 1 (checked by the launcher)
                    const float v = (float)vo;
                    s1 += v;
                    s2 = __builtin_fmaf(v, v, s2);
                    tile[row * 32 + l31] = __builtin_bit_cast(unsigned short, vo);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // same-wave LDS hand-off
            const int rr = lane >> 2, ch = lane & 3;
            const int n = n0 + wn * 32 + ch * 8;
            const bool part0 = __builtin_amdgcn_readfirstlane(n0 + wn * 32) < a.n1;
            int gpc = 0, gp = 0;
            const unsigned short* gaux = nullptr;
            if constexpr (GS) {
                // A wave's 32 channels lie in one part (n1 % 32 == 0).  aux is read eight bytes (four channels) at a time, in two
                // passes over the tile: 16-byte reads with eight channels of partial sums per lane put the kernel over its 256 VGPRs
                // (the weights were spilled inside the MFMA loop)
                gp = part0 ? 0 : 1;                        // wave-uniform: pointers, pitches and the part test stay in SGPRs
                gpc = gp ? a.nout - a.n1 : a.n1;
                gaux = a.gred[gp] ? (const unsigned short*)a.gaux[gp] : nullptr;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 16 + rr;
                const u32x4 v = *(const u32x4*)(tile + row * 32 + (ch << 3));
                const int py = 4 * wm + (row >> 4), px = row & 15;
                const unsigned opix = (unsigned)((img * a.hi + (y0 + py)) * a.wi + (x0 + px));
                // the wave's 32 channels lie in one output part (n1 % 32 == 0): a scalar branch -- a per-lane choice of the buffer
                // descriptor makes hipcc wrap every store in a readfirstlane (waterfall) loop
                if constexpr (abl::nostore)
                    asm volatile("" ::"v"(v), "v"(opix));                           // timing only
                else if (part0)
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsy, (opix * (unsigned)a.ldy + (unsigned)n) * 2u, 0, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsy2, (opix * (unsigned)a.ldy2 + (unsigned)(n - a.n1)) * 2u, 0, 0);
            }
            if constexpr (GS) {
                if (gaux) {                                               // wave-uniform
                    // recompute the lane's coordinates per patch: hoisted out of the patch loop they (and every address derived
                    // from them) stay live across the MFMA loop, which has no registers to spare
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
                    const int rr = ln >> 2, ch = ln & 3;
                    const int gnl = n0 + wn * 32 + ch * 8 - (gp ? a.n1 : 0);
                    const int slot = (int)(blockIdx.x % (unsigned)a.gslots);
                    double* const dst = a.gred[gp] + ((size_t)slot * a.gbatch * gpc + (size_t)img * gpc + gnl) * 2;
                    // (aux has the extent of its output part, which the launcher checked to be below 4 GiB: 32-bit offsets)
                    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)gaux, 0, 0xfffffff0u, 0x00020000);
                    const unsigned ldab = (unsigned)a.ldgaux[gp] * 2u;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        u32x2 av[4];
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int row = it * 16 + rr;
                            const unsigned opix = (unsigned)((img * a.hi + (y0 + 4 * wm + (row >> 4))) * a.wi + (x0 + (row & 15)));
                            av[it] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsa, opix * ldab + (unsigned)(gnl + 4 * half) * 2u, 0, 0));
                        }
                        float t[8];            // t[0..3] = sum v, t[4..7] = sum v * aux of channels 4 half .. 4 half + 3
#pragma unroll
                        for (int e = 0; e < 8; ++e) t[e] = 0.f;
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int row = it * 16 + rr;
                            const u32x2 v = *(const u32x2*)(tile + row * 32 + (ch << 3) + 4 * half);
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const float v0 = __uint_as_float(v[e] << 16), v1 = __uint_as_float(v[e] & 0xffff0000u);
                                const float a0 = __uint_as_float(av[it][e] << 16), a1 = __uint_as_float(av[it][e] & 0xffff0000u);
                                t[2 * e] += v0;
                                t[2 * e + 1] += v1;
                                t[4 + 2 * e] += v0 * a0;
                                t[4 + 2 * e + 1] += v1 * a1;
                            }
                        }
                        // reduce-scatter of the eight sums over the sixteen lanes rr of a channel group: three halving steps over
                        // lane bits 5, 4, 3 leave value index rr >> 1 (bit 2 of rr = moment, bits 1-0 = channel), a last add over
                        // lane bit 2 completes it; the even-rr lane adds it: one atomic instruction per pass
#pragma unroll
                        for (int st = 0; st < 3; ++st) {
                            const int hf = 4 >> st, bit = 32 >> st;
                            const bool up = (ln & bit) != 0;
#pragma unroll
                            for (int e = 0; e < hf; ++e) {
                                const float keep = up ? t[hf + e] : t[e];
                                const float send = up ? t[e] : t[hf + e];
                                t[e] = keep + __shfl_xor(send, bit, 64);
                            }
                        }
                        const float tot = t[0] + __shfl_xor(t[0], 4, 64);
                        const int vi = rr >> 1;                                   // 0..3: sum v of channel vi; 4..7: sum v * aux of channel vi - 4
                        if ((rr & 1) == 0) atomicAdd(dst + (size_t)(4 * half + (vi & 3)) * 2 + (vi >> 2), (double)tot);
                    }
                }
            }
        } else if constexpr (!abl::noepi) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int py = 4 * wm + (row >> 4), px = row & 15;
                    const unsigned opix = (unsigned)((img * a.hi + (y0 + py)) * a.wi + (x0 + px));
                    const float u = acc[i][r];
                    const float v = shm_lrelu_max(u, a.slope);
                    s1 += v;
                    s2 = __builtin_fmaf(v, v, s2);
                    if (__builtin_amdgcn_readfirstlane(n0 + wn * 32) < a.n1) {        // wave-uniform (n1 % 32 == 0): no waterfall loop
                        if (ncol < a.nout)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsy, (opix * (unsigned)a.ldy + (unsigned)ncol) * 4u, 0, 0);
                    } else if (ncol < a.nout) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsy2, (opix * (unsigned)a.ldy2 + (unsigned)(ncol - a.n1)) * 4u, 0, 0);
                    }
                }
        }
        S1 += s1;
        S2 += s2;
        // halo(q + 1) was issued at the top of this patch; the only younger operations of this wave are this epilogue's
        // stores (bf16 outputs: exactly four 16-byte store instructions, plus the rare statistics flush), which stay in flight
        if constexpr (abl::nostore || abl::noepi) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (kWide && GS) {
            if (a.gred[n0 + wn * 32 < a.n1 ? 0 : 1])                 // wave-uniform: four stores and the gsum atomic
                asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if constexpr (kWide)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (NM)
            if (q + 1 < q1) norm_a(q + 1, buf ^ 1);
    };
    if constexpr (NM == 2) {                 // one weight copy per image -> one segment of the patch range per image
        int q = q0;
        while (q < q1) {
            const int qe = min(q1, (q / ppi + 1) * ppi);
            if (q != q0) load_w(q / ppi);
            for (; q < qe; ++q) patch(q);
        }
    } else {
        for (int q = q0; q < q1; ++q) patch(q);
    }
    if (a.stats) flush(simg);
}

// ------------------------------------------------------------------------------------------
// The fp32 counterpart of tapgemm_wreg_kernel: 3x3 / stride-1 tap GEMM for K <= 64 input channels with the weights in
// registers, on v_mfma_f32_16x16x4_f32 (exact fp32, the same 64 FLOP/clk/SIMD as the 32x32x2 form).
//
// With 16-column MFMA tiles a wave's slice of the weight tensor is 9 taps x 64 channels x 16 columns = 144 VGPRs; one block
// per CU (8 waves: 2 (M) x 4 (N), wave tile 64 pixels x 16 channels, 8 x 16-pixel patches) keeps it for its whole range of
// patches.  A K step (one tap, 16 channels) is four 16-byte fragment reads and sixteen MFMAs per wave, there is no weight
// traffic and ONE barrier per patch (576 MFMAs = 18 432 MFMA cycles per wave): the per-K-step barrier / DMA-issue /
// first-ds_read bubble that holds the 128x64 DMA tile at 65-70 % of the fp32 peak on the Cout <= 64 layers does not exist.
// Outputs are stored straight from the accumulators (lane = channel: 64-byte segments, four pixel rows per instruction);
// InstanceNorm sums are carried in registers (f64) across the patches of an image.  LDS rows are 64 bytes (16 channels) with
// the DMA source-side swizzle chunk' = (chunk + (R >> 1)) & 3 on the halo row R: conflict free for this instruction's lane
// groups (pixel = lane & 15, chunk = lane >> 4) over all nine taps (tools/probes/halo_swizzle_check.py).
// GS: the gsum epilogue (input-gradient launches, see TapGemmArgs): S2 carries sum(v * aux) instead of sum(v * v).
// NM: "norm" (see tapgemm_halo_kernel) -- the source is the un-normalised activation of an InstanceNorm block; a wave normalises
// the halo items it DMA'd itself at the end of the patch in front (they have landed by then), from its own 1 KiB copy of the
// image's (mean, inv, beta) planes, which travels with the halo DMA.
// LDS pitch (halo rows per patch row, 18 of them used) and DMA items per 16-channel chunk of tapgemm_wreg_f32_kernel.  The 64-channel
// form (WN = 4, the hot one) pads its halo image to 24 rows per patch row: the chunk swizzle (lq + (R >> 1)) & 3 then repeats from one
// patch row to the next (12 = 0 mod 4), so the four M tiles of a tap read at ONE address register plus immediates -- 9 fragment address
// registers instead of 36, which is what lets the gsum form keep them across patches (recomputing them per patch, as it had to at
// pitch 18, was 4 % of the kernel) -- for 15 instead of 12 DMA items per chunk (the padding rows are out-of-range reads: zeros, no
// memory traffic).  The narrow forms keep pitch 18 (their 18- and 34-row halos would not fit at 24).
constexpr int wreg32_pitch(int wn) { return wn == 4 ? 24 : 18; }
constexpr int wreg32_nit(int wn) { return ((32 / wn + 2) * wreg32_pitch(wn) + 15) / 16; }
// T = bf16_t (round 3, "tapgemm.wreg16"): the same kernel on bf16 operands and outputs -- the LDS image, the DMA and every address are
// the fp32 kernel's (64-byte rows = 32 channels, a lane's 16-byte fragment = 8 channels = ONE v_mfma_f32_16x16x32_bf16 where fp32
// issues four 16x16x4), the weights of a 16-column wave tile are 9 x NCH x 4 registers -- 72 at 64 input channels, against 144 in the
// four-wave tapgemm_wreg_kernel -- so the kernel fits 128 VGPRs and a SIMD holds four waves of two blocks instead of two
// (profiles/r03_bf16_wreg_ablation.txt: at two waves per SIMD the MFMA phase and the epilogue / store / DMA-wait phase of that kernel add up
// instead of overlapping).  Plain forward form only (no gsum, no norm, one source).
template <int NCH, int WN = 4, bool TWO = false, bool GS = false, int NM = 0, typename T = float>
__global__ __launch_bounds__(512, sizeof(T) == 2 ? 4 : 2) void tapgemm_wreg_f32_kernel(const TapGemmArgs a, const int npatch) {
    static_assert(!NM || (!TWO && !GS), "norm: one source, forward form");
    static_assert(sizeof(T) == 4 || (WN == 4 && !TWO && !GS && !NM), "bf16: plain 64-channel form");
    constexpr int ESZ = sizeof(T), CHE = 16 / ESZ, BKE = 64 / ESZ;       // channels per 16-byte fragment / per 64-byte row
    // WN waves along N (16 columns each), WM = 8 / WN along M (four patch rows each): 64 / 32 / 16 output channels per block on
    // patches of 8 / 16 / 32 rows -- the narrow forms serve SpecSeg's 16- and 32-channel layers without idle N waves
    constexpr int WM = 8 / WN, PH = 4 * WM, HC = 18, HP = wreg32_pitch(WN), NIT = wreg32_nit(WN);     // halo (PH + 2) x 18 pixels at pitch HP, in DMA items of 16 rows
    constexpr int ASTG = NIT * 256;                     // floats per 16-channel chunk
    constexpr int ABUF = NCH * ASTG;                    // floats per halo buffer
    extern __shared__ __attribute__((aligned(1024))) float smem[];      // two halo buffers
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const TapPhase& P = a.ph[0];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * (16 * WN);
    const int ppr = a.wi >> 4, ppi = (a.hi / PH) * ppr;

    const int per = (npatch + gridDim.x - 1) / gridDim.x;
    const int q0 = blockIdx.x * per, q1 = min(npatch, q0 + per);
    if (q0 >= q1) return;

    // ---- weights -> registers: lane (l15, lq) holds W[tap][n][c*16 + 4 lq .. +3] for its column n; MFMA e of a K step
    // contracts channel 4 k' + e of the chunk over k' = lane >> 4 (the same permutation on the A side)
    const int ncol = n0 + wn * 16 + l15;
    f32x4 bw[9][NCH];
    float bias;
    // (NM = 2, SHM_NORM_SCALED: the weight copy and the bias row of image `img`, re-read when the block's patch range moves on to the
    // next image -- outside the patch loop, so that hipcc's waitcnt pass drains these loads in the loop's preheader, not at every use)
    auto load_w = [&](int img) {
        const T* wp = (const T*)a.w + (NM == 2 ? (size_t)img * (a.wimg / ESZ) : (size_t)0);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < NCH; ++c) bw[t][c] = *(const f32x4*)(wp + ((size_t)P.widx[t] * a.nout + ncol) * a.K + c * BKE + lq * CHE);
        bias = a.bias ? a.bias[(NM == 2 ? (size_t)img * a.bias_img : (size_t)0) + ncol] : 0.f;
    };
    load_w(NM == 2 ? q0 / ppi : 0);

    // ---- halo DMA: item (c, ri) = 16-channel chunk c, halo rows [16 ri, 16 ri + 16) of the LDS image; wave w owns row items w, w + 8, ...
    // of EVERY chunk, so a lane's pixel inside the halo depends on the row item only: per lane and row item one pixel offset and five
    // edge bits (top / bottom / left / right edge of the halo, "nothing to fetch"), per patch four scalar edge bits and the origin --
    // an add, a masked test, a multiply-add and one select per DMA instruction.  (Until round 3 every patch recomputed coordinates,
    // range tests and exec-masked selects per item, ~25 VALU instructions each, from the lane id -- the registers to keep them were
    // not there before the fragment addresses went from 36 to 9, see wreg32_pitch.)
    constexpr int NR = (NIT + 7) / 8;                    // row items per wave (the last one may be idle)
    static_assert(NR <= 6, "five edge bits per row item in one register");
    const int drow = lane >> 2, dq = lane & 3;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000);
    const unsigned pixb = (unsigned)a.ldx * (unsigned)ESZ, pixb2 = (unsigned)a.ldx2 * (unsigned)ESZ;
    const int nc1 = a.c1 / BKE;                           // TWO: chunks [0, nc1) come from x, the rest from x2 (Concatenate)
    float* const tbl = smem + 2 * ABUF + wave * 256;     // NM: this wave's copy of the planes of the image of the halo in flight
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void*)a.nt, 0, NM ? a.ntbytes : 0u, 0x00020000);
    // LDS chunk dq of row R holds channel chunk (dq - (R >> 1)) & 3; items start at multiples of 16 rows, so the term depends on the lane only
    const unsigned swb = (unsigned)(((dq - (drow >> 1)) & 3) << 4);
    unsigned po[NR], bm = 0;
#pragma unroll
    for (int jr = 0; jr < NR; ++jr) {
        const int ri = wave + 8 * jr;
        const int hrow = 16 * ri + drow;
        const int hr = hrow / HP, hc = hrow - hr * HP;
        po[jr] = (unsigned)(hr * a.wi + hc);
        const unsigned bits = (ri >= NIT || hr >= PH + 2 || hc >= HC) ? 16u : (hr == 0 ? 1u : 0u) | (hr == PH + 1 ? 2u : 0u) | (hc == 0 ? 4u : 0u) | (hc == HC - 1 ? 8u : 0u);
        bm |= bits << (5 * jr);
    }
    auto patch_edges = [&](int y0, int x0) {             // which edges of the image the halo of the patch at (y0, x0) sticks out of (+ bit 4)
        return 16u | (y0 == 0 ? 1u : 0u) | (y0 + PH == a.hi ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + 16 == a.wi ? 8u : 0u);
    };
    auto dma = [&](int q, int buf) {
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
        float* dst = smem + buf * ABUF;
        // 4 x ntc <= 256 floats (checked by the launcher); the previous table was last read a patch ago.  NM = 2: only a patch on the
        // image border reads it (the `ring` plane)
        if constexpr (NM)
            if (NM == 1 || y0 == 0 || y0 + PH == a.hi || x0 == 0 || x0 + 16 == a.wi)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsn, (lds_ptr)tbl, 16, (int)((unsigned)img * 16u * (unsigned)a.ntc + (unsigned)lane * 16u), 0, 0, 0);
        const unsigned edges = patch_edges(y0, x0);
        const unsigned basepix = (unsigned)((img * a.hi + y0 - 1) * a.wi + x0 - 1);          // pixel index of halo (0, 0); may wrap below zero
#pragma unroll
        for (int jr = 0; jr < NR; ++jr) {
            const int ri = wave + 8 * jr;                // wave-uniform
            if (jr < NR - 1 || ri < NIT) {
                const bool out = (bm & (edges << (5 * jr))) != 0;
                const unsigned pp = po[jr] + basepix;
                const unsigned o1 = pp * pixb + swb;
                [[maybe_unused]] const unsigned o2 = TWO ? pp * pixb2 + swb : 0u;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (!TWO || c < nc1) {
                        const unsigned off = out ? 0xffffffffu : o1 + (unsigned)(c * 64);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst + (c * NIT + ri) * 256), 16, (int)off, 0, 0, 0);
                    } else {
                        const unsigned off = out ? 0xffffffffu : o2 + (unsigned)((c - nc1) * 64);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx2, (lds_ptr)(dst + (c * NIT + ri) * 256), 16, (int)off, 0, 0, 0);
                    }
                }
            }
        }
    };

    // NM: normalise this wave's items of halo(q) in buffer buf (landed: the caller waited)
    [[maybe_unused]] auto norm_a = [&](int q, int buf) {
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
        if (NM == 2 && !(y0 == 0 || y0 + PH == a.hi || x0 == 0 || x0 + 16 == a.wi)) return;       // block-uniform: no out-of-image halo entry
        const unsigned edges = patch_edges(y0, x0);
        float* dst = smem + buf * ABUF + lane * 4;
#pragma unroll
        for (int jr = 0; jr < NR; ++jr) {
            const int ri = wave + 8 * jr;
            if (jr < NR - 1 || ri < NIT) {
                const unsigned m = (bm >> (5 * jr)) & 31u;
                const bool halo = (m & 16u) == 0;                  // (not a padding row of the LDS image)
                const bool inside = (m & edges & 15u) == 0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    float* p = dst + (c * NIT + ri) * 256;
                    if constexpr (NM == 2) {        // SHM_NORM_SCALED: `ring` over the out-of-image entries (see tapgemm_halo_kernel)
                        if (halo && !inside) *(f32x4*)p = *(const f32x4*)(tbl + 3 * a.ntc + c * 16 + (swb >> 2));
                    } else if (halo && inside) {
                        const float* tb = tbl + c * 16 + (swb >> 2);
                        f32x4 x = *(const f32x4*)p;
                        const f32x4 mean = *(const f32x4*)tb, inv = *(const f32x4*)(tb + a.ntc), beta = *(const f32x4*)(tb + 2 * a.ntc);
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = shm_in_norm(x[e], mean[e], inv[e], beta[e]);
                        *(f32x4*)p = x;
                    }
                }
            }
        }
    };

    // ---- fragment addressing: M tile m = patch row 4 wm + m, pixel = l15; halo row of the centre tap
    const int hb0 = (4 * wm + 1) * HP + l15 + 1;
    int tsh[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tsh[t] = P.dh[t] * HP + P.dw[t];

    double S1 = 0.0, S2 = 0.0;
    int simg = q0 / ppi;
    // gsum: the wave's 16 columns lie in one part (n1 % 16 == 0): gp, the pitch and the "this part takes sums" test are scalars
    const int gp = __builtin_amdgcn_readfirstlane(n0 + wn * 16) < a.n1 ? 0 : 1;
    const int gpc = gp ? a.nout - a.n1 : a.n1, gnl = ncol - (gp ? a.n1 : 0);
    const bool gson = GS && a.gred[gp] != nullptr;
    // (aux has the extent of its output part, which the launcher checked to be below 4 GiB: 32-bit offsets, scalar descriptor)
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a.gaux[gp], 0, gson ? 0xfffffff0u : 0u, 0x00020000);
    const unsigned ldab = (unsigned)a.ldgaux[gp] * 4u;
    const float* const gaux = gson ? (const float*)a.gaux[gp] : nullptr;
    auto flush = [&](int img) {
        double t1 = S1 + __shfl_xor(S1, 16, 64), t2 = S2 + __shfl_xor(S2, 16, 64);
        t1 += __shfl_xor(t1, 32, 64);
        t2 += __shfl_xor(t2, 32, 64);
        if (lane < 16) {
            if constexpr (GS) {
                if (gaux) {
                    double* dst = a.gred[gp] + ((size_t)((WM * blockIdx.x + wm) % a.gslots) * a.gbatch * gpc + (size_t)img * gpc + gnl) * 2;
                    atomicAdd(dst, t1);
                    atomicAdd(dst + 1, t2);
                }
            } else {
                double* dst = a.stats + (size_t)((WM * blockIdx.x + wm) % a.stats_slots) * a.stats_stride + ((size_t)img * a.nout + ncol) * 2;
                atomicAdd(dst, t1);
                atomicAdd(dst + 1, t2);
            }
        }
        S1 = S2 = 0.0;
    };
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy2 = __builtin_amdgcn_make_buffer_rsrc(a.y2, 0, a.y2bytes, 0x00020000);
    const bool part0 = __builtin_amdgcn_readfirstlane(n0 + wn * 16) < a.n1;

    dma(q0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (NM) norm_a(q0, 0);
    auto patch = [&](const int q) {
        const int buf = (q - q0) & 1;
        SHM_LDS_BARRIER();                   // halo(q) landed for every wave; everyone is done with the other buffer
        asm volatile("" ::: "memory");
        if (q + 1 < q1) dma(q + 1, buf ^ 1);
        // gsum: aux at the sixteen output positions of this lane (the same 64-byte segments as the epilogue's stores), issued here so
        // that their latency passes under the 576 MFMAs of the patch.  A part without sums has a zero-length descriptor: zeros.
        float gq[4][4];
        if constexpr (GS) {
            const int img = q / ppi, prem = q - img * ppi;
            const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
            // one address register (the lane's pixel of tile 0, register 0); tile m / register r is a scalar offset
            const unsigned ao = (unsigned)((img * a.hi + (y0 + 4 * wm)) * a.wi + (x0 + 4 * lq)) * ldab + (unsigned)gnl * 4u;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    gq[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, ao, (unsigned)(m * a.wi + r) * ldab, 0));
        }

        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][r] = bias;
        const float* Ab = smem + buf * ABUF;
        // gsum form: the 36 fragment addresses are formed per patch -- hoisted out of the patch loop (as hipcc does) they no longer fit
        // beside the sixteen aux values, and the spills landed in the DMA issue path (a scratch reload + vmcnt(0) in front of every
        // halo DMA: the DMAs of a patch ran one after the other, 84 instead of 131 TFLOP/s)
        int hbq = hb0;
        if constexpr (GS && HP % 8 != 0) asm volatile("" : "+v"(hbq));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // rows of the four M tiles are HP halo rows apart.  Pitch 18: the swizzle term (R >> 1) grows by 9 per tile, per-tile addresses;
            // pitch 24: by 12, the same chunk -- one address per tap, the tiles are immediates
            int fa[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int hrow = hbq + (HP % 8 == 0 ? 0 : m * HP) + tsh[t];
                fa[m] = hrow * 16 + (((lq + (hrow >> 1)) & 3) << 2) + (HP % 8 == 0 ? m * HP * 16 : 0);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if constexpr (ESZ == 4) {
                    f32x4 av[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) av[m] = *(const f32x4*)(Ab + c * ASTG + fa[m]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][e], bw[t][c][e], acc[m], 0, 0, 0);
                } else {
                    // two fragments at a time (128 VGPRs: 72 of weights, 16 accumulators)
#pragma unroll
                    for (int mh = 0; mh < 4; mh += 2) {
                        const f32x4 a0 = *(const f32x4*)(Ab + c * ASTG + fa[mh]), a1 = *(const f32x4*)(Ab + c * ASTG + fa[mh + 1]);
                        acc[mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bw[t][c]), acc[mh], 0, 0, 0);
                        acc[mh + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bw[t][c]), acc[mh + 1], 0, 0, 0);
                        asm volatile("" ::: "memory");
                    }
                }
            }
        }

        // ---- epilogue of patch q: accumulator register r of tile m = pixel (row 4 wm + m, column 4 lq + r), channel ncol
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
        if ((GS || a.stats) && img != simg) {
            flush(simg);
            simg = img;
        }
        float s1 = 0.f, s2 = 0.f;
        // the wave's 16 channels lie in one output part (n1 % 16 == 0): descriptor, pitch and channel offset are scalar selects (a per-lane
        // choice of the descriptor makes hipcc wrap every store in a readfirstlane loop); one address register -- the lane's pixel of
        // tile 0, register 0 -- and a scalar offset per (tile, register)
        const unsigned ldyb = (unsigned)(part0 ? a.ldy : a.ldy2) * (unsigned)ESZ;
        const unsigned yo = (unsigned)((img * a.hi + (y0 + 4 * wm)) * a.wi + (x0 + 4 * lq)) * ldyb + (unsigned)(part0 ? ncol : ncol - a.n1) * (unsigned)ESZ;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u = acc[m][r];
                const T vo = (T)shm_lrelu_max(u, a.slope);           // LeakyReLU for 0 <= slope <= 1 (checked by the launcher)
                const float v = (float)vo;                           // statistics of the value as stored
                s1 += v;
                if constexpr (GS) s2 += v * gq[m][r];
                else s2 = __builtin_fmaf(v, v, s2);
                if constexpr (abl::nostore)
                    asm volatile("" ::"v"(v));                           // timing only
                else if constexpr (ESZ == 4)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vo), part0 ? rsy : rsy2, yo, (unsigned)(m * a.wi + r) * ldyb, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, vo), part0 ? rsy : rsy2, yo, (unsigned)(m * a.wi + r) * ldyb, 0);
            }
        S1 += (double)s1;
        S2 += (double)s2;
        // halo(q + 1) was issued at the top of this patch; younger: this epilogue's sixteen stores (plus the rare flush; the gsum
        // form's aux loads were issued right behind the halo and have been consumed: loads return in order)
        if constexpr (abl::nostore)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // timing only: no stores behind the halo DMA
        else
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if constexpr (NM)
            if (q + 1 < q1) norm_a(q + 1, buf ^ 1);
    };
    if constexpr (NM == 2) {                 // one weight copy per image -> one segment of the patch range per image
        int q = q0;
        while (q < q1) {
            const int qe = min(q1, (q / ppi + 1) * ppi);
            if (q != q0) load_w(q / ppi);
            for (; q < qe; ++q) patch(q);
        }
    } else {
        for (int q = q0; q < q1; ++q) patch(q);
    }
    if (GS || a.stats) flush(simg);
}

// ------------------------------------------------------------------------------------------
// The four output phases of a 3x3 / stride-2 transposed product in ONE block (Conv2DTranspose forward, input gradient of
// the stride-2 convolution): tapgemm_dma_kernel runs them as four grid slices with 4 + 2 + 2 + 1 taps, i.e. K loops of one
// to four taps -- at 128 input channels a block lives for 8..32 K steps between its prologue and its epilogue, and every
// phase fetches the same input rows again (fp32 89-119, bf16 200-460 TFLOP/s).  Here a block owns 16 x 16 INPUT pixels
// (-> 32 x 32 output pixels) x 64 output channels: per 64-byte channel chunk the 18 x 18 halo (17 x 17 used) and the NINE
// weight slices are DMA'd once, two stages deep, ONE barrier per chunk; the nine (phase, tap) steps are unrolled with static
// fragment addresses (conflict-free halo swizzle of the static-tap halo kernel) and accumulate into the accumulators of
// their phase: 8 waves = 4 (M: 64 pixels) x 2 (N: 32 channels), 4 phases x 2 tiles = 128 accumulator registers.
// Phases arrive sorted by tap count 4, 2, 2, 1 (fill_s2_phases with pad_before = 0).  LDS: 2 x 24 KiB halo + 2 x 36 KiB weights.
template <typename T, typename TO>
__global__ __launch_bounds__(512, 2) void tapgemm_phase4_kernel(const TapGemmArgs a) {
    constexpr int ESZ = sizeof(T), CHE = 16 / ESZ, BKE = 64 / ESZ;
    constexpr int HC = 18, NIT = 24, NHR = NIT * 16;
    constexpr int ASTG = NHR * 16, BTAP = 64 * 16, BSTG = 9 * BTAP;      // floats per stage
    extern __shared__ __attribute__((aligned(1024))) float psm[];
    float* const sA = psm;
    float* const sB = psm + 2 * ASTG;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ppr = a.wi >> 4, ppi = (a.hi >> 4) * ppr;
    const int img = blockIdx.x / ppi, prem = blockIdx.x - img * ppi;
    const int y0 = (prem / ppr) << 4, x0 = (prem % ppr) << 4;
    const int n0 = blockIdx.y * 64;

    // ---- DMA lane constants: halo items wave, wave + 8, wave + 16 (16 halo rows each); weight items wave + 8 j < 36
    // (item = 4 * step + row group: 16 of the 64 weight rows of (phase, tap) step `item >> 2`)
    const int drow = lane >> 2, dq = lane & 3;
    unsigned arow[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int hrow = 16 * (wave + 8 * j) + drow;
        const int hr = hrow / HC, hc = hrow - hr * HC;
        const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
        const bool v = hrow < HC * HC && (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
        const int pix = (img * a.hi + iy) * a.wi + ix;
        arow[j] = v ? (unsigned)(pix * a.ldx + (dq ^ (((hrow >> 1) + hr) & 3)) * CHE) * (unsigned)ESZ : 0xffffffffu;
    }
    unsigned wrow[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int item = wave + 8 * j;
        const int step = item >> 2;                           // 0..3 phase 0, 4..5 phase 1, 6..7 phase 2, 8 phase 3
        const int ph = step < 4 ? 0 : step < 6 ? 1 : step < 8 ? 2 : 3;
        const int tp = step < 4 ? step : step < 6 ? step - 4 : step < 8 ? step - 6 : 0;
        const int row = (item & 3) * 16 + drow;
        const int nn = n0 + row;
        const bool v = item < 36 && nn < a.nout;
        const int wi_ = item < 36 ? a.ph[ph].widx[tp] : 0;
        wrow[j] = v ? (unsigned)((wi_ * a.nout + nn) * a.K + (dq ^ ((row >> 2) & 3)) * CHE) * (unsigned)ESZ : 0xffffffffu;
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
    const int nch = a.K / BKE;

    auto dma = [&](int chunk) {
        const unsigned cb = (unsigned)(chunk * BKE) * (unsigned)ESZ;
        float* da = sA + (chunk & 1) * ASTG + wave * 256;
        float* db = sB + (chunk & 1) * BSTG + wave * 256;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned off = arow[j] == 0xffffffffu ? arow[j] : arow[j] + cb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(da + j * 8 * 256), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j < 4 || wave < 4) {                           // wave-uniform: items 32..35 belong to waves 0..3
                const unsigned off = wrow[j] == 0xffffffffu ? wrow[j] : wrow[j] + cb;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(db + j * 8 * 256), 16, (int)off, 0, 0, 0);
            }
        }
    };

    f32x16 acc[4][2][1];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][i][0][r] = 0.f;

    // fragment addresses (floats, relative to the stage): A per (step, tile); k group 1 = address ^ 8.  B: rows wn*32 + l31
    int fs[9][2];
    {
        int hb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) hb[i] = (4 * wm + 2 * i + (l31 >> 4) + 1) * HC + (l31 & 15) + 1;
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            const int ph = st < 4 ? 0 : st < 6 ? 1 : st < 8 ? 2 : 3;
            const int tp = st < 4 ? st : st < 6 ? st - 4 : st < 8 ? st - 6 : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hrow = hb[i] + a.ph[ph].dh[tp] * HC + a.ph[ph].dw[tp];
                fs[st][i] = hrow * 16 + ((h ^ (((hrow >> 1) + hrow / HC) & 3)) << 2);
            }
        }
    }
    const int swb = (l31 >> 2) & 3;
    const int fb0 = (wn * 32 + l31) * 16 + ((0 + h) ^ swb) * 4, fb1 = (wn * 32 + l31) * 16 + ((2 + h) ^ swb) * 4;
    typedef const __attribute__((address_space(3))) f32x4* lds_f4;
    const unsigned sA_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)sA;

    dma(0);
    for (int chunk = 0; chunk < nch; ++chunk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SHM_LDS_BARRIER();
        asm volatile("" ::: "memory");
        if (chunk + 1 < nch) dma(chunk + 1);                   // the other stage: last read before this barrier
        const unsigned Ab = sA_lds + (unsigned)((chunk & 1) * ASTG * 4);
        const float* Bb = sB + (chunk & 1) * BSTG;
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            constexpr int kPhase[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x4 av[2], bv[1];
#pragma unroll
                for (int i = 0; i < 2; ++i) av[i] = *(lds_f4)(size_t)((Ab + (unsigned)(fs[st][i] << 2)) ^ (unsigned)(kk << 5));
                bv[0] = *(const f32x4*)(Bb + st * BTAP + (kk ? fb1 : fb0));
                tap_mfma<T, 2, 1>(av, bv, acc[kPhase[st]]);
            }
        }
        asm volatile("" ::: "memory");
    }

    // ---- epilogue: bias + LeakyReLU, phase p of input pixel (y, x) -> output pixel (2y + oph, 2x + opw)
    constexpr bool kWide = sizeof(TO) == 2;
    const bool wide = kWide && (a.ldy % 8 == 0) && (((size_t)a.y & 15) == 0);
    const int ncol = n0 + wn * 32 + l31;
    const float bcol = (a.bias && ncol < a.nout) ? a.bias[ncol] : 0.f;
    if constexpr (kWide) if (wide) {
        __syncthreads();                                         // every wave is past its last fragment read
        unsigned short* tile = (unsigned short*)psm + wave * (64 * 32);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const TO vo = (TO)shm_lrelu(acc[p][i][0][r] + bcol, a.slope);
                    // 16-byte chunk c of row `row` lives at chunk c ^ ((row >> 1) & 3)
                    tile[row * 32 + ((((l31 >> 3) ^ ((row >> 1) & 3)) << 3) | (l31 & 7))] = __builtin_bit_cast(unsigned short, vo);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave hand-off (ds ops of a wave complete in order)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = it * 64 + lane, row = q >> 2, ch = q & 3;
                const u32x4 v = *(const u32x4*)(tile + row * 32 + ((ch ^ ((row >> 1) & 3)) << 3));
                const int i = row >> 5, r32 = row & 31;
                const int py = 4 * wm + 2 * i + (r32 >> 4), px = r32 & 15;
                const size_t opix = ((size_t)img * a.ho + (2 * (y0 + py) + a.ph[p].oph)) * a.wo + (2 * (x0 + px) + a.ph[p].opw);
                const int n = n0 + wn * 32 + ch * 8;
                if (n < a.nout) *(u32x4*)((unsigned short*)a.y + opix * a.ldy + n) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    // Element stores: a lane's output address is ONE register per (phase, tile) -- its pixel of accumulator row 0 -- plus a scalar
    // offset per row (row r of a lane is input pixel (r >> 3, 8 ((r >> 2) & 1) + (r & 3)) of the tile's two patch rows, i.e. twice
    // that in output pixels), through a scalar descriptor when the output is below 4 GiB: no per-element address arithmetic
    // (it was ~5 VALU instructions per element, 128 elements per lane, in a kernel whose epilogue nobody overlaps: one block per CU)
    const bool ybuf = a.ybytes != 0;
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000);
    const unsigned ldyb = (unsigned)a.ldy * (unsigned)sizeof(TO), nyb = (unsigned)(ncol < a.nout ? ncol : 0) * (unsigned)sizeof(TO);
    auto lane_pix0 = [&](int p, int i) {                  // output pixel of the lane's accumulator row 0 of tile i, phase p
        return (unsigned)((img * a.ho + (2 * (y0 + 4 * wm + 2 * i) + a.ph[p].oph)) * a.wo + (2 * (x0 + 4 * h) + a.ph[p].opw));
    };
    auto row_pix = [&](int r) { return (unsigned)(2 * (r >> 3) * a.wo + 16 * ((r >> 2) & 1) + 2 * (r & 3)); };       // scalar
    if (!wide && a.gred[0] == nullptr && ybuf) {
        if (ncol < a.nout) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned yo = lane_pix0(p, i) * ldyb + nyb;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const TO vo = (TO)shm_lrelu(acc[p][i][0][r] + bcol, a.slope);
                        if constexpr (sizeof(TO) == 4)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vo), rsy, yo, row_pix(r) * ldyb, 0);
                        else
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, vo), rsy, yo, row_pix(r) * ldyb, 0);
                    }
                }
        }
    }
    if (!wide && a.gred[0] == nullptr && !ybuf) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int py = 4 * wm + 2 * i + (row >> 4), px = row & 15;
                    const size_t opix = ((size_t)img * a.ho + (2 * (y0 + py) + a.ph[p].oph)) * a.wo + (2 * (x0 + px) + a.ph[p].opw);
                    if (ncol < a.nout) ((TO*)a.y)[opix * a.ldy + ncol] = (TO)shm_lrelu(acc[p][i][0][r] + bcol, a.slope);
                }
    }
    if (!wide && a.gred[0] != nullptr) {
        // gsum (fp32 outputs; the launcher does not fuse bf16 ones): the stride-2 input gradient of a discriminator block writes
        // the gradient at the previous block's InstanceNorm output.  Sixteen aux loads per 32 x 32 tile in front of its stores
        // (scalar descriptor, 32-bit offsets); the block's 32 x 32 output pixels belong to one image: one pair of atomics per column.
        const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a.gaux[0], 0, 0xfffffff0u, 0x00020000);
        const unsigned ldab = (unsigned)a.ldgaux[0] * (unsigned)sizeof(T), nlb = (unsigned)(ncol < a.nout ? ncol : 0) * (unsigned)sizeof(T);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned pix0 = lane_pix0(p, i);
                const unsigned ao = pix0 * ldab + nlb, yo = pix0 * ldyb + nyb;
                float q[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (sizeof(T) == 4)
                        q[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, ao, row_pix(r) * ldab, 0));
                    else
                        q[r] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsa, ao, row_pix(r) * ldab, 0) << 16);
                }
                if (ncol < a.nout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const TO vo = (TO)shm_lrelu(acc[p][i][0][r] + bcol, a.slope);
                        const float v = (float)vo;
                        s1 += v;
                        s2 += v * q[r];
                        if constexpr (sizeof(TO) == 4)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vo), rsy, yo, row_pix(r) * ldyb, 0);
                        else
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, vo), rsy, yo, row_pix(r) * ldyb, 0);
                    }
                }
            }
        const float t1 = s1 + __shfl_xor(s1, 32, 64), t2 = s2 + __shfl_xor(s2, 32, 64);
        if (h == 0 && ncol < a.nout) {
            const int slot = (prem * 4 + wm) % a.gslots;
            double* dst = a.gred[0] + ((size_t)slot * a.gbatch * a.nout + (size_t)img * a.nout + ncol) * 2;
            atomicAdd(dst, (double)t1);
            atomicAdd(dst + 1, (double)t2);
        }
    }
}

static thread_local double* g_conv_stats = nullptr;     // set by shm_conv2d_in_fwd around its conv launch
static thread_local int g_conv_hw = 0, g_conv_slots = 1;

// gsum request of the *_gsum entry points around their launch, and whether the kernel that ran took it (otherwise the entry
// point follows up with the stand-alone reduce pass, shm_gsum_reduce_internal)
struct GsumReq {
    const void* aux[2];
    int ld[2];
    double* red[2];
};
static thread_local GsumReq g_gsum = {};
static thread_local bool g_gsum_fused = false;

// norm request of shm_conv2d_in_fwd_norm around its launch (TapGemmArgs::nt); query = shm_conv2d_norm_supported's dry run: the
// launcher goes through its variant choice, records whether that kernel can normalise its source in LDS, and launches nothing
struct NormReq {
    const float* nt;
    int part, c, mode;
    bool query, query_ok;
};
static thread_local NormReq g_norm = {};

// Variant choice.  `forced` (shm_set_tuning("tapgemm.variant", SHM_TG_*)) overrides the automatic choice; a forced
// variant the shape is not eligible for is an error (SHM_E_SHAPE), so a parity test that forces a variant knows it ran.
template <typename T, typename TO>
static int launch_tapgemm_t(const TapGemmArgs& a_in, int batch, int nphase, hipStream_t st, const char* who) {
    TapGemmArgs a = a_in;
    constexpr int BKE = 64 / (int)sizeof(T);
    const char* tn = sizeof(T) == 4 ? "float" : "__bf16";
    const char* ton = sizeof(TO) == 4 ? "float" : "__bf16";
    const int forced = shm_tune(SHM_TUNE_TAPGEMM_VARIANT);
    // the 16x16-patch halo kernels: unit-stride 3x3 (forward or flipped taps), whole patches
    bool halo_ok = nphase == 1 && a.is == 1 && a.os == 1 && a.ph[0].ntaps == 9 && a.hi % 16 == 0 && a.wi % 16 == 0 && a.hg == a.hi && a.wg == a.wi;
    if (halo_ok)
        for (int t = 0; t < 9; ++t)             // every tap within the 1-pixel halo
            halo_ok = halo_ok && a.ph[0].dh[t] >= -1 && a.ph[0].dh[t] <= 1 && a.ph[0].dw[t] >= -1 && a.ph[0].dw[t] <= 1;
    const bool bk32_ok = a.K % (2 * BKE) == 0 && (a.x2 == nullptr || a.c1 % (2 * BKE) == 0);
    // weights-in-registers kernel: bf16, one source tensor with 32 or 64 channels
    const bool wreg_ok = sizeof(T) == 2 && halo_ok && a.x2 == nullptr && (a.K == 32 || a.K == 64) && a.ybytes != 0 && (a.y2 == nullptr || a.y2bytes != 0) &&
                         a.slope >= 0.f && a.slope <= 1.f &&
                         (sizeof(TO) == 4 || (a.nout % 64 == 0 && a.n1 % 32 == 0 && a.ldy % 8 == 0 && ((size_t)a.y & 15) == 0 &&
                                              (a.y2 == nullptr || (a.ldy2 % 8 == 0 && ((size_t)a.y2 & 15) == 0))));
    // ... and its fp32 form: 16 or 64 input channels
    // (also 16 / 32 output channels on 32- / 16-row patches, K = 32, and SpecSeg's Concatenate of two 16-channel tensors into 16)
    const int wreg32_wn = a.nout % 64 == 0 ? 4 : a.nout == 32 ? 2 : a.nout == 16 ? 1 : 0;
    const bool wreg32_ok = sizeof(T) == 4 && sizeof(TO) == 4 && halo_ok && (a.x2 == nullptr || (a.c1 == 16 && a.K == 32 && a.nout == 16)) && a.ybytes != 0 &&
                           (a.y2 == nullptr || a.y2bytes != 0) && a.slope >= 0.f && a.slope <= 1.f && a.n1 % 16 == 0 &&
                           ((wreg32_wn == 4 && (a.K == 16 || a.K == 32 || a.K == 64)) || (wreg32_wn == 2 && a.hi % 16 == 0 && (a.K == 16 || a.K == 32)) ||
                            (wreg32_wn == 1 && a.hi % 32 == 0 && (a.K == 16 || a.K == 32)));
    // the four phases of a stride-2 transposed product fused in one block: 16 x 16 input patches, 64-channel output slices
    bool phase4_ok = nphase == 4 && a.is == 1 && a.os == 2 && a.x2 == nullptr && a.y2 == nullptr && a.stats == nullptr && a.hi % 16 == 0 &&
                     a.wi % 16 == 0 && a.hg == a.hi && a.wg == a.wi && a.ho == 2 * a.hi && a.wo == 2 * a.wi && a.nout % 64 == 0 &&
                     a.ph[0].ntaps == 4 && a.ph[1].ntaps == 2 && a.ph[2].ntaps == 2 && a.ph[3].ntaps == 1;
    if (phase4_ok)
        for (int p = 0; p < 4; ++p)
            for (int t = 0; t < a.ph[p].ntaps; ++t)
                phase4_ok = phase4_ok && a.ph[p].dh[t] >= -1 && a.ph[p].dh[t] <= 1 && a.ph[p].dw[t] >= -1 && a.ph[p].dw[t] <= 1;
    int v = forced;
    if (v == SHM_TG_AUTO) {
        const long tiles128 = (long)shm_cdiv(a.M, 128) * shm_cdiv(a.nout, 128) * nphase;
        if (wreg_ok || wreg32_ok) {
            v = SHM_TG_WREG;
        } else if (phase4_ok && (long)batch * (a.hi / 16) * (a.wi / 16) * (a.nout / 64) >= shm_tune(SHM_TUNE_TAPGEMM_PHASE4_MIN) &&
                   (sizeof(T) == 2 || a.K <= 256)) {
            // tools/bench_phase4.py, n = 40 / 8: Conv2DTranspose 128 -> 64 fp32 832 vs 1153 us (bf16 184 vs 304), 256 -> 128 751 vs 863
            // (135 vs 229); stride-2 input gradient 64 <- 128, n = 96: 498 vs 645 (113 vs 148).  One 8-wave block per CU (120 KiB of
            // LDS), so from 512 input channels on the fp32 DMA tiles (two blocks per CU, long K loops) are level or ahead
            // (512 -> 256: 850 vs 820 us); bf16 stays ahead (129 vs 176) down to one round of blocks.
            v = SHM_TG_PHASE4;
        } else if (halo_ok) {
            // The static-tap halo kernels beat the DMA tiles on every unit-stride 3x3 layer they can take, small grids included
            // (round-2 A/B, tools/bench_variants.py: fp32 n = 8 maps 120-134 vs 107-121 TFLOP/s).  128 or 64 output channels per
            // block: the 128-wide block is ~3 % faster when both fill the chip, but it has half the blocks -- two 8-wave (or
            // 4-wave) blocks fit a CU, i.e. 512 slots -- so below two rounds (bf16: below one) the choice goes by how full the last
            // round is (bf16 n = 8 at 32 x 32: 62 us on the 64-wide block against 76).
            const long np16 = (long)batch * (a.hi / 16) * (a.wi / 16);
            const long nb128 = np16 * shm_cdiv(a.nout, 128), nb64 = np16 * shm_cdiv(a.nout, 64);
            auto fill = [](long nb) { return (double)nb / (double)(((nb + 511) / 512) * 512); };
            if (sizeof(T) == 4 && nb64 < 256)
                // fewer 64-wide halo blocks than CUs (SpecSeg's deep layers at n = 8, any model at batch 1): the 64 x 64 DMA tile
                // has four times the blocks -- fp32 n = 8: 128 -> 128 @32x32 99 -> 50 us, 256 -> 256 @16x16 165 -> 77,
                // 128 -> 64 @64x64 100 -> 70; level with the halo block in bf16, where it is not taken
                v = SHM_TG_DMA_64x64;
            else if (a.nout <= 64)
                v = SHM_TG_HALO64_ST;
            else if (sizeof(T) == 4 && nb128 < shm_tune(SHM_TUNE_TAPGEMM_HALO_MIN))
                // fp32 is MFMA bound, so what counts is the busiest CU: blocks are dealt out over 256 CUs, a 64-wide block is half
                // the work at ~4 % less efficiency.  Matches every A/B point of tools/bench_variants.py (n = 2..40 on the 128-, 256-
                // and 512-channel layers), e.g. 320 blocks (n = 40, 32 x 32, 256 <- 512): 2 units against 3 x 0.52 -- 1085 vs 865 us
                v = (double)((nb128 + 255) / 256) <= (double)((nb64 + 255) / 256) * 0.52 ? SHM_TG_HALO128_ST : SHM_TG_HALO64_ST;
            else if (nb128 >= (sizeof(T) == 2 ? 512 : shm_tune(SHM_TUNE_TAPGEMM_HALO_MIN)) || fill(nb128) * 1.03 >= fill(nb64))
                v = SHM_TG_HALO128_ST;
            else
                v = SHM_TG_HALO64_ST;
        } else if ((long)shm_cdiv(a.M, 64) * shm_cdiv(a.nout, 128) * nphase < 256) {
            v = SHM_TG_DMA_64x64;           // not even one 64x128 tile per CU
        } else if (nphase == 1 && a.nout > 64 && a.K <= (sizeof(T) == 2 ? 256 : 128) &&
                   (long)shm_cdiv(a.M, 256) * shm_cdiv(a.nout, 128) >= 256) {
            // stride-2 forward products with a short K loop (discriminator blocks, Conv2DTranspose input gradients): eight waves on a
            // 256 x 128 tile amortise the per-step barrier and the weight slice over twice the rows (tools/bench_s2.py: bf16 64 -> 128
            // @256x256 n = 40 230 -> 183 us, n = 96 @128x128 135 -> 110; fp32 859 -> 801, 522 -> 490; from K = 256 (fp32) / 512 (bf16) on it loses)
            v = SHM_TG_DMA_256x128;
        } else if (nphase == 1 && a.nout > 64 && bk32_ok && (sizeof(T) == 2 ? a.K >= 256 : (a.K >= 256 && a.K < 512 && a.is == 2 && tiles128 >= 512))) {
            // long K: twice the channels per K step -- twice the MFMAs per barrier (bf16 256 -> 512 @32x32 n = 96: 109 -> 88 us, 512 -> 1024
            // @16x16: 108 -> 83; fp32 stride-2 forward 256 -> 512 @64x64 n = 40: 807 -> 749 us, @32x32 n = 96: 486 -> 462, round 3 tools/bench_s2.py)
            v = SHM_TG_DMA_128x128_BK32;
        } else if (a.nout > 64 && tiles128 < shm_tune(SHM_TUNE_TAPGEMM_SMALL_GRID)) {
            // small grids (16x16 maps, the n = 8 pass of the stride-2 / transposed layers): 64-row tiles double the number of
            // blocks, so a CU holds two waves per SIMD instead of one and the K-step bubbles of one wave hide behind the other's MFMAs
            v = SHM_TG_DMA_64x128;
        } else if (a.nout > 64) {
            v = SHM_TG_DMA_128x128;
        } else {
            v = SHM_TG_DMA_128x64;
        }
    }
    // gsum (input-gradient launches): which kernels take the sums in their epilogue.  The LDS-staged bf16 epilogues read aux 16
    // bytes at a time; the DMA tiles need whole wave tiles per sample; the weights-in-registers kernels have gsum instantiations for
    // 64 output channels per block; the fused four-phase kernel has none.  Anything else: the entry point runs the reduce pass.
    const bool want_gs = a.gred[0] || a.gred[1];
    bool gs_fused = false;
    if (want_gs) {
        bool al = true;
        for (int p = 0; p < 2; ++p)
            if (a.gred[p]) al = al && (sizeof(TO) == 4 || (a.ldgaux[p] % 8 == 0 && ((size_t)a.gaux[p] & 15) == 0));
        al = al && (a.y2 == nullptr || a.n1 % 32 == 0);
        const bool small = a.ybytes != 0 && (a.y2 == nullptr || a.y2bytes != 0);        // 32-bit offsets into aux (it has the output's extent)
        // bf16 outputs: the sums are taken in the LDS-staged 16-byte store path, which has its own alignment conditions
        const bool wide_ok = sizeof(TO) == 4 || ((a.nout % 8 == 0) && (a.n1 % 8 == 0) && (a.ldy % 8 == 0) && (((size_t)a.y & 15) == 0) &&
                                                 (a.y2 == nullptr || ((a.ldy2 % 8 == 0) && (((size_t)a.y2 & 15) == 0))));
        al = al && wide_ok;
        switch (v) {
        case SHM_TG_HALO128_ST: case SHM_TG_HALO64_ST:          // (the other halo forms are forced-only variants: reduce pass)
            gs_fused = al && small && (a.y2 == nullptr || a.n1 % 64 == 0);
            break;
        case SHM_TG_DMA_128x128: case SHM_TG_DMA_64x128: case SHM_TG_DMA_128x64: case SHM_TG_DMA_64x64: case SHM_TG_DMA_256x64:
        case SHM_TG_DMA_256x128: case SHM_TG_DMA_128x128_BK32: case SHM_TG_DMA_128x128_NST4:
            gs_fused = al && (a.hg * a.wg) % 64 == 0;
            break;
        case SHM_TG_WREG:
            gs_fused = al && small && ((wreg_ok && sizeof(TO) == 2) || (wreg32_ok && wreg32_wn == 4 && a.x2 == nullptr));
            break;
        case SHM_TG_PHASE4:                                     // fp32 outputs only (the element-store epilogue)
            gs_fused = sizeof(TO) == 4 && small && a.gred[1] == nullptr;
            break;
        default:
            break;
        }
        if (!gs_fused) a.gred[0] = a.gred[1] = nullptr;
    }
    g_gsum_fused = gs_fused;
    // norm: the kernels that stage the A operand as a halo image in LDS (static-tap halo blocks, weights-in-registers kernels) can
    // normalise it there; the part's channels must fit the LDS table and be whole 64-byte rows
    const bool want_nm = a.nt != nullptr;
    if (want_nm) {
        const int pc = a.ntpart ? a.K - a.c1 : a.c1;
        bool ok = !want_gs && nphase == 1 && a.ntc == pc && pc % BKE == 0 && pc <= SHM_NT_MAXC && (a.ntpart == 0 || a.x2 != nullptr);
        switch (v) {
        case SHM_TG_HALO128_ST: case SHM_TG_HALO64_ST:
            ok = ok && halo_ok && sizeof(TO) == sizeof(T);
            break;
        case SHM_TG_WREG:           // one source; its planes travel as one 1 KiB DMA piece per wave
            ok = ok && a.ntpart == 0 && pc <= 64 && ((wreg_ok && sizeof(TO) == 2) || (wreg32_ok && wreg32_wn == 4 && a.x2 == nullptr));
            break;
        default:
            ok = false;
        }
        if (g_norm.query) {
            g_norm.query_ok = ok;
            return SHM_OK;
        }
        SHM_REQUIRE(ok, SHM_E_SHAPE,
                    "%s: the kernel this shape runs on (tapgemm variant %d) cannot normalise its source in LDS (unit-stride 3x3 on a map that is a "
                    "multiple of 16, normalised part of at most %d channels; ask shm_conv2d_norm_supported) -- use shm_in_apply", who, v, SHM_NT_MAXC);
    } else if (g_norm.query) {
        g_norm.query_ok = false;
        return SHM_OK;
    }
    auto grid1d = [&](int bm, int bn) { return dim3(shm_cdiv(a.M, bm), shm_cdiv(a.nout, bn), nphase); };
    const int npatch = batch * (a.hi / 16) * (a.wi / 16);
    // "conv.f32_split" (opt-in): the fp32 static-tap halo layers (and the weights-in-registers layers) as six bf16 MFMA products (conv_fwd_x3.hip)
    if constexpr (sizeof(T) == 4 && sizeof(TO) == 4) {
        if (shm_tune(SHM_TUNE_CONV_F32_SPLIT) == 1 && halo_ok && (!want_nm || a.ntmode == 0) &&
            (v == SHM_TG_HALO128_ST || v == SHM_TG_HALO64_ST || (v == SHM_TG_WREG && forced == SHM_TG_AUTO)) && (!want_gs || gs_fused) && shm_x3_fwd_eligible(a))
            return shm_x3_fwd_launch(a, batch, gs_fused, st, who);
    }
    switch (v) {
    case SHM_TG_HALO128:
        SHM_REQUIRE(halo_ok, SHM_E_SHAPE, "%s: forced variant halo128 needs a unit-stride 3x3 layer on a map that is a multiple of 16", who);
        hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 128>), dim3(npatch, shm_cdiv(a.nout, 128), 1), dim3(512), 0, st, a);
        shm_set_last_kernel("tapgemm_halo_kernel<%s, %s, 128, 16, false, 2>", tn, ton);
        break;
    case SHM_TG_HALO64:
        SHM_REQUIRE(halo_ok, SHM_E_SHAPE, "%s: forced variant halo64 needs a unit-stride 3x3 layer on a map that is a multiple of 16", who);
        hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 64>), dim3(npatch, shm_cdiv(a.nout, 64), 1), dim3(256), 0, st, a);
        shm_set_last_kernel("tapgemm_halo_kernel<%s, %s, 64, 16, false, 2>", tn, ton);
        break;
    case SHM_TG_HALO128_ST:
        SHM_REQUIRE(halo_ok, SHM_E_SHAPE, "%s: forced variant halo128/static-taps needs a unit-stride 3x3 layer on a map that is a multiple of 16", who);
        if (gs_fused)
            hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 128, 16, true, 2, true>), dim3(npatch, shm_cdiv(a.nout, 128), 1), dim3(512), 0, st, a);
        else if (want_nm) {
            if constexpr (sizeof(T) == sizeof(TO)) {
                if (a.ntmode)
                    hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 128, 16, true, 2, false, 2>), dim3(npatch, shm_cdiv(a.nout, 128), 1), dim3(512), 0, st, a);
                else
                    hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 128, 16, true, 2, false, 1>), dim3(npatch, shm_cdiv(a.nout, 128), 1), dim3(512), 0, st, a);
            }
        } else
            hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 128, 16, true>), dim3(npatch, shm_cdiv(a.nout, 128), 1), dim3(512), 0, st, a);
        shm_set_last_kernel(gs_fused ? "tapgemm_halo_kernel<%s, %s, 128, 16, true, 2, true>"
                            : want_nm ? (a.ntmode ? "tapgemm_halo_kernel<%s, %s, 128, 16, true, 2, false, 2>" : "tapgemm_halo_kernel<%s, %s, 128, 16, true, 2, false, 1>")
                                      : "tapgemm_halo_kernel<%s, %s, 128, 16, true, 2>", tn, ton);
        break;
    case SHM_TG_HALO128_ST_W4:
        SHM_REQUIRE(halo_ok, SHM_E_SHAPE, "%s: forced variant halo128/static-taps/4 waves needs a unit-stride 3x3 layer on a map that is a multiple of 16", who);
        hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 128, 16, true, 4>), dim3(npatch, shm_cdiv(a.nout, 128), 1), dim3(256), 0, st, a);
        shm_set_last_kernel("tapgemm_halo_kernel<%s, %s, 128, 16, true, 4>", tn, ton);
        break;
    case SHM_TG_HALO64_ST:
        SHM_REQUIRE(halo_ok, SHM_E_SHAPE, "%s: forced variant halo64/static-taps needs a unit-stride 3x3 layer on a map that is a multiple of 16", who);
        if (gs_fused)
            hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 64, 16, true, 2, true>), dim3(npatch, shm_cdiv(a.nout, 64), 1), dim3(256), 0, st, a);
        else if (want_nm) {
            if constexpr (sizeof(T) == sizeof(TO)) {
                if (a.ntmode)
                    hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 64, 16, true, 2, false, 2>), dim3(npatch, shm_cdiv(a.nout, 64), 1), dim3(256), 0, st, a);
                else
                    hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 64, 16, true, 2, false, 1>), dim3(npatch, shm_cdiv(a.nout, 64), 1), dim3(256), 0, st, a);
            }
        } else
            hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 64, 16, true>), dim3(npatch, shm_cdiv(a.nout, 64), 1), dim3(256), 0, st, a);
        shm_set_last_kernel(gs_fused ? "tapgemm_halo_kernel<%s, %s, 64, 16, true, 2, true>"
                            : want_nm ? (a.ntmode ? "tapgemm_halo_kernel<%s, %s, 64, 16, true, 2, false, 2>" : "tapgemm_halo_kernel<%s, %s, 64, 16, true, 2, false, 1>")
                                      : "tapgemm_halo_kernel<%s, %s, 64, 16, true, 2>", tn, ton);
        break;
    case SHM_TG_HALO128_PH8:
        // 8-row patches (3 four-wave blocks per CU, 4-wave barriers): measured equal or slower than 16-row patches in
        // bf16 (806-975 vs 795-994 TFLOP/s over the four big layer shapes).  Kept selectable.
        SHM_REQUIRE(halo_ok && sizeof(T) == 2, SHM_E_SHAPE, "%s: forced variant halo128/ph8 is bf16, unit-stride 3x3, map multiple of 16", who);
        if constexpr (sizeof(T) == 2) {
            hipLaunchKernelGGL((tapgemm_halo_kernel<T, TO, 128, 8>), dim3(batch * (a.hi / 8) * (a.wi / 16), shm_cdiv(a.nout, 128), 1), dim3(256), 0, st, a);
            shm_set_last_kernel("tapgemm_halo_kernel<%s, %s, 128, 8, false, 2>", tn, ton);
        }
        break;
    case SHM_TG_PHASE4: {
        SHM_REQUIRE(phase4_ok, SHM_E_SHAPE,
                    "%s: forced variant phase4 needs a four-phase stride-2 transposed 3x3 product on a 16-aligned input map, one source and one "
                    "destination tensor, Cout %% 64 == 0, no fused statistics", who);
        constexpr unsigned kLds = (2 * 24 * 16 * 16 + 2 * 9 * 64 * 16) * sizeof(float);       // 120 KiB
        static const hipError_t attr = hipFuncSetAttribute((const void*)tapgemm_phase4_kernel<T, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        SHM_REQUIRE(attr == hipSuccess, SHM_E_HIP, "%s: cannot reserve 120 KiB of LDS: %s", who, hipGetErrorString(attr));
        hipLaunchKernelGGL((tapgemm_phase4_kernel<T, TO>), dim3(npatch, a.nout / 64, 1), dim3(512), kLds, st, a);
        shm_set_last_kernel("tapgemm_phase4_kernel<%s, %s>", tn, ton);
        break;
    }
    case SHM_TG_WREG: {
        SHM_REQUIRE(wreg_ok || wreg32_ok, SHM_E_SHAPE,
                    "%s: forced variant wreg needs a unit-stride 3x3 layer on a map that is a multiple of 16, slope in [0,1] and: bf16 -- one source "
                    "tensor with 32/64 channels, Cout %% 64 == 0; fp32 -- one source with 16/32/64 input channels and "
                    "Cout %% 64 == 0, or 16/32 input channels with Cout = 32 (map a multiple of 16) or 16 (map a multiple of 32; also 16 + 16 from two tensors)", who);
        static const int ncu = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            return n;
        }();
        const int np8 = batch * (a.hi / 8) * (a.wi / 16), ny = shm_cdiv(a.nout, 64);
        if constexpr (sizeof(T) == 2) {
            int gx = 2 * ncu / ny;             // two 4-wave blocks per CU (LDS, VGPRs)
            if (gx < 1) gx = 1;
            if (gx > np8) gx = np8;
            if constexpr (sizeof(TO) == 2) {
                if (gs_fused && a.K == 64)
                    hipLaunchKernelGGL((tapgemm_wreg_kernel<TO, 2, true>), dim3(gx, ny, 1), dim3(256), 0, st, a, np8);
                else if (gs_fused)
                    hipLaunchKernelGGL((tapgemm_wreg_kernel<TO, 1, true>), dim3(gx, ny, 1), dim3(256), 0, st, a, np8);
            }
            if constexpr (sizeof(TO) == 2) {
                if (want_nm && a.K == 64 && a.ntmode)
                    hipLaunchKernelGGL((tapgemm_wreg_kernel<TO, 2, false, 2>), dim3(gx, ny, 1), dim3(256), 0, st, a, np8);
                else if (want_nm && a.K == 64)
                    hipLaunchKernelGGL((tapgemm_wreg_kernel<TO, 2, false, 1>), dim3(gx, ny, 1), dim3(256), 0, st, a, np8);
                else if (want_nm && a.ntmode)
                    hipLaunchKernelGGL((tapgemm_wreg_kernel<TO, 1, false, 2>), dim3(gx, ny, 1), dim3(256), 0, st, a, np8);
                else if (want_nm)
                    hipLaunchKernelGGL((tapgemm_wreg_kernel<TO, 1, false, 1>), dim3(gx, ny, 1), dim3(256), 0, st, a, np8);
            }
            // "tapgemm.wreg16": the eight-wave kernel with 16-column wave tiles and line-wide stores (tapgemm_wreg16_bf16_kernel): plain bf16 -> bf16
            // launches from one source tensor whose 64-channel blocks lie in one output part
            bool w16 = false;
            if constexpr (sizeof(TO) == 2)
                w16 = shm_tune(SHM_TUNE_TAPGEMM_WREG16) != 0 && !gs_fused && !want_nm && a.nout % 64 == 0 && (a.y2 == nullptr || a.n1 % 64 == 0) && a.x2 == nullptr && a.ybytes != 0 && (a.y2 == nullptr || a.y2bytes != 0);
            if constexpr (sizeof(TO) == 2) if (w16) {
                // "tapgemm.wreg16" = 2 (default): the K = 64 layers on maps of 8 x 32-pixel patches take the ping-pong kernel (conv_pingpong.hip)
                const int rc16 = shm_tune(SHM_TUNE_TAPGEMM_WREG16) == 2 && shm_pp_eligible(a) ? shm_pp_launch(a, batch, ncu, st, who) : shm_wreg16_launch(a, np8, ncu, st, who);
                if (rc16 != SHM_OK) return rc16;
            }
            if (gs_fused || want_nm || w16) {
            } else if (a.K == 64)
                hipLaunchKernelGGL((tapgemm_wreg_kernel<TO, 2>), dim3(gx, ny, 1), dim3(256), 0, st, a, np8);
            else
                hipLaunchKernelGGL((tapgemm_wreg_kernel<TO, 1>), dim3(gx, ny, 1), dim3(256), 0, st, a, np8);
            if (!w16)
            shm_set_last_kernel(gs_fused ? "tapgemm_wreg_kernel<%s, %d, true>"
                                : want_nm ? (a.ntmode ? "tapgemm_wreg_kernel<%s, %d, false, 2>" : "tapgemm_wreg_kernel<%s, %d, false, 1>") : "tapgemm_wreg_kernel<%s, %d>",
                                ton, a.K / 32);
        } else if constexpr (sizeof(TO) == 4) {
            // one 8-wave block per CU; patches of 8 (64 channels per block), 16 (32) or 32 (16) rows
            const int ph = 32 / wreg32_wn, npw = batch * (a.hi / ph) * (a.wi / 16), nyw = a.nout / (16 * wreg32_wn);
            const int nch = a.K / 16;
            // one 8-wave block per CU; the plain 16-input-channel, 64-column form (100 VGPRs, 30 KiB of LDS) fits two: 399 -> 382 us on the
            // generator's first layer (16 -> 64 @256^2, n = 40)
            const int per_cu = (nch == 1 && wreg32_wn == 4 && !want_nm && !gs_fused) ? 2 : 1;
            int gx = per_cu * ncu / nyw;
            if (gx < 1) gx = 1;
            if (gx > npw) gx = npw;
            const unsigned lds = 2u * (unsigned)nch * (unsigned)wreg32_nit(wreg32_wn) * 1024u + (want_nm ? 8u * 1024u : 0u);      // two halo buffers (+ norm: 1 KiB of planes per wave)
            hipError_t attr = hipSuccess;
#define SHM_WREG32_LAUNCH2(NCH_, WN_, TWO_)                                                                                              \
    do {                                                                                                                                 \
        static const hipError_t at_ = hipFuncSetAttribute((const void*)tapgemm_wreg_f32_kernel<NCH_, WN_, TWO_>,                         \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize,                                    \
                                                          2 * NCH_ * wreg32_nit(WN_) * 1024);                        \
        attr = at_;                                                                                                                      \
        if (attr == hipSuccess)                                                                                                          \
            hipLaunchKernelGGL((tapgemm_wreg_f32_kernel<NCH_, WN_, TWO_>), dim3(gx, nyw, 1), dim3(512), lds, st, a, npw);                \
    } while (0)
#define SHM_WREG32_LAUNCH(NCH_, WN_) SHM_WREG32_LAUNCH2(NCH_, WN_, false)
#define SHM_WREG32_LAUNCH_GS(NCH_)                                                                                                       \
    do {                                                                                                                                 \
        static const hipError_t at_ = hipFuncSetAttribute((const void*)tapgemm_wreg_f32_kernel<NCH_, 4, false, true>,                    \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NCH_ * wreg32_nit(4) * 1024);             \
        attr = at_;                                                                                                                      \
        if (attr == hipSuccess)                                                                                                          \
            hipLaunchKernelGGL((tapgemm_wreg_f32_kernel<NCH_, 4, false, true>), dim3(gx, nyw, 1), dim3(512), lds, st, a, npw);           \
    } while (0)
#define SHM_WREG32_LAUNCH_NM(NCH_, MODE_)                                                                                                \
    do {                                                                                                                                 \
        static const hipError_t at_ = hipFuncSetAttribute((const void*)tapgemm_wreg_f32_kernel<NCH_, 4, false, false, MODE_>,            \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NCH_ * wreg32_nit(4) * 1024 + 8 * 1024);  \
        attr = at_;                                                                                                                      \
        if (attr == hipSuccess)                                                                                                          \
            hipLaunchKernelGGL((tapgemm_wreg_f32_kernel<NCH_, 4, false, false, MODE_>), dim3(gx, nyw, 1), dim3(512), lds, st, a, npw);   \
    } while (0)
            if (want_nm && nch == 4 && a.ntmode) SHM_WREG32_LAUNCH_NM(4, 2);
            else if (want_nm && nch == 4) SHM_WREG32_LAUNCH_NM(4, 1);
            else if (want_nm && nch == 2 && a.ntmode) SHM_WREG32_LAUNCH_NM(2, 2);
            else if (want_nm && nch == 2) SHM_WREG32_LAUNCH_NM(2, 1);
            else if (want_nm && a.ntmode) SHM_WREG32_LAUNCH_NM(1, 2);
            else if (want_nm) SHM_WREG32_LAUNCH_NM(1, 1);
            else if (gs_fused && nch == 4) SHM_WREG32_LAUNCH_GS(4);
            else if (gs_fused && nch == 2) SHM_WREG32_LAUNCH_GS(2);
            else if (gs_fused) SHM_WREG32_LAUNCH_GS(1);
            else if (wreg32_wn == 4 && nch == 4) SHM_WREG32_LAUNCH(4, 4);
            else if (wreg32_wn == 4 && nch == 2) SHM_WREG32_LAUNCH(2, 4);
            else if (wreg32_wn == 4) SHM_WREG32_LAUNCH(1, 4);
            else if (wreg32_wn == 2 && nch == 2) SHM_WREG32_LAUNCH(2, 2);
            else if (wreg32_wn == 2) SHM_WREG32_LAUNCH(1, 2);
            else if (nch == 2 && a.x2) SHM_WREG32_LAUNCH2(2, 1, true);
            else if (nch == 2) SHM_WREG32_LAUNCH(2, 1);
            else SHM_WREG32_LAUNCH(1, 1);
#undef SHM_WREG32_LAUNCH2
#undef SHM_WREG32_LAUNCH
#undef SHM_WREG32_LAUNCH_GS
#undef SHM_WREG32_LAUNCH_NM
            SHM_REQUIRE(attr == hipSuccess, SHM_E_HIP, "%s: cannot reserve %u bytes of LDS: %s", who, lds, hipGetErrorString(attr));
            shm_set_last_kernel(gs_fused ? "tapgemm_wreg_f32_kernel<%d, %d, %s, true>"
                                : want_nm ? (a.ntmode ? "tapgemm_wreg_f32_kernel<%d, %d, %s, false, 2>" : "tapgemm_wreg_f32_kernel<%d, %d, %s, false, 1>")
                                          : "tapgemm_wreg_f32_kernel<%d, %d, %s>", nch, wreg32_wn,
                                a.x2 ? "true" : "false");
        }
        break;
    }
    case SHM_TG_DMA_128x128:
        hipLaunchKernelGGL((tapgemm_dma_kernel<T, TO, 128, 128, 2, 2, 3, 16>), grid1d(128, 128), dim3(256), 0, st, a);
        shm_set_last_kernel("tapgemm_dma_kernel<%s, %s, 128, 128, 2, 2, 3, 16>", tn, ton);
        break;
    case SHM_TG_DMA_64x128:
        hipLaunchKernelGGL((tapgemm_dma_kernel<T, TO, 64, 128, 2, 2, 3, 16>), grid1d(64, 128), dim3(256), 0, st, a);
        shm_set_last_kernel("tapgemm_dma_kernel<%s, %s, 64, 128, 2, 2, 3, 16>", tn, ton);
        break;
    case SHM_TG_DMA_128x64:
        hipLaunchKernelGGL((tapgemm_dma_kernel<T, TO, 128, 64, 2, 2, 3, 16>), grid1d(128, 64), dim3(256), 0, st, a);
        shm_set_last_kernel("tapgemm_dma_kernel<%s, %s, 128, 64, 2, 2, 3, 16>", tn, ton);
        break;
    case SHM_TG_DMA_64x64:
        hipLaunchKernelGGL((tapgemm_dma_kernel<T, TO, 64, 64, 2, 2, 3, 16>), grid1d(64, 64), dim3(256), 0, st, a);
        shm_set_last_kernel("tapgemm_dma_kernel<%s, %s, 64, 64, 2, 2, 3, 16>", tn, ton);
        break;
    case SHM_TG_DMA_256x64:
        hipLaunchKernelGGL((tapgemm_dma_kernel<T, TO, 256, 64, 4, 1, 3, 16>), grid1d(256, 64), dim3(256), 0, st, a);
        shm_set_last_kernel("tapgemm_dma_kernel<%s, %s, 256, 64, 4, 1, 3, 16>", tn, ton);
        break;
    case SHM_TG_DMA_256x128:
        hipLaunchKernelGGL((tapgemm_dma_kernel<T, TO, 256, 128, 4, 2, 3, 16>), grid1d(256, 128), dim3(512), 0, st, a);
        shm_set_last_kernel("tapgemm_dma_kernel<%s, %s, 256, 128, 4, 2, 3, 16>", tn, ton);
        break;
    case SHM_TG_DMA_128x128_BK32:
        SHM_REQUIRE(bk32_ok, SHM_E_SHAPE, "%s: forced variant bk32 needs channel counts that are multiples of %d", who, 2 * BKE);
        hipLaunchKernelGGL((tapgemm_dma_kernel<T, TO, 128, 128, 2, 2, 2, 32>), grid1d(128, 128), dim3(256), 0, st, a);
        shm_set_last_kernel("tapgemm_dma_kernel<%s, %s, 128, 128, 2, 2, 2, 32>", tn, ton);
        break;
    case SHM_TG_DMA_128x128_NST4:
        hipLaunchKernelGGL((tapgemm_dma_kernel<T, TO, 128, 128, 2, 2, 4, 16>), grid1d(128, 128), dim3(256), 0, st, a);
        shm_set_last_kernel("tapgemm_dma_kernel<%s, %s, 128, 128, 2, 2, 4, 16>", tn, ton);
        break;
    default:
        SHM_REQUIRE(false, SHM_E_SHAPE, "%s: unknown tapgemm.variant %d", who, v);
    }
    return SHM_OK;
}

static int launch_tapgemm(TapGemmArgs& a, int batch, int nphase, int dtype, hipStream_t st, const char* who) {
    SHM_REQUIRE(dtype == SHM_F32 || dtype == SHM_BF16 || dtype == SHM_BF16_GF32, SHM_E_DTYPE,
                "%s: dtype %d not in {SHM_F32, SHM_BF16, SHM_BF16_GF32}", who, dtype);
    a.stats = g_conv_stats;
    a.hw = g_conv_hw;
    a.stats_slots = g_conv_slots;
    a.stats_stride = (unsigned)batch * (unsigned)a.nout * 2u;
    for (int p = 0; p < 2; ++p) {
        a.gaux[p] = g_gsum.aux[p];
        a.ldgaux[p] = g_gsum.ld[p];
        a.gred[p] = g_gsum.red[p];
    }
    a.gslots = SHM_GSUM_SLOTS;
    a.gbatch = batch;
    a.nt = g_norm.nt;
    a.ntpart = g_norm.part;
    a.ntc = g_norm.c;
    a.ntmode = g_norm.mode;
    a.ntbytes = (unsigned)((size_t)batch * SHM_NT_PLANES * g_norm.c * sizeof(float));
    a.wimg = 0;
    a.bias_img = 0;
    g_gsum_fused = false;
    if (a.gred[0] || a.gred[1]) {
        SHM_REQUIRE(a.stats == nullptr, SHM_E_SHAPE, "%s: fused forward statistics and gsum are exclusive", who);
        a.hw = a.hg * a.wg;           // pixels per sample in the M index space of one phase
    }
    SHM_REQUIRE(dtype != SHM_BF16_GF32 || a.stats == nullptr, SHM_E_DTYPE, "%s: SHM_BF16_GF32 has no fused statistics", who);
    const int esz = dtype == SHM_F32 ? 4 : 2, bke = 64 / esz, che = 16 / esz;
    SHM_REQUIRE(a.K % bke == 0 && a.K > 0, SHM_E_SHAPE, "%s: contraction channels %d must be a multiple of %d", who, a.K, bke);
    SHM_REQUIRE(a.c1 % bke == 0, SHM_E_SHAPE, "%s: concat split %d must be a multiple of %d", who, a.c1, bke);
    SHM_REQUIRE(a.ldx % che == 0 && (a.x2 == nullptr || a.ldx2 % che == 0), SHM_E_SHAPE, "%s: input pitch must be a multiple of 16 bytes", who);
    SHM_REQUIRE((size_t)batch * a.hi * a.wi < (1u << 31) && (size_t)batch * a.ho * a.wo < (1u << 31), SHM_E_SHAPE, "%s: pixel count overflows int32", who);
    a.M = batch * a.hg * a.wg;
    if (a.M == 0 || a.nout == 0) return SHM_OK;
    {
        const size_t lim = 0xfffffff0ull;
        size_t xb = (size_t)batch * a.hi * a.wi * a.ldx * esz, x2b = a.x2 ? (size_t)batch * a.hi * a.wi * a.ldx2 * esz : 0;
        size_t wb = 0;
        for (int p = 0; p < nphase; ++p)
            for (int t = 0; t < a.ph[p].ntaps; ++t) {
                size_t e = (size_t)(a.ph[p].widx[t] + 1) * a.nout * a.K * esz;
                if (e > wb) wb = e;
            }
        if (a.nt && a.ntmode) {                   // SHM_NORM_SCALED: one weight copy and one bias row per sample
            SHM_REQUIRE(nphase == 1 && a.bias, SHM_E_SHAPE, "%s: SHM_NORM_SCALED takes per-sample weights AND per-sample bias rows", who);
            a.wimg = (unsigned)((size_t)a.ph[0].ntaps * a.nout * a.K * esz);
            a.bias_img = a.nout;
            wb = (size_t)batch * a.wimg;
        }
        SHM_REQUIRE(xb < lim && x2b < lim && wb < lim, SHM_E_SHAPE, "%s: operand larger than 4 GiB (32-bit buffer offsets)", who);
        a.xbytes = (unsigned)xb;
        a.x2bytes = (unsigned)x2b;
        a.wbytes = (unsigned)wb;
        const int oesz = dtype == SHM_BF16 ? 2 : 4;
        const size_t yb = (size_t)batch * a.ho * a.wo * a.ldy * oesz, y2b = a.y2 ? (size_t)batch * a.ho * a.wo * a.ldy2 * oesz : 0;
        // ("tapgemm.flat_epilogue": the >= 4 GiB behaviour on demand -- keeps the 64-bit-address epilogues and the variant choice without
        // the buffer-store kernels under test at sizes a test can afford)
        const bool flat = shm_tune(SHM_TUNE_TAPGEMM_FLAT_EPILOGUE) != 0;
        a.ybytes = (yb < lim && !flat) ? (unsigned)yb : 0u;
        a.y2bytes = (y2b < lim && !flat) ? (unsigned)y2b : 0u;
    }
    // the 3-channel stride-2 first layer of the discriminator on the compact image layout (conv_rgb.hip); a forced tapgemm.variant keeps
    // the generic kernels (which read K channels per tap from the 16-byte pixels: the neighbours' values times the zero weight columns)
    if (nphase == 1 && a.is == 2 && a.os == 1 && a.ph[0].ntaps == 9 && a.ph[0].dh[0] == 0 && a.ph[0].dw[0] == 0 && !a.x2 && !a.y2 && !a.gred[0] && !a.gred[1] &&
        !a.nt && !g_norm.query && a.ldx * esz == 16 && a.K * esz == 64 && a.ybytes != 0 && dtype != SHM_BF16_GF32 &&
        shm_tune(SHM_TUNE_TAPGEMM_VARIANT) == SHM_TG_AUTO) {
        const int r = shm_rgb_s2_fwd_launch(a.x, a.ldx, a.w, a.K, a.bias, a.y, a.ldy, batch, a.hi, a.wi, a.nout, a.slope, a.stats, a.stats_slots, a.stats_stride,
                                            a.xbytes, a.ybytes, dtype, st);
        if (r < 0) return r;
        if (r == 1) return SHM_OK;
    }
    int rc;
    if (dtype == SHM_BF16)
        rc = launch_tapgemm_t<bf16_t, bf16_t>(a, batch, nphase, st, who);
    else if (dtype == SHM_BF16_GF32)
        rc = launch_tapgemm_t<bf16_t, float>(a, batch, nphase, st, who);
    else
        rc = launch_tapgemm_t<float, float>(a, batch, nphase, st, who);
    if (rc) return rc;
    if (g_norm.query) return SHM_OK;
    SHM_LAUNCH_CHECK(who);
    return SHM_OK;
}

// ------------------------------------------------------------------------------------
// [ntaps][rows][cols] -> [ntaps][cols][rows_pad]
template <typename T>
__global__ void transpose_taps_kernel(const float* __restrict__ w, T* __restrict__ wt, int rows, int cols, int rows_pad) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const float* src = w + (size_t)t * rows * cols;
    T* dst = wt + (size_t)t * cols * rows_pad;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < rows_pad) dst[(size_t)c * rows_pad + r] = (T)tile[threadIdx.x][i];
    }
}

extern "C" int shm_transpose_taps(const float* w, void* wt, int ntaps, int rows, int cols, int rows_pad, int dtype, void* stream) {
    SHM_REQUIRE(rows_pad >= rows && ntaps > 0 && rows > 0 && cols > 0, SHM_E_SHAPE, "shm_transpose_taps: bad shape");
    SHM_REQUIRE(dtype == SHM_F32 || dtype == SHM_BF16, SHM_E_DTYPE, "shm_transpose_taps: bad dtype %d", dtype);
    dim3 grid(shm_cdiv(cols, 32), shm_cdiv(rows_pad, 32), ntaps);
    if (dtype == SHM_BF16)
        hipLaunchKernelGGL(transpose_taps_kernel<bf16_t>, grid, dim3(32, 8), 0, (hipStream_t)stream, w, (bf16_t*)wt, rows, cols, rows_pad);
    else
        hipLaunchKernelGGL(transpose_taps_kernel<float>, grid, dim3(32, 8), 0, (hipStream_t)stream, w, (float*)wt, rows, cols, rows_pad);
    SHM_LAUNCH_CHECK("shm_transpose_taps");
    return SHM_OK;
}

// Every layer's transpose of one model in ONE launch (the per-layer launches are 4.5 us each, 27 per step, in front of the
// forward pass): block -> (layer, tap, 32 x 32 tile) through a prefix table in the kernel arguments.
constexpr int kMaxTransposes = 48;
struct TransposeBatch {
    const float* w[kMaxTransposes];
    void* wt[kMaxTransposes];
    int rows[kMaxTransposes], cols[kMaxTransposes], rows_pad[kMaxTransposes];
    int tx[kMaxTransposes], ty[kMaxTransposes];      // tiles along cols / rows_pad
    int block0[kMaxTransposes + 1];                  // first block of each layer
    int count;
};

template <typename T>
__global__ void transpose_taps_multi_kernel(const TransposeBatch b) {
    __shared__ float tile[32][33];
    int l = 0;
    while (l + 1 < b.count && (int)blockIdx.x >= b.block0[l + 1]) ++l;            // block-uniform scan (<= 48 entries)
    const int rows = b.rows[l], cols = b.cols[l], rows_pad = b.rows_pad[l];
    int rem = (int)blockIdx.x - b.block0[l];
    const int per_tap = b.tx[l] * b.ty[l];
    const int t = rem / per_tap;
    rem -= t * per_tap;
    const int r0 = (rem / b.tx[l]) * 32, c0 = (rem % b.tx[l]) * 32;
    const float* src = b.w[l] + (size_t)t * rows * cols;
    T* dst = (T*)b.wt[l] + (size_t)t * cols * rows_pad;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < rows_pad) dst[(size_t)c * rows_pad + r] = (T)tile[threadIdx.x][i];
    }
}

extern "C" int shm_transpose_taps_multi(int count, const void* const* w, void* const* wt, const int* ntaps, const int* rows, const int* cols,
                                        const int* rows_pad, int dtype, void* stream) {
    SHM_REQUIRE(count >= 0 && count <= kMaxTransposes, SHM_E_SHAPE, "shm_transpose_taps_multi: %d layers (at most %d)", count, kMaxTransposes);
    SHM_REQUIRE(dtype == SHM_F32 || dtype == SHM_BF16, SHM_E_DTYPE, "shm_transpose_taps_multi: bad dtype %d", dtype);
    if (count == 0) return SHM_OK;
    SHM_REQUIRE(w && wt && ntaps && rows && cols && rows_pad, SHM_E_SHAPE, "shm_transpose_taps_multi: null table");
    TransposeBatch b{};
    b.count = count;
    int total = 0;
    for (int l = 0; l < count; ++l) {
        SHM_REQUIRE(w[l] && wt[l] && rows_pad[l] >= rows[l] && ntaps[l] > 0 && rows[l] > 0 && cols[l] > 0, SHM_E_SHAPE,
                    "shm_transpose_taps_multi: bad shape of layer %d", l);
        b.w[l] = (const float*)w[l];
        b.wt[l] = wt[l];
        b.rows[l] = rows[l];
        b.cols[l] = cols[l];
        b.rows_pad[l] = rows_pad[l];
        b.tx[l] = shm_cdiv(cols[l], 32);
        b.ty[l] = shm_cdiv(rows_pad[l], 32);
        b.block0[l] = total;
        total += ntaps[l] * b.tx[l] * b.ty[l];
    }
    b.block0[count] = total;
    if (dtype == SHM_BF16)
        hipLaunchKernelGGL(transpose_taps_multi_kernel<bf16_t>, dim3(total), dim3(32, 8), 0, (hipStream_t)stream, b);
    else
        hipLaunchKernelGGL(transpose_taps_multi_kernel<float>, dim3(total), dim3(32, 8), 0, (hipStream_t)stream, b);
    SHM_LAUNCH_CHECK("shm_transpose_taps_multi");
    return SHM_OK;
}

// ------------------------------------------------------------------------------------
// SHM_NORM_SCALED operands of a convolution whose source part [part_lo, part_lo + c) is the un-normalised activation a of an
// InstanceNorm block with table nt (common.h): conv(w, (a - mean) * inv + beta) = conv(w * inv, a) + sum w * (beta - mean * inv) inside
// the image, so per sample n
//   wk_n[n][tap][co][k] = wk[tap][co][k] * inv_n[k - part_lo]  (k in the part; the other channels are copied)
//   bias_n[n][co]       = bias[co] + sum_{tap, k in part} wk[tap][co][k] * (beta[k'] - mean_n[k'] * inv_n[k'])
// (the kernels write `ring` over out-of-image taps, whose product with the scaled weight cancels that tap's share of bias_n).
// One block per (co, sample).
template <typename T>
__global__ __launch_bounds__(256) void norm_prepare_kernel(const T* __restrict__ wk, const float* __restrict__ bias, const float* __restrict__ nt, int c,
                                                           int part_lo, T* __restrict__ wk_n, float* __restrict__ bias_n, int ntaps, int cout, int K) {
    const int co = blockIdx.x, n = blockIdx.y;
    const float* mean = nt + (size_t)n * SHM_NT_PLANES * c;
    const float *inv = mean + c, *beta = mean + 2 * c;
    float acc = 0.f;
    for (int i = threadIdx.x; i < ntaps * K; i += 256) {
        const int tap = i / K, k = i - tap * K;
        const size_t src = ((size_t)tap * cout + co) * K + k;
        const float w = (float)wk[src];
        float o = w;
        const int kp = k - part_lo;
        if (kp >= 0 && kp < c) {
            o = w * inv[kp];
            acc += w * (beta[kp] - mean[kp] * inv[kp]);
        }
        wk_n[(size_t)n * ntaps * cout * K + src] = (T)o;
    }
    __shared__ float red[4];
    acc = shm_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bias_n[(size_t)n * cout + co] = (bias ? bias[co] : 0.f) + ((red[0] + red[1]) + (red[2] + red[3]));
}

extern "C" int shm_conv2d_norm_prepare(const void* wk, const float* bias, const float* nt, int c, int part_lo, void* wk_n, float* bias_n, int batch,
                                       int cin, int cout, int ksize, int dtype, void* stream) {
    SHM_REQUIRE(wk && nt && wk_n && bias_n, SHM_E_SHAPE, "shm_conv2d_norm_prepare: null pointer");
    SHM_REQUIRE(ksize == 1 || ksize == 3, SHM_E_SHAPE, "shm_conv2d_norm_prepare: ksize %d not in {1,3}", ksize);
    SHM_REQUIRE(c > 0 && part_lo >= 0 && part_lo + c <= cin, SHM_E_SHAPE, "shm_conv2d_norm_prepare: part [%d, %d) outside %d channels", part_lo, part_lo + c, cin);
    if (batch == 0 || cout == 0) return SHM_OK;
    SHM_DISPATCH(dtype, "shm_conv2d_norm_prepare",
                 hipLaunchKernelGGL(norm_prepare_kernel<T>, dim3(cout, batch), dim3(256), 0, (hipStream_t)stream, (const T*)wk, bias, nt, c, part_lo, (T*)wk_n,
                                    bias_n, ksize * ksize, cout, cin));
    SHM_LAUNCH_CHECK("shm_conv2d_norm_prepare");
    return SHM_OK;
}

// ------------------------------------------------------------------------------------
extern "C" int shm_conv2d_fwd(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* wk,
                              const float* bias, void* y, int ldy, int batch, int hi, int wi, int cin,
                              int cout, int ksize, int stride, float slope, int dtype, void* stream) {
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
    return launch_tapgemm(a, batch, 1, dtype, (hipStream_t)stream, "shm_conv2d_fwd");
}

int shm_in_finalize_internal(double* stats, double* part, int nslot, int total, int hw, double eps, float* nt, const float* beta, int c, hipStream_t st);

extern "C" int shm_conv2d_in_fwd(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* wk,
                                 const float* bias, void* y, int ldy, int batch, int hi, int wi, int cin,
                                 int cout, int ksize, int stride, float slope, double* stats, double* scratch,
                                 float eps, int dtype, void* stream) {
    return shm_conv2d_in_fwd_norm(x, x2, c1, ldx, ldx2, nullptr, nullptr, SHM_NORM_EXACT, wk, bias, y, ldy, batch, hi, wi, cin, cout, ksize, stride, slope, stats,
                                  scratch, eps, nullptr, nullptr, dtype, stream);
}

// Does the kernel that shm_conv2d_in_fwd_norm would run for this shape normalise its source in LDS?  A dry run of the launcher's
// variant choice (which depends on the shape, the batch and the tuning knobs): nothing is launched.
extern "C" int shm_conv2d_norm_supported(int batch, int hi, int wi, int cin, int c1, int cout, int ksize, int stride, int norm_part, int dtype) {
    if (dtype != SHM_F32 && dtype != SHM_BF16) return 0;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || batch <= 0 || cin <= 0 || cout <= 0) return 0;
    if (norm_part != 0 && norm_part != 1) return 0;
    const bool two = c1 > 0 && c1 < cin;
    if (norm_part == 1 && !two) return 0;
    // operands are never dereferenced in a dry run; any aligned non-null address serves
    static const __attribute__((aligned(256))) char dummy[256] = {};
    const void* px = dummy;
    g_norm.nt = (const float*)dummy;
    g_norm.part = norm_part;
    g_norm.c = two ? (norm_part ? cin - c1 : c1) : cin;
    g_norm.query = true;
    g_norm.query_ok = false;
    const int r = shm_conv2d_fwd(px, two ? px : nullptr, two ? c1 : 0, two ? c1 : cin, two ? cin - c1 : 0, px, nullptr, (void*)dummy, cout, batch, hi, wi, cin, cout,
                                 ksize, stride, 0.2f, dtype, nullptr);
    const bool ok = r == SHM_OK && g_norm.query_ok;
    g_norm = NormReq{};
    return ok ? 1 : 0;
}

extern "C" int shm_conv2d_in_fwd_norm(const void* x, const void* x2, int c1, int ldx, int ldx2, const float* nt_x, const float* nt_x2, int norm_mode,
                                      const void* wk, const float* bias, void* y, int ldy, int batch, int hi, int wi, int cin, int cout, int ksize,
                                      int stride, float slope, double* stats, double* scratch, float eps, float* nt_out, const float* beta_out, int dtype,
                                      void* stream) {
    SHM_REQUIRE(stats, SHM_E_SHAPE, "shm_conv2d_in_fwd: null stats");
    SHM_REQUIRE(norm_mode == SHM_NORM_EXACT || norm_mode == SHM_NORM_SCALED, SHM_E_SHAPE, "shm_conv2d_in_fwd_norm: norm_mode %d", norm_mode);
    SHM_REQUIRE(!(nt_x && nt_x2), SHM_E_SHAPE, "shm_conv2d_in_fwd_norm: at most one source can be normalised on the fly");
    SHM_REQUIRE(!nt_x2 || x2, SHM_E_SHAPE, "shm_conv2d_in_fwd_norm: nt_x2 without a second source");
    SHM_REQUIRE(!nt_out || beta_out, SHM_E_SHAPE, "shm_conv2d_in_fwd_norm: nt_out needs beta_out");
    int ho, wo, pt;
    shm_same_pad(hi, ksize, stride, &ho, &pt);
    shm_same_pad(wi, ksize, stride, &wo, &pt);
    const int hw = ho * wo;
    const bool norm_in = nt_x || nt_x2;
    struct NormScope {             // the request lives for the conv launch of this call only
        ~NormScope() { g_norm = NormReq{}; }
    } norm_scope;
    if (norm_in) {
        g_norm.nt = nt_x ? nt_x : nt_x2;
        g_norm.part = nt_x ? 0 : 1;
        g_norm.c = x2 ? (nt_x ? c1 : cin - c1) : cin;
        g_norm.mode = norm_mode;
    }
    if (!shm_tune(SHM_TUNE_STATS_FUSION) || hw % 64 != 0) {       // tiny maps: separate statistics pass
        int r = shm_conv2d_fwd(x, x2, c1, ldx, ldx2, wk, bias, y, ldy, batch, hi, wi, cin, cout, ksize, stride, slope, dtype, stream);
        g_norm = NormReq{};
        if (r) return r;
        r = shm_in_stats(y, ldy, stats, batch, hw, cout, eps, dtype, stream);
        if (r == SHM_OK && nt_out) r = shm_in_norm_table(stats, beta_out, nt_out, batch, cout, stream);
        return r;
    }
    // Every wave tile of a sample adds its column sums with f64 atomics: on one copy that is hw/64 atomics
    // per address, a serial chain worth ~100 us at 256x256 whatever the batch (measured, bf16 and fp32).
    // With `scratch` the chain is cut SHM_STATS_SLOTS-fold and the finalize kernel sums the copies.
    // `scratch` is zero on entry by contract and left zero (the finalize kernel clears what it sums); without it the
    // sums go to `stats`, which is zeroed here.
    double* acc = scratch ? scratch : stats;
    const int slots = scratch ? SHM_STATS_SLOTS : 1;
    int r = scratch ? SHM_OK : shm_zero(stats, (size_t)batch * cout * 2 * sizeof(double), stream);
    if (r) return r;
    g_conv_stats = acc;
    g_conv_slots = slots;
    g_conv_hw = hw;
    r = shm_conv2d_fwd(x, x2, c1, ldx, ldx2, wk, bias, y, ldy, batch, hi, wi, cin, cout, ksize, stride, slope, dtype, stream);
    g_conv_stats = nullptr;
    g_conv_slots = 1;
    g_norm = NormReq{};
    if (r == SHM_OK) r = shm_in_finalize_internal(stats, scratch, slots, batch * cout, hw, (double)eps, nt_out, beta_out, cout, (hipStream_t)stream);
    // "zero on entry, zero on return" also on the error path: a failed launch must not leave sums behind
    if (r != SHM_OK && scratch) (void)hipMemsetAsync(scratch, 0, (size_t)SHM_STATS_SLOTS * batch * cout * 2 * sizeof(double), (hipStream_t)stream);
    return r;
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

extern "C" int shm_conv2d_dgrad(const void* dy, int lddy, const void* w, void* dx, void* dx2, int n1,
                                int lddx, int lddx2, int batch, int hi, int wi, int cin, int cout,
                                int ksize, int stride, int dtype, void* stream) {
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
        return launch_tapgemm(a, batch, 1, dtype, (hipStream_t)stream, "shm_conv2d_dgrad");
    }
    SHM_REQUIRE(hi % 2 == 0 && wi % 2 == 0, SHM_E_SHAPE, "shm_conv2d_dgrad: stride 2 needs even input size");
    a.hg = hi / 2;
    a.wg = wi / 2;
    a.os = 2;
    fill_s2_phases(a, pt, pl);
    return launch_tapgemm(a, batch, 4, dtype, (hipStream_t)stream, "shm_conv2d_dgrad");
}

extern "C" int shm_conv2d_transpose_fwd(const void* x, int ldx, const void* w, const float* bias, void* y,
                                        int ldy, int batch, int hi, int wi, int cin, int cout, float slope,
                                        int dtype, void* stream) {
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
    return launch_tapgemm(a, batch, 4, dtype, (hipStream_t)stream, "shm_conv2d_transpose_fwd");
}

// Keras Conv2DTranspose(k=2, strides=2) (SpecSeg.py:63,69,75,81): non-overlapping, every output
// phase (ph, pw) is a 1x1 product with its own tap: y[2a+ph, 2b+pw] = bias + x[a, b] . w[ph][pw].
extern "C" int shm_conv2d_transpose2x2_fwd(const void* x, int ldx, const void* w, const float* bias, void* y,
                                           int ldy, int batch, int hi, int wi, int cin, int cout, float slope,
                                           int dtype, void* stream) {
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
    return launch_tapgemm(a, batch, 4, dtype, (hipStream_t)stream, "shm_conv2d_transpose2x2_fwd");
}

// ------------------------------------------------------------------------------------
// gsum entry points: an input-gradient product that also delivers, per (sample, channel) of its output, sum(g) and sum(g * aux)
// -- the two sums the InstanceNorm backward of the block whose OUTPUT gradient it writes would otherwise collect in a pass of its
// own over g and the stored activation (shm_in_bwd's reduce pass).  red = f64 [SHM_GSUM_SLOTS][batch][channels][2], zero on entry
// (slot copies cut the per-address atomic chains; shm_in_bwd_apply sums and clears them).  Kernels that cannot take the sums in
// their epilogue (launch_tapgemm_t lists which can) are followed by the stand-alone reduce pass: callers always get the sums.
int shm_gsum_reduce_internal(const void* g, int ldg, const void* aux, int ldaux, double* red, int batch, int hw, int c, int dtype, hipStream_t st);

extern "C" int shm_conv2d_dgrad_gsum(const void* dy, int lddy, const void* w, void* dx, void* dx2, int n1, int lddx, int lddx2, int batch, int hi,
                                     int wi, int cin, int cout, int ksize, int stride, const void* aux, int ldaux, double* red, const void* aux2,
                                     int ldaux2, double* red2, int dtype, void* stream) {
    const char* who = "shm_conv2d_dgrad_gsum";
    SHM_REQUIRE((aux != nullptr) == (red != nullptr) && (aux2 != nullptr) == (red2 != nullptr), SHM_E_SHAPE, "%s: aux and red come in pairs", who);
    SHM_REQUIRE(red || red2, SHM_E_SHAPE, "%s: no sums requested (use shm_conv2d_dgrad)", who);
    SHM_REQUIRE(!red2 || dx2, SHM_E_SHAPE, "%s: sums of the second part need dx2", who);
    g_gsum = GsumReq{{aux, aux2}, {ldaux, ldaux2}, {red, red2}};
    int r = shm_conv2d_dgrad(dy, lddy, w, dx, dx2, n1, lddx, lddx2, batch, hi, wi, cin, cout, ksize, stride, dtype, stream);
    const bool fused = g_gsum_fused;
    g_gsum = GsumReq{};
    g_gsum_fused = false;
    if (r == SHM_OK && !fused) {
        const int c0 = dx2 ? n1 : cin;
        if (red) r = shm_gsum_reduce_internal(dx, lddx, aux, ldaux, red, batch, hi * wi, c0, dtype, (hipStream_t)stream);
        if (r == SHM_OK && red2) r = shm_gsum_reduce_internal(dx2, lddx2, aux2, ldaux2, red2, batch, hi * wi, cin - n1, dtype, (hipStream_t)stream);
    }
    return r;
}

// The stride-2 forward form is the input gradient of Conv2DTranspose (model.py runs it with slope 1 and no bias); any forward
// product may ask for the sums of its output.
extern "C" int shm_conv2d_fwd_gsum(const void* x, const void* x2, int c1, int ldx, int ldx2, const void* wk, const float* bias, void* y, int ldy,
                                   int batch, int hi, int wi, int cin, int cout, int ksize, int stride, float slope, const void* aux, int ldaux,
                                   double* red, int dtype, void* stream) {
    const char* who = "shm_conv2d_fwd_gsum";
    SHM_REQUIRE(aux && red, SHM_E_SHAPE, "%s: null aux / red", who);
    g_gsum = GsumReq{{aux, nullptr}, {ldaux, 0}, {red, nullptr}};
    int r = shm_conv2d_fwd(x, x2, c1, ldx, ldx2, wk, bias, y, ldy, batch, hi, wi, cin, cout, ksize, stride, slope, dtype, stream);
    const bool fused = g_gsum_fused;
    g_gsum = GsumReq{};
    g_gsum_fused = false;
    if (r == SHM_OK && !fused) {
        int ho, wo, pt;
        shm_same_pad(hi, ksize, stride, &ho, &pt);
        shm_same_pad(wi, ksize, stride, &wo, &pt);
        r = shm_gsum_reduce_internal(y, ldy, aux, ldaux, red, batch, ho * wo, cout, dtype, (hipStream_t)stream);
    }
    return r;
}
