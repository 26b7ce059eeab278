// The north-star block as a PING-PONG kernel (round 5): conv3x3 64 -> 64, unit stride, bf16, + bias + LeakyReLU + InstanceNorm statistics
// (ShmGANwithSSpecSeg.py:244-245), and its input gradient.
//
// Why another kernel.  tapgemm_wreg16_bf16_kernel (conv_wreg16.hip) runs two eight-wave blocks per CU whose waves move in lockstep through
// [halo DMA issue | MFMA loop | epilogue | staging barrier | stores]; its timing ablations are ADDITIVE (LABNOTES 10.2: remove the MFMAs, the
// LDS reads, the DMA or the stores and it loses 50-75 us each, any two ~150 of 240), and round 5's cut of its epilogue from ~215 to ~150
// non-MFMA vector instructions per patch and wave left the forward launch where it was (225 -> 216-225 us, same box): the phases do not overlap,
// the matrix pipe idles whenever both resident blocks are outside their MFMA loops.  Here the overlap is built in:
//
//   * ONE eight-wave block per CU (two waves per SIMD, 256 registers each), split into two GROUPS of four waves.  A group owns a stream of
//     8 x 32-pixel patches (all 64 output channels: 2 (rows 0-3 / 4-7) x 2 (channels 0-31 / 32-63) waves) and alternates between an X segment
//     -- the patch's 288 v_mfma_f32_16x16x32_bf16 per wave (2 channel tiles x 2 pixel tiles x 4 rows of 16 x 16 outputs), operands: weights in 144
//     registers, activation fragments from the group's halo image in LDS -- and a Y segment: halo DMA of its NEXT patch, epilogue of the patch just computed (bias, LeakyReLU, bf16, LDS staging),
//     whole-line stores.  The two groups run half a period apart, separated by workgroup barriers (two per segment: the Y segment's staging
//     barrier falls behind step 48 of the other group's 72 X steps): on every SIMD one wave is in X while its partner is in Y, by construction.
//   * A fragment read (1 KiB per wave: 16 pixels x 32 channels) feeds up to six MFMAs of 16 cycles (two channel tiles x the halo-row walk: halo row
//     R, column shift cs serves patch rows R - kh).  The first version ran v_mfma_f32_32x32x16_bf16 (half the instructions, same registers); the
//     launch is clock-bound and the chip holds a lower clock on that shape (bare loops, same FLOPs, sustained: 121 us at 1.33 GHz against 109 us
//     at 1.67 GHz), item 6 below.
//   * Halo image per group: [10 rows][34 pixels][128 bytes], single-buffered (it is refilled in the group's Y segment, after the barrier that ends
//     its X segment), filled by LDS-DMA: wave w4 of the group fetches the 8-pixel column segment w4 of every halo row (ten 1 KiB items whose
//     per-lane source offset is ONE lane constant plus a uniform term) and the two-pixel tails of rows w4, w4 + 4, w4 + 8; 16-byte chunk c of
//     halo column hc sits at position c ^ (hc & 6): conflict-free for the 16x16x32 fragment read at every column shift, and the DMA write is linear.
//     (Round 6: the first swizzle, c ^ ((hc >> 1) & 7), was laid out for the 32x32x16 shape's 32-pixel fragment; under ds_read_b128's lane groups --
//     {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... : sixteen pixels, the middle eight with the NEXT k chunk -- it was 2-way conflicted at column
//     shifts 1 and 2: 6.67 LDS cycles per read instead of 4, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.40 in profiles/r05_*_sq_pmc.json.
//     tools/probes/pp_lds_conflicts.py models every LDS access of this kernel against the guide's lane groups.)  Out-of-image rows fall outside a per-image buffer descriptor (zeros), out-of-image columns get an out-of-range offset.
//   * Epilogue as in conv_wreg16.hip: a lane's accumulator quads are four consecutive channels of one pixel -> 8-byte pieces into a
//     [256 pixels][128 bytes] staging image (chunk XOR pixel & 7), barrier, 1 KiB stores of whole lines.  InstanceNorm sums on the matrix pipe
//     from the staging image (transposed reads: mfma(F, F) diagonal = sum y^2, mfma(F, ones) = sum y; see conv_wreg16.hip), of the group's
//     PREVIOUS patch, interleaved into the first X steps of the next one.
//
// What the phase stamps (tools/probes/pp_stamps.py; cycles per patch and wave, n = 40 at 256 x 256) taught on the way, in the order found:
//   1. hipcc sinks a one-step fragment look-ahead back to its use (ds_read, lgkmcnt(0), MFMA): an X segment took 7 000 cycles for 4 608 of
//      MFMA.  Reads are issued two / three steps ahead behind sched_barriers, and halo rows are walked 0 / 2 interleaved, 1 | 5 / 3
//      interleaved, 4, so that a one-MFMA fragment is followed by a three-MFMA one.
//   2. hipcc waits vmcnt(0) in front of every LDS access that follows an LDS-DMA in program order (it cannot tell the addresses apart): the
//      epilogue's first staging write waited 3 200 cycles for the whole halo to land.  The Y segment's LDS accesses are inline asm with
//      waits placed by hand, the end-of-Y wait is the BUILTIN s_waitcnt (visible to hipcc's tracker, which otherwise drains the stores in front
//      of the next fragment read), and the Y segment sits under ONE branch (two made the tracker see a path "DMA issued, stores skipped").
//   3. A wave's sixteen statistics MFMAs took 2 400 cycles in the Y segment (no free slot on a pipe the X wave keeps busy) and 1 100 as a
//      block in front of the X steps; interleaved into them they cost ~500.
//   4. v_pk_add_f32 / v_pk_mul_f32 (formed by the SLP vectoriser) beside the partner's MFMAs: the epilogue took 3 300 cycles, 1 750 with
//      scalar instructions (-fno-slp-vectorize for this file).  s_setprio for either segment changes nothing.
//   5. With the fragment address formed in the loop (v_mov, v_xad, ds_read behind each other) the chain of a one-MFMA step does not fit under
//      its MFMA: +14 cycles per read.  The twelve addresses are registers for the length of an X half (re-formed per half: kept across the Y
//      segment they spill).  A second, statistics-free copy of the first X half in the loop made hipcc shuffle 84 register pairs per patch
//      to reconcile two allocations: the statistics are a template parameter and the first patch runs them on garbage and drops the result.
//   6. v_mfma_f32_32x32x16_bf16 -> 16x16x32: 13 570 -> 14 050 cycles per patch pair (the Y wave gets fewer issue slots: epilogue 1 800 -> 2 170), the
//      sustained in-kernel clock 1.40-1.46 -> 1.56 GHz: forward + statistics 217-223 -> 214-218 us, input gradient 180 -> 171-176 us, n = 8 -3..-5 %.
//   State: X 4 540 + 1 910 cycles (ideal 3 330 + 1 540), Y 1 360 (DMA issue) + 2 170 (epilogue) + 950 (stores), period 14 050 cycles at
//   1.56 GHz sustained; the launch 214-218 us against tapgemm_wreg16_bf16_kernel's 229-235 (forward + statistics) and 171-176 against 186-190
//   (input gradient), same box, sustained; n = 8: 52.5 against 52.7-57.9 / 39.7 against 41.6-42.0.  The ~12 us prologue (36 weight loads per
//   lane, first halo) is 6 % at n = 40.
//
// LDS: 2 x 43 KiB halo + 2 x 32 KiB staging + bias = 150.25 KiB.  Eligible: K = 64, one source tensor, map height % 8 == 0 and width % 32 == 0,
// Cout % 64 == 0 with 64-channel blocks inside one output part, outputs below 4 GiB, an image below 2 GiB.
#include "tapgemm.h"

#include <stdlib.h>

namespace {
constexpr int PP_PH = 8, PP_PW = 32, PP_HR = PP_PH + 2, PP_HC = PP_PW + 2;
constexpr int PP_ITEMS = (PP_HR * PP_HC + 7) / 8;                     // 43 DMA items of 8 pixels x 128 bytes
constexpr int PP_HALO = PP_ITEMS * 1024;                               // bytes per group
constexpr int PP_STG = PP_PH * PP_PW * 128;                            // 32 KiB per group
constexpr unsigned PP_LDS = 2u * PP_HALO + 2u * PP_STG + 256u;
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// the X segment's walk over (halo row rr, column shift cs, 32-channel step k32, 16-pixel tile pt): rows 0 / 2 interleaved, row 1 | rows 5 / 3
// interleaved, row 4; within a row (cs, k32, pt) with pt fastest
struct PPStep {
    int rr, cs, k32, pt;
};
constexpr PPStep pp_step(int t) {
    const int half = t / 36, u = t % 36;
    const int idx = u < 24 ? u >> 1 : u - 24;
    const int rr = half == 0 ? (u < 24 ? ((u & 1) ? 2 : 0) : 1) : (u < 24 ? ((u & 1) ? 3 : 5) : 4);
    return PPStep{rr, idx >> 2, (idx >> 1) & 1, idx & 1};
}
// is step t the first one that accumulates into the tiles of (patch row m, pixel tile pt) (their MFMAs then take C = 0)?
constexpr bool pp_touches(int m, int pt, int t) { return pp_step(t).pt == pt && pp_step(t).rr - m >= 0 && pp_step(t).rr - m <= 2; }
constexpr bool pp_first(int m, int pt, int t) {
    if (!pp_touches(m, pt, t)) return false;
    for (int u = 0; u < t; ++u)
        if (pp_touches(m, pt, u)) return false;
    return true;
}
template <int V>
struct PPInt {
    static constexpr int value = V;
};
// compile-time loop: f(PPInt<T0>{}), ..., f(PPInt<T1 - 1>{}).  The X steps index register arrays (accumulators, weights) by functions of the step
// number: with an unrolled run-time loop hipcc did not always fold them (the 16x16x32 form went to scratch with waterfall loops), here every
// index is a constant expression by construction
template <int T0, int T1, class F>
__device__ __forceinline__ void pp_static_for(F&& f) {
    if constexpr (T0 < T1) {
        f(PPInt<T0>{});
        pp_static_for<T0 + 1, T1>(f);
    }
}
}  // namespace

// MODE 0: plain product (input-gradient launches: no bias, slope 1, no statistics); 1: bias + LeakyReLU; 2: bias + LeakyReLU + InstanceNorm statistics
template <int MODE, int XSPLIT>
__global__ __launch_bounds__(512, 2) void tapgemm_pp_bf16_kernel(const TapGemmArgs a, const int npatch, const int dbg, unsigned long long* __restrict__ clk) {
    constexpr bool EPI = MODE != 0, STATS = MODE == 2;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    [[maybe_unused]] unsigned long long tk0 = 0, rk0 = 0, tl0 = 0, rl0 = 0;
    if constexpr (abl::stamp) {
        tk0 = __builtin_amdgcn_s_memtime();
        rk0 = __builtin_amdgcn_s_memrealtime();
    }
    const TapPhase& P = a.ph[0];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.y * 64;
    char* const halo = smem + G * PP_HALO;
    char* const stg = smem + 2 * PP_HALO + G * PP_STG;
    float* const sbias = (float*)(smem + 2 * PP_HALO + 2 * PP_STG);

    // patch ranges: group gi of 2 * gridDim.x walks [q0, q1); blocks that share an XCD (blockIdx.x % 8 equal) get neighbouring ranges
    const int nb = gridDim.x;
    const int bx = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int per = (npatch + 2 * nb - 1) / (2 * nb);
    const int q0 = min(npatch, (2 * bx + G) * per), q1 = min(npatch, q0 + per);
    const int ppr = a.wi / PP_PW, ppi = (a.hi / PP_PH) * ppr;

    // ---- weights -> registers: A operand of v_mfma_f32_16x16x32_bf16, lane (l15, lq) holds row n0 + 32 wn + 16 ct + l15, k = 32 k32 + 8 lq ... + 7.
    // (Round 5: the first version of this kernel ran v_mfma_f32_32x32x16_bf16 -- same registers, half the instructions.  The launch is clock-bound
    // (LABNOTES 11.3) and the chip holds a lower clock on that shape: the bare loops of bench.py's ceiling probe, same FLOPs, sustained: 121 us at
    // 1.33 GHz against 109 us at 1.67 GHz for 16x16x32.)
    const bool flip = P.dh[0] > 0;                       // input-gradient launch: taps arrive as (1 - kh, 1 - kw)
    f32x4 bw[9][2][2];                                   // [tap][k32][channel tile]
    {
        const bf16_t* wp = (const bf16_t*)a.w;
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int wsl = flip ? P.widx[8 - u] : P.widx[u];
#pragma unroll
            for (int k32 = 0; k32 < 2; ++k32)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    bw[u][k32][ct] = *(const f32x4*)(wp + ((size_t)wsl * a.nout + n0 + 32 * wn + 16 * ct + l15) * a.K + k32 * 32 + lq * 8);
        }
    }
    if (tid < 64) sbias[tid] = a.bias ? a.bias[n0 + tid] : 0.f;

    // ---- halo DMA of patch q into the group's image (header comment).  Full item (row j, segment w4): LDS halo + j * 4352 + w4 * 1024; lane =
    // (pixel p8 = lane >> 3 of the segment, position lane & 7): halo column hc = 8 w4 + p8, source chunk position ^ (hc & 6) = position ^ (p8 & 6)
    const unsigned pixb = (unsigned)a.ldx * 2u;
    const unsigned imgb = (unsigned)(a.hi * a.wi) * pixb;
    auto dma = [&](int q) {
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PP_PH, x0 = (prem % ppr) * PP_PW;
        // rows above / below the image lie outside this descriptor (the offset wraps below zero or passes imgb): zeros
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((char*)a.x + (size_t)img * imgb, 0, imgb, 0x00020000);
        int ln = lane;
        asm volatile("" : "+v"(ln));                     // the lane constants are re-formed per patch, not kept across the MFMA segment
        const int p8 = ln >> 3, pos = ln & 7;
        const unsigned base = (unsigned)((y0 - 1) * a.wi + x0 - 1) * pixb;           // halo (0, 0); wraps below zero on the first row / column
        const unsigned lc = base + (unsigned)(8 * w4 + p8) * pixb + (unsigned)((pos ^ (p8 & 6)) << 4);
        // column -1 (segment 0, p8 = 0) of a patch on the left edge: the previous row's last pixel, not padding -> out of range by hand
        const bool lcut = w4 == 0 && x0 == 0;            // wave-uniform
        const unsigned rowb = (unsigned)a.wi * pixb;
#pragma unroll
        for (int j = 0; j < PP_HR; ++j) {
            unsigned off = lc + (unsigned)j * rowb;
            if (lcut) off = p8 == 0 ? 0xffffffffu : off;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(halo + j * (PP_HC * 128) + w4 * 1024), 16, (int)off, 0, 0, 0);
        }
        // tails: halo columns 32, 33 of rows w4, w4 + 4, w4 + 8 (sixteen lanes; chunk = position: (32 + p8) & 6 = 0)
        if (ln < 16) {
            const bool rcut = x0 + PP_PW == a.wi;        // column wi: the next row's first pixel
            const unsigned lt = base + (unsigned)(32 + p8) * pixb + (unsigned)(pos << 4);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int hr = w4 + 4 * j;               // wave-uniform
                if (j < 2 || hr < PP_HR) {
                    const unsigned off = (rcut && p8 == 1) ? 0xffffffffu : lt + (unsigned)hr * rowb;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(halo + hr * (PP_HC * 128) + 4096), 16, (int)off, 0, 0, 0);
                }
            }
        }
    };

    // ---- fragment addresses (B operand: lane (l15, lq) holds pixel 16 pt + l15 + cs of halo row 4 wm + rr, channels 32 k32 + 8 lq ... + 7 = chunk
    // 4 k32 + lq): position chunk ^ ((l15 + cs) & 6) -- the tile's 16 pt leaves the swizzle alone; rr and pt are immediates, k32 an XOR of bit 6
    int fa[3];
#pragma unroll
    for (int cs = 0; cs < 3; ++cs) fa[cs] = G * PP_HALO + ((4 * wm * PP_HC) + l15 + cs) * 128 + ((lq ^ ((l15 + cs) & 6)) << 4);

    const bool part0 = n0 < a.n1;                        // block-uniform: the 64 channels lie in one output part (launcher)
    const __amdgpu_buffer_rsrc_t rsy = part0 ? __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc(a.y2, 0, a.y2bytes, 0x00020000);
    const unsigned ldyb = (unsigned)(part0 ? a.ldy : a.ldy2) * 2u;
    const unsigned ycol0 = (unsigned)(part0 ? n0 : n0 - a.n1) * 2u;

    // InstanceNorm statistics on the matrix pipe: wave w4 owns channel group cg = w4 of the group's patches, both sums.  The two 16 x 16 tiles
    // live for one patch only; their diagonal (channel l15: register l15 & 3 of the lanes with (l15 >> 2) == lq) is folded into two
    // per-lane floats that persist over the group's patches of an image (eight VALU per patch instead of eight registers across the X segment)
    float s1acc = 0.f, s2acc = 0.f;
    int simg = q0 / ppi;
    auto flush = [&](int img) {
        const int l15 = lane & 15, lq = lane >> 4;
        if ((l15 >> 2) == lq) {
            double* sp = a.stats + (size_t)((2 * blockIdx.x + G) % a.stats_slots) * a.stats_stride + ((size_t)img * a.nout + n0 + 16 * w4 + l15) * 2;
            atomicAdd(sp, (double)s1acc);
            atomicAdd(sp + 1, (double)s2acc);
        }
        s1acc = 0.f;
        s2acc = 0.f;
    };

    // timing-only build (abl::stamp, tools/probes/pp_stamps.py): cycles between the boundaries below, summed over the group's patches, per wave
    [[maybe_unused]] unsigned long long tph[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int k) {
        if constexpr (abl::stamp) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) tph[k] += t - tlast;
            tlast = t;
        }
    };
    f32x4 acc[4][2][2];                                  // [patch row m][pixel tile pt][channel tile ct]: four channels (4 lq + i) of pixel l15
    // X segment, steps [t0, t1) of the walk pp_step (72 steps; the first 36 are halo rows 0 - 2): the fragment of (halo row rr, column shift cs,
    // k step ks) feeds patch rows m = rr - kh, kh = 0 .. 2.  Fragments are read two steps ahead; the sched_barriers pin "issue the read, then
    // this step's MFMAs" (header comment)
    auto xsteps = [&](auto T0, auto T1, auto DEPTH, auto&& pre, auto&& post) {
        constexpr int t0 = decltype(T0)::value, t1 = decltype(T1)::value, depth = decltype(DEPTH)::value;       // fragments in flight: 2 or 3
        // the twelve (cs, ks) fragment addresses live in registers for the length of this call only: re-formed from fa[] here (the empty asm
        // stops hipcc from hoisting them out of the patch loop, where they would be twelve more registers across the Y segment: spills), so that
        // a step's read is ONE instruction -- with the XOR in the loop (v_mov, v_xad, ds_read behind each other) the address chain of a
        // one-MFMA step did not fit under its MFMA (stamps: +14 cycles per read)
        int fl[3] = {fa[0], fa[1], fa[2]};
        asm volatile("" : "+v"(fl[0]), "+v"(fl[1]), "+v"(fl[2]));
        auto frag = [&](const int t) {               // t is a constant at every call site (always inlined)
            const PPStep st = pp_step(t < t1 ? t : t1 - 1);
            if constexpr (abl::nolds) return f32x4{1.f, 2.f, 3.f, 4.f};         // timing only: no fragment reads
            return *(const f32x4*)(smem + ((fl[st.cs] ^ (st.k32 << 6)) + st.rr * (PP_HC * 128) + st.pt * 2048));
        };
        f32x4 f0 = frag(t0), f1 = frag(t0 + 1), f2 = f1;
        if constexpr (depth == 3) f2 = frag(t0 + 2);
        pp_static_for<t0, t1>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            constexpr PPStep st = pp_step(t);
            const f32x4 fn = frag(t + depth);
            pre(t);
            __builtin_amdgcn_sched_barrier(0);
            pp_static_for<0, 4>([&](auto MM) {
                constexpr int m = decltype(MM)::value, kh = st.rr - m;
                if constexpr (kh >= 0 && kh <= 2) {
                    constexpr bool first = pp_first(m, st.pt, t);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const f32x4 c0 = first ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[m][st.pt][ct];
                        acc[m][st.pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[kh * 3 + st.cs][st.k32][ct]), __builtin_bit_cast(bf16x8, f0), c0, 0, 0, 0);
                    }
                }
            });
            post(t);
            __builtin_amdgcn_sched_barrier(0);
            f0 = f1;
            if constexpr (depth == 3) {
                f1 = f2;
                f2 = fn;
            } else {
                f1 = fn;
            }
        });
    };
    auto nohook = [](int) {};

    // hipcc cannot tell an LDS access from the destination of an LDS-DMA in flight: in front of every ds_read / ds_write that follows the halo DMA in
    // program order it waits vmcnt(0) -- for the whole halo to land (the round-5 stamps: 3 200 cycles in front of the epilogue's first staging
    // write, 4 500 with the DMA issue, against 2 600 for half an X segment).  The Y segment's LDS accesses behind the DMA issue (staging writes,
    // staging reads of the store pass) are therefore inline asm with waits placed by hand; values read that way are threaded through the
    // waiting asm ("+v"), so that no use is scheduled in front of it.  The bias vector is read BEFORE the DMA issue, as ordinary loads.
    const int stg_a = 2 * PP_HALO + G * PP_STG;           // LDS byte address of the group's staging image (smem starts at 0: the only LDS object)
    // Y segment, first half: next halo, epilogue of patch q into the staging image
    auto yhead = [&](const int q) {
        f32x4 b4[2];                                     // the lane's four channels 4 lq + i of each channel tile
        if constexpr (EPI) {
            int lb = lane;
            asm volatile("" : "+v"(lb));
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) b4[ct] = *(const f32x4*)(sbias + 32 * wn + 16 * ct + 4 * (lb >> 4));
        }
        if (q + 1 < q1 && !(abl::stamp && (dbg & 4))) dma(q + 1);          // dbg: timing-only ablations of the stamped build (4: no halo DMA after the first)
        stamp(4);                                // [4] bias reads + halo DMA issue
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int e15 = ln & 15, eq = ln >> 4;
        // staging: pixel p = (4 wm + m) * 32 + 16 pt + e15, 16-byte chunk 4 wn + 2 ct + (eq >> 1) of its 128-byte row XOR p & 7 (= e15 & 7), half eq & 1
        const int st_w = stg_a + (4 * wm * 32 + e15) * 128 + (((4 * wn + (eq >> 1)) ^ (e15 & 7)) << 4) + ((eq & 1) << 3);      // + m * 4096 + pt * 2048, ct: ^ (2 << 4)
        if (abl::stamp && (dbg & 8)) return;     // 8: no epilogue
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                u32x2_t pk[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x2 lo = {acc[m][pt][ct][0], acc[m][pt][ct][1]}, hi = {acc[m][pt][ct][2], acc[m][pt][ct][3]};
                    if constexpr (EPI) {         // bias, LeakyReLU with 0 <= slope <= 1 (launcher): max(u, u * slope).  Scalar instructions on purpose (this
                        // file is built with -fno-slp-vectorize): beside the partner wave's MFMAs a v_pk_add_f32 / v_pk_mul_f32 costs four times a v_add_f32
                        lo = f32x2{shm_lrelu_max(lo.x + b4[ct][0], a.slope), shm_lrelu_max(lo.y + b4[ct][1], a.slope)};
                        hi = f32x2{shm_lrelu_max(hi.x + b4[ct][2], a.slope), shm_lrelu_max(hi.y + b4[ct][3], a.slope)};
                    }
                    pk[m] = u32x2_t{__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_t)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_t))};
                }
                const int aw = (st_w ^ (ct << 5)) + pt * 2048;
                // patch rows m, m + 1 are 4096 bytes = 8 x (64 x 8 bytes) apart
                asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:0 offset1:8" ::"v"(aw), "v"(pk[0]), "v"(pk[1]) : "memory");
                asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:16 offset1:24" ::"v"(aw), "v"(pk[2]), "v"(pk[3]) : "memory");
            }
    };
    // Y segment, second half: whole-line stores of the group's patch
    auto ytail = [&](const int q) {
        if (abl::stamp && (dbg & 16)) return;    // 16: no store pass
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PP_PH, x0 = (prem % ppr) * PP_PW;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int p8 = ln >> 3, c = ln & 7;
        // store instruction i of wave w4 covers patch pixels 64 w4 + 8 i + p8 = row 2 w4 + (i >> 2), column 8 (i & 3) + p8
        const int st_r = stg_a + (64 * w4 + p8) * 128 + ((c ^ p8) << 4);                        // + i * 1024
        const unsigned yo = (unsigned)((img * a.hi + y0 + 2 * w4) * a.wi + x0 + p8) * ldyb + ycol0 + (unsigned)c * 16u;
        u32x4 v[8];
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                     "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(st_r)
                     : "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // reads return in order: instruction i needs all but the 7 - i youngest
            if (i == 0) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            if (i == 1) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            if (i == 2) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            if (i == 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            if (i == 4) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            if (i == 5) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            if (i == 6) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(v[6]), "+v"(v[7]));
            if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[7]));
            __builtin_amdgcn_raw_buffer_store_b128(v[i], rsy, yo + (unsigned)((i >> 2) * a.wi + 8 * (i & 3)) * ldyb, 0, 2);       // nt: read next by another kernel
        }
    };
    // InstanceNorm sums of the patch whose staging image is complete and not yet overwritten: the group's PREVIOUS patch, inside the first half of
    // an X segment (the Y wave has no free slot on the matrix pipe: with the X wave ahead in the arbitration its sixteen MFMAs took 2 400
    // cycles; as a block in front of the X steps they cost the X wave 1 100).  Interleaved: the two transposed reads of fragment f at step 2 f,
    // its two MFMAs (mfma(F, F): diagonal = sum y^2; mfma(F, ones) = sum y) behind the main MFMAs of step 2 f + 2, the diagonals folded into the
    // per-lane sums at step 17.  X1 has room for it: patch row 3's accumulator is not live before step 36.
    const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
    auto tr_base = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int pix = ln >> 2, tp = ln & 3;             // transposed read: lane 4 q + p of group g supplies pixel 4 g + q, 8-byte piece p of the channel group
        return (const unsigned short*)(stg + pix * 128 + (((2 * w4 + (tp >> 1)) ^ (pix & 7)) << 4) + ((tp & 1) << 3));
    };
    auto tr_frag = [&](const unsigned short* tb, int f) {
        const s16x4_t f0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tb + f * 2048));            // pixels 32 f + 4 g + q
        const s16x4_t f1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tb + f * 2048 + 1024));     // + 16
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(f0, f1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto fold = [&](const f32x4& S1, const f32x4& S2, const bool keep) {       // keep = false (uniform): the sums are discarded by a SELECT (they may be NaN)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int i = ln & 3;            // = l15 & 3
        const float v1 = i == 0 ? S1[0] : i == 1 ? S1[1] : i == 2 ? S1[2] : S1[3];
        const float v2 = i == 0 ? S2[0] : i == 1 ? S2[1] : i == 2 ? S2[2] : S2[3];
        s1acc += keep ? v1 : 0.f;
        s2acc += keep ? v2 : 0.f;
    };
    auto next_image = [&](const int img) {
        if (img != simg) {
            flush(simg);
            simg = img;
        }
    };
    auto xstats_tail = [&](const int img) {      // the group's last patch, after the loop
        next_image(img);
        const unsigned short* tb = tr_base();
        f32x4 S1 = {0.f, 0.f, 0.f, 0.f}, S2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const bf16x8 F = tr_frag(tb, f);
            S2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F, F, S2, 0, 0, 0);
            S1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F, ones, S1, 0, 0, 0);
        }
        fold(S1, S2, true);
    };

    // ---- prologue: first halo of both groups; group 1 then runs one segment (two barriers) behind group 0
    if (q0 < q1) dma(q0);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), as a builtin: visible to hipcc's wait tracker (see the end-of-Y wait)
    SHM_LDS_BARRIER();
    if (G == 1) {
        SHM_LDS_BARRIER();
        SHM_LDS_BARRIER();
    }
    if constexpr (abl::stamp) {
        tl0 = __builtin_amdgcn_s_memtime();
        rl0 = __builtin_amdgcn_s_memrealtime();
    }
    // clock probe (shm_set_clock_probe, bench.py's north-star ceiling): shader-clock and 100 MHz counters around the patch loop of one wave
    unsigned long long ck0 = 0, cr0 = 0;
    if (clk) {
        ck0 = __builtin_amdgcn_s_memtime();
        cr0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int i = 0; i < per; ++i) {
        const int q = q0 + i;
        const bool act = q < q1;                 // group-uniform
        // ===== X segment (the other group is in its Y segment)
        stamp(-1);
        if (act) {
            if (abl::stamp && (dbg & 3) == 1) __builtin_amdgcn_s_setprio(1);      // wave priority: no effect either way (round-5 stamps), left to the stamped build
            if constexpr (STATS) {
                // ONE form of the first half per instantiation (with a second, statistics-free copy of the 36 steps in the loop hipcc shuffled 84
                // register pairs per patch to reconcile the two allocations): the first patch of a group runs the statistics instructions on
                // whatever the staging image holds and drops the result
                const bool keep = i > 0;
                if (keep) next_image((q - 1) / ppi);
                const unsigned short* tb = tr_base();
                bf16x8 Fa = ones, Fb = ones;             // fragment in use / fragment in flight
                f32x4 S1 = {0.f, 0.f, 0.f, 0.f}, S2 = {0.f, 0.f, 0.f, 0.f};
                xsteps(PPInt<0>{}, PPInt<XSPLIT>{}, PPInt<2>{},
                       [&](int t) {
                           if (t <= 16 && (t & 1) == 0) {
                               Fa = Fb;
                               if (t < 16) Fb = tr_frag(tb, t >> 1);
                           }
                       },
                       [&](int t) {
                           if (t >= 2 && t <= 16 && (t & 1) == 0) {
                               S2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Fa, Fa, S2, 0, 0, 0);
                               S1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Fa, ones, S1, 0, 0, 0);
                           }
                           if (t == 17) fold(S1, S2, keep);
                       });
            } else {
                xsteps(PPInt<0>{}, PPInt<XSPLIT>{}, PPInt<3>{}, nohook, nohook);
            }
        }
        if constexpr (abl::stamp) asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[1][0][0]), "v"(acc[2][0][0]));
        stamp(0);                                // [0] first half of the MFMA steps
        SHM_LDS_BARRIER();
        stamp(1);                                // [1] wait at the mid-X barrier (= the other group's staging barrier)
        if (act) {
            xsteps(PPInt<XSPLIT>{}, PPInt<72>{}, PPInt<3>{}, nohook, nohook);
            if (abl::stamp && (dbg & 3) == 1) __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (abl::stamp) asm volatile("" ::"v"(acc[0][1][1]), "v"(acc[1][1][1]), "v"(acc[2][1][1]), "v"(acc[3][1][1]));
        stamp(2);                                // [2] second half
        SHM_LDS_BARRIER();                       // every wave of the group has read its last fragment: the halo image may be refilled
        stamp(3);                                // [3] wait at the end-of-X barrier
        // ===== Y segment (the other group is in its X segment).  One branch around the whole segment, barriers included (s_barrier counts arrivals,
        // not program addresses): with yhead and ytail under separate `if (act)` hipcc's wait tracker sees a path "DMA issued, stores skipped",
        // on which vmcnt(8) does not cover the DMA, and drains vmcnt(0) in front of the next LDS read
        if (act) {
            if (abl::stamp && (dbg & 3) == 2) __builtin_amdgcn_s_setprio(1);
            yhead(q);
            stamp(9);                            // [9] epilogue + staging writes
            SHM_LDS_BARRIER();                   // the staging image of patch q is complete
            stamp(5);                            // [5] wait at the staging barrier
            ytail(q);
            stamp(6);                            // [6] staging reads + stores
            // the next halo's items were issued in yhead; younger: this tail's eight stores.  The BUILTIN wait (vmcnt(8), nothing else): hipcc's wait
            // tracker sees it and knows the halo DMA has landed -- behind an inline-asm wait it would drain the stores too (vmcnt(0)) in front of
            // the next X segment's first fragment read
            __builtin_amdgcn_s_waitcnt(0x0F78);
            if (abl::stamp && (dbg & 3) == 2) __builtin_amdgcn_s_setprio(0);
            stamp(7);                            // [7] wait for the next halo
            SHM_LDS_BARRIER();                   // the group's next halo has landed for all four waves
            stamp(8);                            // [8] wait at the end-of-Y barrier
        } else {
            SHM_LDS_BARRIER();
            SHM_LDS_BARRIER();
        }
    }
    if (G == 0) {
        SHM_LDS_BARRIER();
        SHM_LDS_BARRIER();
    }
    if (clk && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && tid == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - ck0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - cr0;
    }
    if constexpr (STATS)
        if (q0 < q1) {
            xstats_tail((q1 - 1) / ppi);
            flush(simg);
        }
    if constexpr (abl::stamp)
        if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && lane == 0 && a.bias) {
            unsigned* dump = (unsigned*)a.bias + 64;          // behind the 64 bias values (the probe passes a longer buffer)
#pragma unroll
            for (int k = 0; k < 11; ++k) dump[wave * 16 + k] = (unsigned)tph[k];
            dump[wave * 16 + 11] = (unsigned)(q1 - q0);
            const unsigned long long te = __builtin_amdgcn_s_memtime(), re = __builtin_amdgcn_s_memrealtime();
            dump[wave * 16 + 12] = (unsigned)(tl0 - tk0);          // cycles before the loop
            dump[wave * 16 + 13] = (unsigned)(te - tl0);           // cycles of the loop and the tail
            dump[wave * 16 + 14] = (unsigned)(rl0 - rk0);          // the same in 10 ns ticks
            dump[wave * 16 + 15] = (unsigned)(re - rl0);
        }
}

int shm_pp_eligible(const TapGemmArgs& a) {
    return a.K == 64 && a.x2 == nullptr && a.hi % PP_PH == 0 && a.wi % PP_PW == 0 && a.nout % 64 == 0 && (a.y2 == nullptr || a.n1 % 64 == 0) && a.ybytes != 0 &&
           (a.y2 == nullptr || a.y2bytes != 0) && (size_t)a.hi * a.wi * a.ldx * 2 < (size_t)1 << 31 && a.slope >= 0.f && a.slope <= 1.f;
}

int shm_pp_launch(const TapGemmArgs& a, int batch, int ncu, hipStream_t st, const char* who) {
    const int npatch = batch * (a.hi / PP_PH) * (a.wi / PP_PW), nyw = a.nout / 64;
    int gx = ncu / nyw;                  // one eight-wave block per CU
    if (gx < 1) gx = 1;
    if (gx > (npatch + 1) / 2) gx = (npatch + 1) / 2;
    const bool epi = a.slope != 1.f || a.stats != nullptr || a.bias != nullptr;
    // SHM_PP_DBG: switches of the stamped timing build only (tools/probes/pp_stamps.py): bits 0-1 wave priority (1: X segment, 2: Y segment), 4 no halo
    // DMA, 8 no epilogue, 16 no store pass.  The product build ignores the value.
    static const int prio = [] {
        const char* e = getenv("SHM_PP_DBG");
        return e ? atoi(e) : 0;
    }();
    hipError_t att = hipSuccess;
#define PP_LAUNCH(EPI_, XS_)                                                                                                            \
    do {                                                                                                                                \
        static const hipError_t at_ = hipFuncSetAttribute((const void*)tapgemm_pp_bf16_kernel<EPI_, XS_>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS); \
        att = at_;                                                                                                                      \
        if (att == hipSuccess) hipLaunchKernelGGL((tapgemm_pp_bf16_kernel<EPI_, XS_>), dim3(gx, nyw, 1), dim3(512), PP_LDS, st, a, npatch, prio, shm_clock_probe()); \
    } while (0)
    const int mode = !epi ? 0 : a.stats ? 2 : 1;
    if (mode == 2) PP_LAUNCH(2, 48);
    else if (mode == 1) PP_LAUNCH(1, 48);
    else PP_LAUNCH(0, 48);
#undef PP_LAUNCH
    SHM_REQUIRE(att == hipSuccess, SHM_E_HIP, "%s: cannot reserve %u bytes of LDS: %s", who, PP_LDS, hipGetErrorString(att));
    shm_set_last_kernel("tapgemm_pp_bf16_kernel<%d>", mode);
    return SHM_OK;
}
