// fp32 unit-stride 3x3 forward / input-gradient convolutions as SIX bf16 MFMA products ("conv.f32_split" = 1; opt-in, round 5).
//
//   y[p][n] = sum_{tap, k} X[p + tap][k] * W[tap][n][k]            (ShmGANwithSSpecSeg.py:244-323: Conv2D 3x3 'same' of the generator blocks and,
//                                                                   with the taps mirrored and W^T, their input gradients)
//
// The exact-fp32 path (tapgemm_halo_kernel<float>) runs v_mfma_f32_32x32x2_f32 at 0.92-0.95 of a 157 TFLOP/s pipe.  As in conv_wgrad_x3.hip
// every fp32 operand splits EXACTLY into three bf16 planes by truncation and the six plane products with i + j <= 2 carry everything above
// 2^-24 of the product; bf16 x bf16 is exact in the MFMA's fp32 accumulators: six v_mfma_f32_32x32x16_bf16 of 32 cycles per 32 x 32 x 16
// instead of eight fp32 MFMAs of 64.
//
// Structure = tapgemm_halo_kernel's static-tap form (block: 16 x 16 output pixels x 128 output channels, eight waves of 64 x 64; per 32-channel
// chunk the 18 x 18 halo sits in LDS once and the nine taps read it through nine shifted fragment addresses; bf16 LDS rows of 64 bytes with
// that kernel's swizzles), with the operand planes made on the way in:
//   A: the halo chunk comes from HBM as fp32 into REGISTERS (three items of 16 halo rows per wave, two 16-byte loads per lane and item), is
//      split (4 VALU per value + 1.5 to pack) and written as three plane images; ONE LDS stage -- the next chunk's loads are in flight in
//      registers during this chunk's nine taps;
//   B: the tap's weight slice (BN rows x 32 channels, fp32) travels through registers one tap ahead, is split there and written as three planes
//      into one of two LDS stages (a plane tensor made by a kernel of its own in front of every launch cost a launch, a workspace argument and
//      0.4 ms per step; splitting 8 values per lane and tap costs ~45 VALU beside 48 MFMAs).
// Per (tap, 16-channel K step) a wave reads 2 x 3 A and 2 x 3 B fragments and issues 24 MFMAs (x2 w0, x1 w1, x0 w2, x1 w0, x0 w1, x0 w0: the
// small products first).  One barrier per tap, one more per chunk.  LDS: 3 x 24 KiB + 2 x 3 x 8 KiB = 120 KiB, one block per CU.
// Epilogues as tapgemm_halo_kernel's for fp32 outputs: bias + LeakyReLU + element stores through buffer descriptors, InstanceNorm statistics,
// and the gsum form (TapGemmArgs); a normalise-on-load source (SHM_NORM_EXACT) is normalised in the stage registers before the split.  Layers of
// at most 64 output channels run a block of 32 x 16 pixels x 64 channels (eight waves along M).  Not taken (the exact kernels run): SHM_NORM_SCALED
// sources, outputs beyond 4 GiB, K % 32 != 0.
#include "tapgemm.h"
#include "x3split.h"

namespace {
// Two block shapes, eight waves of 64 x 64 each: BN = 128 output channels x 16 x 16 pixels (waves 4 (M) x 2 (N)) and, for the layers of at most 64
// output channels, BN = 64 x a patch of 32 x 16 pixels (waves 8 x 1) -- the same fragment reads per MFMA
template <int BN>
struct X3Shape {
    static constexpr int HC = 18, NW = 8, PH = BN == 128 ? 16 : 32, WGN = BN / 64, WGM = NW / WGN;
    static constexpr int NROW = (PH + 2) * HC, NIT = (NROW + 15) / 16, NA = (NIT + NW - 1) / NW;       // 24 / 39 items of 16 halo rows, 3 / 5 per wave
    static constexpr int NBI = BN / 16;                                                               // weight items: 8 / 4
    static constexpr int ASTG = NIT * 256;             // 4-byte words per A plane (rows of 64 bytes)
    static constexpr int BSTG = BN * 16;               // words per B plane stage
    static constexpr unsigned LDS = (3u * ASTG + 6u * BSTG) * 4u;       // 120 KiB / 141 KiB: one block per CU
};

}  // namespace

template <bool GS, int BN>
__global__ __launch_bounds__(512, 2) void tapgemm_halo_x3_kernel(const TapGemmArgs a) {
    typedef X3Shape<BN> SH;
    constexpr int HC = SH::HC, NW = SH::NW, NA = SH::NA, NIT = SH::NIT, ASTG = SH::ASTG, BSTG = SH::BSTG, PH = SH::PH, WGN = SH::WGN, WGM = SH::WGM;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    float* const sA = smem;                      // [plane][ASTG]
    float* const sB = smem + 3 * ASTG;           // [stage][plane][BSTG]

    const TapPhase& P = a.ph[0];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const int ppr = a.wi >> 4, ppi = (a.hi / PH) * ppr;
    const int img = blockIdx.x / ppi, prem = blockIdx.x - img * ppi;
    const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
    const int n0 = blockIdx.y * BN;

    // ---- staging lane constants.  A item it = halo rows [16 it, 16 it + 16) of the (PH + 2) x 18 halo; wave w owns items w, w + 8, ...; lane ->
    // (row drow of the item, 16-byte bf16 chunk dq of its 64-byte row) = eight channels = 32 bytes of fp32 source; LDS chunk dq holds source
    // chunk dq ^ (((R >> 1) + R / 18) & 3) (tapgemm_halo_kernel's conflict-free static-tap swizzle)
    const int drow = lane >> 2, dq = lane & 3;
    unsigned arow1[NA], arow2[NA];
    unsigned inimg = 0;                            // bit j: the lane's row of item j is a pixel of the image (what a normalising source touches)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int hrow = 16 * (wave + NW * j) + drow;
        const int hr = hrow / HC, hc = hrow - hr * HC;
        const int iy = y0 - 1 + hr, ix = x0 - 1 + hc;
        const bool v = hrow < SH::NROW && (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
        inimg |= (v ? 1u : 0u) << j;
        const int pix = (img * a.hi + iy) * a.wi + ix;
        const int coff = (dq ^ (((hrow >> 1) + hr) & 3)) * 8;
        arow1[j] = v ? (unsigned)(pix * a.ldx + coff) * 4u : 0xffffffffu;
        arow2[j] = v ? (unsigned)(pix * a.ldx2 + coff) * 4u : 0xffffffffu;
    }
    // the lane's first channel within a 32-channel chunk, item by item (two bits each): only the swizzle differs between items
    unsigned coffs = 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int hrow = 16 * (wave + NW * j) + drow;
        coffs |= (unsigned)(dq ^ (((hrow >> 1) + hrow / HC) & 3)) << (2 * j);
    }
    // B item of wave w (< BN / 16) = weight rows [16 w, 16 w + 16) of the block's BN: LDS chunk dq of row r holds source chunk dq ^ ((r >> 2) & 3)
    const int brow = wave * 16 + drow;
    const bool bwave = wave < SH::NBI;             // wave-uniform
    const unsigned wrow = n0 + brow < a.nout ? (unsigned)((n0 + brow) * a.K + (dq ^ ((brow >> 2) & 3)) * 8) * 4u : 0xffffffffu;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.wbytes, 0x00020000);
    const int nch = a.K >> 5;

    // stage registers.  Item wave + 8 j of the halo exists for every wave when NIT is a multiple of 8 (24); with 39 items wave 7 has no fifth one
    f32x4 ar[NA][2], br[2];
    auto item_exists = [&](int j) { return NA * NW <= NIT || j < NA - 1 || wave + NW * j < NIT; };
    // normalise-on-load (TapGemmArgs::nt, SHM_NORM_EXACT): source a.ntpart is the UN-normalised activation of an InstanceNorm block; shm_in_norm
    // on the in-image values of the stage registers before the split (padding stays zero), the (mean, inv, beta) rows of the block's sample
    // straight from the table (L2): 6 x 16 bytes per lane, item and chunk
    const float* const ntb = a.nt ? a.nt + (size_t)img * SHM_NT_PLANES * a.ntc : nullptr;
    int ncs = -1;                                  // first channel (within the normalised part) of the chunk in the stage registers, -1: as stored
    auto load_a = [&](int chunk) {
        const int c0 = chunk << 5;
        const bool second = c0 >= a.c1;                          // block-uniform
        const unsigned cb = (unsigned)(second ? c0 - a.c1 : c0) * 4u;
        ncs = (ntb && (int)second == a.ntpart) ? (second ? c0 - a.c1 : c0) : -1;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if (item_exists(j)) {
                const unsigned r = second ? arow2[j] : arow1[j];
                const unsigned off = r == 0xffffffffu ? r : r + cb;
                const unsigned off2 = r == 0xffffffffu ? r : r + cb + 16u;
                ar[j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? rsx2 : rsx, off, 0, 0));
                ar[j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? rsx2 : rsx, off2, 0, 0));
            }
        }
    };
    auto spill_a = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if (!item_exists(j)) continue;
            if (ncs >= 0 && ((inimg >> j) & 1u)) {
                const float* t = ntb + ncs + 8 * ((coffs >> (2 * j)) & 3u);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const f32x4 m4 = *(const f32x4*)(t + 4 * hf), i4 = *(const f32x4*)(t + a.ntc + 4 * hf), b4 = *(const f32x4*)(t + 2 * a.ntc + 4 * hf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ar[j][hf][e] = shm_in_norm(ar[j][hf][e], m4[e], i4[e], b4[e]);
                }
            }
            u32x4 p0, p1, p2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned q0, q1, q2;
                x3_split_pair(ar[j][e >> 1][2 * (e & 1)], ar[j][e >> 1][2 * (e & 1) + 1], q0, q1, q2);
                p0[e] = q0;
                p1[e] = q1;
                p2[e] = q2;
            }
            float* dst = sA + (wave + NW * j) * 256 + lane * 4;
            *(u32x4*)dst = p0;
            *(u32x4*)(dst + ASTG) = p1;
            *(u32x4*)(dst + 2 * ASTG) = p2;
        }
    };
    auto load_b = [&](int t_wi, int chunk) {
        if (!bwave) return;
        const unsigned wbase = (unsigned)((t_wi * a.nout) * a.K + (chunk << 5)) * 4u;
        const unsigned off = wrow == 0xffffffffu ? wrow : wrow + wbase;
        br[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0));
        br[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wrow == 0xffffffffu ? wrow : off + 16u, 0, 0));
    };
    auto spill_b = [&](int stage) {
        if (!bwave) return;
        u32x4 p0, p1, p2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned q0, q1, q2;
            x3_split_pair(br[e >> 1][2 * (e & 1)], br[e >> 1][2 * (e & 1) + 1], q0, q1, q2);
            p0[e] = q0;
            p1[e] = q1;
            p2[e] = q2;
        }
        float* dst = sB + stage * 3 * BSTG + wave * 256 + lane * 4;
        *(u32x4*)dst = p0;
        *(u32x4*)(dst + BSTG) = p1;
        *(u32x4*)(dst + 2 * BSTG) = p2;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment addresses (words).  A: lane -> patch pixel (4 wm + 2 i + (l31 >> 4), l31 & 15), K half h: the halo row of tap t is a
    // register for the whole block, tile i sits 36 halo rows (2304 bytes) further on (the swizzle does not change), K step kk = XOR 8 words.
    // B: row l31 of the wave's 64 (+ 32 j), chunk (2 kk + h) ^ ((l31 >> 2) & 3)
    const int hb0 = (4 * wm + (l31 >> 4) + 1) * HC + (l31 & 15) + 1;
    int fs[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int hrow = hb0 + P.dh[t] * HC + P.dw[t];
        fs[t] = hrow * 16 + ((h ^ (((hrow >> 1) + hrow / HC) & 3)) << 2);
    }
    const int swb = (l31 >> 2) & 3;
    const int fb0 = wn * 64 * 16 + l31 * 16 + ((0 + h) ^ swb) * 4, fb1 = wn * 64 * 16 + l31 * 16 + ((2 + h) ^ swb) * 4;
    const int tl = lane < 9 ? lane : 0;
    const int tapw_v = P.widx[tl];
    int tw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tw[t] = __builtin_amdgcn_readlane(tapw_v, t);

    load_a(0);
    load_b(tw[0], 0);
    for (int chunk = 0; chunk < nch; ++chunk) {
        if (chunk) SHM_LDS_BARRIER();                // every wave is past its last fragment read of the previous chunk's planes
        spill_a();
        if (chunk + 1 < nch) load_a(chunk + 1);      // in flight during this chunk's nine taps
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // B stage parity of step s = 9 chunk + tap: the previous step's fragments came from the other stage, and every wave finished them
            // before it arrived at that step's barrier
            const int stage = (chunk + tap) & 1;
            spill_b(stage);
            if (tap < 8)
                load_b(tw[tap + 1], chunk);
            else if (chunk + 1 < nch)
                load_b(tw[0], chunk + 1);
            SHM_LDS_BARRIER();                       // the step's weight planes (and, at tap 0, the chunk's halo planes) are complete
            const float* Bb = sB + stage * 3 * BSTG;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x4 av[3][2], bv[3][2];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) av[p][i] = *(const f32x4*)(sA + p * ASTG + ((fs[tap] ^ (kk << 3)) + i * (2 * HC * 16)));
#pragma unroll
                    for (int j = 0; j < 2; ++j) bv[p][j] = *(const f32x4*)(Bb + p * BSTG + j * 512 + (kk ? fb1 : fb0));
                }
                // the small products first
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[PA[q]][i]), __builtin_bit_cast(bf16x8, bv[PB[q]][j]),
                                                                               acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue (tapgemm_halo_kernel's, fp32 outputs): bias + LeakyReLU + element stores (+ InstanceNorm statistics / gsum)
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    float bj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        bj[j] = (a.bias && n < a.nout) ? a.bias[n] : 0.f;
        asm volatile("" : "+v"(bj[j]));
    }
    // gsum: every aux value of the wave's four 32 x 32 tiles first (64 loads per lane in one round trip: one block per CU, nothing else hides the
    // latency; the fragment and stage registers are dead by now), then the stores
    [[maybe_unused]] float q[2][2][16];
    if constexpr (GS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = __builtin_amdgcn_readfirstlane(n0 + wn * 64 + j * 32);
            const int gp = nb < a.n1 ? 0 : 1;
            const int n = nb + l31;
            const bool on = a.gred[gp] != nullptr && nb < a.nout;
            const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a.gaux[gp], 0, on ? 0xfffffff0u : 0u, 0x00020000);
            const unsigned ldab = (unsigned)a.ldgaux[gp] * 4u, nab = (unsigned)(n < a.nout ? n - (gp ? a.n1 : 0) : 0) * 4u;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned ao = (unsigned)((img * a.hi + (y0 + 4 * wm + 2 * i)) * a.wi + x0 + 4 * h) * ldab + nab;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned cr = (unsigned)((r >> 3) * a.wi + 8 * ((r >> 2) & 1) + (r & 3));         // scalar
                    q[j][i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, ao, cr * ldab, 0));
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        // the 32 columns of a (wave, j) group lie in one output part (n1 % 32 == 0, launcher): part, pitch and descriptors are scalars; a lane's
        // address is one register per 32 x 32 tile (its pixel of accumulator row 0) plus a scalar offset per row
        const int nb = __builtin_amdgcn_readfirstlane(n0 + wn * 64 + j * 32);
        const int gp = nb < a.n1 ? 0 : 1;
        const int n = nb + l31;
        const int nl = n - (gp ? a.n1 : 0);
        const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(gp ? a.y2 : a.y, 0, gp ? a.y2bytes : a.ybytes, 0x00020000);
        const unsigned ldyb = (unsigned)(gp ? a.ldy2 : a.ldy) * 4u, nyb = (unsigned)(n < a.nout ? nl : 0) * 4u;
        [[maybe_unused]] const bool on = GS && a.gred[gp] != nullptr && nb < a.nout;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned pix0 = (unsigned)((img * a.hi + (y0 + 4 * wm + 2 * i)) * a.wi + x0 + 4 * h);
            const unsigned yo = pix0 * ldyb + nyb;
            if (n < a.nout) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned cr = (unsigned)((r >> 3) * a.wi + 8 * ((r >> 2) & 1) + (r & 3));
                    const float v = shm_lrelu(acc[i][j][r] + bj[j], a.slope);
                    s1[j] += v;
                    if constexpr (GS)
                        s2[j] += v * q[j][i][r];
                    else
                        s2[j] = __builtin_fmaf(v, v, s2[j]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsy, yo, cr * ldyb, 0);
                }
            }
        }
        const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64), t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
        if constexpr (GS) {
            if (on && h == 0 && n < a.nout) {
                const int pc = gp ? a.nout - a.n1 : a.n1;
                double* dst = a.gred[gp] + ((size_t)((prem * WGM + wm) % a.gslots) * a.gbatch * pc + (size_t)img * pc + nl) * 2;
                atomicAdd(dst, (double)t1);
                atomicAdd(dst + 1, (double)t2);
            }
        } else {
            if (a.stats && h == 0 && n < a.nout) {
                double* dst = a.stats + (size_t)((prem * WGM + wm) % a.stats_slots) * a.stats_stride + ((size_t)img * a.nout + n) * 2;
                atomicAdd(dst, (double)t1);
                atomicAdd(dst + 1, (double)t2);
            }
        }
    }
}

// gs_fused: the launch takes the gsum sums in its epilogue (a.gred as the launcher left them).  Layers of at most 64 output channels take the
// 32 x 16-pixel block (map height % 32 == 0); a normalising source only in SHM_NORM_EXACT mode.
int shm_x3_fwd_eligible(const TapGemmArgs& a) {
    return a.K % 32 == 0 && (a.x2 == nullptr || a.c1 % 32 == 0) && (a.nout > 64 || a.hi % 32 == 0) && (a.nt == nullptr || (a.ntmode == 0 && a.ntc % 32 == 0)) &&
           a.ybytes != 0 && (a.y2 == nullptr || (a.y2bytes != 0 && a.n1 % 32 == 0));
}

template <bool GS, int BN>
static int x3_launch_t(const TapGemmArgs& a, int batch, hipStream_t st, const char* who) {
    typedef X3Shape<BN> SH;
    static const hipError_t attr = hipFuncSetAttribute((const void*)tapgemm_halo_x3_kernel<GS, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, SH::LDS);
    SHM_REQUIRE(attr == hipSuccess, SHM_E_HIP, "%s: cannot reserve %u bytes of LDS: %s", who, SH::LDS, hipGetErrorString(attr));
    const dim3 grid(batch * (a.hi / SH::PH) * (a.wi / 16), shm_cdiv(a.nout, BN), 1);
    hipLaunchKernelGGL((tapgemm_halo_x3_kernel<GS, BN>), grid, dim3(512), SH::LDS, st, a);
    shm_set_last_kernel("tapgemm_halo_x3_kernel<%s, %d>", GS ? "true" : "false", BN);
    return SHM_OK;
}

int shm_x3_fwd_launch(const TapGemmArgs& a, int batch, bool gs_fused, hipStream_t st, const char* who) {
    if (a.nout > 64) return gs_fused ? x3_launch_t<true, 128>(a, batch, st, who) : x3_launch_t<false, 128>(a, batch, st, who);
    return gs_fused ? x3_launch_t<true, 64>(a, batch, st, who) : x3_launch_t<false, 64>(a, batch, st, who);
}
