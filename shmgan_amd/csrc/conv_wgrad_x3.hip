// fp32 weight gradient of the 3x3 unit-stride layers as SIX bf16 MFMA products ("wgrad.f32_split" = 1; bench.py --dtype f32x3; opt-in, round 5).
//
//   dW[tap][ci][co] = sum_p X[src(p, tap)][ci] * dY[p][co]          (ShmGANwithSSpecSeg.py:859-872: the gradients both optimizers apply)
//
// The exact-fp32 path (wgrad_halo_kernel) runs on v_mfma_f32_32x32x2_f32 at 0.90 of a 157 TFLOP/s pipe; the bf16 pipe is sixteen times as
// fast.  Every fp32 operand splits EXACTLY into three bf16 planes by truncation -- x0 = x & 0xffff0000, r = x - x0 (exact), x1 = r & 0xffff0000,
// x2 = r - x1 (eight significant bits left: a bf16 value as it stands) -- and the product of two fp32 numbers is the sum of the nine plane
// products, of which the six with i + j <= 2 carry everything above 2^-24 of it.  bf16 x bf16 is exact in the MFMA's fp32 accumulators, so
// the six-product sum agrees with an fp32 dot product on random-sign operands (rel-L2 against float64 on a 512 x 576 x 128 product: 1.2e-7, the f32
// MFMA's own 3.1e-7; LABNOTES 10.8): six MFMAs of 32 cycles instead of eight of 64.  NOT bit-faithful fp32 (round 6): the bf16 MFMA truncates below its
// alignment window on every accumulation, towards zero -- on one-signed operands a one-sided shrink of ~0.3 ulp per step of the chain, -5.3e-6 over
// the step's longest reduction (tests/test_x3_gpu.py::test_x3_wgrad_same_sign_operands_over_the_longest_reduction; exact-fp32 MFMA: 2.4e-7).
//
// Structure = wgrad_halo_bf16_kernel<2> (block: 64 ci x 64 co x 9 taps over a slice of 2 x 16-pixel patches, wave w the 32 x 32 sub-tile
// (w >> 1, w & 1) of every tap, operands by ds_read_b64_tr_b16 from [pixel][64 channels] rows with the half-swap swizzle, deterministic
// split-K slabs) with the operand planes made IN the kernel: a stage's 4 x 18 x-halo pixels and 2 x 16 dY pixels come from HBM as fp32 into
// registers (two 16-byte loads per lane and item, out-of-image = out-of-range offset = zeros), are split (4 VALU per value + 1.5 to pack) and
// written as three x planes and three dY planes of the bf16 kernel's LDS image (42 KiB per stage, ONE LDS stage: the next stage's loads are in
// flight in registers during this stage's 108 MFMAs per wave; two blocks per CU cover each other's split / write phase).  Splitting in the
// kernel costs ~150 vector instructions per lane and stage beside 3 456 matrix-pipe cycles; as a pass of its own it would write and
// re-read 6 bytes per element of every activation and gradient tensor (~20 ms per step).  Per (K step of 16 pixels, tap): one x fragment per
// plane meets the K step's three dY fragments: x0 three MFMAs, x1 two, x2 one.
#include "wgrad.h"
#include "x3split.h"


// NM: the x source a.ntpart is the UN-normalised activation of an InstanceNorm block (SHM_NORM_EXACT, as wgrad_halo_kernel<1>): shm_in_norm on the in-image
// pixels of the stage registers before the split; the block's 64 (mean, inv, beta) triples of the current sample sit in LDS behind the plane images
// S2 (round 6): the stride-2 layers (ShmGANwithSSpecSeg.py:353-361 the discriminator's convolutions, :298-319 the Conv2DTranspose weight gradients with the
// operands' roles swapped) -- patches of 2 x 16 OUTPUT pixels, a 5 x 33 input halo kept as wgrad_halo8_bf16_kernel<1> keeps it: per halo row an even run
// (columns 0, 2, .., 32 at LDS rows 0 .. 16) and an odd run (columns 1, 3, .., 31 at rows 20 .. 35), so that the sixteen pixels of a K step are sixteen
// consecutive LDS rows for every kw (kw = 1: the odd run, kw = 2: the even run one row on).  TF SAME padding of an even map: nothing before the first
// row / column, one zero row / column behind the last (out-of-image = out-of-range offset = zeros).
template <int R, bool NM = false, bool S2 = false>
__global__ __launch_bounds__(256, 2) void wgrad_halo_x3_kernel(const WgradHaloArgs a) {
    static_assert(!S2 || (R == 2 && !NM), "stride 2: stages of two output rows, plain sources");
    constexpr int PW = 16, HP = 20;                     // patch R x 16; S1: halo R + 2 rows, LDS pitch 20 (18 valid); S2: pitch of the even run
    constexpr int HRP = S2 ? 36 : HP;                   // LDS rows per halo row
    constexpr int NHROW = S2 ? 2 * R + 1 : R + 2;
    constexpr int NHR = NHROW * HRP, NPX = R * PW;      // S1, R = 2: 80 halo rows, 32 dY rows; S2: 180 halo rows
    constexpr int XP = NHR * 64, DP = NPX * 64;         // bf16 elements per x plane / dY plane
    constexpr int NXI = (NHR + 7) / 8, NDI = NPX / 8;   // items of 8 rows x 64 channels: 10 + 4 (S2: 23 + 4; the last x item's rows 180 .. 183 are not written)
    constexpr int NIT = NXI + NDI, NJ = (NIT + 3) / 4;  // items per wave: waves below NIT % 4 (or all) take NJ, the others NJ - 1
    extern __shared__ __attribute__((aligned(1024))) unsigned short smem[];       // [x plane 0..2][dY plane 0..2][NM: mean, inv, beta x 64]
    [[maybe_unused]] float* const tab = (float*)(smem + 3 * XP + 3 * DP);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int mi = wave >> 1, ni = wave & 1;
    const Blk3 blk = xcd_block_order();
    const int ci0 = blk.x * 64, co0 = blk.y * 64;
    const int pid0 = blk.z * a.patches_per_split;
    const int pid1 = min(a.npatch, pid0 + a.patches_per_split);
    const int nstages = pid1 - pid0;

    // item lane mapping (as the bf16 kernel's DMA): lane -> (row l >> 3 of the item, 16-byte bf16 chunk l & 7 = eight channels); LDS chunk j of
    // row r holds source chunk j ^ (4 * bit1(r)); items are 8 rows, so bit1(r) = bit1(l >> 3)
    const int drow = lane >> 3;
    const int sch = (lane & 7) ^ (((drow >> 1) & 1) << 2);
    const bool second = ci0 >= a.c1;
    const int ldX = second ? a.ldx2 : a.ldx;
    const int cX = ci0 + sch * 8;
    const bool xvalid = cX < a.cin_ld;
    const int ccX = second ? cX - a.c1 : cX;
    const int coD = co0 + sch * 8;
    const bool dvalid = coD < a.cout;
    const __amdgpu_buffer_rsrc_t rsx = second ? __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, a.x2bytes, 0x00020000)
                                              : __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);

    const int ho = S2 ? a.h / 2 : a.h, wo = S2 ? a.w / 2 : a.w;
    int n, pr, pc;                                      // patch origin in OUTPUT pixels
    {
        const int ppc = ho / R, ppi = ppc * (wo / PW);             // patches numbered down the columns of an image, see wgrad_halo_kernel
        const int p = pid0 < a.npatch ? pid0 : 0;
        n = p / ppi;
        const int r = p - n * ppi;
        pc = (r / ppc) * PW;
        pr = (r % ppc) * R;
    }
    // per-lane constant byte offset + patch origin, edge bits (five per item, one register), as in wgrad_halo_bf16_kernel -- fp32 elements here
    unsigned off0[NJ], bm = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int item = wave + 4 * j;
        unsigned bits;
        if (item < NXI) {
            const int hp = 8 * item + drow;
            if constexpr (S2) {
                const int r_ = hp / HRP, t_ = hp - r_ * HRP;
                const int c_ = t_ < HP ? 2 * t_ : 2 * (t_ - HP) + 1;
                const bool ok = hp < NHR && (t_ < HP ? t_ <= PW : true);
                off0[j] = (unsigned)((r_ * a.w + c_) * ldX + ccX) * 4u;
                bits = !(xvalid && ok) ? 16u : (r_ == NHROW - 1 ? 2u : 0u) | (c_ == 2 * PW ? 8u : 0u);
            } else {
                const int r_ = hp / HP, c_ = hp - r_ * HP;
                off0[j] = (unsigned)((r_ * a.w + c_) * ldX + ccX) * 4u;
                bits = !(xvalid && c_ < PW + 2) ? 16u : (r_ == 0 ? 1u : 0u) | (r_ == R + 1 ? 2u : 0u) | (c_ == 0 ? 4u : 0u) | (c_ == PW + 1 ? 8u : 0u);
            }
        } else {
            const int q = 8 * (item - NXI) + drow;
            off0[j] = (unsigned)(((q >> 4) * wo + (q & 15)) * a.lddy + coD) * 4u;
            bits = (dvalid && item < NIT) ? 0u : 16u;
        }
        bm |= bits << (5 * j);
    }
    // stage registers: eight fp32 channels per item and lane
    f32x4 sr[NJ][2];
    [[maybe_unused]] int sn = 0, spr = 0, spc = 0;             // (NM) sample and origin of the patch the stage registers hold
    auto load = [&]() {
        if constexpr (NM) {
            sn = n;
            spr = pr;
            spc = pc;
        }
        const int org = S2 ? (n * a.h + 2 * pr) * a.w + 2 * pc : (n * a.h + pr - 1) * a.w + (pc - 1);       // input pixel index of halo (0,0)
        const unsigned edges = 16u | ((!S2 && pr == 0) ? 1u : 0u) | (pr + R == ho ? 2u : 0u) | ((!S2 && pc == 0) ? 4u : 0u) | (pc + PW == wo ? 8u : 0u);
        const unsigned xb = (unsigned)(org * ldX) * 4u, db = (unsigned)(((n * ho + pr) * wo + pc) * a.lddy) * 4u;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int item = wave + 4 * j;
            if (j < NJ - 1 || item < NIT) {
                const bool isx = item < NXI;                   // wave-uniform
                const unsigned off = (bm & (edges << (5 * j))) ? 0xffffffffu : off0[j] + (isx ? xb : db);
                sr[j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(isx ? rsx : rsd, off, 0, 0));
                sr[j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(isx ? rsx : rsd, off == 0xffffffffu ? off : off + 16u, 0, 0));
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
    // split the stage registers and write the six plane images
    const bool nm_on = NM && a.nt != nullptr && (int)second == a.ntpart;        // block-uniform
    [[maybe_unused]] int nimg = -1;
    auto spill = [&]() {
        [[maybe_unused]] float nmean[8], ninv[8], nbeta[8];
        if constexpr (NM)
            if (nm_on) {
                int lc = lane;
                asm volatile("" : "+v"(lc));             // re-read per stage (24 registers across the MFMA loop would spill)
                const int c8 = ((lc & 7) ^ ((((lc >> 3) >> 1) & 1) << 2)) * 8;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const f32x4 m4 = *(const f32x4*)(tab + c8 + 4 * hf), i4 = *(const f32x4*)(tab + 64 + c8 + 4 * hf), b4 = *(const f32x4*)(tab + 128 + c8 + 4 * hf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        nmean[4 * hf + e] = m4[e];
                        ninv[4 * hf + e] = i4[e];
                        nbeta[4 * hf + e] = b4[e];
                    }
                }
            }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int item = wave + 4 * j;
            if (j < NJ - 1 || item < NIT) {
                u32x4 p0, p1, p2;
                float v[8] = {sr[j][0][0], sr[j][0][1], sr[j][0][2], sr[j][0][3], sr[j][1][0], sr[j][1][1], sr[j][1][2], sr[j][1][3]};
                if constexpr (NM)
                    if (nm_on && item < NXI) {
                        const int hp = 8 * item + drow;
                        const int r_ = hp / HP, c_ = hp - r_ * HP;
                        const int iy = spr - 1 + r_, ix = spc - 1 + c_;
                        if (xvalid && c_ < PW + 2 && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w) {       // padding stays zero
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = shm_in_norm(v[e], nmean[e], ninv[e], nbeta[e]);
                        }
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned q0, q1, q2;
                    x3_split_pair(v[2 * e], v[2 * e + 1], q0, q1, q2);
                    p0[e] = q0;
                    p1[e] = q1;
                    p2[e] = q2;
                }
                const bool isx = item < NXI;                   // wave-uniform
                unsigned short* dst = smem + (isx ? item * 512 : 3 * XP + (item - NXI) * 512) + lane * 8;
                const int ps = isx ? XP : DP;
                // S2: the x planes end inside the last item (180 rows = 22.5 items): its upper half belongs to the next plane
                if (!S2 || !isx || 8 * item + drow < NHR) {
                    *(u32x4*)(dst) = p0;
                    *(u32x4*)(dst + ps) = p1;
                    *(u32x4*)(dst + 2 * ps) = p2;
                }
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read addresses (elements), as in wgrad_halo_bf16_kernel: the lane supplies row 8 hh + (i >> 2) [+ 4 for the second read] and
    // channels [32 tile + 16 (g & 1) + 4 (i & 3), + 4); tap and K step enter as immediates, kw shifts the row and with it bit 1 of the row index
    const int fq = 8 * hh + ((lane & 15) >> 2);
    const int fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    int fa[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        // S1: kw shifts the row; S2: kw = 0 / 2 read the even run (2 = one row on), kw = 1 the odd run HP rows further (same bit 1: HP % 4 == 0)
        const int row = fq + (S2 ? (kw == 2 ? 1 : 0) : kw);
        fa[kw] = row * 64 + ((mi * 32 + fcol) ^ (((row >> 1) & 1) << 5)) + ((S2 && kw == 1) ? HP * 64 : 0);
    }
    const int fb = 3 * XP + fq * 64 + ((ni * 32 + fcol) ^ (((fq >> 1) & 1) << 5));
    auto compute = [&]() {
#pragma unroll
        for (int qr = 0; qr < R; ++qr) {
            bf16x8 d[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) d[p] = tr_frag(smem + p * DP + fb + qr * PW * 64);
            // the x fragments of halo row qr + kh are shared by (qr, kh) and (qr + 1, kh - 1): hipcc would keep them all in registers (12 rows x
            // 3 planes x 4 registers beside 144 accumulators: 57 spills); an opaque copy of the addresses per K step makes them reads again
            int fx[3] = {fa[0], fa[1], fa[2]};
            asm volatile("" : "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]));
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                bf16x8 x[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) x[p] = tr_frag(smem + p * XP + fx[t % 3] + ((S2 ? 2 * qr : qr) + t / 3) * HRP * 64);
                // the small products first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], d[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], d[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], d[2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], d[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], d[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], d[0], acc[t], 0, 0, 0);
            }
        }
    };

    if (nstages > 0) {
        load();
        for (int s = 0; s < nstages; ++s) {
            SHM_LDS_BARRIER();                   // every wave has read the last fragment of the previous stage
            if constexpr (NM)
                if (nm_on && sn != nimg) {       // block-uniform: the stage in registers belongs to another sample than the table in LDS
                    nimg = sn;
                    if (tid < 192) {
                        const int pl = tid >> 6, ch = ci0 - (second ? a.c1 : 0) + (tid & 63);
                        tab[tid] = ch < a.ntc ? a.nt[((size_t)sn * SHM_NT_PLANES + pl) * a.ntc + ch] : 0.f;
                    }
                    SHM_LDS_BARRIER();
                }
            spill();                             // (hipcc waits for the stage registers' loads here)
            if (s + 1 < nstages) load();         // the next stage's loads fly during this stage's MFMAs
            SHM_LDS_BARRIER();                   // the six plane images are complete
            compute();
        }
    }

    float* out = a.part + (size_t)blk.z * 9 * a.cin * a.cout;
    const int con = co0 + ni * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (ci < a.cin && con < a.cout) out[((size_t)t * a.cin + ci) * a.cout + con] = acc[t][r];
        }
    }
}

int shm_wgrad_x3_launch(const WgradHaloArgs& hgs, int cin, int cout, int nsplit, int rows, hipStream_t st, bool stride2) {
    const dim3 grid(shm_cdiv(cin, 64), shm_cdiv(cout, 64), nsplit);
    const bool nm = hgs.nt != nullptr;               // SHM_NORM_EXACT source
    if (stride2) {
        constexpr unsigned kLds = (3u * 180 + 3u * 32) * 128u;       // 79.5 KiB: two blocks per CU, just
        SHM_REQUIRE(!nm && rows == 2, SHM_E_SHAPE, "shm_conv2d_wgrad: the stride-2 x3 form takes plain sources and stages of two rows");
        static const hipError_t attr = hipFuncSetAttribute((const void*)wgrad_halo_x3_kernel<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        SHM_REQUIRE(attr == hipSuccess, SHM_E_HIP, "shm_conv2d_wgrad: cannot reserve 79.5 KiB of LDS: %s", hipGetErrorString(attr));
        hipLaunchKernelGGL((wgrad_halo_x3_kernel<2, false, true>), grid, dim3(256), kLds, st, hgs);
        shm_set_last_kernel("wgrad_halo_x3_kernel<2, false, true>");
        return SHM_OK;
    }
    if (rows == 4) {
        constexpr unsigned kLds = (3u * (6 * 20) + 3u * (4 * 16)) * 128u + 768u;      // 69 KiB: two blocks per CU
        SHM_REQUIRE(!nm, SHM_E_SHAPE, "shm_conv2d_wgrad: the normalising x3 form runs stages of two rows");
        static const hipError_t attr = hipFuncSetAttribute((const void*)wgrad_halo_x3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        SHM_REQUIRE(attr == hipSuccess, SHM_E_HIP, "shm_conv2d_wgrad: cannot reserve 69 KiB of LDS: %s", hipGetErrorString(attr));
        hipLaunchKernelGGL((wgrad_halo_x3_kernel<4>), grid, dim3(256), kLds, st, hgs);
    } else {
        constexpr unsigned kLds = (3u * (4 * 20) + 3u * (2 * 16)) * 128u + 768u;      // 42 KiB
        if (nm) hipLaunchKernelGGL((wgrad_halo_x3_kernel<2, true>), grid, dim3(256), kLds, st, hgs);
        else hipLaunchKernelGGL((wgrad_halo_x3_kernel<2>), grid, dim3(256), kLds, st, hgs);
    }
    shm_set_last_kernel(nm ? "wgrad_halo_x3_kernel<%d, true>" : "wgrad_halo_x3_kernel<%d>", rows == 4 ? 4 : 2);
    return SHM_OK;
}
