// The bf16 weights-in-registers tap GEMM (K <= 64 input channels, 3x3, unit stride): BASELINE.json's north-star block -- conv 64 -> 64 at 256 x 256
// + bias + LeakyReLU + InstanceNorm statistics (ShmGANwithSSpecSeg.py:244-245), its input gradient, and the generator's first layer.  A translation
// unit of its own since round 5 (it is the kernel that gets rebuilt most; conv_igemm.hip takes 90 s to compile).
#include "tapgemm.h"

// ------------------------------------------------------------------------------------------
// Round 4: the bf16 weights-in-registers kernel (K <= 64 input channels, 3x3, unit stride; the north star's 64 -> 64 block), rebuilt
// around what the timing-only ablations of its predecessor (tapgemm_wreg_f32_kernel<..., bf16_t>, round 3) showed: with the MFMAs
// removed that kernel still took 70 % of its time, and of that the OUTPUT STORES were half -- sixteen 2-byte stores per wave and patch
// (2.6 M store instructions per launch at n = 40; the vector-memory unit takes a store instruction per ~16 cycles and CU whatever
// its width), each touching four 32-byte pieces of four lines.  Same block shape (eight waves = 2 (M) x 4 (N), 8 x 16-pixel patches,
// 16 output channels per wave, v_mfma_f32_16x16x32_bf16, 72 weight registers, 128 VGPRs: two blocks per CU), three changes:
//  * Transposed product.  The WEIGHTS are the A operand and the activation fragment the B operand, so a lane's four accumulator
//    registers are four consecutive CHANNELS of one pixel (not four pixels of one channel): they pack into 8 bytes with two
//    v_cvt_pk_bf16_f32, and the bias vector is the accumulators' initial value as it stands.
//  * Outputs through LDS.  Each wave writes its 8-byte pieces into a 16 KiB staging image [128 pixels][64 channels] (XOR swizzle on the
//    16-byte chunk), one barrier, then wave w stores patch row w: two 16-byte-per-lane instructions of whole 128-byte lines -- 16 store
//    instructions per patch and block instead of 128.
//  * Halo rows instead of taps.  The MFMA loop walks the six halo rows of the wave's four patch rows: the fragment of halo row r, column
//    shift cs feeds every (patch row m, tap row r - m) pair -- up to three MFMAs -- where the tap loop read it once per pair: 36 fragment
//    reads per 72 MFMAs instead of 72 (at one 1 KiB ds_read_b128 per 16-cycle MFMA and wave the eight waves asked the LDS for all of
//    its 256 B/clk at the full MFMA rate).  bw[u] is the weight slice of the tap whose SOURCE pixel is (u / 3 - 1, u % 3 - 1): tap u of
//    a forward launch, tap 8 - u of an input-gradient launch.  LDS pitch 20: the chunk swizzle (lq + (R >> 1)) & 3 of halo row R then
//    depends on the parity of r only ((20 r) >> 1 = 2 r mod 4): two fragment addresses per column shift, rows and chunks are immediates.
// InstanceNorm statistics (round 5): ON THE MATRIX PIPE, from the staging image.  The kernel is bound by the SIMDs' vector issue port
// (LABNOTES 10.2), and the round-4 epilogue spent ~95 of its ~215 non-MFMA vector instructions per patch and wave on the sums -- unpacking the
// stored bf16 pairs, an add and an fma per value, 32 DPP adds + 8 selects of cross-lane reduction.  Now: after the staging barrier wave w
// fetches the four operand fragments F of the stored values [32 pixels][16 channels] of channel group w & 3 with two ds_read_b64_tr_b16
// each -- a lane then holds eight PIXELS of one channel, which is the A operand of Y^T and, register for register, the B operand of Y --
// and issues four MFMAs into ONE accumulator tile that persists over the block's patches of an image: waves 0-3 mfma(F, F), the 16 x 16
// Gram matrix of the channel group, whose DIAGONAL is sum y^2; waves 4-7 mfma(F, ones), every column = sum y.  8 LDS reads + 4 MFMAs per
// patch and wave, four registers.  The sums are of the values as stored (bf16 products are exact in the MFMA's fp32 accumulators), in fp32
// over the block's pixels of an image (<= a few thousand terms), combined in double at the flush.  EPI = false: input-gradient launches
// (slope 1, no statistics).
// One source tensor, outputs below 4 GiB, a 64-channel block inside one output part.
constexpr int W16_HP = 20, W16_NIT = ((8 + 2) * W16_HP + 15) / 16;        // LDS rows per halo row, 16-row DMA items per 32-channel chunk (13)
constexpr unsigned w16_lds_bytes(int nch) { return 2u * (unsigned)nch * W16_NIT * 1024u + 16384u + 256u; }        // halo x 2, output staging, bias
template <int NCH, bool EPI>
__global__ __launch_bounds__(512, 4) void tapgemm_wreg16_bf16_kernel(const TapGemmArgs a, const int npatch) {
    constexpr int PH = 8, HC = 18, HP = W16_HP, NIT = W16_NIT;
    constexpr int ASTG = NIT * 256;                     // floats per 32-channel chunk of a halo buffer (items of 16 rows x 64 bytes)
    constexpr int ABUF = NCH * ASTG;                    // floats per halo buffer
    extern __shared__ __attribute__((aligned(1024))) float smem[];      // two halo buffers, then the 16 KiB output staging image
    typedef __attribute__((address_space(3))) void* lds_ptr;
    char* const stg = (char*)(smem + 2 * ABUF);

    const TapPhase& P = a.ph[0];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int n0 = blockIdx.y * 64;
    const int ppr = a.wi >> 4, ppi = (a.hi / PH) * ppr;
    const int per = (npatch + gridDim.x - 1) / gridDim.x;
    const int q0 = blockIdx.x * per, q1 = min(npatch, q0 + per);
    if (q0 >= q1) return;

    // ---- weights -> registers (A operand: lane (l15, lq) holds row n = column-of-the-layer n0 + 16 wn + l15, k = 8 lq .. 8 lq + 7 of each chunk)
    const bool flip = P.dh[0] > 0;                       // input-gradient launch: taps arrive as (1 - kh, 1 - kw)
    f32x4 bw[9][NCH];
    {
        const bf16_t* wp = (const bf16_t*)a.w;
        const int ncol = n0 + wn * 16 + l15;
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int wsl = flip ? P.widx[8 - u] : P.widx[u];
#pragma unroll
            for (int c = 0; c < NCH; ++c) bw[u][c] = *(const f32x4*)(wp + ((size_t)wsl * a.nout + ncol) * a.K + c * 32 + lq * 8);
        }
    }
    // the block's 64 bias values live in LDS behind the staging image and are re-read (16 bytes per lane) at the top of every patch: as
    // registers they would be four more live values across the MFMA loop, where 72 weight registers leave no room (the 64-channel form spilled)
    float* const sbias = (float*)(stg + 16384);
    if (tid < 64) sbias[tid] = a.bias ? a.bias[n0 + tid] : 0.f;

    // ---- halo DMA, as in tapgemm_wreg_f32_kernel: item (c, ri) = chunk c, halo rows [16 ri, 16 ri + 16) of the LDS image; wave w owns row
    // items w and w + 8 of every chunk; per lane and row item one pixel offset and five edge bits, per patch the origin and four edge bits
    // (Per-lane constants of the EPILOGUE are re-formed per patch from the lane id -- `asm volatile("" : "+v")` keeps hipcc from hoisting
    // them -- instead of living in registers across the MFMA loop: with 72 weight registers the 64-channel form has few to spare.  The
    // kernel is bound by the SIMDs' vector issue (phase stamps, tools/probes/wreg16_stamps.py: a v_mfma_f32_16x16x32_bf16 holds the issue
    // port for 8 of its 16 cycles, every other vector instruction for 4), so what IS kept in registers is chosen by instructions saved.)
    constexpr int NR = (NIT + 7) / 8;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.xbytes, 0x00020000);
    const unsigned pixb = (unsigned)a.ldx * 2u;
    const int drow = lane >> 2, dq = lane & 3;
    const unsigned swb = (unsigned)(((dq - (drow >> 1)) & 3) << 4);        // LDS chunk dq of row R holds channel chunk (dq - (R >> 1)) & 3
    unsigned po[NR], bm = 0;
#pragma unroll
    for (int jr = 0; jr < NR; ++jr) {
        const int ri = wave + 8 * jr;
        const int hrow = 16 * ri + drow;
        const int hr = hrow / HP, hc = hrow - hr * HP;
        po[jr] = (unsigned)(hr * a.wi + hc) * pixb + swb;
        const unsigned bits = (ri >= NIT || hr >= PH + 2 || hc >= HC) ? 16u : (hr == 0 ? 1u : 0u) | (hr == PH + 1 ? 2u : 0u) | (hc == 0 ? 4u : 0u) | (hc == HC - 1 ? 8u : 0u);
        bm |= bits << (5 * jr);
    }
    auto dma = [&](int q, int buf) {
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
        float* dst = smem + buf * ABUF;
        const unsigned edges = 16u | (y0 == 0 ? 1u : 0u) | (y0 + PH == a.hi ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + 16 == a.wi ? 8u : 0u);
        const unsigned baseb = (unsigned)((img * a.hi + y0 - 1) * a.wi + x0 - 1) * pixb;          // byte offset of halo (0, 0); may wrap below zero
#pragma unroll
        for (int jr = 0; jr < NR; ++jr) {
            const int ri = wave + 8 * jr;                // wave-uniform
            if (jr < NR - 1 || ri < NIT) {
                const bool out = (bm & (edges << (5 * jr))) != 0;
                const unsigned o1 = po[jr] + baseb;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const unsigned off = out ? 0xffffffffu : o1 + (unsigned)(c * 64);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst + (c * NIT + ri) * 256), 16, (int)off, 0, 0, 0);
                }
            }
        }
    };

    // ---- fragment addresses (floats): [column shift][row parity]
    int fc[3][2];
#pragma unroll
    for (int cs = 0; cs < 3; ++cs)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int hrow = (4 * wm + par) * HP + l15 + cs;
            fc[cs][par] = hrow * 16 + (((lq + (hrow >> 1)) & 3) << 2);
        }
    const bool part0 = n0 < a.n1;                        // block-uniform: the 64 channels lie in one output part (checked by the launcher)
    const __amdgpu_buffer_rsrc_t rsy = part0 ? __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc(a.y2, 0, a.y2bytes, 0x00020000);
    const unsigned ldyb = (unsigned)(part0 ? a.ldy : a.ldy2) * 2u;
    const unsigned ycol0 = (unsigned)(part0 ? n0 : n0 - a.n1) * 2u;

    // InstanceNorm statistics on the matrix pipe (header comment).  Wave w owns channel group cg = w & 3 over the patch's 128 pixels, the
    // squares if w < 4, the sums otherwise; transposed-read addresses: lane 4 q + p of group g supplies pixel P = 4 g + q (+ 16, + 32, ...
    // + 112 for the other seven reads: the same swizzle, so one address register and immediates), 8-byte piece p of the group's 32 bytes.
    const int cg = wave & 3;
    const bool sq = wave < 4;                            // wave-uniform
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
    int simg = q0 / ppi;
    auto flush = [&](int img) {
        // D[row 4 lq + r][col l15]: the diagonal element (sum y^2; in the ones product: sum y) of channel l15 of the group sits in register
        // l15 & 3 of the lanes with (l15 >> 2) == lq
        const int r = l15 & 3;
        const float sv = r == 0 ? S[0] : r == 1 ? S[1] : r == 2 ? S[2] : S[3];
        if ((l15 >> 2) == lq)
            atomicAdd(a.stats + (size_t)(blockIdx.x % a.stats_slots) * a.stats_stride + ((size_t)img * a.nout + n0 + 16 * cg + l15) * 2 + (sq ? 1 : 0), (double)sv);
        S = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // timing-only build (abl::stamp): cycles between the phase boundaries of a patch, summed over the block's patches, per wave
    [[maybe_unused]] unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int k) {
        if constexpr (abl::stamp) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (k >= 0) tph[k] += t - tlast;
            tlast = t;
        }
    };
    dma(q0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // waves 4-7 of a workgroup lose the SIMD's issue arbitration to waves 0-3 on every phase (stamps: MFMA loop 4100-4200 cycles against
    // 3200, and waves 0-3 then wait 1900 cycles at the staging barrier): one static priority step for that half (guide, "static priority")
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    for (int q = q0; q < q1; ++q) {
        const int buf = (q - q0) & 1;
        stamp(-1);
        SHM_LDS_BARRIER();                   // halo(q) landed for every wave; the other halo buffer and the staging image are free
        asm volatile("" ::: "memory");
        stamp(0);                            // [0] wait at the top barrier
        if constexpr (!abl::nodma)
            if (q + 1 < q1) dma(q + 1, buf ^ 1);
        stamp(1);                            // [1] halo DMA issue

        f32x4 acc[4];
        {
            const f32x4 bias4 = *(const f32x4*)(sbias + wn * 16 + lq * 4);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = bias4;
        }
        const float* Ab = smem + buf * ABUF;
        f32x4 abl_frag = {0.f, 0.f, 0.f, 0.f};
        if constexpr (abl::nolds) abl_frag = *(const f32x4*)(Ab + lane * 4);        // timing only: one read per patch
        auto frag = [&](int s) {                        // step s = (halo row r, column shift cs, chunk c), c fastest
            const int c = s % NCH, cs = (s / NCH) % 3, r = s / (3 * NCH);
            if constexpr (abl::nolds) return abl_frag;
            else return *(const f32x4*)(Ab + c * ASTG + fc[cs][r & 1] + (r & ~1) * HP * 16);
        };
        constexpr int NS = 6 * 3 * NCH;
        f32x4 cur = frag(0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f32x4 nxt = frag(s + 1 < NS ? s + 1 : s);
            const int c = s % NCH, cs = (s / NCH) % 3, r = s / (3 * NCH);
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (r - m >= 0 && r - m <= 2) {
                    if constexpr (abl::nomfma)
                        asm volatile("" ::"v"(cur), "v"(bw[(r - m) * 3 + cs][c]));      // timing only
                    else
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[(r - m) * 3 + cs][c]), __builtin_bit_cast(bf16x8, cur), acc[m], 0, 0, 0);
                }
            asm volatile("" ::: "memory");
            cur = nxt;
        }

        if constexpr (abl::stamp) asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
        stamp(2);                            // [2] MFMA loop (fragment reads + MFMA issue)
        // ---- epilogue of patch q: register r of tile m = channel cb + r of pixel (row 4 wm + m, column l15)
        const int img = q / ppi, prem = q - img * ppi;
        const int y0 = (prem / ppr) * PH, x0 = (prem % ppr) << 4;
        // output staging: pixel p = (4 wm + m) * 16 + l15 of the patch, 16-byte chunk 2 wn + (lq >> 1) of its 128-byte row, XOR p & 7
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int e15 = ln & 15, eq = ln >> 4;
        const int st_w = (4 * wm * 16 + e15) * 128 + (((2 * wn + (eq >> 1)) ^ (e15 & 7)) << 4) + ((eq & 1) << 3);      // + m * 2048
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // two values -> one dword of two bf16 in ONE v_cvt_pk_bf16_f32 (the vector conversion; cast one by one, hipcc converts each with a
            // dummy partner and merges the halves with a third instruction).  Compiler-visible on purpose: an inline-asm conversion reading
            // the MFMA accumulators gets no MFMA-result wait states from hipcc (advisor finding, round 4).  A NaN stays a NaN.
            f32x2 lo = {acc[m][0], acc[m][1]}, hi = {acc[m][2], acc[m][3]};
            if constexpr (EPI) {             // LeakyReLU, 0 <= slope <= 1 (launcher): two instructions per value
                lo = f32x2{shm_lrelu_max(lo.x, a.slope), shm_lrelu_max(lo.y, a.slope)};
                hi = f32x2{shm_lrelu_max(hi.x, a.slope), shm_lrelu_max(hi.y, a.slope)};
            }
            const unsigned pk0 = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_t));
            const unsigned pk1 = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_t));
            if constexpr (!abl::nostore) *(uint2*)(stg + st_w + m * 2048) = make_uint2(pk0, pk1);
            else asm volatile("" ::"v"(pk0), "v"(pk1));
        }
        stamp(3);                            // [3] epilogue arithmetic + staging writes
        if constexpr (!abl::nostore) {
            SHM_LDS_BARRIER();               // the staging image of patch q is complete
            asm volatile("" ::: "memory");
            stamp(4);                        // [4] wait at the staging barrier
            const int st_r = (16 * wave + (ln >> 3)) * 128 + (((ln & 7) ^ (ln >> 3)) << 4);                            // + j * 1024
            const unsigned yo = (unsigned)((img * a.hi + y0 + wave) * a.wi + x0 + (ln >> 3)) * ldyb + ycol0 + (unsigned)(ln & 7) * 16u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 v = *(const u32x4*)(stg + st_r + j * 1024);
                // aux 2 = nt: 1 KiB of whole lines per instruction, read next by another kernel (round 4 A/B: the block 257 -> 242 us, step -0.09 ms;
                // the same hint on the halo kernels' 64-byte pieces and on the elementwise kernels' stores changed nothing)
                __builtin_amdgcn_raw_buffer_store_b128(v, rsy, yo, (unsigned)(8 * j) * ldyb, 2);
            }
            if constexpr (EPI)
                if (a.stats) {               // block-uniform
                    if (img != simg) {
                        flush(simg);
                        simg = img;
                    }
                    const int pix = ln >> 2, tp = ln & 3;          // 4 g + q
                    const unsigned short* tb = (const unsigned short*)(stg + pix * 128 + (((2 * cg + (tp >> 1)) ^ (pix & 7)) << 4) + ((tp & 1) << 3));
                    bf16x8 F[4];
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const s16x4_t f0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tb + f * 2048));            // pixels 32 f + 4 g + q
                        const s16x4_t f1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(tb + f * 2048 + 1024));     // + 16
                        F[f] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(f0, f1, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                    if (sq) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[f], F[f], S, 0, 0, 0);
                    } else {
                        const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
#pragma unroll
                        for (int f = 0; f < 4; ++f) S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[f], ones, S, 0, 0, 0);
                    }
                }
            stamp(5);                        // [5] staging reads + store issue (+ statistics fragments and MFMAs)
            // halo(q + 1) was issued at the top of this patch; younger: the two stores (and the rare flush, which only makes the wait stricter)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            stamp(6);                        // [6] wait for the next halo
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // timing only
        }
    }
    if constexpr (EPI)
        if (a.stats) flush(simg);
    if constexpr (abl::stamp)
        if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && lane == 0 && a.bias) {
            unsigned* dump = (unsigned*)a.bias;
#pragma unroll
            for (int k = 0; k < 7; ++k) dump[wave * 8 + k] = (unsigned)tph[k];
            dump[wave * 8 + 7] = (unsigned)(q1 - q0);
        }
}


int shm_wreg16_launch(const TapGemmArgs& a, int np8, int ncu, hipStream_t st, const char* who) {
    const int nyw = a.nout / 64;
    int gxw = 2 * ncu / nyw;            // two eight-wave blocks per CU (68 KiB of LDS, 128 VGPRs each)
    if (gxw < 1) gxw = 1;
    if (gxw > np8) gxw = np8;
    static const hipError_t at16 = [] {
        hipError_t e = hipFuncSetAttribute((const void*)tapgemm_wreg16_bf16_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, w16_lds_bytes(2));
        return e != hipSuccess ? e : hipFuncSetAttribute((const void*)tapgemm_wreg16_bf16_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, w16_lds_bytes(2));
    }();
    SHM_REQUIRE(at16 == hipSuccess, SHM_E_HIP, "%s: cannot reserve %u bytes of LDS: %s", who, w16_lds_bytes(2), hipGetErrorString(at16));
    // EPI = false: no activation and no statistics (the input-gradient launches)
    const bool epi = a.slope != 1.f || a.stats != nullptr;
    if (a.K == 64 && epi)
        hipLaunchKernelGGL((tapgemm_wreg16_bf16_kernel<2, true>), dim3(gxw, nyw, 1), dim3(512), w16_lds_bytes(2), st, a, np8);
    else if (a.K == 64)
        hipLaunchKernelGGL((tapgemm_wreg16_bf16_kernel<2, false>), dim3(gxw, nyw, 1), dim3(512), w16_lds_bytes(2), st, a, np8);
    else if (epi)
        hipLaunchKernelGGL((tapgemm_wreg16_bf16_kernel<1, true>), dim3(gxw, nyw, 1), dim3(512), w16_lds_bytes(1), st, a, np8);
    else
        hipLaunchKernelGGL((tapgemm_wreg16_bf16_kernel<1, false>), dim3(gxw, nyw, 1), dim3(512), w16_lds_bytes(1), st, a, np8);
    shm_set_last_kernel("tapgemm_wreg16_bf16_kernel<%d, %s>", a.K / 32, epi ? "true" : "false");
    return SHM_OK;
}
