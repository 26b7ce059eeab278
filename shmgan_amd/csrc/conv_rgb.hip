// The discriminator's first layer -- Conv2D(3x3, stride 2, 'same', no bias) + LeakyReLU on a 3-channel image, Keras `Conv_LReLU_IN`
// (/root/reference/ShmGANwithSSpecSeg.py:353, 386-389) -- and its weight gradient, on a COMPACT image layout: one 16-byte chunk per pixel
// (float32: r, g, b, 0; bfloat16: r, g, b and five zeros) instead of the 64-byte MFMA staging row the tap-GEMM kernels read.
//
// Why (round 4): through the generic kernels the layer multiplied 3 real channels inside a 16- / 32-channel K block (17-19 TFLOP/s on the
// MFMA, 290-320 us per step and dtype) and the padded image tensor was 403 MB written by the colour kernels and read twice (forward, weight
// gradient) for 38-75 MB of pixels.  Both products are tiny (5.4 GFLOP at 96 images of 256 x 256): what they need is to stream.
//
// Forward (transposed product, weights = A operand: a lane ends up with four consecutive output channels of one pixel): the image arrives as
// 16-byte pixels, one window tap per lane quarter, straight into the B operands -- bfloat16 three v_mfma_f32_16x16x32_bf16 per 16 pixels and 16
// output channels (K slot = (tap, channel 0..7)), float32 seven v_mfma_f32_16x16x4_f32 (K slot = one channel of a lane quarter's tap); a wave
// walks a run of pixel groups of ONE image behind a per-image buffer descriptor (the padding row is out of range by itself: no branch in the
// loop), fragments a group or two ahead in a register ring; the tile goes through a wave-private LDS stage and out as linear, non-temporal 1 KiB
// stores (64- / 32-byte pieces per pixel and instruction held the kernel at 2.4 TB/s: LABNOTES 10.7); InstanceNorm sums per lane in float,
// one f64 atomic per (wave, channel, moment).  n = 96 at 256 x 256: 97 us fp32, 59 us bf16 (generic kernels on the staging layout: 331 / 276).
// Weight gradient: pixels are the contraction index; float32 MFMA for both dtypes (bf16 values are exact in fp32): rows = the 27 (tap, ci)
// pairs in two 16-row tiles, columns = output channels in the permutation co = 4 * lane + tile so that a lane's 16-byte dz load feeds the four
// column tiles; the window strip of a run of 64 output pixels is copied to LDS with linear loads and the A operands are picked from there;
// split over blocks into the slabs shm_conv2d_wgrad_reduce sums.  116 us fp32, 87 us bf16 with the reduction (generic: 161 / 256).
#include "common.h"
#include "ablate.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct RgbFwdArgs {
    const void* x;
    const void* wk;          // [9][cout][K] (the layer's K-contiguous weight copy; K = 16 fp32 / 32 bf16, channels >= 3 are zero)
    const float* bias;       // [cout] or null
    void* y;
    int ldx, K, ldy, batch, hi, wi, ho, wo, cout;
    float slope;
    double* stats;           // optional [slot][batch][cout][2]
    int stats_slots;
    unsigned stats_stride;
    unsigned ybytes;
    int groups_per_wave;     // divides the groups of an image: a wave's run lies in one image
};

__device__ __forceinline__ float rgb_row16_sum(float v) {           // sum over the 16 lanes of a DPP row, in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));       // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));       // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));      // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));      // row_mirror
    return v;
}

// NT = cout / 16 column tiles (1..4); PD = groups whose image fragments are in flight ahead of the one being multiplied.
// A wave walks `groups_per_wave` runs of 16 consecutive output pixels of ONE image.  Its buffer descriptor covers that image alone, so the
// window row below the image (2 oh + 2 == hi: SAME padding of an even map puts one row after, none before) is out of range by itself
// and reads 0; the column right of the image is sent out of range by a per-lane select.  No branch in the loop: the image loads of the
// next groups, the MFMAs and the stores overlap (a first version with `cond ? offset : ~0u` was compiled into scalar branches with
// s_waitcnt vmcnt(0) inside and ran at 2 TB/s).
template <typename T, int NT, bool HAS_BIAS, int PD, int UNR, bool STAGED>
__global__ __launch_bounds__(256) void conv3x3s2_rgb_fwd_kernel(const RgbFwdArgs a) {
    constexpr int ESZ = sizeof(T);
    // The image arrives as 16-byte pixels, one window tap per lane quarter: load g (of NL) brings tap 4 g + lq.
    // bfloat16: three loads (taps 0-3, 4-7, 8 and three dead quarters) are the B operands of three v_mfma_f32_16x16x32_bf16 as they come (K slot =
    // (tap, channel 0..7)).  float32: seven v_mfma_f32_16x16x4_f32 -- K block m < 6 = channel m % 3 of load m / 3 (K slot lq = tap 4 (m / 3) + lq: a
    // component of the loaded pixel, no shuffling), K block 6 = the three channels of tap 8 (slot lq = channel lq), fetched as one dword per lane.
    // (A first version gathered all 27 (tap, channel) pairs as dwords: ~37 cycles of the texture path per wave-load, 41 us of 114.)
    constexpr int NM = ESZ == 4 ? 7 : 3, NL = ESZ == 4 ? 2 : 3;
    constexpr unsigned OOB = 0x80000000u;                // added to an offset: past any image (the launcher keeps images below 2 GiB)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int gpr = a.wo >> 4, gpi = a.ho * gpr, rpi = gpi / a.groups_per_wave;
    const int wid = blockIdx.x * 4 + wave;
    if (wid >= a.batch * rpi) return;
    const int img = wid / rpi, gl0 = (wid - img * rpi) * a.groups_per_wave;

    // ---- weights -> registers (A operand: row = output channel 16 j + l15, k = lq of the K block); this lane's window displacements
    f32x4 wa4[ESZ == 2 ? NM * NT : 1];
    float wa1[ESZ == 4 ? NM * NT : 1];
    unsigned cin_[NL], cedge[NL];                        // byte displacement of this lane's tap of load g from the window's first pixel
#pragma unroll
    for (int g = 0; g < NL; ++g) {
        const int tap = 4 * g + lq;
        const bool live = tap < 9;
        const int kh = tap / 3, kw = tap - 3 * kh;
        cin_[g] = live ? (unsigned)((kh * a.wi + kw) * 16) : OOB;
        cedge[g] = (live && kw < 2) ? cin_[g] : OOB;     // for the pixel in the last column of the map
    }
    const unsigned c8 = (unsigned)((2 * a.wi + 2) * 16 + 4 * lq);          // float32: channel lq of tap 8 (out of the map for the last column)
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = 16 * j + l15;
            if constexpr (ESZ == 4) {
                const int tap = m < 6 ? 4 * (m / 3) + lq : 8, ch = m < 6 ? m % 3 : lq;      // (the K-padded weight copy holds zeros in channel 3)
                wa1[m * NT + j] = ((const float*)a.wk)[((size_t)tap * a.cout + co) * a.K + ch];
            } else {
                const int tap = 4 * m + lq;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (tap < 9) v = *(const f32x4*)((const bf16_t*)a.wk + ((size_t)tap * a.cout + co) * a.K);
                wa4[m * NT + j] = v;
            }
        }
    f32x4 bias4[HAS_BIAS ? NT : 1];
    if constexpr (HAS_BIAS) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[j][r] = a.bias[16 * j + 4 * lq + r];
    }
    const unsigned imgbytes = (unsigned)a.hi * a.wi * 16u;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((char*)a.x + (size_t)img * imgbytes, 0, imgbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.ybytes, 0x00020000);
    const unsigned lane_x = (unsigned)l15 * 32u;                                              // pixel 2 (owb + l15) of the row
    const unsigned lane_y = ((unsigned)l15 * a.ldy + 4u * lq) * ESZ;

    // the walk over the run: (oh, owb) of group gl, advanced without divisions
    struct Pos {
        int oh, owb;
    };
    auto advance = [&](Pos& p) {
        p.owb += 16;
        if (p.owb == a.wo) {
            p.owb = 0;
            ++p.oh;
        }
    };
    // image fragment of a group: fp32 -- seven floats; bf16 -- three 16-byte chunks
    struct Frag {
        u32x4 b[NL];
        float f8;
    };
    auto load_x = [&](const Pos& p, Frag& fr) {
        const unsigned base = (unsigned)(2 * p.oh * a.wi + 2 * p.owb) * 16u + lane_x;
        const bool edge = p.owb + l15 + 1 == a.wo;       // per lane: v_cndmask, not a branch
#pragma unroll
        for (int g = 0; g < NL; ++g) {
            const unsigned off = abl::noload ? OOB : base + (edge ? cedge[g] : cin_[g]);
            fr.b[g] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)off, 0, 0);
        }
        if constexpr (ESZ == 4) {
            const unsigned off = abl::noload ? OOB : base + (edge ? OOB : c8);
            fr.f8 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, (int)off, 0, 0));
        }
    };

    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[j][r] = s2[j][r] = 0.f;
    // STAGED (ldy == cout): a pixel's NT * 16 channels are PIXB contiguous bytes, staged at a pitch of PIXB + 16 (the 16 lanes of
    // a write pass -- one lq, 16 pixels -- then fall into 16 different bank quads)
    constexpr int PIXB = NT * 16 * ESZ, PITCH = PIXB + 16;
    __shared__ __attribute__((aligned(16))) char stage_all[STAGED ? 4 * 16 * PITCH : 16];
    char* const stg = stage_all + (STAGED ? wave * 16 * PITCH : 0);

    Pos pl;                                              // position of the next group to load
    pl.oh = gl0 / gpr;
    pl.owb = (gl0 - pl.oh * gpr) << 4;
    Pos pc = pl;                                         // position of the group being computed
    static_assert(UNR > PD, "ring");                     // the loop is unrolled over the fragment ring (no register copies); groups_per_wave % UNR == 0
    Frag fr[UNR];
    const int ng = a.groups_per_wave;
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        load_x(pl, fr[d]);
        advance(pl);
    }
    // (entering the loop with the first fragments landed: otherwise hipcc merges "pending, nothing younger" from here with "pending, four stores
    // younger" from the back edge into s_waitcnt vmcnt(0) at the top of the loop -- a drain of the stores in every round)
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
    auto compute = [&](const Frag& f) {
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if constexpr (HAS_BIAS) acc[j] = bias4[j];
            else acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (ESZ == 4) {
            float bop[NM];                               // the B operands: the three channels of the two loaded pixels, then tap 8's
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const u32x4 px = f.b[g];
                bop[3 * g] = __uint_as_float(px.x);
                bop[3 * g + 1] = __uint_as_float(px.y);
                bop[3 * g + 2] = __uint_as_float(px.z);
            }
            bop[6] = f.f8;
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa1[m * NT + j], bop[m], acc[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa4[m * NT + j]), __builtin_bit_cast(bf16x8, f.b[m]), acc[j], 0, 0, 0);
        }
        const unsigned yg = (unsigned)((img * a.ho + pc.oh) * a.wo + pc.owb) * (unsigned)a.ldy * ESZ;       // the group's first pixel
        const unsigned yo = yg + lane_y;
        advance(pc);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // (the tile's offset in the instruction's immediate field, NOT as an SGPR soffset: see shm_lrelu_max in common.h)
            if constexpr (ESZ == 4) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = shm_lrelu_max(acc[j][r], a.slope);
                    s1[j][r] += v[r];
                    s2[j][r] = __builtin_fmaf(v[r], v[r], s2[j][r]);
                }
                if constexpr (STAGED) *(f32x4*)(stg + l15 * PITCH + 64 * j + 16 * lq) = v;
                else if constexpr (!abl::nostore)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsy, yo + (unsigned)(16 * j) * ESZ, 0, 0);
            } else {
                unsigned pk[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float u0 = shm_lrelu_max(acc[j][2 * h], a.slope), u1 = shm_lrelu_max(acc[j][2 * h + 1], a.slope);
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
                    pk[h] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_{u0, u1}), bf16x2_));           // one v_cvt_pk_bf16_f32
                    const float v0 = __uint_as_float(pk[h] << 16), v1 = __uint_as_float(pk[h] & 0xffff0000u);       // the values as stored
                    s1[j][2 * h] += v0;
                    s2[j][2 * h] = __builtin_fmaf(v0, v0, s2[j][2 * h]);
                    s1[j][2 * h + 1] += v1;
                    s2[j][2 * h + 1] = __builtin_fmaf(v1, v1, s2[j][2 * h + 1]);
                }
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                if constexpr (STAGED) *(u32x2*)(stg + l15 * PITCH + 32 * j + 8 * lq) = u32x2{pk[0], pk[1]};
                else if constexpr (!abl::nostore)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk[0], pk[1]}, rsy, yo + (unsigned)(16 * j) * ESZ, 0, 0);
            }
        }
        if constexpr (STAGED) {
            // the group's 16 pixels are PIXB * 16 contiguous bytes of y: back out of the wave's staging rows in linear order, 1 KiB per store
            // instruction (same wave, LDS operations complete in order: no barrier).  Partial-line stores -- 64 (fp32) or 32 (bf16) bytes of
            // each pixel per instruction -- ran the kernel at 2.4 TB/s; see LABNOTES 10.7
            constexpr int TOTAL = 16 * PIXB;
#pragma unroll
            for (int i = 0; i < (TOTAL + 1023) / 1024; ++i) {
                const int off = i * 1024 + lane * 16;
                if (TOTAL % 1024 == 0 || off < TOTAL) {
                    const u32x4 v = *(const u32x4*)(stg + (off / PIXB) * PITCH + off % PIXB);
                    if constexpr (!abl::nostore) __builtin_amdgcn_raw_buffer_store_b128(v, rsy, yg + (unsigned)off, 0, 2);      // aux 2 = nt: written once, read by the next kernel
                }
            }
        }
    };
    for (int g = 0; g < ng; g += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            // (past the run's end the loads still go out -- inside the image or out of range, both harmless -- so that the loop has no branch)
            load_x(pl, fr[(u + PD) % UNR]);
            advance(pl);
            compute(fr[u]);
        }
    }
    if (a.stats) {
        double* base = a.stats + (size_t)(wid % a.stats_slots) * a.stats_stride + ((size_t)img * a.cout + 4 * lq) * 2;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float mine = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t1 = rgb_row16_sum(s1[j][r]), t2 = rgb_row16_sum(s2[j][r]);
                mine = l15 == 2 * r ? t1 : l15 == 2 * r + 1 ? t2 : mine;
            }
            if (l15 < 8) atomicAdd(base + (size_t)16 * j * 2 + l15, (double)mine);       // channels 16 j + 4 lq + (l15 >> 1), sum / sum of squares
        }
    }
}

// returns 1 = launched, 0 = this shape is not this kernel's (the caller goes on to the generic path), < 0 = error
int shm_rgb_s2_fwd_launch(const void* x, int ldx, const void* wk, int K, const float* bias, void* y, int ldy, int batch, int hi, int wi, int cout, float slope,
                          double* stats, int stats_slots, unsigned stats_stride, size_t xbytes, size_t ybytes, int dtype, hipStream_t st) {
    const int esz = dtype == SHM_BF16 ? 2 : 4;
    if (dtype != SHM_F32 && dtype != SHM_BF16) return 0;
    if (ldx * esz != 16 || K * esz != 64 || hi % 2 || wi % 2 || (wi / 2) % 16 || cout % 16 || cout > 64 || cout < 16) return 0;
    if (ldy % (16 / esz) || ((size_t)y & 15) || ((size_t)x & 15) || ((size_t)wk & 15)) return 0;
    if (!(slope >= 0.f && slope <= 1.f) || (size_t)hi * wi * 16 >= 0x7fffff00ull || ybytes >= 0xfffffff0ull) return 0;
    if (batch == 0) return 1;
    RgbFwdArgs a{};
    a.x = x;
    a.wk = wk;
    a.bias = bias;
    a.y = y;
    a.ldx = ldx;
    a.K = K;
    a.ldy = ldy;
    a.batch = batch;
    a.hi = hi;
    a.wi = wi;
    a.ho = hi / 2;
    a.wo = wi / 2;
    a.cout = cout;
    a.slope = slope;
    a.stats = stats;
    a.stats_slots = stats_slots < 1 ? 1 : stats_slots;
    a.stats_stride = stats_stride;
    a.ybytes = (unsigned)ybytes;
    // a wave walks a run of 16-pixel groups of one image: the largest power of two up to 32 that divides the groups of an image and leaves
    // ~1024 blocks (four per CU)
    const int gpi = a.ho * (a.wo / 16);
    const int nt = cout / 16;
    // prefetch distance / ring: float32 one group ahead (2 slots), bfloat16 two (4 slots) -- round-4 sweep at n = 96, 256 x 256: fp32 99 / 100 / 96 us
    // for 1 / 2 / 3 ahead, bf16 63 / 60 / 61
    const int unr = dtype == SHM_F32 ? 2 : 4;
    if (gpi % unr) return 0;
    // a wave walks a run of 16-pixel groups of one image: the largest power of two up to 32 that divides the groups of an image and leaves
    // 4096 waves (four blocks per CU)
    int gpw = unr;
    while (gpw < 32 && gpi % (2 * gpw) == 0 && (long)batch * (gpi / (2 * gpw)) >= 4096) gpw *= 2;
    a.groups_per_wave = gpw;
    const long waves = (long)batch * (gpi / gpw);
    const dim3 grid((unsigned)((waves + 3) / 4));
    const bool staged = ldy == cout;                      // a group's 16 pixels are contiguous in y: linear stores out of LDS
#define SHM_RGB_FWD4(T_, NT_, B_, PD_, UNR_)                                                                                  \
    do {                                                                                                                      \
        if (staged) hipLaunchKernelGGL((conv3x3s2_rgb_fwd_kernel<T_, NT_, B_, PD_, UNR_, true>), grid, dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((conv3x3s2_rgb_fwd_kernel<T_, NT_, B_, PD_, UNR_, false>), grid, dim3(256), 0, st, a);        \
    } while (0)
#define SHM_RGB_FWD3(T_, NT_, PD_, UNR_)                  \
    do {                                                  \
        if (bias) SHM_RGB_FWD4(T_, NT_, true, PD_, UNR_); \
        else SHM_RGB_FWD4(T_, NT_, false, PD_, UNR_);     \
    } while (0)
    if (dtype == SHM_F32) {
        if (nt == 4) SHM_RGB_FWD3(float, 4, 1, 2);
        else if (nt == 3) SHM_RGB_FWD3(float, 3, 1, 2);
        else if (nt == 2) SHM_RGB_FWD3(float, 2, 1, 2);
        else SHM_RGB_FWD3(float, 1, 1, 2);
    } else {
        if (nt == 4) SHM_RGB_FWD3(bf16_t, 4, 2, 4);
        else if (nt == 3) SHM_RGB_FWD3(bf16_t, 3, 2, 4);
        else if (nt == 2) SHM_RGB_FWD3(bf16_t, 2, 2, 4);
        else SHM_RGB_FWD3(bf16_t, 1, 2, 4);
    }
#undef SHM_RGB_FWD3
#undef SHM_RGB_FWD4
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        shm_set_error("conv3x3s2_rgb_fwd: launch failed: %s", hipGetErrorString(e));
        return SHM_E_HIP;
    }
    shm_set_last_kernel("conv3x3s2_rgb_fwd_kernel<%s, %d>", dtype == SHM_BF16 ? "__bf16" : "float", nt);
    return 1;
}

// ------------------------------------------------------------------------------------------ weight gradient
struct RgbWgradArgs {
    const void* x;
    const void* dy;
    float* part;             // [blocks][9][cin][cout]
    int ldx, lddy, batch, hi, wi, ho, wo, cin, cout;
    int rq;                  // quads (runs of four output pixels) per unit; a unit = rq quads of one output row
    int units, units_per_wave;
    unsigned dybytes;
};

// NT = cout / 16.  Row tile i, lane row l15 = (tap, ci) pair 16 i + l15 of the 9 * cin <= 32; column tile j, lane column l15 = output channel
// 4 * l15 + j (NT == 4; NT * l15 + j in general), so that the lane's four consecutive dz values are the B operands of the four tiles.
// A wave takes `units_per_wave` units.  Per unit it copies the three image rows of the unit's window strip -- 8 rq + 1 sixteen-byte pixels each --
// into its own piece of LDS with linear 16-byte loads, and picks its A operands out of LDS (the first version gathered them from global memory,
// two 4- / 2-byte loads per quad and lane: the texture path spends ~37 cycles on such a wave-load); dz streams through a register ring, three quads
// ahead.  The descriptor of x covers one image, so the row below the image reads 0 by itself; the column right of it is written as zeros.
template <typename T, int NT, int UNR>
__global__ __launch_bounds__(256) void conv3x3s2_rgb_wgrad_kernel(const RgbWgradArgs a) {
    constexpr int ESZ = sizeof(T);
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int W = 8 * a.rq + 1, rowp = (8 * a.rq + 4) * 16;                // strip: pixels per row, bytes per row
    char* const stg = smem + wave * 3 * rowp;
    const int rows = 9 * a.cin, upr = (a.wo >> 2) / a.rq;                   // units per output row
    // this lane's two A rows: LDS byte offset of (kh, kw, ci) from the strip pixel of output pixel 4 q + lq (rows past the 9 * cin: any valid
    // address -- a row of the product depends on its own A row only, and those rows are not stored)
    unsigned aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * i + l15;
        const int tap = row < rows ? row / a.cin : 0, ci = row < rows ? row - tap * a.cin : 0;
        const int kh = tap / 3, kw = tap - 3 * kh;
        aoff[i] = (unsigned)(kh * rowp + (2 * lq + kw) * 16 + ci * ESZ);
    }
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dybytes, 0x00020000);
    const unsigned imgbytes = (unsigned)a.hi * a.wi * 16u;
    const int u0 = min(a.units, (blockIdx.x * 4 + wave) * a.units_per_wave), u1 = min(a.units, u0 + a.units_per_wave);

    struct Dz {
        u32x4 v;             // float32: four channels; bfloat16: .x, .y = four channels
    };
    for (int u = u0; u < u1; ++u) {
        const int rowid = u / upr, run = u - rowid * upr;                   // (img, oh) and the run inside the output row
        const int img = rowid / a.ho, oh = rowid - img * a.ho;
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((char*)a.x + (size_t)img * imgbytes, 0, imgbytes, 0x00020000);
        // dz of quad q: pixel (oh, 4 (run rq + q) + lq), channels NT l15 ...
        const unsigned dz0 = ((unsigned)((img * a.ho + oh) * a.wo + 4 * run * a.rq + lq) * (unsigned)a.lddy + (unsigned)(NT * l15)) * ESZ;
        const unsigned dzq = 4u * (unsigned)a.lddy * ESZ;
        auto load_dz = [&](int q, Dz& d) {
            const unsigned off = dz0 + (unsigned)q * dzq;
            if constexpr (NT == 4) {
                if constexpr (ESZ == 4) d.v = __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)off, 0, 0);
                else {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsd, (int)off, 0, 0);
                    d.v = u32x4{t.x, t.y, 0u, 0u};
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned t = 0u;
                    if (j < NT) {
                        if constexpr (ESZ == 4) t = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsd, (int)(off + 4u * j), 0, 0);
                        else t = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsd, (int)(off + 2u * j), 0, 0) << 16;
                    }
                    d.v[j] = t;
                }
            }
        };
        Dz ring[UNR];                                    // UNR - 1 quads ahead
#pragma unroll
        for (int d = 0; d < UNR - 1; ++d) load_dz(d, ring[d]);
        // the strip: rows 2 oh .. 2 oh + 2, columns 8 run rq .. + 8 rq (the column right of the image: zeros)
        const int c0 = 8 * run * a.rq;
        for (int c = lane; c < W; c += 64) {
            const unsigned colb = (c0 + c < a.wi) ? (unsigned)(c0 + c) * 16u : OOB;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const u32x4 px = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)((unsigned)((2 * oh + r) * a.wi) * 16u + colb), 0, 0);
                *(u32x4*)(stg + r * rowp + c * 16) = px;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the loop starts with nothing pending (see the forward kernel)
        auto compute = [&](int q, const Dz& d) {
            float xa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* p = stg + aoff[i] + q * 128;
                if constexpr (ESZ == 4) xa[i] = *(const float*)p;
                else xa[i] = __uint_as_float((unsigned)*(const unsigned short*)p << 16);
            }
            float dzf[4];
            if constexpr (NT == 4 && ESZ == 2) {
                dzf[0] = __uint_as_float(d.v.x << 16);
                dzf[1] = __uint_as_float(d.v.x & 0xffff0000u);
                dzf[2] = __uint_as_float(d.v.y << 16);
                dzf[3] = __uint_as_float(d.v.y & 0xffff0000u);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dzf[j] = __uint_as_float(d.v[j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], dzf[j], acc[i][j], 0, 0, 0);
        };
        for (int q = 0; q < a.rq; q += UNR) {            // rq % UNR == 0
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                load_dz(q + k + UNR - 1, ring[(k + UNR - 1) % UNR]);       // (past the unit's end: the next pixels of dz, or out of range -- loaded, never used)
                compute(q + k, ring[k]);
            }
        }
    }
    // the block's slab = the sum of its four waves' tiles in a fixed order, (w0 + w2) + (w1 + w3): accumulator register r of tile (i, j) = row
    // 16 i + 4 lq + r, column l15 -> channel NT * l15 + j.  Two steps through a two-wave buffer (half the LDS: it fits in the strips' space)
    __syncthreads();                                     // the strips are done with: their LDS is the reduction buffer now
    float (*red)[32][16 * NT + 1] = (float (*)[32][16 * NT + 1])smem;
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave - 2][16 * i + 4 * lq + r][NT * l15 + j] = acc[i][j][r];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][16 * i + 4 * lq + r][NT * l15 + j] += acc[i][j][r];
    }
    __syncthreads();
    float* out = a.part + (size_t)blockIdx.x * rows * a.cout;
    for (int e = threadIdx.x; e < rows * a.cout; e += 256) {
        const int row = e / a.cout, co = e - row * a.cout;
        out[e] = red[0][row][co] + red[1][row][co];
    }
}

// returns 1 = launched (*nsplit_out slabs written), 0 = not this kernel's shape, < 0 = error
int shm_rgb_s2_wgrad_launch(const void* x, int ldx, const void* dy, int lddy, float* part, size_t ws_bytes, int batch, int hi, int wi, int cin, int cout,
                            size_t xbytes, size_t dybytes, int dtype, int* nsplit_out, hipStream_t st) {
    const int esz = dtype == SHM_BF16 ? 2 : 4;
    if (dtype != SHM_F32 && dtype != SHM_BF16) return 0;
    if (ldx * esz != 16 || 9 * cin > 32 || cin < 1 || hi % 2 || wi % 2 || (wi / 2) % 16 || cout % 16 || cout > 64 || cout < 16) return 0;
    if (lddy % (16 / esz) || ((size_t)dy & 15) || ((size_t)x & 15) || (size_t)hi * wi * 16 >= 0x7fffff00ull || dybytes >= 0xfffffff0ull) return 0;
    if (batch == 0) return 0;
    const int ho = hi / 2, wo = wi / 2, qpr = wo / 4;
    int rq = qpr;                                         // quads per unit: at most 16 (a 6 KiB strip per wave), a multiple of 4 that divides the row
    while (rq > 16 && rq % 2 == 0) rq /= 2;
    if (rq > 16 || rq % 4) return 0;
    const long units = (long)batch * ho * (qpr / rq);
    // one slab per block of four waves: as many blocks as the workspace allows, at most 2048 (eight per CU), at least two units per wave
    const size_t slab = (size_t)9 * cin * cout * sizeof(float);
    long nblk_l = (long)(ws_bytes / slab);
    if (nblk_l > 2048) nblk_l = 2048;
    if (nblk_l > units / 8) nblk_l = units / 8 > 0 ? units / 8 : 1;
    if (nblk_l < 1) return 0;
    const long upw = (units + 4 * nblk_l - 1) / (4 * nblk_l);
    const int nblk = (int)((units + 4 * upw - 1) / (4 * upw));
    if ((size_t)nblk * slab > ws_bytes) return 0;
    RgbWgradArgs a{};
    a.x = x;
    a.dy = dy;
    a.part = part;
    a.ldx = ldx;
    a.lddy = lddy;
    a.batch = batch;
    a.hi = hi;
    a.wi = wi;
    a.ho = ho;
    a.wo = wo;
    a.cin = cin;
    a.cout = cout;
    a.rq = rq;
    a.units = (int)units;
    a.units_per_wave = (int)upw;
    a.dybytes = (unsigned)dybytes;
    const int nt = cout / 16;
    const size_t strips = (size_t)4 * 3 * (8 * rq + 4) * 16, redb = (size_t)2 * 32 * (16 * nt + 1) * sizeof(float);
    const size_t lds = strips > redb ? strips : redb;     // <= 25 KiB
    const dim3 grid(nblk);
    // (dz ring: three quads ahead; seven ahead measured no better -- bf16 87.4 vs 88.3 us, fp32 116 vs 124)
#define SHM_RGB_WG(T_, NT_) hipLaunchKernelGGL((conv3x3s2_rgb_wgrad_kernel<T_, NT_, 4>), grid, dim3(256), lds, st, a)
    if (dtype == SHM_F32) {
        if (nt == 4) SHM_RGB_WG(float, 4);
        else if (nt == 3) SHM_RGB_WG(float, 3);
        else if (nt == 2) SHM_RGB_WG(float, 2);
        else SHM_RGB_WG(float, 1);
    } else {
        if (nt == 4) SHM_RGB_WG(bf16_t, 4);
        else if (nt == 3) SHM_RGB_WG(bf16_t, 3);
        else if (nt == 2) SHM_RGB_WG(bf16_t, 2);
        else SHM_RGB_WG(bf16_t, 1);
    }
#undef SHM_RGB_WG
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        shm_set_error("conv3x3s2_rgb_wgrad: launch failed: %s", hipGetErrorString(e));
        return SHM_E_HIP;
    }
    *nsplit_out = nblk;
    shm_set_last_kernel("conv3x3s2_rgb_wgrad_kernel<%s, %d>", dtype == SHM_BF16 ? "__bf16" : "float", nt);
    return 1;
}
