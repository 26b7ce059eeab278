// Convolution weight gradient on v_mfma_f32_32x32x2_f32 (gfx950, exact fp32).
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
#include "common.h"

struct WgradArgs {
    const float* x;
    const float* x2;
    int c1, ldx, ldx2;
    const float* dy;
    int lddy;
    float* part;
    int hi, wi, ho, wo;
    int cin_ld, cin, cout;
    int is, ntaps;
    int dh[9], dw[9];
    int M, pix_per_split;
};

template <int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    constexpr int BKP = 16;
    __shared__ __attribute__((aligned(16))) float Xs[NT][BKP][64];
    __shared__ __attribute__((aligned(16))) float Ds[BKP][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int mi = wave >> 1, ni = wave & 1;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    const int p_begin = blockIdx.z * a.pix_per_split;
    const int p_end = min(a.M, p_begin + a.pix_per_split);

    const int k = tid >> 4, c4 = tid & 15;
    const int c = ci0 + c4 * 4;
    const bool xvalid = c < a.cin_ld;
    const float* src = a.x;
    int ld = a.ldx, cc = c;
    if (c >= a.c1) {
        src = a.x2;
        ld = a.ldx2;
        cc = c - a.c1;
    }
    const int co = co0 + c4 * 4;
    const bool dvalid = co < a.cout;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    f32x4 rx[NT], rd;
    auto gload = [&](int p0) {
        const int p = p0 + k;
        const bool ok = p < p_end;
        const int pp = ok ? p : 0;
        const int ow = pp % a.wo, t2 = pp / a.wo;
        const int oh = t2 % a.ho, n = t2 / a.ho;
        const int ihb = oh * a.is, iwb = ow * a.is;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ih = ihb + a.dh[t], iw = iwb + a.dw[t];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok && xvalid && (unsigned)ih < (unsigned)a.hi && (unsigned)iw < (unsigned)a.wi)
                v = *(const f32x4*)(src + ((size_t)(n * a.hi + ih) * a.wi + iw) * ld + cc);
            rx[t] = v;
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok && dvalid) v = *(const f32x4*)(a.dy + (size_t)p * a.lddy + co);
        rd = v;
    };

    if (p_begin < p_end) gload(p_begin);
    for (int p0 = p_begin; p0 < p_end; p0 += BKP) {
        __syncthreads();   // previous step's LDS reads are done
#pragma unroll
        for (int t = 0; t < NT; ++t) *(f32x4*)(&Xs[t][k][c4 * 4]) = rx[t];
        *(f32x4*)(&Ds[k][c4 * 4]) = rd;
        __syncthreads();
        if (p0 + BKP < p_end) gload(p0 + BKP);
#pragma unroll
        for (int kk = 0; kk < BKP / 2; ++kk) {
            const int kr = 2 * kk + h;
            const float bv = Ds[kr][ni * 32 + l31];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float av = Xs[t][kr][mi * 32 + l31];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }

    // partial slab [split][tap][cin][cout]
    float* out = a.part + (size_t)blockIdx.z * NT * a.cin * a.cout;
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

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, size_t n, int nsplit, int accumulate) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? dw[i] : 0.f;
    for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * n + i];
    dw[i] = s;
}

static int wgrad_splits(int batch, int ho, int wo, int cin, int cout) {
    long M = (long)batch * ho * wo;
    int tiles = shm_cdiv(cin, 64) * shm_cdiv(cout, 64);
    int want = shm_cdiv(1024, tiles);            // ~4 blocks per CU in flight
    long maxs = (M + 255) / 256;                 // at least 256 pixels per split
    if (want > maxs) want = (int)maxs;
    if (want < 1) want = 1;
    return want;
}

extern "C" size_t shm_conv2d_wgrad_workspace(int batch, int ho, int wo, int cin, int cout, int ksize) {
    int ns = wgrad_splits(batch, ho, wo, cin, cout);
    return (size_t)ns * ksize * ksize * cin * cout * sizeof(float);
}

extern "C" int shm_conv2d_wgrad(const float* x, const float* x2, int c1, int ldx, int ldx2, const float* dy,
                                int lddy, float* dw, int batch, int hi, int wi, int cin, int cin_ld,
                                int cout, int ksize, int stride, int accumulate, void* workspace,
                                size_t ws_bytes, void* stream) {
    SHM_REQUIRE(ksize == 1 || ksize == 3, SHM_E_SHAPE, "shm_conv2d_wgrad: ksize %d not in {1,3}", ksize);
    SHM_REQUIRE(stride == 1 || stride == 2, SHM_E_SHAPE, "shm_conv2d_wgrad: stride %d not in {1,2}", stride);
    SHM_REQUIRE(x && dy && dw && workspace, SHM_E_SHAPE, "shm_conv2d_wgrad: null pointer");
    SHM_REQUIRE(cin_ld % 4 == 0 && cin_ld >= cin && cout % 4 == 0, SHM_E_SHAPE,
                "shm_conv2d_wgrad: cin_ld %d / cout %d must be multiples of 4", cin_ld, cout);
    SHM_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && (!x2 || (ldx2 % 4 == 0 && c1 % 4 == 0)), SHM_E_SHAPE,
                "shm_conv2d_wgrad: pitches must be multiples of 4");
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
    int ns = wgrad_splits(batch, ho, wo, cin, cout);
    size_t need = (size_t)ns * a.ntaps * cin * cout * sizeof(float);
    SHM_REQUIRE(ws_bytes >= need, SHM_E_WORKSPACE, "shm_conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    int pps = shm_cdiv(a.M, ns);
    pps = (pps + 15) / 16 * 16;
    ns = shm_cdiv(a.M, pps);
    a.pix_per_split = pps;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(shm_cdiv(cin, 64), shm_cdiv(cout, 64), ns);
    if (ksize == 3)
        hipLaunchKernelGGL((wgrad_kernel<9>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((wgrad_kernel<1>), grid, dim3(256), 0, st, a);
    SHM_LAUNCH_CHECK("shm_conv2d_wgrad");
    size_t n = (size_t)a.ntaps * cin * cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(shm_cdiv((long)n, 256)), dim3(256), 0, st, (const float*)workspace, dw, n, ns, accumulate);
    SHM_LAUNCH_CHECK("shm_conv2d_wgrad(reduce)");
    return SHM_OK;
}
