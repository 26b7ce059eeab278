// Image-space generator losses of SHMGAN's train_step and their gradient wrt the generated
// Y planes (SHM.py:744-826): L1 cycle loss, tf.image.ssim on rescale_01'd YUV (utils.py:190),
// content + gram-matrix style loss.  Everything here is tiny next to the convolutions
// (5B x 3 x S x S values), so the kernels favour clarity: one pixel pass for the sums and
// min/max, a direct 11x11 gaussian from an LDS tile for SSIM forward and backward.
//
// loss slots (f64 sums over the batch):
//   0 L1(gen_rgb, origED)   1..5 L1(cyc_rgb_k, orig_k)   6..10 ssim_k   11..15 -log((1+ssim_k)/2)
//   (0 where flag k)   16 content   17 style (factor applied)
#include "common.h"

#define Y2R_V_R 1.13988303f
#define Y2R_U_G -0.394642334f
#define Y2R_V_G -0.58062185f
#define Y2R_U_B 2.03206185f

constexpr int NSUM = 19;    // 6 L1 + content + 6 gram(cycED) + 6 gram(ds4)
constexpr int WIN = 11;
constexpr int TILE = 16;
constexpr int HALO = TILE + WIN - 1;   // 26

struct ImgArgs {
    const float* gen_rgb;
    const float* cyc_rgb;
    const float* cyc_y;
    const float* cbcr;
    const float* orig[5];
    const float* ds[5];
    int flags;
    float style_factor;
    double* loss;
    float* dgen_y;
    float* dcyc_y;
    // workspace carve
    double* sums;       // [B][NSUM]
    unsigned* mm;       // [B][10][2] ordered-uint min/max: slots 0..4 cyc_yuv_k, 5..9 ds_k
    float* gd;          // [B][4]  gram diffs D00, D01, D02
    double* ssim_sum;   // [B][5]
    float* gcoef;       // [B][5]  dLoss/dS_p scaling
    double* rsum;       // [B][5][2]  sum dx, sum dx*r
    int* argpos;        // [B][5][2]
    float* dmaps;       // [B*5*3][3][HO*WO]
    int batch, s;
};

__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ double block_sum_d2(double v) {
    __shared__ double ws[4];
    v = shm_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    return ws[0] + ws[1] + ws[2] + ws[3];
}

// ------------------------------------------------------------------- pass A: pixel sums
__global__ __launch_bounds__(256) void img_pass_a(const ImgArgs a) {
    const int b = blockIdx.y, B = a.batch;
    const size_t npix = (size_t)a.s * a.s;
    const float invB = 1.0f / B, inv3n = 1.0f / (3.0f * (float)npix);
    double sm[NSUM];
#pragma unroll
    for (int i = 0; i < NSUM; ++i) sm[i] = 0.0;
    float mn[10], mx[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { mn[i] = 3.0e38f; mx[i] = -3.0e38f; }

    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
        const size_t bp = (size_t)b * npix + p;
        const float u = a.cbcr[bp * 2], v = a.cbcr[bp * 2 + 1];
        // L1(gen_rgb, origED): coefficient 10 * (1/5)
        {
            float g = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float d = a.gen_rgb[bp * 3 + c] - a.orig[4][bp * 3 + c];
                sm[0] += fabs((double)d);
                g += (d > 0.f) ? 1.f : (d < 0.f ? -1.f : 0.f);
            }
            a.dgen_y[bp] = 10.0f * 0.2f * invB * inv3n * g;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const size_t ip = ((size_t)k * B + b) * npix + p;
            float g = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float d = a.cyc_rgb[ip * 3 + c] - a.orig[k][bp * 3 + c];
                sm[1 + k] += fabs((double)d);
                g += (d > 0.f) ? 1.f : (d < 0.f ? -1.f : 0.f);
            }
            float grad = 10.0f * (k == 4 ? 10.0f : 0.2f) * invB * inv3n * g;
            const float y = a.cyc_y[ip];
            mn[k] = fminf(mn[k], fminf(y, fminf(u, v)));
            mx[k] = fmaxf(mx[k], fmaxf(y, fmaxf(u, v)));
            const float d0 = a.ds[k][bp * 3], d1 = a.ds[k][bp * 3 + 1], d2 = a.ds[k][bp * 3 + 2];
            mn[5 + k] = fminf(mn[5 + k], fminf(d0, fminf(d1, d2)));
            mx[5 + k] = fmaxf(mx[5 + k], fmaxf(d0, fmaxf(d1, d2)));
            if (k == 4) {
                // content = mean((cycED_yuv - ds1_yuv)^2): weight 10 (NST) * 1
                const float e0 = y - a.ds[0][bp * 3], e1 = u - a.ds[0][bp * 3 + 1], e2 = v - a.ds[0][bp * 3 + 2];
                sm[6] += (double)e0 * e0 + (double)e1 * e1 + (double)e2 * e2;
                grad += 10.0f * invB * 2.0f * e0 * inv3n;
                // gram sums: (00,01,02,11,12,22)
                sm[7] += (double)y * y;  sm[8] += (double)y * u;  sm[9] += (double)y * v;
                sm[10] += (double)u * u; sm[11] += (double)u * v; sm[12] += (double)v * v;
                sm[13] += (double)d0 * d0; sm[14] += (double)d0 * d1; sm[15] += (double)d0 * d2;
                sm[16] += (double)d1 * d1; sm[17] += (double)d1 * d2; sm[18] += (double)d2 * d2;
            }
            a.dcyc_y[ip] = grad;
        }
    }
#pragma unroll
    for (int i = 0; i < NSUM; ++i) {
        double s = block_sum_d2(sm[i]);
        if (threadIdx.x == 0) atomicAdd(&a.sums[(size_t)b * NSUM + i], s);
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        float lo = shm_wave_min(mn[i]), hi = shm_wave_max(mx[i]);
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&a.mm[((size_t)b * 10 + i) * 2], f2ord(lo));
            atomicMax(&a.mm[((size_t)b * 10 + i) * 2 + 1], f2ord(hi));
        }
    }
}

// ---------------------------------------------------------- pass B: per-sample finalize
__global__ void img_pass_b(const ImgArgs a) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double npix = (double)a.s * a.s;
    for (int b = 0; b < a.batch; ++b) {
        const double* sm = a.sums + (size_t)b * NSUM;
        for (int i = 0; i < 6; ++i) a.loss[i] += sm[i] / (3.0 * npix);
        a.loss[16] += sm[6] / (3.0 * npix);
        double D[6];
        for (int i = 0; i < 6; ++i) D[i] = (sm[7 + i] - sm[13 + i]) / npix;
        // full 3x3: diag (00,11,22) once, off-diag (01,02,12) twice
        double sq = D[0] * D[0] + D[3] * D[3] + D[5] * D[5] + 2.0 * (D[1] * D[1] + D[2] * D[2] + D[4] * D[4]);
        a.loss[17] += (double)a.style_factor * sq / 9.0;
        a.gd[b * 4 + 0] = (float)D[0];
        a.gd[b * 4 + 1] = (float)D[1];
        a.gd[b * 4 + 2] = (float)D[2];
    }
}

// style gradient: d style/dY_p = 4 f / (9 S^2) * (D00*Y + D01*U + D02*V); weight 10*100/B
__global__ void img_style_grad(const ImgArgs a) {
    const size_t npix = (size_t)a.s * a.s;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)a.batch * npix) return;
    const int b = (int)(idx / npix);
    const size_t ip = (size_t)4 * a.batch * npix + idx;
    const float y = a.cyc_y[ip], u = a.cbcr[idx * 2], v = a.cbcr[idx * 2 + 1];
    const float coef = 1000.0f / a.batch * 4.0f * a.style_factor / (9.0f * (float)npix);
    a.dcyc_y[ip] += coef * (a.gd[b * 4] * y + a.gd[b * 4 + 1] * u + a.gd[b * 4 + 2] * v);
}

// ------------------------------------------------------------------------ SSIM forward
__device__ __forceinline__ void gauss1d(float* w) {   // 11 taps, sigma 1.5, normalised
    float s = 0.f;
    for (int i = 0; i < WIN; ++i) {
        float c = (float)i - 5.0f;
        w[i] = expf(-0.5f * c * c / 2.25f);
        s += w[i];
    }
    for (int i = 0; i < WIN; ++i) w[i] /= s;
}

// value of channel c of cyc_yuv_k at (b,p)
__device__ __forceinline__ float cyc_val(const ImgArgs& a, int b, int k, int c, size_t p, size_t npix) {
    if (c == 0) return a.cyc_y[((size_t)k * a.batch + b) * npix + p];
    return a.cbcr[((size_t)b * npix + p) * 2 + (c - 1)];
}

// grid: (tiles_x*tiles_y, B*5, 3); block 16x16
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const ImgArgs a) {
    __shared__ float xs[HALO][HALO + 1], ys[HALO][HALO + 1];
    __shared__ float hx[HALO][TILE + 1], hy[HALO][TILE + 1], hxy[HALO][TILE + 1], hsq[HALO][TILE + 1];
    __shared__ float w1[WIN];
    const int S = a.s, HO = S - WIN + 1;
    const int tiles_x = (HO + TILE - 1) / TILE;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int bk = blockIdx.y, b = bk / 5, k = bk % 5, c = blockIdx.z;
    const size_t npix = (size_t)S * S;
    const float xmn = ord2f(a.mm[((size_t)b * 10 + k) * 2]), xmx = ord2f(a.mm[((size_t)b * 10 + k) * 2 + 1]);
    const float ymn = ord2f(a.mm[((size_t)b * 10 + 5 + k) * 2]), ymx = ord2f(a.mm[((size_t)b * 10 + 5 + k) * 2 + 1]);
    const float xr = xmx > xmn ? xmx - xmn : 0.f, yr = ymx > ymn ? ymx - ymn : 0.f;
    if (threadIdx.x == 0) gauss1d(w1);
    const int oy0 = ty * TILE, ox0 = tx * TILE;
    for (int i = threadIdx.x; i < HALO * HALO; i += 256) {
        int r = i / HALO, cc = i % HALO;
        int yy = oy0 + r, xx = ox0 + cc;
        float xv = 0.f, yv = 0.f;
        if (yy < S && xx < S) {
            size_t p = (size_t)yy * S + xx;
            float xraw = cyc_val(a, b, k, c, p, npix);
            float yraw = a.ds[k][((size_t)b * npix + p) * 3 + c];
            xv = xr > 0.f ? (xraw - xmn) / xr : 0.f;       // divide_no_nan
            yv = yr > 0.f ? (yraw - ymn) / yr : 0.f;
        }
        xs[r][cc] = xv;
        ys[r][cc] = yv;
    }
    __syncthreads();
    // the window is an outer product: row sums of the four moments once per (halo row, output column) -- 26 x 16 of them, shared
    // by the eleven output rows that use a halo row -- then eleven taps down the column (same operations, same order as the
    // 121-tap loop per output pixel this replaces: bit-identical, 6x fewer FMAs)
    for (int i = threadIdx.x; i < HALO * TILE; i += 256) {
        const int r = i / TILE, cx = i % TILE;
        float rx = 0.f, ry = 0.f, rxy = 0.f, rsq = 0.f;
        for (int j = 0; j < WIN; ++j) {
            float xv = xs[r][cx + j], yv = ys[r][cx + j], w = w1[j];
            rx += w * xv;
            ry += w * yv;
            rxy += w * xv * yv;
            rsq += w * (xv * xv + yv * yv);
        }
        hx[r][cx] = rx;
        hy[r][cx] = ry;
        hxy[r][cx] = rxy;
        hsq[r][cx] = rsq;
    }
    __syncthreads();
    const int ly = threadIdx.x / TILE, lx = threadIdx.x % TILE;
    const int oy = oy0 + ly, ox = ox0 + lx;
    double sval = 0.0;
    if (oy < HO && ox < HO) {
        float mx_ = 0.f, my_ = 0.f, exy = 0.f, esq = 0.f;
        for (int i = 0; i < WIN; ++i) {
            mx_ += w1[i] * hx[ly + i][lx];
            my_ += w1[i] * hy[ly + i][lx];
            exy += w1[i] * hxy[ly + i][lx];
            esq += w1[i] * hsq[ly + i][lx];
        }
        const float c1 = 0.0025f, c2 = 0.0225f;      // (0.01*5)^2, (0.03*5)^2: max_val = 5 (SHM.py:759)
        const float A1 = 2.f * mx_ * my_ + c1, B1 = mx_ * mx_ + my_ * my_ + c1;
        const float A2 = 2.f * exy - 2.f * mx_ * my_ + c2, B2 = esq - mx_ * mx_ - my_ * my_ + c2;
        const float l = A1 / B1, cs = A2 / B2;
        sval = (double)(l * cs);
        const float dmx = cs * (2.f * my_ * B1 - A1 * 2.f * mx_) / (B1 * B1) + l * (-2.f * my_ * B2 + A2 * 2.f * mx_) / (B2 * B2);
        const float dexy = l * 2.f / B2;
        const float desq = -l * A2 / (B2 * B2);
        const size_t no = (size_t)HO * HO;
        float* dm = a.dmaps + ((size_t)bk * 3 + c) * 3 * no + (size_t)oy * HO + ox;
        dm[0] = dmx;
        dm[no] = dexy;
        dm[2 * no] = desq;
    }
    sval = block_sum_d2(sval);
    if (threadIdx.x == 0) atomicAdd(&a.ssim_sum[bk], sval);
}

__global__ void ssim_finalize_kernel(const ImgArgs a) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int HO = a.s - WIN + 1;
    const double cnt = 3.0 * (double)HO * HO;
    for (int b = 0; b < a.batch; ++b)
        for (int k = 0; k < 5; ++k) {
            double ssim = a.ssim_sum[b * 5 + k] / cnt;
            bool fl = (a.flags >> k) & 1;
            a.loss[6 + k] += ssim;
            double sl = fl ? 0.0 : -log((1.0 + ssim) / 2.0);
            a.loss[11 + k] += sl;
            // total_G has 10 * ssim_cyc_loss, ssim_cyc_loss = (l1+l2+l3+l4+10*l5)/5, mean over batch
            double coef = 10.0 * (k == 4 ? 2.0 : 0.2) / a.batch;
            a.gcoef[b * 5 + k] = fl ? 0.f : (float)(coef * (-1.0 / (1.0 + ssim)) / cnt);
        }
}

// ----------------------------------------------------------------------- SSIM backward
// grid: (tiles over the S x S input, B*5, 3); block 16x16 input pixels.
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const ImgArgs a) {
    __shared__ float d0[HALO][HALO + 1], d1[HALO][HALO + 1], d2[HALO][HALO + 1];
    __shared__ float h0[HALO][TILE + 1], h1[HALO][TILE + 1], h2[HALO][TILE + 1];
    __shared__ float w1[WIN];
    const int S = a.s, HO = S - WIN + 1;
    const int tiles_x = (S + TILE - 1) / TILE;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int bk = blockIdx.y, b = bk / 5, k = bk % 5, c = blockIdx.z;
    const float gco = a.gcoef[bk];
    if (gco == 0.f) return;                       // flagged view: no ssim term
    const size_t npix = (size_t)S * S, no = (size_t)HO * HO;
    const float xmn = ord2f(a.mm[((size_t)b * 10 + k) * 2]), xmx = ord2f(a.mm[((size_t)b * 10 + k) * 2 + 1]);
    const float ymn = ord2f(a.mm[((size_t)b * 10 + 5 + k) * 2]), ymx = ord2f(a.mm[((size_t)b * 10 + 5 + k) * 2 + 1]);
    const float xr = xmx > xmn ? xmx - xmn : 0.f, yr = ymx > ymn ? ymx - ymn : 0.f;
    if (threadIdx.x == 0) gauss1d(w1);
    const int qy0 = ty * TILE, qx0 = tx * TILE;
    // outputs p in [q-10, q]: LDS tile origin = q0 - 10
    const float* dm = a.dmaps + ((size_t)bk * 3 + c) * 3 * no;
    for (int i = threadIdx.x; i < HALO * HALO; i += 256) {
        int r = i / HALO, cc = i % HALO;
        int py = qy0 - (WIN - 1) + r, px = qx0 - (WIN - 1) + cc;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (py >= 0 && py < HO && px >= 0 && px < HO) {
            size_t o = (size_t)py * HO + px;
            v0 = dm[o];
            v1 = dm[no + o];
            v2 = dm[2 * no + o];
        }
        d0[r][cc] = v0;
        d1[r][cc] = v1;
        d2[r][cc] = v2;
    }
    __syncthreads();
    // separable, as in the forward kernel: row sums per (tile row of the LDS image, input column), then eleven taps down the column
    for (int i = threadIdx.x; i < HALO * TILE; i += 256) {
        const int rr = i / TILE, cx = i % TILE;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        for (int j = 0; j < WIN; ++j) {
            const int cc = cx + (WIN - 1) - j;
            const float w = w1[j];
            r0 += w * d0[rr][cc];
            r1 += w * d1[rr][cc];
            r2 += w * d2[rr][cc];
        }
        h0[rr][cx] = r0;
        h1[rr][cx] = r1;
        h2[rr][cx] = r2;
    }
    __syncthreads();
    const int ly = threadIdx.x / TILE, lx = threadIdx.x % TILE;
    const int qy = qy0 + ly, qx = qx0 + lx;
    double sdx = 0.0, sdxr = 0.0;
    if (qy < S && qx < S) {
        // input q contributes to output p = q - (i,j) with window weight w[i][j]
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        for (int i = 0; i < WIN; ++i) {
            const int rr = ly + (WIN - 1) - i;
            g0 += w1[i] * h0[rr][lx];
            g1 += w1[i] * h1[rr][lx];
            g2 += w1[i] * h2[rr][lx];
        }
        const size_t p = (size_t)qy * S + qx;
        const float xraw = cyc_val(a, b, k, c, p, npix);
        const float yraw = a.ds[k][((size_t)b * npix + p) * 3 + c];
        const float xv = xr > 0.f ? (xraw - xmn) / xr : 0.f;
        const float yv = yr > 0.f ? (yraw - ymn) / yr : 0.f;
        const float dx = gco * (g0 + yv * g1 + 2.f * xv * g2);      // dLoss / d rescaled x
        sdx = (double)dx;
        sdxr = (double)dx * (double)xv;
        if (c == 0 && xr > 0.f) {
            a.dcyc_y[((size_t)k * a.batch + b) * npix + p] += dx / xr;
            if (xraw == xmn) a.argpos[bk * 2] = (int)p;
            if (xraw == xmx) a.argpos[bk * 2 + 1] = (int)p;
        }
    }
    sdx = block_sum_d2(sdx);
    sdxr = block_sum_d2(sdxr);
    if (threadIdx.x == 0) {
        atomicAdd(&a.rsum[bk * 2], sdx);
        atomicAdd(&a.rsum[bk * 2 + 1], sdxr);
    }
}

// min / max sub-gradients of rescale_01: r = (x-mn)/(mx-mn)
//   dL/dmn = -sum dx (1-r) / (mx-mn),  dL/dmx = -sum dx r / (mx-mn)
__global__ void ssim_minmax_kernel(const ImgArgs a) {
    int bk = blockIdx.x * blockDim.x + threadIdx.x;
    if (bk >= a.batch * 5) return;
    const int b = bk / 5, k = bk % 5;
    if (a.gcoef[bk] == 0.f) return;
    const float xmn = ord2f(a.mm[((size_t)b * 10 + k) * 2]), xmx = ord2f(a.mm[((size_t)b * 10 + k) * 2 + 1]);
    if (!(xmx > xmn)) return;
    const double inv = 1.0 / ((double)xmx - (double)xmn);
    const double sdx = a.rsum[bk * 2], sdxr = a.rsum[bk * 2 + 1];
    const size_t npix = (size_t)a.s * a.s;
    float* dy = a.dcyc_y + ((size_t)k * a.batch + b) * npix;
    const int pmin = a.argpos[bk * 2], pmax = a.argpos[bk * 2 + 1];
    if (pmin >= 0) dy[pmin] += (float)(-(sdx - sdxr) * inv);
    if (pmax >= 0) dy[pmax] += (float)(-sdxr * inv);
}

// ------------------------------------------------------------------------------- host
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct WsPlan {
    size_t sums, mm, gd, ssim_sum, gcoef, rsum, argpos, dmaps, total, zero_bytes;
};

static WsPlan plan_ws(int batch, int s) {
    WsPlan w;
    size_t off = 0;
    w.sums = off;     off = align256(off + (size_t)batch * NSUM * 8);
    w.ssim_sum = off; off = align256(off + (size_t)batch * 5 * 8);
    w.rsum = off;     off = align256(off + (size_t)batch * 10 * 8);
    w.gd = off;       off = align256(off + (size_t)batch * 4 * 4);
    w.gcoef = off;    off = align256(off + (size_t)batch * 5 * 4);
    w.zero_bytes = off;                       // everything above starts at zero
    w.mm = off;       off = align256(off + (size_t)batch * 20 * 4);
    w.argpos = off;   off = align256(off + (size_t)batch * 10 * 4);
    w.dmaps = off;
    int ho = s - WIN + 1;
    off = align256(off + (size_t)batch * 5 * 3 * 3 * ho * ho * 4);
    w.total = off;
    return w;
}

extern "C" size_t shm_image_losses_workspace(int batch, int s) {
    if (s < WIN) return 0;
    return plan_ws(batch, s).total;
}

__global__ void init_mm_kernel(unsigned* mm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mm[i] = (i & 1) ? 0u : 0xffffffffu;       // [min, max] pairs
}

extern "C" int shm_image_losses(const float* gen_rgb, const float* cyc_rgb, const float* cyc_y, const float* cbcr,
                                const float* const* orig, const float* const* ds, int flags_mask, float style_factor,
                                double* loss, float* dgen_y, float* dcyc_y, void* ws, size_t ws_bytes, int batch, int s,
                                void* stream) {
    SHM_REQUIRE(s >= WIN, SHM_E_SHAPE, "shm_image_losses: image size %d < 11 (ssim window)", s);
    SHM_REQUIRE(batch > 0 && batch * 5 <= 65535, SHM_E_SHAPE, "shm_image_losses: bad batch %d", batch);
    WsPlan w = plan_ws(batch, s);
    SHM_REQUIRE(ws && ws_bytes >= w.total, SHM_E_WORKSPACE, "shm_image_losses: workspace %zu < %zu bytes", ws_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    ImgArgs a{};
    a.gen_rgb = gen_rgb;
    a.cyc_rgb = cyc_rgb;
    a.cyc_y = cyc_y;
    a.cbcr = cbcr;
    for (int i = 0; i < 5; ++i) {
        a.orig[i] = orig[i];
        a.ds[i] = ds[i];
    }
    a.flags = flags_mask;
    a.style_factor = style_factor;
    a.loss = loss;
    a.dgen_y = dgen_y;
    a.dcyc_y = dcyc_y;
    a.sums = (double*)(base + w.sums);
    a.mm = (unsigned*)(base + w.mm);
    a.gd = (float*)(base + w.gd);
    a.ssim_sum = (double*)(base + w.ssim_sum);
    a.gcoef = (float*)(base + w.gcoef);
    a.rsum = (double*)(base + w.rsum);
    a.argpos = (int*)(base + w.argpos);
    a.dmaps = (float*)(base + w.dmaps);
    a.batch = batch;
    a.s = s;

    int r = shm_zero(base, w.zero_bytes, stream);
    if (r) return r;
    r = shm_zero(loss, 32 * sizeof(double), stream);
    if (r) return r;
    hipError_t e = hipMemsetAsync(a.argpos, 0xff, (size_t)batch * 10 * 4, st);
    SHM_REQUIRE(e == hipSuccess, SHM_E_HIP, "shm_image_losses: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(init_mm_kernel, dim3(shm_cdiv(batch * 20, 64)), dim3(64), 0, st, a.mm, batch * 20);
    SHM_LAUNCH_CHECK("shm_image_losses(init)");

    const size_t npix = (size_t)s * s;
    int nblk = (int)((npix + 1023) / 1024);
    if (nblk > 256) nblk = 256;
    hipLaunchKernelGGL(img_pass_a, dim3(nblk, batch), dim3(256), 0, st, a);
    SHM_LAUNCH_CHECK("shm_image_losses(pass a)");
    hipLaunchKernelGGL(img_pass_b, dim3(1), dim3(64), 0, st, a);
    SHM_LAUNCH_CHECK("shm_image_losses(pass b)");
    hipLaunchKernelGGL(img_style_grad, dim3(shm_cdiv((long)(batch * npix), 256)), dim3(256), 0, st, a);
    SHM_LAUNCH_CHECK("shm_image_losses(style)");
    const int ho = s - WIN + 1;
    const int tf = shm_cdiv(ho, TILE), tb = shm_cdiv(s, TILE);
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(tf * tf, batch * 5, 3), dim3(256), 0, st, a);
    SHM_LAUNCH_CHECK("shm_image_losses(ssim fwd)");
    hipLaunchKernelGGL(ssim_finalize_kernel, dim3(1), dim3(64), 0, st, a);
    SHM_LAUNCH_CHECK("shm_image_losses(ssim finalize)");
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3(tb * tb, batch * 5, 3), dim3(256), 0, st, a);
    SHM_LAUNCH_CHECK("shm_image_losses(ssim bwd)");
    hipLaunchKernelGGL(ssim_minmax_kernel, dim3(shm_cdiv(batch * 5, 64)), dim3(64), 0, st, a);
    SHM_LAUNCH_CHECK("shm_image_losses(minmax)");
    return SHM_OK;
}
