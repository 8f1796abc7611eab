// Winograd F(4x4, 3x3) for the DEEP 3x3 convolutions (>= 256 channels at <= 64x64 pixels per image: down2/down3/bottleneck.2/
// dec4/dec3 of /root/reference/Main_Final.py:235-270 and their autograd): 4x fewer multiplies than the direct convolution
// (36 per 4x4 output tile and channel pair instead of 144), 1.78x fewer than the fused F(2x2,3x3) kernel of conv_winograd.hip.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A          6x6 input patch d (stride 4, 1-pixel halo), 3x3 filter g, 4x4 output tile
//
// Unfused on purpose: at these depths the Winograd-domain tensors are small next to the arithmetic (V = 2.25x the input, a few
// tens of MB), so three streaming passes cost little, while the 36 position-GEMMs [tiles x K] x [K x N] run on the LDS-tiled
// 128x128 MFMA GEMM (runet_gemm_batched) instead of a kernel that also has to transform.  The high-resolution layers (64/128
// channels) stay on the fused F(2x2) kernel, where the Winograd-domain traffic would dominate.
//   wino4_input_kernel<0>: V[36][T][C]  = B^T d B        wino4_input_kernel<1>: Z[36][T][C] = A dY A^T   (weight gradient)
//   wino4_output_kernel  : y = A^T M A (+ bias, + y)
//   wino4_weight_kernel  : U[36][K][N]  = G g G^T   (forward: g = w[.,.,k,n]; data gradient: g = rot180(w)[.,.,n,k])
//   wino4_wgrad_out_kernel: dw = G^T (sum_splits dU) G
// Interpolation points 0, +-1, +-2, inf (Lavin & Gray); fp32 throughout, error ~1e-5 relative to the direct convolution.
#include "runet_common.h"
#include "../../include/runet_hip.h"
#include "derive_weights.h"
#include <stdlib.h>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ void bt6(T& d0, T& d1, T& d2, T& d3, T& d4, T& d5) {     // B^T d (in place)
    const T a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
    const T t0 = 4.f * d0 - 5.f * d2 + d4, t5 = 4.f * d1 - 5.f * d3 + d5;
    d0 = t0; d1 = a + b; d2 = a - b; d3 = c + e; d4 = c - e; d5 = t5;
}
template <typename T> __device__ __forceinline__ void a6(const T y0, const T y1, const T y2, const T y3, T (&z)[6]) {   // A y  (A = (A^T)^T, 6x4)
    const T p = y0 + y2, q = y1 + y3, r = y0 + 4.f * y2, s = 2.f * y1 + 8.f * y3;
    z[0] = y0; z[1] = p + q; z[2] = p - q; z[3] = r + s; z[4] = r - s; z[5] = y3;
}
template <typename T> __device__ __forceinline__ void at4(const T m0, const T m1, const T m2, const T m3, const T m4, const T m5, T (&o)[4]) {   // A^T m
    const T p = m1 + m2, q = m1 - m2, r = m3 + m4, s = m3 - m4;
    o[0] = m0 + p + r; o[1] = q + 2.f * s; o[2] = p + 4.f * r; o[3] = q + 8.f * s + m5;
}

// A convolution with dilation D over an H x W image is D*D independent dilation-1 convolutions over its (H/D) x (W/D) sub-images
// (rows oy, oy+D, ... and columns ox, ox+D, ...): the padding D of the full image is the padding 1 of each sub-image.  The transform
// kernels therefore see n_img*D*D "images" of H x W = sub-image size and only their addressing knows about D (pix()).
struct W4Geom { int n_img, H, W, TY, TX, D, WF; long T, img_px; };

__device__ __forceinline__ long pix(const W4Geom& g, int n, int ih, int iw) {        // pixel index in the full NHWC tensor
    if (g.D == 1) return ((long)n * g.H + ih) * g.W + iw;
    const int dd = g.D * g.D;
    const int nf = n / dd, o = n - nf * dd, oy = o / g.D, ox = o - oy * g.D;
    return (long)nf * g.img_px + (long)(ih * g.D + oy) * g.WF + (iw * g.D + ox);
}

// The elementwise producer of a transform's input, folded into its loads (ACT) so that the tensor between the two is never written:
//   MODE 0: src = the convolution output t in front of BatchNorm + ReLU (+ Dropout2d): the patch is max(t * scale + shift, 0) * factor[n][c],
//           i.e. bn_apply_kernel(relu) on the fly (same bn_pre -> bit-identical), zero outside the image;
//   MODE 1: src = dy of that activation, x = t: the tile is the BatchNorm-backward dx of bn_bwd_apply_kernel(relu_shift) on the fly.
struct W4Pre {
    const float *scale, *shift, *factor;
    const float *x, *mean, *invstd, *sums;
    int ldx;
    float inv_m;
};

// MODE 0: src = conv input (or dy for the data gradient), 6x6 patch at (4ty-1, 4tx-1);  MODE 1: src = dy, 4x4 tile at (4ty, 4tx)
template <int MODE, bool ACT = false>
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ src, int ld, int C, W4Geom g, float* __restrict__ V, W4Pre pre = W4Pre{}) {
    const int C2 = C >> 1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.T * C2) return;
    const long tile = idx / C2;
    const int c = (int)(idx - tile * C2) * 2;
    const int per = g.TY * g.TX;
    const int n = (int)(tile / per);
    const int rem = (int)(tile - (long)n * per);
    const int ty = rem / g.TX, tx = rem - ty * g.TX;
    f32x2 d[6][6];
    f32x2 psc = {1.f, 1.f}, psh = {0.f, 0.f}, pfa = {1.f, 1.f}, pca = {0.f, 0.f}, pcb = {0.f, 0.f};
    if constexpr (ACT) {
        const int nf = n / (g.D * g.D);                    // image of the full tensor (factor is per image and channel)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            psc[q] = pre.scale[c + q]; psh[q] = pre.shift[c + q];
            pfa[q] = pre.factor ? pre.factor[(long)nf * C + c + q] : 1.f;
            if constexpr (MODE == 1) {
                float a, b;
                bn_bwd_coef(psc[q], pre.mean[c + q], pre.invstd[c + q], pre.sums[c + q], pre.sums[C + c + q], pre.inv_m, a, b);
                pca[q] = a; pcb[q] = b;
            }
        }
    }
    if constexpr (MODE == 0) {
        const int h0 = 4 * ty - 1, w0 = 4 * tx - 1;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int ih = h0 + i, iw = w0 + j;
                const bool ok = (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
                const long off = ok ? pix(g, n, ih, iw) * ld + c : 0;
                f32x2 v = *reinterpret_cast<const f32x2*>(src + off);
                if constexpr (ACT) {
                    v[0] = fmaxf(bn_pre(v[0], psc[0], psh[0]), 0.f) * pfa[0];
                    v[1] = fmaxf(bn_pre(v[1], psc[1], psh[1]), 0.f) * pfa[1];
                }
                d[i][j] = ok ? v : f32x2{0.f, 0.f};
            }
#pragma unroll
        for (int j = 0; j < 6; ++j) bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
        for (int i = 0; i < 6; ++i) bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5]);
    } else {
        f32x2 y[4][4], t[6][4];
        const float* base = src + pix(g, n, 4 * ty, 4 * tx) * ld + c;
        const long rs = (long)g.D * g.WF * ld, ps = (long)g.D * ld;          // row / pixel stride inside the sub-image
        const float* xbase = ACT ? pre.x + pix(g, n, 4 * ty, 4 * tx) * pre.ldx + c : nullptr;
        const long xrs = (long)g.D * g.WF * pre.ldx, xps = (long)g.D * pre.ldx;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 v = *reinterpret_cast<const f32x2*>(base + i * rs + j * ps);
                if constexpr (ACT) {
                    const f32x2 xv = *reinterpret_cast<const f32x2*>(xbase + i * xrs + j * xps);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float gg = (bn_pre(xv[q], psc[q], psh[q]) > 0.f) ? v[q] * pfa[q] : 0.f;
                        v[q] = bn_bwd_dx(gg, psc[q], xv[q], pca[q], pcb[q]);
                    }
                }
                y[i][j] = v;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 z[6];
            a6(y[0][j], y[1][j], y[2][j], y[3][j], z);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][j] = z[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) a6(t[i][0], t[i][1], t[i][2], t[i][3], d[i]);
    }
    float* out = V + tile * C + c;
    const long xs = g.T * C;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2*>(out + (long)(i * 6 + j) * xs) = d[i][j];
}

// STATS: the BatchNorm statistics of what is stored ride along - (count, mean, M2) of the thread's 16 pixels per channel by a two-pass in
// registers, Chan-combined in tile order over the tiles of the block that hold the same channel pair (N/2 a power of two <= 256: 256 / (N/2)
// tiles per block, one partial row per block; N/2 >= 256: one row per tile).  stats [wino4_stats_parts][N][3]; fixed order: reproducible.
template <bool STATS>
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ M, int N, W4Geom g, const float* __restrict__ bias,
                                                           float* __restrict__ y, int ldy, int accumulate, float* __restrict__ stats) {
    const int N2 = N >> 1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx < g.T * N2;
    if (!STATS && !valid) return;
    const long tile = valid ? idx / N2 : 0;
    const int c = valid ? (int)(idx - tile * N2) * 2 : 0;
    f32x2 vals[4][4];
    if (valid) {
        const int per = g.TY * g.TX;
        const int n = (int)(tile / per);
        const int rem = (int)(tile - (long)n * per);
        const int ty = rem / g.TX, tx = rem - ty * g.TX;
        const float* in = M + tile * N + c;
        const long xs = g.T * N;
        f32x2 s[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x2 m[6], o[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const f32x2*>(in + (long)(i * 6 + j) * xs);
            at4(m[0], m[1], m[2], m[3], m[4], m[5], o);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i][j] = o[i];
        }
        f32x2 bv = {0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x2*>(bias + c);
        float* dst = y + pix(g, n, 4 * ty, 4 * tx) * ldy + c;
        const long rs = (long)g.D * g.WF * ldy, ps = (long)g.D * ldy;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 o[4];
            at4(s[i][0], s[i][1], s[i][2], s[i][3], s[i][4], s[i][5], o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* p = dst + i * rs + j * ps;
                f32x2 v = o[j] + bv;
                if (accumulate) v += *reinterpret_cast<const f32x2*>(p);
                *reinterpret_cast<f32x2*>(p) = v;
                vals[i][j] = v;
            }
        }
    }
    if constexpr (STATS) {
        f32x2 mean = {0.f, 0.f}, m2 = {0.f, 0.f};
        if (valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mean += vals[i][j];
            mean *= (1.f / 16.f);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const f32x2 d = vals[i][j] - mean; m2 += d * d; }
        }
        if (N2 >= 256) {                          // one tile slice per block: the thread's pair is the only holder of its channels in row `tile`
            if (valid) {
                float* o = stats + (tile * N + c) * 3;
                o[0] = 16.f; o[1] = mean[0]; o[2] = m2[0]; o[3] = 16.f; o[4] = mean[1]; o[5] = m2[1];
            }
            return;
        }
        __shared__ float sst[256 * 5];            // [thread][count, mean0, m2_0, mean1, m2_1]
        float* me = sst + threadIdx.x * 5;
        me[0] = valid ? 16.f : 0.f; me[1] = mean[0]; me[2] = m2[0]; me[3] = mean[1]; me[4] = m2[1];
        __syncthreads();
        if ((int)threadIdx.x < N2) {              // 256 % N2 == 0: thread t holds channel pair t of the block's first tile, t + N2 of the next, ...
            float cnt = 0.f, mu0 = 0.f, q0 = 0.f, mu1 = 0.f, q1 = 0.f;
            for (int t = threadIdx.x; t < 256; t += N2) {
                const float* o = sst + t * 5;
                const float nb = o[0];
                if (nb > 0.f) {
                    const float nt = cnt + nb, f = nb / nt, w = cnt * f;
                    const float d0 = o[1] - mu0, d1 = o[3] - mu1;
                    mu0 += d0 * f; q0 += o[2] + d0 * d0 * w;
                    mu1 += d1 * f; q1 += o[4] + d1 * d1 * w;
                    cnt = nt;
                }
            }
            float* o = stats + ((long)blockIdx.x * N + 2 * threadIdx.x) * 3;
            o[0] = cnt; o[1] = mu0; o[2] = q0; o[3] = cnt; o[4] = mu1; o[5] = q1;
        }
    }
}

// ---- data gradient as the ADJOINT of the forward algorithm:  y = A^T [U .* (B^T d B)] A   =>   dd = B [U^T .* (A dy A^T)] B^T, overlap-added.
// Z = A dy A^T is the transform the weight gradient needs anyway (wino4_input_kernel<1>): one pass over dy serves both, and the filter needs
// no second (rotated) transform.  M'[36][T][C] = Z . U^T comes from the position GEMMs; this kernel turns it into dx in GATHER form: thread =
// (4x4 block of output pixels = tile (a, b), channel pair).  The tile's own 6x6 patch B M' B^T covers the block with its inner 4x4; the block's
// first / last row and column also receive row 5 / row 0 (column 5 / column 0) of the neighbouring tiles' patches, and its corners one value of the
// diagonal neighbours.  Rows 0 and 5 of B M' B^T need only rows 0 and 5 of M' (B's first and last rows are 4 e_0 and e_5), so a thread reads
// 36 + 4 x 6 + 4 = 64 values instead of 36; nothing is scattered, every output pixel is written once (or accumulated into once).
template <typename T> __device__ __forceinline__ void b6(const T v0, const T v1, const T v2, const T v3, const T v4, const T v5, T (&o)[6]) {      // B v
    o[0] = 4.f * v0;
    o[1] = 4.f * (v2 - v1) + 2.f * (v4 - v3) + 4.f * v5;
    o[2] = -5.f * v0 - 4.f * (v1 + v2) - (v3 + v4);
    o[3] = (v1 - v2) + 2.f * (v3 - v4) - 5.f * v5;
    o[4] = v0 + v1 + v2 + v3 + v4;
    o[5] = v5;
}
__global__ __launch_bounds__(256) void wino4_output_adj_kernel(const float* __restrict__ M, int N, W4Geom g, float* __restrict__ y, int ldy, int accumulate) {
    const int N2 = N >> 1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.T * N2) return;
    const long tile = idx / N2;
    const int c = (int)(idx - tile * N2) * 2;
    const int per = g.TY * g.TX;
    const int n = (int)(tile / per);
    const int rem = (int)(tile - (long)n * per);
    const int ty = rem / g.TX, tx = rem - ty * g.TX;
    const long xs = g.T * N;
    const float* in = M + tile * N + c;
    auto ld = [&](const float* base, int pos) { return *reinterpret_cast<const f32x2*>(base + (long)pos * xs); };
    const bool up = ty > 0, down = ty < g.TY - 1, left = tx > 0, right = tx < g.TX - 1;
    const long dT = (long)g.TX * N;                      // one tile row
    // All 28 neighbour values are loaded up front from clamped addresses (a missing neighbour reads this tile's own patch and is discarded):
    // one batch of loads in flight instead of eight dependent load -> wait -> add groups.
    const float* qu = up ? in - dT : in;
    const float* qd = down ? in + dT : in;
    const float* ql = left ? in - N : in;
    const float* qr = right ? in + N : in;
    f32x2 nu[6], nd[6], nl[6], nr[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        nu[k] = ld(qu, 30 + k);                          // row 5 of the tile above
        nd[k] = ld(qd, k);                               // row 0 of the tile below
        nl[k] = ld(ql, 6 * k + 5);                       // column 5 of the tile to the left
        nr[k] = ld(qr, 6 * k);                           // column 0 of the tile to the right
    }
    const f32x2 cul = ld(up && left ? in - dT - N : in, 35), cur = ld(up && right ? in - dT + N : in, 30);
    const f32x2 cdl = ld(down && left ? in + dT - N : in, 5), cdr = ld(down && right ? in + dT + N : in, 0);
    float* dst = y + pix(g, n, 4 * ty, 4 * tx) * ldy + c;
    const long rs = (long)g.D * g.WF * ldy, ps = (long)g.D * ldy;
    f32x2 old[4][4];
    if (accumulate) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) old[i][j] = *reinterpret_cast<const f32x2*>(dst + i * rs + j * ps);
    }
    // own patch: columns first (B applied down each column), then rows; only the inner 4 x 4 is used
    f32x2 t[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x2 o[6];
        b6(ld(in, j), ld(in, 6 + j), ld(in, 12 + j), ld(in, 18 + j), ld(in, 24 + j), ld(in, 30 + j), o);
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i][j] = o[i];
    }
    f32x2 out[4][4];
#pragma unroll
    for (int i = 1; i < 5; ++i) {
        f32x2 o[6];
        b6(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], o);
#pragma unroll
        for (int j = 1; j < 5; ++j) out[i - 1][j - 1] = o[j];
    }
    const f32x2 zero = {0.f, 0.f};
    {   // vertical neighbours: their row 5 (tile above) / 4 x row 0 (tile below), transformed along the row
        f32x2 o[6];
        b6(nu[0], nu[1], nu[2], nu[3], nu[4], nu[5], o);
#pragma unroll
        for (int j = 1; j < 5; ++j) out[0][j - 1] += up ? o[j] : zero;
        b6(nd[0], nd[1], nd[2], nd[3], nd[4], nd[5], o);
#pragma unroll
        for (int j = 1; j < 5; ++j) out[3][j - 1] += down ? 4.f * o[j] : zero;
        // horizontal neighbours: their column 5 (tile to the left) / 4 x column 0 (tile to the right), transformed along the column
        b6(nl[0], nl[1], nl[2], nl[3], nl[4], nl[5], o);
#pragma unroll
        for (int i = 1; i < 5; ++i) out[i - 1][0] += left ? o[i] : zero;
        b6(nr[0], nr[1], nr[2], nr[3], nr[4], nr[5], o);
#pragma unroll
        for (int i = 1; i < 5; ++i) out[i - 1][3] += right ? 4.f * o[i] : zero;
    }
    // diagonal neighbours: one corner value each
    out[0][0] += (up && left) ? cul : zero;
    out[0][3] += (up && right) ? 4.f * cur : zero;
    out[3][0] += (down && left) ? 4.f * cdl : zero;
    out[3][3] += (down && right) ? 16.f * cdr : zero;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 v = out[i][j];
            if (accumulate) v += old[i][j];
            *reinterpret_cast<f32x2*>(dst + i * rs + j * ps) = v;
        }
}

// thread = (k, n) with n fastest (coalesced U stores).  forward: g[r][s] = w[r][s][k][n];  dgrad: g[r][s] = w[2-r][2-s][n][k]
__global__ __launch_bounds__(256) void wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int cin, int cout, int dgrad) {
    const int K = dgrad ? cout : cin, N = dgrad ? cin : cout;
    const long kn = (long)K * N;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= kn) return;
    const int k = (int)(i / N), n = (int)(i - (long)k * N);
    float gm[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            gm[r][s] = dgrad ? w[((long)((2 - r) * 3 + (2 - s)) * cin + n) * cout + k] : w[((long)(r * 3 + s) * cin + k) * cout + n];
    float t[6][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float g0 = gm[0][s], g1 = gm[1][s], g2 = gm[2][s];
        const float e = (g0 + g2) * (1.f / 6.f), f = g0 * (1.f / 24.f) + g2 * (1.f / 6.f);
        t[0][s] = 0.25f * g0; t[1][s] = -e - g1 * (1.f / 6.f); t[2][s] = -e + g1 * (1.f / 6.f);
        t[3][s] = f + g1 * (1.f / 12.f); t[4][s] = f - g1 * (1.f / 12.f); t[5][s] = g2;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const float g0 = t[a][0], g1 = t[a][1], g2 = t[a][2];
        const float e = (g0 + g2) * (1.f / 6.f), f = g0 * (1.f / 24.f) + g2 * (1.f / 6.f);
        float* o = U + (long)(a * 6) * kn + i;
        o[0] = 0.25f * g0; o[kn] = -e - g1 * (1.f / 6.f); o[2 * kn] = -e + g1 * (1.f / 6.f);
        o[3 * kn] = f + g1 * (1.f / 12.f); o[4 * kn] = f - g1 * (1.f / 12.f); o[5 * kn] = g2;
    }
}

// The same filter transform written STRAIGHT into the split-plane packing the bf16 matrix-core position GEMMs read (gemm_split.hip:
// Up[36][plane 3][K/8][N][8] bf16, x = h + m + l exactly): one pass instead of wino4_weight_kernel + x3_pack_kernel - the fp32 U (4x the
// weights) is never written or read (2 x 28 launches and 4.6 GB of the step's 105 GB were this round trip).  thread = (k octet, n), n fastest.
typedef __bf16 w4_bf16x8 __attribute__((ext_vector_type(8)));
// dgrad: 0 forward (K = cin);  1 rotated filter for the data gradient computed as a convolution (K = cout);  2 the forward's U TRANSPOSED over
// (k, n) and not rotated, for the data gradient computed as the ADJOINT of the forward algorithm (K = cout; wino4_output_adj_kernel)
__global__ __launch_bounds__(256) void wino4_weight_x3_kernel(const float* __restrict__ w, __bf16* __restrict__ Up, int cin, int cout, int dgrad) {
    derive::wino4_weight_x3_body(w, Up, cin, cout, dgrad, blockIdx.x);
}

// dw[r][s][k][n] = (G^T (sum_splits dU[split][36][k][n]) G)[r][s]
__device__ __forceinline__ void gt3(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5, float (&r)[3]) {
    const float p = m1 + m2, q = m3 + m4;
    r[0] = 0.25f * m0 - p * (1.f / 6.f) + q * (1.f / 24.f);
    r[1] = (m2 - m1) * (1.f / 6.f) + (m3 - m4) * (1.f / 12.f);
    r[2] = (q - p) * (1.f / 6.f) + m5;
}
__global__ __launch_bounds__(256) void wino4_wgrad_out_kernel(const float* __restrict__ dU, int splits, long kn, float* __restrict__ dw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= kn) return;
    float m[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) m[x] = 0.f;
    for (int s = 0; s < splits; ++s) {
        const float* p = dU + (long)s * 36 * kn + i;
        float v[36];                           // all 36 loads of a split in flight before the first add (the fused form waited per pair of loads)
#pragma unroll
        for (int x = 0; x < 36; ++x) v[x] = p[(long)x * kn];
#pragma unroll
        for (int x = 0; x < 36; ++x) m[x] += v[x];
    }
    float t[3][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float r[3];
        gt3(m[j], m[6 + j], m[12 + j], m[18 + j], m[24 + j], m[30 + j], r);
        t[0][j] = r[0]; t[1][j] = r[1]; t[2][j] = r[2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float r[3];
        gt3(t[a][0], t[a][1], t[a][2], t[a][3], t[a][4], t[a][5], r);
        dw[(long)(a * 3 + 0) * kn + i] = r[0];
        dw[(long)(a * 3 + 1) * kn + i] = r[1];
        dw[(long)(a * 3 + 2) * kn + i] = r[2];
    }
}

W4Geom geom(int n_img, int h, int w, int dil) {
    W4Geom g{};
    g.D = dil; g.WF = w; g.img_px = (long)h * w;
    g.n_img = n_img * dil * dil; g.H = h / dil; g.W = w / dil; g.TY = g.H / 4; g.TX = g.W / 4; g.T = (long)g.n_img * g.TY * g.TX;
    return g;
}
bool dil_ok(int h, int w, int dil) { return dil >= 1 && dil <= 8 && h % dil == 0 && w % dil == 0; }

// rows (tiles) per split of the weight-gradient GEMMs: enough splits for ~512 blocks, at least 64 tiles each, 16-aligned
int wgrad_rows_per_split(long T, int cin, int cout) {
    static const int target = getenv("RUNET_W4_WGRAD_BLOCKS") ? atoi(getenv("RUNET_W4_WGRAD_BLOCKS")) : 512;      // measurement knob (256: 582.1, 512: 580.9, 1024: 576.1, 2048: 572.1 img/s)
    const long blocks = (long)cdiv(cin, 128) * cdiv(cout, 128) * 36;
    long s = cdiv(target, blocks);
    const long maxs = T / 64 > 0 ? T / 64 : 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    return (int)((cdiv(T, s) + 15) / 16 * 16);
}

}  // namespace

extern "C" int runet_wino4_supported(int h, int w, int k, int n) {
    return (h >= 4 && w >= 4 && h % 4 == 0 && w % 4 == 0 && k >= 16 && k % 4 == 0 && n >= 4 && n % 4 == 0) ? 1 : 0;
}

extern "C" long runet_wino4_workspace_floats(int n_img, int h, int w, int k, int n) {
    return 36L * n_img * (h / 4) * (w / 4) * ((long)k + n);
}

extern "C" int runet_wino4_weights(const float* w_hwio, float* U, int cin, int cout, int dgrad, void* stream) {
    RUNET_REQUIRE(w_hwio && U && cin > 0 && cout > 0, "bad arguments");
    hipLaunchKernelGGL(wino4_weight_kernel, dim3(cdiv((long)cin * cout, 256)), dim3(256), 0, (hipStream_t)stream, w_hwio, U, cin, cout, dgrad);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_wino4_weights_x3(const float* w_hwio, void* Upacked, int cin, int cout, int dgrad, void* stream) {
    const int k = dgrad ? cout : cin, n = dgrad ? cin : cout;
    RUNET_REQUIRE(w_hwio && Upacked && cin > 0 && cout > 0 && k % 8 == 0, "bad arguments (the contraction side must be a multiple of 8 channels)");
    RUNET_REQUIRE(((uintptr_t)Upacked % 16) == 0, "alignment");
    hipLaunchKernelGGL(wino4_weight_x3_kernel, dim3(cdiv((long)(k / 8) * n, 256)), dim3(256), 0, (hipStream_t)stream, w_hwio, (__bf16*)Upacked, cin, cout, dgrad);
    RUNET_CHECK_LAUNCH();
}

// The three stages of runet_wino4_conv / runet_wino4_wgrad as separate entry points (profiling, reuse of V between forward and
// weight gradient).  mode 0: V[36][T][c] = B^T d B of the 6x6 patches of src;  mode 1: Z[36][T][c] = A dY A^T of the 4x4 tiles.
extern "C" int runet_wino4_input(const float* src, int ld, int c, int n_img, int h, int w, int dil, int mode, float* V, void* stream) {
    RUNET_REQUIRE(src && V && dil_ok(h, w, dil) && runet_wino4_supported(h / dil, w / dil, 16, 4) && c > 0 && c % 2 == 0 && ld >= c && ld % 2 == 0, "bad arguments");
    RUNET_REQUIRE(((uintptr_t)src % 8) == 0 && ((uintptr_t)V % 8) == 0 && (mode == 0 || mode == 1), "alignment / mode");
    const W4Geom g = geom(n_img, h, w, dil);
    if (mode == 0) hipLaunchKernelGGL(wino4_input_kernel<0>, dim3(cdiv(g.T * (c / 2), 256)), dim3(256), 0, (hipStream_t)stream, src, ld, c, g, V, W4Pre{});
    else hipLaunchKernelGGL(wino4_input_kernel<1>, dim3(cdiv(g.T * (c / 2), 256)), dim3(256), 0, (hipStream_t)stream, src, ld, c, g, V, W4Pre{});
    RUNET_CHECK_LAUNCH();
}

// runet_wino4_input(mode 0) of a1 = max(t * scale + shift, 0) * factor_nc[n][c] (runet_bn_apply with relu = 1) without a1 ever being written:
// V is bit-identical to the two-step form.  factor_nc may be NULL.
extern "C" int runet_wino4_input_act(const float* t, int ld, int c, int n_img, int h, int w, int dil, const float* scale, const float* shift,
                                     const float* factor_nc, float* V, void* stream) {
    RUNET_REQUIRE(t && V && scale && shift && dil_ok(h, w, dil) && runet_wino4_supported(h / dil, w / dil, 16, 4) && c > 0 && c % 2 == 0 && ld >= c && ld % 2 == 0,
                  "bad arguments");
    RUNET_REQUIRE(((uintptr_t)t % 8) == 0 && ((uintptr_t)V % 8) == 0, "alignment");
    const W4Geom g = geom(n_img, h, w, dil);
    W4Pre pre{};
    pre.scale = scale; pre.shift = shift; pre.factor = factor_nc;
    hipLaunchKernelGGL((wino4_input_kernel<0, true>), dim3(cdiv(g.T * (c / 2), 256)), dim3(256), 0, (hipStream_t)stream, t, ld, c, g, V, pre);
    RUNET_CHECK_LAUNCH();
}

// runet_wino4_input(mode 1) of dx = runet_bn_bwd_apply(dy, x, ..., relu_shift = shift) without dx ever being written: Z = A dx A^T, bit-identical
// to the two-step form.  sums [2c] = (sum g * xhat | sum g) as runet_bn_bwd_reduce leaves them (zeros in eval mode), m_total as there (0: the
// tensor's own pixel count).
extern "C" int runet_wino4_input_bn_bwd(const float* dy, int lddy, const float* x, int ldx, int c, int n_img, int h, int w, int dil, const float* mean,
                                        const float* invstd, const float* scale, const float* shift, const float* sums, const float* factor_nc,
                                        long m_total, float* Z, void* stream) {
    RUNET_REQUIRE(dy && x && Z && mean && invstd && scale && shift && sums, "null pointer");
    RUNET_REQUIRE(dil_ok(h, w, dil) && runet_wino4_supported(h / dil, w / dil, 16, 4) && c > 0 && c % 2 == 0 && lddy >= c && lddy % 2 == 0 && ldx >= c &&
                  ldx % 2 == 0, "bad arguments");
    RUNET_REQUIRE(((uintptr_t)dy % 8) == 0 && ((uintptr_t)x % 8) == 0 && ((uintptr_t)Z % 8) == 0, "alignment");
    const W4Geom g = geom(n_img, h, w, dil);
    W4Pre pre{};
    pre.scale = scale; pre.shift = shift; pre.factor = factor_nc; pre.x = x; pre.ldx = ldx; pre.mean = mean; pre.invstd = invstd; pre.sums = sums;
    pre.inv_m = 1.0f / (float)(m_total > 0 ? m_total : (long)n_img * h * w);
    hipLaunchKernelGGL((wino4_input_kernel<1, true>), dim3(cdiv(g.T * (c / 2), 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, c, g, Z, pre);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_wino4_output(const float* M, int n, int n_img, int h, int w, int dil, const float* bias, float* y, int ldy, int accumulate,
                                  void* stream) {
    RUNET_REQUIRE(M && y && dil_ok(h, w, dil) && runet_wino4_supported(h / dil, w / dil, 16, 4) && n > 0 && n % 2 == 0 && ldy >= n && ldy % 2 == 0, "bad arguments");
    RUNET_REQUIRE(((uintptr_t)M % 8) == 0 && ((uintptr_t)y % 8) == 0 && (!bias || ((uintptr_t)bias % 8) == 0), "alignment");
    const W4Geom g = geom(n_img, h, w, dil);
    hipLaunchKernelGGL(wino4_output_kernel<false>, dim3(cdiv(g.T * (n / 2), 256)), dim3(256), 0, (hipStream_t)stream, M, n, g, bias, y, ldy, accumulate, (float*)nullptr);
    RUNET_CHECK_LAUNCH();
}

// Rows of the statistics partials runet_wino4_output_stats writes for n output channels, or 0 when it cannot take them (n / 2 must be a power of
// two: the channel pairs of a block then repeat with period n / 2; dilated sub-image launches are not supported)
extern "C" int runet_wino4_output_stats_parts(int n_img, int h, int w, int n, int dil) {
    const int n2 = n / 2;
    if (dil != 1 || n <= 0 || n % 2 || (n2 & (n2 - 1)) || !runet_wino4_supported(h, w, 16, 4)) return 0;
    const long T = (long)n_img * (h / 4) * (w / 4);
    const long parts = n2 >= 256 ? T : cdiv(T * n2, 256);
    return parts > (1L << 30) ? 0 : (int)parts;
}

// runet_wino4_output that also leaves the BatchNorm statistics partials of y behind: stats [runet_wino4_output_stats_parts][n][3] = (count, mean, M2)
// for runet_bn_stats_finalize (no pass of runet_bn_stats over y)
extern "C" int runet_wino4_output_stats(const float* M, int n, int n_img, int h, int w, const float* bias, float* y, int ldy, int accumulate, float* stats,
                                        void* stream) {
    RUNET_REQUIRE(M && y && stats && runet_wino4_output_stats_parts(n_img, h, w, n, 1) > 0 && ldy >= n && ldy % 2 == 0, "bad arguments (n / 2 a power of two)");
    RUNET_REQUIRE(((uintptr_t)M % 8) == 0 && ((uintptr_t)y % 8) == 0 && (!bias || ((uintptr_t)bias % 8) == 0), "alignment");
    const W4Geom g = geom(n_img, h, w, 1);
    hipLaunchKernelGGL(wino4_output_kernel<true>, dim3(cdiv(g.T * (n / 2), 256)), dim3(256), 0, (hipStream_t)stream, M, n, g, bias, y, ldy, accumulate, stats);
    RUNET_CHECK_LAUNCH();
}

// data gradient by the adjoint form: M' [36][T][n] (= Z . U^T from the position GEMMs) -> dx [n_img, h, w, n] (+= when accumulate)
extern "C" int runet_wino4_output_adj(const float* M, int n, int n_img, int h, int w, int dil, float* y, int ldy, int accumulate, void* stream) {
    RUNET_REQUIRE(M && y && dil_ok(h, w, dil) && runet_wino4_supported(h / dil, w / dil, 16, 4) && n > 0 && n % 2 == 0 && ldy >= n && ldy % 2 == 0, "bad arguments");
    RUNET_REQUIRE(((uintptr_t)M % 8) == 0 && ((uintptr_t)y % 8) == 0, "alignment");
    const W4Geom g = geom(n_img, h, w, dil);
    hipLaunchKernelGGL(wino4_output_adj_kernel, dim3(cdiv(g.T * (n / 2), 256)), dim3(256), 0, (hipStream_t)stream, M, n, g, y, ldy, accumulate);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_wino4_wgrad_output(const float* dU, int splits, int cin, int cout, float* dw, void* stream) {
    RUNET_REQUIRE(dU && dw && splits >= 1 && cin > 0 && cout > 0, "bad arguments");
    const long kn = (long)cin * cout;
    hipLaunchKernelGGL(wino4_wgrad_out_kernel, dim3(cdiv(kn, 256)), dim3(256), 0, (hipStream_t)stream, dU, splits, kn, dw);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_wino4_wgrad_rows_per_split(int n_img, int h, int w, int cin, int cout) {
    return wgrad_rows_per_split((long)n_img * (h / 4) * (w / 4), cin, cout);
}

extern "C" int runet_wino4_conv(const float* x, int ldx, const float* U, const float* bias, float* y, int ldy, int n_img, int h, int w, int k, int n,
                                int dil, int accumulate, float* workspace, long workspace_floats, void* stream) {
    RUNET_REQUIRE(x && U && y && workspace, "null pointer");
    RUNET_REQUIRE(dil_ok(h, w, dil) && runet_wino4_supported(h / dil, w / dil, k, n),
                  "shape not supported by the F(4x4,3x3) path (H/dil, W/dil multiples of 4; K, N multiples of 4)");
    RUNET_REQUIRE(ldx >= k && ldx % 2 == 0 && ldy >= n && ldy % 2 == 0, "pixel strides must be even and cover the channels");
    RUNET_REQUIRE(((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)U % 16) == 0 && ((uintptr_t)workspace % 16) == 0 &&
                  (!bias || ((uintptr_t)bias % 8) == 0), "alignment");
    RUNET_REQUIRE(workspace_floats >= runet_wino4_workspace_floats(n_img, h, w, k, n), "workspace too small (runet_wino4_workspace_floats)");
    const W4Geom g = geom(n_img, h, w, dil);
    hipStream_t st = (hipStream_t)stream;
    float* V = workspace;
    float* M = workspace + 36L * g.T * k;
    hipLaunchKernelGGL(wino4_input_kernel<0>, dim3(cdiv(g.T * (k / 2), 256)), dim3(256), 0, st, x, ldx, k, g, V, W4Pre{});
    const int rc = runet_gemm_batched(V, k, g.T * k, U, (long)k * n, M, n, g.T * n, 36, (int)g.T, k, n, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(wino4_output_kernel<false>, dim3(cdiv(g.T * (n / 2), 256)), dim3(256), 0, st, M, n, g, bias, y, ldy, accumulate, (float*)nullptr);
    RUNET_CHECK_LAUNCH();
}

// The same composite with the 36 position-GEMMs on the bf16 matrix cores (gemm_split.hip: exact three-way operand split, fp32-accurate).
// Upacked: runet_wino4_weights -> runet_gemm_x3_pack (36 matrices [k][n] -> split planes), once per optimizer step.
extern "C" int runet_wino4_conv_x3(const float* x, int ldx, const void* Upacked, const float* bias, float* y, int ldy, int n_img, int h, int w, int k,
                                   int n, int dil, int accumulate, float* workspace, long workspace_floats, void* stream) {
    RUNET_REQUIRE(x && Upacked && y && workspace, "null pointer");
    RUNET_REQUIRE(dil_ok(h, w, dil) && runet_wino4_supported(h / dil, w / dil, k, n) && k % 16 == 0,
                  "shape not supported by the split-operand F(4x4,3x3) path (H/dil, W/dil multiples of 4; K multiple of 16, N of 4)");
    RUNET_REQUIRE(ldx >= k && ldx % 2 == 0 && ldy >= n && ldy % 2 == 0, "pixel strides must be even and cover the channels");
    RUNET_REQUIRE(((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)Upacked % 16) == 0 && ((uintptr_t)workspace % 16) == 0 &&
                  (!bias || ((uintptr_t)bias % 8) == 0), "alignment");
    RUNET_REQUIRE(workspace_floats >= runet_wino4_workspace_floats(n_img, h, w, k, n), "workspace too small (runet_wino4_workspace_floats)");
    const W4Geom g = geom(n_img, h, w, dil);
    hipStream_t st = (hipStream_t)stream;
    float* V = workspace;
    float* M = workspace + 36L * g.T * k;
    hipLaunchKernelGGL(wino4_input_kernel<0>, dim3(cdiv(g.T * (k / 2), 256)), dim3(256), 0, st, x, ldx, k, g, V, W4Pre{});
    const int rc = runet_gemm_x3_batched(V, k, g.T * k, Upacked, M, n, g.T * n, 36, (int)g.T, k, n, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(wino4_output_kernel<false>, dim3(cdiv(g.T * (n / 2), 256)), dim3(256), 0, st, M, n, g, bias, y, ldy, accumulate, (float*)nullptr);
    RUNET_CHECK_LAUNCH();
}

extern "C" long runet_wino4_wgrad_workspace_floats(int n_img, int h, int w, int cin, int cout) {
    const long T = (long)n_img * (h / 4) * (w / 4);
    return 36L * T * ((long)cin + cout) + (long)cdiv(T, wgrad_rows_per_split(T, cin, cout)) * 36 * cin * cout;
}

extern "C" int runet_wino4_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats, int n_img,
                                 int h, int w, int cin, int cout, int dil, void* stream) {
    RUNET_REQUIRE(x && dy && dw && workspace, "null pointer");
    RUNET_REQUIRE(dil_ok(h, w, dil) && runet_wino4_supported(h / dil, w / dil, cin, cout) && cout >= 16, "shape not supported by the F(4x4,3x3) weight gradient");
    RUNET_REQUIRE(ldx >= cin && ldx % 2 == 0 && ldy >= cout && ldy % 2 == 0, "pixel strides must be even and cover the channels");
    RUNET_REQUIRE(((uintptr_t)x % 8) == 0 && ((uintptr_t)dy % 8) == 0 && ((uintptr_t)workspace % 16) == 0, "alignment");
    RUNET_REQUIRE(workspace_floats >= runet_wino4_wgrad_workspace_floats(n_img, h, w, cin, cout), "workspace too small (runet_wino4_wgrad_workspace_floats)");
    const W4Geom g = geom(n_img, h, w, dil);
    hipStream_t st = (hipStream_t)stream;
    float* V = workspace;
    float* Z = V + 36L * g.T * cin;
    float* dU = Z + 36L * g.T * cout;
    const int rps = wgrad_rows_per_split(g.T, cin, cout);
    const int splits = cdiv(g.T, rps);
    hipLaunchKernelGGL(wino4_input_kernel<0>, dim3(cdiv(g.T * (cin / 2), 256)), dim3(256), 0, st, x, ldx, cin, g, V, W4Pre{});
    hipLaunchKernelGGL(wino4_input_kernel<1>, dim3(cdiv(g.T * (cout / 2), 256)), dim3(256), 0, st, dy, ldy, cout, g, Z, W4Pre{});
    const int rc = runet_gemm_tn_batched(V, cin, g.T * cin, Z, cout, g.T * cout, dU, 36, (int)g.T, cin, cout, rps, stream);
    if (rc) return rc;
    const long kn = (long)cin * cout;
    hipLaunchKernelGGL(wino4_wgrad_out_kernel, dim3(cdiv(kn, 256)), dim3(256), 0, st, dU, splits, kn, dw);
    RUNET_CHECK_LAUNCH();
}
