// The RGB stem of the network on the matrix cores: Conv2d(3, C, 3, padding=1) and the 1x1 shortcut Conv2d(3, C, 1) of the first
// ResidualBlock (/root/reference/Main_Final.py:157,172 at in_channels = 3; `inc` :235) and their weight gradients.
//
// With 3 input channels the contraction is 27 deep (3 deep for the 1x1): the general implicit-GEMM kernel pads it to 48 and stages both
// operands through LDS per 16-deep step, the general weight-gradient tile kernel pads cin to 64 - both end up 3-7x away from the only
// real cost of these layers, which is moving the C-channel tensor through HBM once (16 x 256 x 256 x 64 fp32 = 268 MB, ~60 us).
//
//   stem_conv_kernel     block = 8 x 32 pixel tile.  The 10 x 34 x 4-channel input halo (5.4 KB) goes to LDS once; the whole filter bank is
//                        ONE 32 x (32*NT) B matrix held in registers (k = tap*cin + c, zero rows above 9*cin); a wave owns two 32-pixel
//                        row segments and runs 16 k-steps of v_mfma_f32_32x32x2_f32 per 32-channel tile, its A operand gathered from
//                        the halo with one ds_read_b32 per step.  The 1x1 shortcut rides along as further columns of B that are zero
//                        except in the centre tap's rows: both convolutions of the block in one launch, x read once.
//   stem_wgrad_kernel    dW[k][co] = sum_p X[p][k] * dY[p][co] with the PIXELS as the contraction: A[m = k][kk = pixel] is the same halo
//                        gather (lane-constant tap offset, pixel advancing), B[kk = pixel][n = co] is loaded from global memory
//                        straight into the MFMA operand layout (a half-wave reads 32 consecutive channels of one pixel: 128-byte
//                        segments), a whole tile row ahead of its use.  Per-block partial slabs, summed in a fixed order.
#include "runet_common.h"
#include "../../include/runet_hip.h"

namespace {

constexpr int TR = 8, TC = 32;                       // pixel tile: rows x columns
constexpr int HC = TC + 2;                           // halo columns
constexpr int HALO_FLOATS = (TR + 2) * HC * 4;       // + one zero slot behind it

struct StemFwd {
    const float* x; int ldx;
    const float* w3; const float* w1;                // [3][3][cin_w][cout], [cin_w][cout] or null
    float* y3; int ldy3; float* y1; int ldy1;
    int H, W, cin_w, cout;
    float* stats3; float* stats1;                    // nullptr, or [blocks][cout][3] (count, mean, M2) partials of y3 / y1 (BatchNorm statistics)
};

// x halo of tile (n, r0, c0) -> LDS [row][col][4] (zero outside the image); 256 threads
__device__ __forceinline__ void load_halo(float* halo, const float* __restrict__ x, int ldx, int n, int r0, int c0, int H, int W) {
    for (int i = threadIdx.x; i < (TR + 2) * HC; i += 256) {
        const int hr = i / HC, hc = i - hr * HC;
        const int ih = r0 + hr - 1, iw = c0 + hc - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = *reinterpret_cast<const f32x4*>(x + (((long)n * H + ih) * W + iw) * ldx);
        *reinterpret_cast<f32x4*>(halo + i * 4) = v;
    }
}

// LDS offset (floats, relative to the pixel's own halo slot at tap (0,0)) of contraction index k = tap*cin_w + c, or -1 for the padding rows
__device__ __forceinline__ int tap_offset(int k, int cin_w, int taps) {
    if (k >= taps * cin_w) return -1;
    const int tap = k / cin_w, c = k - tap * cin_w;
    if (taps == 1) return (HC + 1) * 4 + c;          // 1x1: the centre of the halo
    return ((tap / 3) * HC + (tap % 3)) * 4 + c;
}

template <int NT>
__global__ __launch_bounds__(256) void stem_conv_kernel(StemFwd a) {
    __shared__ __attribute__((aligned(16))) float halo[HALO_FLOATS + 4];
    __shared__ float bmat[32 * NT * 32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 31, kk = lane >> 5;
    const int tiles_x = (a.W + TC - 1) / TC;
    const int n = blockIdx.y, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int r0 = ty * TR, c0 = tx * TC;
    const int n3 = a.cout / 32;                      // 32-channel tiles of the 3x3 output; tiles n3.. belong to the 1x1 output
    // ---- B[k][n]: rows k < 9*cin_w of the 3x3 filter are contiguous in HWIO; the 1x1 filter occupies the centre tap's rows
    for (int i = tid; i < 32 * NT * 32; i += 256) {
        const int k = i / (NT * 32), col = i - k * (NT * 32);
        float v = 0.f;
        if (k < 9 * a.cin_w) {
            if (col < a.cout) v = a.w3[(long)k * a.cout + col];
            else {
                const int tap = k / a.cin_w, c = k - tap * a.cin_w;
                if (tap == 4) v = a.w1[(long)c * a.cout + (col - a.cout)];
            }
        }
        bmat[i] = v;
    }
    load_halo(halo, a.x, a.ldx, n, r0, c0, a.H, a.W);
    if (tid < 4) halo[HALO_FLOATS + tid] = 0.f;
    __syncthreads();
    float breg[16][NT];
    int aoff[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        aoff[s] = tap_offset(2 * s + kk, a.cin_w, 9);
#pragma unroll
        for (int t = 0; t < NT; ++t) breg[s][t] = bmat[(2 * s + kk) * (NT * 32) + t * 32 + li];
    }
    // running (count, mean, M2) of this lane's channel of every tile over the pixels it stores (a.stats3 != nullptr)
    float scnt = 0.f, smean[NT], sm2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { smean[t] = 0.f; sm2[t] = 0.f; }
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
        const int r = 2 * wid + seg;
        if (r0 + r >= a.H) break;
        const int base = (r * HC + li) * 4;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float av = halo[aoff[s] >= 0 ? base + aoff[s] : HALO_FLOATS];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[s][t], acc[t], 0, 0, 0);
        }
        // ---- D[row = pixel column][col = channel]: lane holds channel li, pixels (i&3) + 8*(i>>2) + 4*kk: 128-byte segments per pixel
        const long rowpix = ((long)n * a.H + r0 + r) * a.W + c0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float* y = t < n3 ? a.y3 : a.y1;
            const int ldy = t < n3 ? a.ldy3 : a.ldy1;
            const int ch = (t < n3 ? t : t - n3) * 32 + li;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = (i & 3) + 8 * (i >> 2) + 4 * kk;
                if (c0 + m < a.W) y[(rowpix + m) * ldy + ch] = acc[t][i];
            }
        }
        if (a.stats3) {
            // the segment's up to 16 pixels of this lane: two-pass moments, Chan-combined into the running ones (segment 0 first)
            float cnt = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) cnt += (c0 + (i & 3) + 8 * (i >> 2) + 4 * kk < a.W) ? 1.f : 0.f;
            if (cnt > 0.f) {
                const float nt = scnt + cnt, f = cnt / nt, wgt = scnt * f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float s1 = 0.f, q = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) s1 += (c0 + (i & 3) + 8 * (i >> 2) + 4 * kk < a.W) ? acc[t][i] : 0.f;
                    const float mu = s1 / cnt;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float d = acc[t][i] - mu;
                        q += (c0 + (i & 3) + 8 * (i >> 2) + 4 * kk < a.W) ? d * d : 0.f;
                    }
                    const float dl = mu - smean[t];
                    smean[t] += dl * f;
                    sm2[t] += q + dl * dl * wgt;
                }
                scnt = nt;
            }
        }
    }
    if (a.stats3) {
        // lane halves (pixels + 4), then the four waves in wave order through LDS (bmat's storage: it is dead once every wave holds its B registers)
        float* xch = bmat;                           // [4 waves][NT * 32][3]
        __syncthreads();                             // every wave has loaded its B registers from bmat
        const float ocnt = __shfl_xor(scnt, 32, 64);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float omean = __shfl_xor(smean[t], 32, 64), om2 = __shfl_xor(sm2[t], 32, 64);
            const float nt = scnt + ocnt;
            float mu = smean[t], q = sm2[t];
            if (ocnt > 0.f) { const float f = ocnt / nt, dl = omean - mu; mu += dl * f; q += om2 + dl * dl * (scnt * f); }
            if (kk == 0) { float* o = xch + ((wid * NT + t) * 32 + li) * 3; o[0] = nt; o[1] = mu; o[2] = q; }
        }
        __syncthreads();
        for (int col = tid; col < NT * 32; col += 256) {
            float cnt = 0.f, mu = 0.f, q = 0.f;
            for (int wv = 0; wv < 4; ++wv) {
                const float* o = xch + (wv * NT * 32 + col) * 3;
                if (o[0] > 0.f) { const float nt = cnt + o[0], f = o[0] / nt, dl = o[1] - mu; mu += dl * f; q += o[2] + dl * dl * (cnt * f); cnt = nt; }
            }
            const int t = col >> 5;
            float* dst = (t < n3 ? a.stats3 : a.stats1) + (((long)blockIdx.y * gridDim.x + blockIdx.x) * a.cout + (t < n3 ? t : t - n3) * 32 + (col & 31)) * 3;
            dst[0] = cnt; dst[1] = mu; dst[2] = q;
        }
    }
}

struct StemWgrad {
    const float* x; int ldx; const float* dy; int ldy;
    float* slabs;
    int Nimg, H, W, cin_w, cout, taps;               // taps: 9 (3x3) or 1 (1x1)
    int tiles_y, tiles_x, total_tiles, tiles_per_block;
};

template <int NT>
__global__ __launch_bounds__(256) void stem_wgrad_kernel2(StemWgrad a) {
    __shared__ __attribute__((aligned(16))) float smem[4 * NT * 16 * 64 > HALO_FLOATS + 4 ? 4 * NT * 16 * 64 : HALO_FLOATS + 4];
    float* halo = smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 31, kk = lane >> 5;
    const int K = a.taps * a.cin_w;
    const int moff = tap_offset(li, a.cin_w, a.taps);                 // this lane's row of dW: tap offset in the halo, -1 = padding row
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int t_begin = blockIdx.x * a.tiles_per_block;
    const int t_end = min(a.total_tiles, t_begin + a.tiles_per_block);
    const int per_img = a.tiles_y * a.tiles_x;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int n = tile / per_img, rem = tile - n * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int r0 = ty * TR, c0 = tx * TC;
        __syncthreads();                                              // the previous tile's halo is no longer read
        load_halo(halo, a.x, a.ldx, n, r0, c0, a.H, a.W);
        if (tid < 4) halo[HALO_FLOATS + tid] = 0.f;
        // the wave's two rows of dy, straight into the B-operand layout: b[s][t] = dy[pixel 2s + kk][32t + li]
        float breg[2][16][NT];
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            const int r = r0 + 2 * wid + seg;
            const float* row = a.dy + (((long)n * a.H + r) * a.W + c0) * a.ldy + li;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int c = 2 * s + kk;
                const bool ok = r < a.H && c0 + c < a.W;
#pragma unroll
                for (int t = 0; t < NT; ++t) breg[seg][s][t] = ok ? row[(long)c * a.ldy + t * 32] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            const int base = ((2 * wid + seg) * HC + kk) * 4;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float av = halo[moff >= 0 ? base + 8 * s + moff : HALO_FLOATS];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[seg][s][t], acc[t], 0, 0, 0);
            }
        }
    }
    // ---- the four waves' partial D -> one slab [K][cout]
    __syncthreads();
    float* red = smem;                                                // [wave][t][i][lane]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((wid * NT + t) * 16 + i) * 64 + lane] = acc[t][i];
    __syncthreads();
    float* slab = a.slabs + (long)blockIdx.x * K * a.cout;
    for (int e = tid; e < NT * 16 * 64; e += 256) {
        const int l = e & 63, i = (e >> 6) & 15, t = e >> 10;
        const int m = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
        if (m < K) {
            const float v = (red[((0 * NT + t) * 16 + i) * 64 + l] + red[((1 * NT + t) * 16 + i) * 64 + l]) +
                            (red[((2 * NT + t) * 16 + i) * 64 + l] + red[((3 * NT + t) * 16 + i) * 64 + l]);
            slab[(long)m * a.cout + t * 32 + (l & 31)] = v;
        }
    }
}

// out[i] = sum_b slabs[b][i] in block order (reproducible): 32 lanes of the block list per output group, LDS tree over them
__global__ __launch_bounds__(256) void stem_slab_reduce(const float* __restrict__ slabs, float* __restrict__ out, int n, int nslabs) {
    __shared__ float red[256];
    const int col = threadIdx.x & 7, kl = threadIdx.x >> 3;           // 8 outputs x 32 slab lanes
    const int i = blockIdx.x * 8 + col;
    float s = 0.f;
    if (i < n)
        for (int b = kl; b < nslabs; b += 32) s += slabs[(long)b * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (kl == 0 && i < n) {
        for (int j = 1; j < 32; ++j) s += red[j * 8 + col];
        out[i] = s;
    }
}

int wgrad_blocks(int total_tiles) { return total_tiles < 1024 ? total_tiles : 1024; }

}  // namespace

extern "C" int runet_stem_supported(int cin_w, int cout) { return (cin_w >= 1 && cin_w <= 3 && cout % 32 == 0 && cout >= 32 && cout <= 64) ? 1 : 0; }

static int stem_conv_launch(const float* x, int ldx, const float* w3, const float* w1, float* y3, int ldy3, float* y1, int ldy1, int n_img, int h, int w,
                            int cin_w, int cout, float* stats3, float* stats1, void* stream);

extern "C" int runet_stem_conv(const float* x, int ldx, const float* w3, const float* w1, float* y3, int ldy3, float* y1, int ldy1, int n_img,
                               int h, int w, int cin_w, int cout, void* stream) {
    return stem_conv_launch(x, ldx, w3, w1, y3, ldy3, y1, ldy1, n_img, h, w, cin_w, cout, nullptr, nullptr, stream);
}

extern "C" int runet_stem_conv_stats_parts(int n_img, int h, int w) { return n_img * cdiv(h, TR) * cdiv(w, TC); }

// runet_stem_conv that also leaves the BatchNorm statistics partials of y3 (and y1) behind: stats3 / stats1 [runet_stem_conv_stats_parts][cout][3]
extern "C" int runet_stem_conv_stats(const float* x, int ldx, const float* w3, const float* w1, float* y3, int ldy3, float* y1, int ldy1, int n_img,
                                     int h, int w, int cin_w, int cout, float* stats3, float* stats1, void* stream) {
    RUNET_REQUIRE(stats3 && (!w1 || stats1), "stats3 (and stats1 with a 1x1 filter) must not be NULL");
    return stem_conv_launch(x, ldx, w3, w1, y3, ldy3, y1, ldy1, n_img, h, w, cin_w, cout, stats3, stats1, stream);
}

static int stem_conv_launch(const float* x, int ldx, const float* w3, const float* w1, float* y3, int ldy3, float* y1, int ldy1, int n_img, int h, int w,
                            int cin_w, int cout, float* stats3, float* stats1, void* stream) {
    RUNET_REQUIRE(x && w3 && y3 && (!w1 || y1), "null pointer");
    RUNET_REQUIRE(runet_stem_supported(cin_w, cout), "stem kernel: 1..3 input channels, 32 or 64 output channels");
    RUNET_REQUIRE(ldx >= 4 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0, "x must be NHWC padded to (a multiple of) 4 channels, 16-byte aligned");
    RUNET_REQUIRE(n_img > 0 && h > 0 && w > 0 && ldy3 >= cout && (!w1 || ldy1 >= cout), "bad shape");
    StemFwd a{x, ldx, w3, w1, y3, ldy3, y1, ldy1, h, w, cin_w, cout, stats3, stats1};
    const dim3 grid(cdiv(h, TR) * cdiv(w, TC), n_img);
    const int nt = (cout / 32) * (w1 ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    if (nt == 1) hipLaunchKernelGGL(stem_conv_kernel<1>, grid, dim3(256), 0, st, a);
    else if (nt == 2) hipLaunchKernelGGL(stem_conv_kernel<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(stem_conv_kernel<4>, grid, dim3(256), 0, st, a);
    RUNET_CHECK_LAUNCH();
}

extern "C" long runet_stem_wgrad_workspace_floats(int n_img, int h, int w, int cin_w, int cout, int ksize) {
    return (long)wgrad_blocks(n_img * cdiv(h, TR) * cdiv(w, TC)) * ksize * ksize * cin_w * cout;
}

extern "C" int runet_stem_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats, int n_img,
                                int h, int w, int cin_w, int cout, int ksize, void* stream) {
    RUNET_REQUIRE(x && dy && dw && workspace, "null pointer");
    RUNET_REQUIRE(runet_stem_supported(cin_w, cout) && (ksize == 1 || ksize == 3), "stem kernel: 1..3 input channels, 32 or 64 output channels, 1x1 or 3x3");
    RUNET_REQUIRE(ldx >= 4 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ldy >= cout, "x must be NHWC padded to 4 channels, 16-byte aligned");
    RUNET_REQUIRE(workspace_floats >= runet_stem_wgrad_workspace_floats(n_img, h, w, cin_w, cout, ksize), "workspace too small (runet_stem_wgrad_workspace_floats)");
    StemWgrad a{};
    a.x = x; a.ldx = ldx; a.dy = dy; a.ldy = ldy; a.slabs = workspace;
    a.Nimg = n_img; a.H = h; a.W = w; a.cin_w = cin_w; a.cout = cout; a.taps = ksize * ksize;
    a.tiles_y = cdiv(h, TR); a.tiles_x = cdiv(w, TC); a.total_tiles = n_img * a.tiles_y * a.tiles_x;
    const int blocks0 = wgrad_blocks(a.total_tiles);
    a.tiles_per_block = cdiv(a.total_tiles, blocks0);
    const int blocks = cdiv(a.total_tiles, a.tiles_per_block);
    hipStream_t st = (hipStream_t)stream;
    if (cout == 32) hipLaunchKernelGGL(stem_wgrad_kernel2<1>, dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(stem_wgrad_kernel2<2>, dim3(blocks), dim3(256), 0, st, a);
    const int nout = a.taps * cin_w * cout;
    hipLaunchKernelGGL(stem_slab_reduce, dim3(cdiv(nout, 8)), dim3(256), 0, st, workspace, dw, nout, blocks);
    RUNET_CHECK_LAUNCH();
}
