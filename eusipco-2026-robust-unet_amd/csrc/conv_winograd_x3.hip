// Fused Winograd F(2x2, 3x3) convolution with the sixteen position products on the gfx950 BF16 matrix cores: the kernel of conv_winograd.hip
// (forward and data gradient of the high-resolution 3x3 convolutions, /root/reference/Main_Final.py:157,159) with its multiply stage moved
// from v_mfma_f32_32x32x2_f32 (the FP32 vector rate) to the split-operand scheme of gemm_split.hip - x = h + m + l with three bf16 values,
// six v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulation: fp32-accurate at 6/16 of the matrix time.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// Same block (32 tiles = a 4 x 8 patch of 2x2 tiles, 64 output channels), same halo loader, same input transform in registers and same
// LDS-exchange output transform as wino_conv_kernel<0, false>.  What changes:
//   * transform_store splits every transformed value on its way into LDS: V lives as three bf16 planes [plane][xi 16][tile 32][16 k]
//     (32-byte rows with the two 16-byte k-octets swapped on tiles with bit 3 set - the conflict-free ds_read_b128 image of gemm_split.hip);
//   * the filter comes PRE-SPLIT: runet_wino_weights_x3 writes U straight into Up[xi 16][plane 3][K/8][N][8] bf16, so a lane's B fragment
//     (column n, eight consecutive k) is one 16-byte load from L2 per plane, ringed three steps ahead of its use as before;
//   * wave w multiplies positions 4w .. 4w+3: per 16-channel chunk and position [32 tiles x 16 k] x [16 k x 64 n] = 2 x 6 MFMAs of 32 cycles
//     (the f32 form: 16 MFMAs of 64 cycles).
#include "x3_common.h"
#include "../../include/runet_hip.h"
#include "derive_weights.h"
#include <stdlib.h>

// The two scheduling fences around each step's six MFMAs keep the loads of the steps ahead in front of them; compiled out
// (-DSCHED_FENCE\(\)=\(\(void\)0\)) the kernel needs 219 instead of 238 registers but the step is 1.1 % slower (586.9 vs 580.6 img/s, A/B x3).
#ifndef SCHED_FENCE
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
namespace {

using namespace x3;

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct WinoX3Args {
    const float* x; int ldx;      // [Nimg, H, W, ldx], K channels
    const __bf16* U;              // [16][3][K/8][N][8]
    const float* bias;            // [N] or nullptr
    float* y; int ldy;            // [Nimg, H, W, ldy], N channels
    int K, N;
    int Nimg, H, W, TY, TX;       // TY = H/2, TX = W/2
    int accumulate;
    int npatches, nchunks;        // grid = npatches * nchunks blocks (4x8-tile patches x 64-channel output chunks)
    float* stats;                 // nullptr, or [npatches][N][3]: (count, mean, M2) per output channel over the block's pixels (BatchNorm statistics)
};

constexpr int WT = 32;            // tiles per block
constexpr int WBN = 64;           // output channels per block
constexpr int RH = 10, RW = 18;   // input halo of a 4x8 patch of 2x2 tiles
constexpr int RPS = 24;           // floats per halo pixel in LDS (16 channels + pad: conflict-free ds_read_b64 in the transform)
constexpr int VPLANE = 16 * WT * 32;      // bytes of one bf16 plane of V: 16 positions x 32 tiles x 16 k

__global__ __launch_bounds__(256, 2) void wino_conv_x3_kernel(WinoX3Args g) {
    __shared__ __attribute__((aligned(16))) unsigned char Vp[3 * VPLANE];   // 48 KB; reused as M[16][32][16] fp32 in the epilogue
    __shared__ __attribute__((aligned(16))) float R[RH * RW * RPS];         // 17 KB raw input halo of the current 16-channel chunk
    __shared__ float sred[4 * 4 * 8 * 2 * 3];                               // statistics: [pass][wave][channel pair][2][count, mean, M2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    // blocks that share an input patch (same patch, different output-channel chunk) get ids 8 apart: same XCD, the halo is fetched into one L2
    int bpatch, bchunk;
    {
        const int L = blockIdx.x, nch = g.nchunks, span = 8 * nch;
        const int grp = L / span, r = L - grp * span;
        bpatch = grp * 8 + (r & 7);
        bchunk = r >> 3;
        if (grp * 8 + 8 > g.npatches) {
            const int done = grp * 8, rem = g.npatches - done;
            bpatch = done + r % rem;
            bchunk = r / rem;
        }
    }
    const int n0 = bchunk * WBN;
    const int bxs = (g.TX + 7) >> 3, bys = (g.TY + 3) >> 2;
    const int bimg = bpatch / (bxs * bys);
    const int brem = bpatch - bimg * (bxs * bys);
    const int by = brem / bxs, bx = brem - by * bxs;
    const int h00 = 8 * by - 1, w00 = 16 * bx - 1;                // image coordinates of halo pixel (0,0)

    // ---- halo loader: item = (halo pixel, 4-channel group); 180 x 4 = 720 items, 3 per thread; every read unconditional ----
    int hoff[3], hlds[3];
    bool hok[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const int item = tid + 256 * v;
        const int px = item >> 2, c4 = item & 3;
        const int hy = px / RW, hx = px - hy * RW;
        const int ih = h00 + hy, iw = w00 + hx;
        const bool in_tile = px < RH * RW;
        hok[v] = in_tile && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        hoff[v] = hok[v] ? (int)((((long)bimg * g.H + ih) * g.W + iw) * g.ldx) + c4 * 4 : 0;
        hlds[v] = in_tile ? px * RPS + c4 * 4 : -1;
    }
    f32x4 hreg[3];
    auto load_halo = [&](int c0) {
        const float* xc = g.x + c0;
#pragma unroll
        for (int v = 0; v < 3; ++v) hreg[v] = *reinterpret_cast<const f32x4*>(xc + hoff[v]);
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int v = 0; v < 3; ++v)
            if (hlds[v] >= 0) *reinterpret_cast<f32x4*>(&R[hlds[v]]) = hok[v] ? hreg[v] : f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- transform role: (tile lt, channel pair k2); B^T d B in registers, then split into the three planes ----
    const int lt = tid >> 3, k2 = tid & 7;
    const int rbase = ((2 * (lt >> 3)) * RW + 2 * (lt & 7)) * RPS + 2 * k2;
    const int vbase = (lt * 2 + ((k2 >> 2) ^ ((lt >> 3) & 1))) * 16 + (k2 & 3) * 4;      // bytes inside a position's [32 tiles][32 B] image
    auto put = [&](int xi, const float2 v) {
        const bf16x2 h = {(__bf16)v.x, (__bf16)v.y};
        const float rx = v.x - (float)h[0], ry = v.y - (float)h[1];
        const bf16x2 m = {(__bf16)rx, (__bf16)ry};
        const bf16x2 l = {(__bf16)(rx - (float)m[0]), (__bf16)(ry - (float)m[1])};
        unsigned char* d = Vp + xi * (WT * 32) + vbase;
        *reinterpret_cast<bf16x2*>(d) = h;
        *reinterpret_cast<bf16x2*>(d + VPLANE) = m;
        *reinterpret_cast<bf16x2*>(d + 2 * VPLANE) = l;
    };
    auto transform_store = [&]() {
        float2 raw[16], tmp[16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) raw[a * 4 + b] = *reinterpret_cast<const float2*>(&R[rbase + (a * RW + b) * RPS]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float2 d0 = raw[b], d1 = raw[4 + b], d2 = raw[8 + b], d3 = raw[12 + b];
            tmp[b] = make_float2(d0.x - d2.x, d0.y - d2.y);
            tmp[4 + b] = make_float2(d1.x + d2.x, d1.y + d2.y);
            tmp[8 + b] = make_float2(d2.x - d1.x, d2.y - d1.y);
            tmp[12 + b] = make_float2(d1.x - d3.x, d1.y - d3.y);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float2 v0 = tmp[a * 4], v1 = tmp[a * 4 + 1], v2 = tmp[a * 4 + 2], v3 = tmp[a * 4 + 3];
            put(a * 4 + 0, make_float2(v0.x - v2.x, v0.y - v2.y));
            put(a * 4 + 1, make_float2(v1.x + v2.x, v1.y + v2.y));
            put(a * 4 + 2, make_float2(v2.x - v1.x, v2.y - v1.y));
            put(a * 4 + 3, make_float2(v1.x - v3.x, v1.y - v3.y));
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.f;

    // B fragments: 16-byte units of Up; step t = (position xl = t >> 1, column tile b = t & 1) needs the three planes of (xi, k-octet lh, column)
    const bf16x8* U16 = reinterpret_cast<const bf16x8*>(g.U);
    const int K8 = g.K >> 3;
    const int nchunks = g.K >> 4;
    int uoff[2];              // columns n >= N read unit 0 instead: their products land in output columns that are never stored
#pragma unroll
    for (int b = 0; b < 2; ++b) uoff[b] = (n0 + b * 32 + li < g.N) ? lh * g.N + n0 + b * 32 + li : 0;
    bf16x8 ring[4][3];        // step t lives in slot t & 3, loaded three steps (18 MFMAs) ahead, across the chunk's barriers too
    auto load_step = [&](int t, int c0, bf16x8 (&slot)[3]) {
        const int xi = wid * 4 + (t >> 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) slot[p] = U16[(long)((xi * 3 + p) * K8 + (c0 >> 3)) * g.N + uoff[t & 1]];      // scalar base + lane offset
    };
    const int a_rd = (li * 2 + (lh ^ ((li >> 3) & 1))) * 16;
    auto read_a = [&](int xl, bf16x8 (&a)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(Vp + p * VPLANE + (wid * 4 + xl) * (WT * 32) + a_rd);
    };

    load_halo(0);
    load_step(0, 0, ring[0]);
    load_step(1, 0, ring[1]);
    load_step(2, 0, ring[2]);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * 16;
        const int cn = (ch + 1 < nchunks) ? c0 + 16 : c0;      // last chunk: harmless re-read, keeps the code branch-free
        store_halo();                          // R was last read before the previous chunk's second barrier
        __syncthreads();                       // R complete; previous chunk's MFMA reads of V are done
        transform_store();
        __syncthreads();
        bf16x8 af[2][3];
        read_a(0, af[0]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t + 3 < 8) load_step(t + 3, c0, ring[(t + 3) & 3]);
            else load_step(t + 3 - 8, cn, ring[(t + 3) & 3]);          // first steps of the NEXT chunk, in flight across the barriers
            if (t == 1) load_halo(cn);                                 // next chunk's halo: in flight during the rest of the MFMAs
            if ((t & 1) == 0 && t + 2 < 8) read_a((t >> 1) + 1, af[((t >> 1) + 1) & 1]);      // next position's A fragments
            SCHED_FENCE();
            X3_MMA(acc[t >> 1][t & 1], af[(t >> 1) & 1], ring[t & 3]);
            SCHED_FENCE();
        }
    }

    // ---- epilogue: exchange the 16 positions through LDS (16 output channels per pass), A^T m A, store ----
    float* M = reinterpret_cast<float*>(Vp);    // M[xi][t][16]
    const int et = tid >> 3;                    // tile of the two (tile, channel) items this thread finishes
    const int en = (tid & 7) * 2;               // channels en, en+1 of the pass
    const int e_ty = 4 * by + (et >> 3), e_tx = 8 * bx + (et & 7);
    const bool e_ok = e_ty < g.TY && e_tx < g.TX;
    const int e_h = 2 * e_ty, e_w = 2 * e_tx;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        __syncthreads();
        const int b = pass >> 1;                // which 32-wide N tile
        if ((li >> 4) == (pass & 1)) {          // lanes whose column falls into this 16-channel pass
#pragma unroll
            for (int xl = 0; xl < 4; ++xl)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    M[((wid * 4 + xl) * WT + t) * 16 + (li & 15)] = acc[xl][b][r];
                }
        }
        __syncthreads();
        float2 px[4];
        float npx = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) px[q] = make_float2(0.f, 0.f);
        if (e_ok) {
            float2 m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = *reinterpret_cast<const float2*>(&M[(xi * WT + et) * 16 + en]);
            float2 s[2][4];                     // A^T m A : rows [1,1,1,0],[0,1,-1,-1]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[0][j] = make_float2(m[j].x + m[4 + j].x + m[8 + j].x, m[j].y + m[4 + j].y + m[8 + j].y);
                s[1][j] = make_float2(m[4 + j].x - m[8 + j].x - m[12 + j].x, m[4 + j].y - m[8 + j].y - m[12 + j].y);
            }
            const int nch = n0 + pass * 16 + en;
            if (nch < g.N) {
                float2 bv = make_float2(0.f, 0.f);
                if (g.bias) bv = *reinterpret_cast<const float2*>(g.bias + nch);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float2 o0 = make_float2(s[i][0].x + s[i][1].x + s[i][2].x + bv.x, s[i][0].y + s[i][1].y + s[i][2].y + bv.y);
                    float2 o1 = make_float2(s[i][1].x - s[i][2].x - s[i][3].x + bv.x, s[i][1].y - s[i][2].y - s[i][3].y + bv.y);
                    float* d0 = g.y + (((long)bimg * g.H + e_h + i) * g.W + e_w) * g.ldy + nch;
                    float* d1 = d0 + g.ldy;
                    if (g.accumulate) {
                        const float2 p0 = *reinterpret_cast<const float2*>(d0), p1 = *reinterpret_cast<const float2*>(d1);
                        o0.x += p0.x; o0.y += p0.y; o1.x += p1.x; o1.y += p1.y;
                    }
                    *reinterpret_cast<float2*>(d0) = o0;
                    *reinterpret_cast<float2*>(d1) = o1;
                    px[2 * i] = o0; px[2 * i + 1] = o1;
                }
                npx = 4.f;
            }
        }
        if (g.stats) {
            // BatchNorm statistics of the four pixels x two channels this thread just stored: (count, mean, M2) in registers, Chan-combined
            // over the eight tiles of the wave that share the channel pair (lanes 8, 16, 32 apart), one row per wave into LDS
            float cnt = npx;
            float2 mean = make_float2(0.f, 0.f), m2 = make_float2(0.f, 0.f);
            if (npx > 0.f) {
                mean = make_float2(0.25f * (px[0].x + px[1].x + px[2].x + px[3].x), 0.25f * (px[0].y + px[1].y + px[2].y + px[3].y));
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float dx_ = px[q].x - mean.x, dy_ = px[q].y - mean.y; m2.x += dx_ * dx_; m2.y += dy_ * dy_; }
            }
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                const float c2 = __shfl_xor(cnt, o, 64);
                const float2 mo = make_float2(__shfl_xor(mean.x, o, 64), __shfl_xor(mean.y, o, 64));
                const float2 qo = make_float2(__shfl_xor(m2.x, o, 64), __shfl_xor(m2.y, o, 64));
                const float nt = cnt + c2;
                if (nt > 0.f) {
                    const float f1 = c2 / nt, f2 = cnt * c2 / nt;
                    const float ddx = mo.x - mean.x, ddy = mo.y - mean.y;
                    m2 = make_float2(m2.x + qo.x + ddx * ddx * f2, m2.y + qo.y + ddy * ddy * f2);
                    mean = make_float2(mean.x + ddx * f1, mean.y + ddy * f1);
                }
                cnt = nt;
            }
            if (lane < 8) {
                float* o = &sred[((pass * 4 + wid) * 8 + lane) * 6];
                o[0] = cnt; o[1] = mean.x; o[2] = m2.x; o[3] = cnt; o[4] = mean.y; o[5] = m2.y;
            }
        }
    }
    if (g.stats) {
        __syncthreads();
        if (tid < 64 && n0 + tid < g.N) {           // channel n0 + tid = pass * 16 + pair * 2 + ch: the four waves' rows in wave order
            const int pass = tid >> 4, pair = (tid & 15) >> 1, chn = tid & 1;
            float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                const float* a = &sred[(((pass * 4 + wv) * 8 + pair) * 2 + chn) * 3];
                const float nt = cnt + a[0], dlt = a[1] - mean;
                if (nt > 0.f) { m2 = m2 + a[2] + dlt * dlt * (cnt * a[0] / nt); mean = mean + dlt * (a[0] / nt); }
                cnt = nt;
            }
            float* o = g.stats + ((long)bpatch * g.N + n0 + tid) * 3;
            o[0] = cnt; o[1] = mean; o[2] = m2;
        }
    }
}

// Up[xi][plane][K/8][N][8] = split(G g G^T);  forward: g[r][s] = w[r][s][k][n];  dgrad: g[r][s] = w[2-r][2-s][n][k].  thread = (k octet, n)
__global__ __launch_bounds__(256) void wino_weight_x3_kernel(const float* __restrict__ w, __bf16* __restrict__ Up, int cin, int cout, int dgrad) {
    derive::wino_weight_x3_body(w, Up, cin, cout, dgrad, blockIdx.x);
}

}  // namespace

extern "C" long runet_wino_x3_pack_elems(int k, int n) { return 16L * 3 * k * n; }

extern "C" int runet_wino_weights_x3(const float* w_hwio, void* Upacked, int cin, int cout, int dgrad, void* stream) {
    const int k = dgrad ? cout : cin, n = dgrad ? cin : cout;
    RUNET_REQUIRE(w_hwio && Upacked && cin > 0 && cout > 0 && k % 8 == 0, "bad arguments (the contraction side must be a multiple of 8 channels)");
    RUNET_REQUIRE(((uintptr_t)Upacked % 16) == 0, "alignment");
    hipLaunchKernelGGL(wino_weight_x3_kernel, dim3(cdiv((long)(k / 8) * n, 256)), dim3(256), 0, (hipStream_t)stream, w_hwio, (__bf16*)Upacked, cin, cout, dgrad);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_wino_conv_x3_stats_parts(int n_img, int h, int w) { return n_img * cdiv(h / 2, 4) * cdiv(w / 2, 8); }

static int wino_conv_x3_launch(const float* x, int ldx, const void* Upacked, const float* bias, float* y, int ldy, int n_img, int h, int w, int k,
                               int n, int accumulate, float* stats, void* stream);

extern "C" int runet_wino_conv_x3(const float* x, int ldx, const void* Upacked, const float* bias, float* y, int ldy, int n_img, int h, int w, int k,
                                  int n, int accumulate, void* stream) {
    return wino_conv_x3_launch(x, ldx, Upacked, bias, y, ldy, n_img, h, w, k, n, accumulate, nullptr, stream);
}

// runet_wino_conv_x3 that also leaves the BatchNorm statistics partials of its output behind: stats [runet_wino_conv_x3_stats_parts][n][3]
extern "C" int runet_wino_conv_x3_stats(const float* x, int ldx, const void* Upacked, const float* bias, float* y, int ldy, int n_img, int h, int w,
                                        int k, int n, int accumulate, float* stats, void* stream) {
    RUNET_REQUIRE(stats, "stats must not be NULL");
    return wino_conv_x3_launch(x, ldx, Upacked, bias, y, ldy, n_img, h, w, k, n, accumulate, stats, stream);
}

static int wino_conv_x3_launch(const float* x, int ldx, const void* Upacked, const float* bias, float* y, int ldy, int n_img, int h, int w, int k,
                               int n, int accumulate, float* stats, void* stream) {
    RUNET_REQUIRE(x && Upacked && y, "null pointer");
    RUNET_REQUIRE(runet_wino_supported(h, w, k, n), "shape not supported by the Winograd kernel (H, W even; K multiple of 16; N even)");
    RUNET_REQUIRE(ldx >= k && ldx % 4 == 0 && ldy >= n && ldy % 2 == 0, "pixel strides must cover the channels (ldx: multiple of 4, ldy: even)");
    RUNET_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)Upacked % 16) == 0 && (!bias || ((uintptr_t)bias % 8) == 0), "alignment");
    RUNET_REQUIRE(runet_wino_fits(n_img, h, w, ldx, ldy, k, n), "tensor too large for 32-bit offsets (runet_wino_fits)");
    WinoX3Args a{};
    a.x = x; a.ldx = ldx; a.U = (const __bf16*)Upacked; a.bias = bias; a.y = y; a.ldy = ldy; a.K = k; a.N = n;
    a.Nimg = n_img; a.H = h; a.W = w; a.TY = h / 2; a.TX = w / 2; a.accumulate = accumulate;
    a.npatches = n_img * cdiv(a.TY, 4) * cdiv(a.TX, 8);
    a.nchunks = cdiv(n, WBN);
    a.stats = stats;
    hipLaunchKernelGGL(wino_conv_x3_kernel, dim3(a.npatches * a.nchunks), dim3(256), 0, (hipStream_t)stream, a);
    RUNET_CHECK_LAUNCH();
}
