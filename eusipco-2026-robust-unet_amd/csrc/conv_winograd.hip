// Fused Winograd F(2x2, 3x3) convolution on the fp32 matrix cores: forward and data-gradient of the 3x3 / dilation-1 /
// 'same' convolutions of the ResidualBlocks (/root/reference/Main_Final.py:157,159 and their autograd), which are 89 % of
// the network's FLOPs.  2.25x fewer multiplies than the direct implicit GEMM at fp32 accuracy (rounding differs by a few ulp).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// * `wino_weight_kernel` transforms the HWIO weight once per use into U[16][K/4][N][4] (forward: K = cin, N = cout; data
//   gradient: the 180-degree rotated filter with K = cout, N = cin).
// * `wino_conv_kernel`: one block = 32 tiles (one 32-row MFMA tile) x 64 output channels.  Per 16-channel chunk every thread
//   loads the 4x4 patch of one (tile, channel pair) straight from NHWC global memory (8 lanes cover the 64 contiguous bytes
//   of a pixel), applies B^T d B in registers and writes the 16 transformed values to LDS as V[xi][tile][k]; after a barrier
//   wave w multiplies positions xi = 4w..4w+3 : [32 tiles x 16 k] x [16 k x 64 n] with v_mfma_f32_32x32x2_f32, the U
//   fragments coming straight from global/L2 (each wave needs different positions, so LDS staging would buy no reuse).
//   Default multiply stage (M16): v_mfma_f32_16x16x4_f32, wave w owns 16 of the 64 output channels for all 16 positions, so
//   the 16 position-products of a (tile, channel) sit in one lane and A^T m A runs in registers.  The earlier form (M16 = false,
//   RUNET_WINO_ABL=32) splits the positions over the waves and exchanges them through LDS in four 16-channel passes.
// Algorithmic FLOPs are counted as the direct convolution's (2*9*Cin*Cout per pixel); the MFMA work is 16/36 of that.
#include "runet_common.h"
#include "../../include/runet_hip.h"
#include <stdlib.h>

namespace {

struct WinoArgs {
    const float* x; int ldx;      // [Nimg, H, W, ldx], K channels
    const float* U;               // [16][K/4][N][4]
    const float* bias;            // [N] or nullptr
    float* y; int ldy;            // [Nimg, H, W, ldy], N channels
    int K, N;
    int Nimg, H, W, TY, TX;       // TY = H/2, TX = W/2
    long tiles;                   // Nimg*TY*TX
    int accumulate;
    int npatches, nchunks;        // grid = npatches * nchunks blocks (4x8-tile patches x 64-channel output chunks)
};

constexpr int WT = 32;            // tiles per block
constexpr int WBN = 64;           // output channels per block
constexpr int VLD = 20;           // padded k-stride of a V row (conflict-free ds_read_b128)
constexpr int RH = 10, RW = 18;   // input halo of a 4x8 patch of 2x2 tiles
constexpr int RPS = 24;           // floats per halo pixel in LDS (16 channels + pad: conflict-free ds_read_b64 in the transform)

// M16 = true: the multiply stage uses v_mfma_f32_16x16x4_f32 and wave w owns output channels [16w, 16w+16) of the block for ALL 16
// positions (32 accumulator tiles of 4 registers).  A lane then holds the 16 position-products of its (tile, channel) items, so the
// output transform runs in registers - no exchange through LDS, no barriers in the epilogue - at the price of every wave reading the
// whole V tile (4x the LDS fragment reads of the 32x32x2 form, where a wave reads only its 4 positions).
template <int ABL, bool M16>
__global__ __launch_bounds__(256, 2) void wino_conv_kernel(WinoArgs g) {
    __shared__ __attribute__((aligned(16))) float V[16 * WT * VLD];       // 40 KB; reused as M[16][32][16] in the epilogue
    __shared__ __attribute__((aligned(16))) float R[RH * RW * RPS];       // 17 KB raw input halo of the current 16-channel chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: U offsets stay in SGPRs
    const int li = lane & 31, lh = lane >> 5;
    // 1-D grid.  Blocks that share an input patch (same patch, different output-channel chunk) get ids 8 apart: consecutive ids are
    // dealt to the 8 XCDs round-robin, so they run at the same time on the SAME XCD and the halo is fetched into one L2 once.
    int bpatch, bchunk;
    {
        const int L = blockIdx.x, nch = g.nchunks, span = 8 * nch;
        const int grp = L / span, r = L - grp * span;
        bpatch = grp * 8 + (r & 7);
        bchunk = r >> 3;
        if (grp * 8 + 8 > g.npatches) {      // tail group (npatches not a multiple of 8): plain order over what is left
            const int done = grp * 8, rem = g.npatches - done;
            bpatch = done + r % rem;
            bchunk = r / rem;
        }
    }
    const int n0 = bchunk * WBN;

    // block -> (image, 4x8 patch of tiles): 8x16 output pixels, 10x18 input halo shared by the 32 tiles through LDS
    const int bxs = (g.TX + 7) >> 3, bys = (g.TY + 3) >> 2;
    const int bimg = bpatch / (bxs * bys);
    const int brem = bpatch - bimg * (bxs * bys);
    const int by = brem / bxs, bx = brem - by * bxs;
    const int h00 = 8 * by - 1, w00 = 16 * bx - 1;                // image coordinates of halo pixel (0,0)

    // Every global read is UNCONDITIONAL (a masked lane reads element 0 and the value is then zeroed): no branches around
    // loads, so the compiler's vmcnt bookkeeping stays exact and the prefetches really overlap the MFMAs.
    // ---- halo loader: item = (halo pixel, 4-channel group); 180 x 4 = 720 items, 3 per thread ----
    int hoff[3];                  // element offset (0 when the pixel is outside the image / item is padding)
    int hlds[3];                  // LDS float index, -1 = no item
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
        const float* xc = g.x + c0;            // wave-uniform
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            if constexpr (ABL & 1) hreg[v] = f32x4{(float)c0, 1.f, 2.f, 3.f};
            else hreg[v] = *reinterpret_cast<const f32x4*>(xc + hoff[v]);
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int v = 0; v < 3; ++v)
            if (hlds[v] >= 0) *reinterpret_cast<f32x4*>(&R[hlds[v]]) = hok[v] ? hreg[v] : f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- transform role: (tile lt, channel pair k2); the wave's 8 tiles are one row of the 4x8 patch ----
    const int lt = tid >> 3, k2 = tid & 7;
    const int rbase = ((2 * (lt >> 3)) * RW + 2 * (lt & 7)) * RPS + 2 * k2;      // halo pixel (0,0) of this tile's 4x4 patch
    auto transform_store = [&]() {
        // B^T d B on both channels; V[xi][lt][2*k2 .. 2*k2+1]
        float2 raw[16], tmp[16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) raw[a * 4 + b] = *reinterpret_cast<const float2*>(&R[rbase + (a * RW + b) * RPS]);
        if constexpr (ABL & 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) *reinterpret_cast<float2*>(&V[(i * WT + lt) * VLD + 2 * k2]) = raw[i];
            return;
        }
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
            float2 o[4];
            o[0] = make_float2(v0.x - v2.x, v0.y - v2.y);
            o[1] = make_float2(v1.x + v2.x, v1.y + v2.y);
            o[2] = make_float2(v2.x - v1.x, v2.y - v1.y);
            o[3] = make_float2(v1.x - v3.x, v1.y - v3.y);
#pragma unroll
            for (int b = 0; b < 4; ++b)
                *reinterpret_cast<float2*>(&V[((a * 4 + b) * WT + lt) * VLD + 2 * k2]) = o[b];
        }
    };

    // U is stored [16][K/4][N][4]: lane (j, h) reads ONE float4 = the 4 consecutive k it feeds to 4 MFMAs
    const int K4 = g.K >> 2;
    const f32x4* U4 = reinterpret_cast<const f32x4*>(g.U);
    const int nchunks = g.K >> 4;

    if constexpr (M16) {
        const int l16 = lane & 15, kq = lane >> 4;
        const int ncol = n0 + wid * 16 + l16;
        const int uo = (ncol < g.N) ? kq * g.N + wid * 16 + l16 : 0;      // column >= N: reads element 0, result never stored
        f32x4 acc16[16][2];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc16[xi][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 ring16[8];          // filter fragment of step t (= position t of the chunk) lives in slot t & 7, loaded 6 steps ahead
        auto load16 = [&](int xi, int c0, f32x4& slot) { slot = U4[((long)(xi * K4 + (c0 >> 2)) * g.N + n0) + uo]; };
        auto read_a16 = [&](int xi, f32x4 (&a)[2]) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) a[mb] = *reinterpret_cast<const f32x4*>(&V[(xi * WT + mb * 16 + l16) * VLD + 4 * kq]);
        };
        load_halo(0);
#pragma unroll
        for (int t = 0; t < 6; ++t) load16(t, 0, ring16[t]);
        for (int ch = 0; ch < nchunks; ++ch) {
            const int c0 = ch * 16;
            const int cn = (ch + 1 < nchunks) ? c0 + 16 : c0;
            store_halo();
            __syncthreads();
            transform_store();
            __syncthreads();
            f32x4 af[2][2];
            read_a16(0, af[0]);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (t + 6 < 16) load16(t + 6, c0, ring16[(t + 6) & 7]);
                else load16(t + 6 - 16, cn, ring16[(t + 6) & 7]);
                if (t == 1) load_halo(cn);
                if (t + 1 < 16) read_a16(t + 1, af[(t + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        acc16[t][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t & 1][mb][j], ring16[t & 7][j], acc16[t][mb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue in registers: lane = (channel ncol, tiles mb*16 + 4*kq + r)
        float bv = 0.f;
        if (g.bias && ncol < g.N) bv = g.bias[ncol];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tile = mb * 16 + 4 * kq + r;
                const int e_ty = 4 * by + (tile >> 3), e_tx = 8 * bx + (tile & 7);
                if (e_ty < g.TY && e_tx < g.TX && ncol < g.N) {
                    float m[16];
#pragma unroll
                    for (int xi = 0; xi < 16; ++xi) m[xi] = acc16[xi][mb][r];
                    float sA[2][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sA[0][j] = m[j] + m[4 + j] + m[8 + j];
                        sA[1][j] = m[4 + j] - m[8 + j] - m[12 + j];
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float o0 = sA[i][0] + sA[i][1] + sA[i][2] + bv;
                        float o1 = sA[i][1] - sA[i][2] - sA[i][3] + bv;
                        float* d0 = g.y + (((long)bimg * g.H + 2 * e_ty + i) * g.W + 2 * e_tx) * g.ldy + ncol;
                        if (g.accumulate) { o0 += d0[0]; o1 += d0[g.ldy]; }
                        d0[0] = o0;
                        d0[g.ldy] = o1;
                    }
                }
            }
        return;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.f;

    int uoff[2];              // columns n >= N read element 0 instead: their products land in output columns that are never stored
#pragma unroll
    for (int b = 0; b < 2; ++b) uoff[b] = (n0 + b * 32 + li < g.N) ? lh * g.N + b * 32 + li : 0;
    // U fragments travel through a ring of four half-position slots (one slot = the two float4 of one (position, k-half)):
    // the load for step t+3 is issued in front of step t's 8 MFMAs, i.e. every fragment has ~1500 MFMA cycles to arrive,
    // with the same 32 registers a plain double buffer of whole positions would use.
    f32x4 ring[4][2];
    auto load_step = [&](int t, int c0, f32x4 (&slot)[2]) {
        const int xl = t >> 1, kh = t & 1;
        const f32x4* up = U4 + ((long)((wid * 4 + xl) * K4 + (c0 >> 2) + kh * 2) * g.N + n0);      // wave-uniform
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if constexpr (ABL & 4) slot[b] = f32x4{1.f, 2.f, 3.f, (float)c0};
            else slot[b] = up[uoff[b]];
        }
    };
    // the A fragment (one ds_read_b128) of step t+1 is read in front of step t's MFMAs: only the first read after the barrier is exposed
    auto read_a = [&](int t) {
        const int xl = t >> 1, kh = t & 1;
        return *reinterpret_cast<const f32x4*>(&V[((wid * 4 + xl) * WT + li) * VLD + kh * 8 + 4 * lh]);
    };
    auto mma_step = [&](int t, const f32x4 a, const f32x4 (&slot)[2]) {
        const int xl = t >> 1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[xl][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], slot[b][q], acc[xl][b], 0, 0, 0);
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
        // the sched_barriers pin each load group in FRONT of the MFMAs it overlaps (the scheduler otherwise sinks loads to the
        // end of the region, right in front of their first use)
        f32x4 afrag[2];
        afrag[0] = read_a(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t + 3 < 8) load_step(t + 3, c0, ring[(t + 3) & 3]);
            else load_step(t + 3 - 8, cn, ring[(t + 3) & 3]);          // first steps of the NEXT chunk, in flight across the barriers
            if (t == 1) load_halo(cn);                                 // next chunk's halo: in flight during the rest of the MFMAs
            if (t + 1 < 8) afrag[(t + 1) & 1] = read_a(t + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(t, afrag[t & 1], ring[t & 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: exchange the 16 positions through LDS (16 output channels per pass), A^T m A, store ----
    if constexpr (ABL & 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b) asm volatile("" :: "v"(acc[i][b]));
        return;
    }
    float* M = V;                               // M[xi][t][16]
    const int et = tid >> 3;                    // tile of the two (tile, channel) items this thread finishes
    const int en = (tid & 7) * 2;               // channels en, en+1 of the pass
    const int e_ty = 4 * by + (et >> 3), e_tx = 8 * bx + (et & 7);
    const bool e_ok = e_ty < g.TY && e_tx < g.TX;
    const int e_n = bimg, e_h = 2 * e_ty, e_w = 2 * e_tx;
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
        if (e_ok) {
            float2 m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = *reinterpret_cast<const float2*>(&M[(xi * WT + et) * 16 + en]);
            // A^T m A : rows [1,1,1,0],[0,1,-1,-1]
            float2 s[2][4];
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
                    float* d0 = g.y + (((long)e_n * g.H + e_h + i) * g.W + e_w) * g.ldy + nch;
                    float* d1 = d0 + g.ldy;
                    if (g.accumulate) {
                        const float2 p0 = *reinterpret_cast<const float2*>(d0), p1 = *reinterpret_cast<const float2*>(d1);
                        o0.x += p0.x; o0.y += p0.y; o1.x += p1.x; o1.y += p1.y;
                    }
                    *reinterpret_cast<float2*>(d0) = o0;
                    *reinterpret_cast<float2*>(d1) = o1;
                }
            }
        }
    }
}

// U[xi][k][n] = (G g G^T)[xi] ;  forward: g[r][s] = w[r][s][k][n] ;  dgrad: g[r][s] = w[2-r][2-s][n][k]  (w is [3][3][cin][cout])
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int cin, int cout, int dgrad) {
    const long total = (long)cin * cout;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i / cout), co = (int)(i - (long)ci * cout);
    float gm[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) gm[r][s] = dgrad ? w[((long)((2 - r) * 3 + (2 - s)) * cin + ci) * cout + co] : w[((long)(r * 3 + s) * cin + ci) * cout + co];
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        t[0][s] = gm[0][s];
        t[1][s] = 0.5f * (gm[0][s] + gm[1][s] + gm[2][s]);
        t[2][s] = 0.5f * (gm[0][s] - gm[1][s] + gm[2][s]);
        t[3][s] = gm[2][s];
    }
    const int K = dgrad ? cout : cin, N = dgrad ? cin : cout;
    const int k = dgrad ? co : ci, n = dgrad ? ci : co;
    const long kn = ((long)(k >> 2) * N + n) * 4 + (k & 3);          // blocked layout [K/4][N][4]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
        U[(long)(a * 4 + 0) * K * N + kn] = u0;
        U[(long)(a * 4 + 1) * K * N + kn] = u1;
        U[(long)(a * 4 + 2) * K * N + kn] = u2;
        U[(long)(a * 4 + 3) * K * N + kn] = u3;
    }
}


// ------------------------------------------------------------------------------------------ Winograd weight gradient
//   dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G        (same 2.25x saving as the forward)
// Block = 32 input channels x 64 output channels, all 16 positions (wave w owns positions 4w..4w+3: 8 accumulator tiles).
// Per batch of 8 tiles each thread transforms the 4x4 x patch of one (tile, cin) and the 2x2 dy tile of two (tile, cout) in
// registers and writes V[xi][tile][cin], Z[xi][tile][cout] to LDS; the MFMA K dimension is the tile index.  The block walks its
// range of tiles with the accumulators in registers and writes one Winograd-domain slab [16][cin][cout];
// `wino_wgrad_reduce_kernel` sums the slabs in a fixed order and applies G^T . G.
struct WinoWgradArgs {
    const float* x; int ldx;      // [Nimg, H, W, ldx]
    const float* dy; int ldy;     // [Nimg, H, W, ldy]
    float* slabs;                 // [splits][16][cin][cout]
    int cin, cout, Nimg, H, W, TY, TX, co_chunks, nchunks, nsplits;
    long tiles, tiles_per_split;
};

constexpr int GT = 8;             // tiles per batch
constexpr int GCI = 32, GCO = 64;

__global__ __launch_bounds__(256, 2) void wino_wgrad_kernel(WinoWgradArgs g) {
    __shared__ __attribute__((aligned(16))) float Vs[16 * GT * GCI];    // 16 KB
    __shared__ __attribute__((aligned(16))) float Zs[16 * GT * GCO];    // 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    // 1-D grid, XCD-aware: the (cin-chunk, cout-chunk) blocks of one tile range get ids 8 apart -> same XCD, same time: x and dy of the
    // range go through one L2 (PMC: 1.2 GB of HBM traffic per launch against 0.54-0.8 GB algorithmic with the plain order)
    int bsplit, bchunk;
    {
        const int L = blockIdx.x, nch = g.nchunks, span = 8 * nch;
        const int grp = L / span, r = L - grp * span;
        bsplit = grp * 8 + (r & 7);
        bchunk = r >> 3;
        if (grp * 8 + 8 > g.nsplits) {
            const int done = grp * 8, rem = g.nsplits - done;
            bsplit = done + r % rem;
            bchunk = r / rem;
        }
    }
    const int ci0 = (bchunk / g.co_chunks) * GCI, co0 = (bchunk % g.co_chunks) * GCO;
    const long t_begin = (long)bsplit * g.tiles_per_split;
    const long t_end = t_begin + g.tiles_per_split < g.tiles ? t_begin + g.tiles_per_split : g.tiles;
    const int per = g.TY * g.TX;
    const int rowx = g.W * g.ldx, rowy = g.W * g.ldy;

    const int xt = tid >> 5, xc = tid & 31;            // x loader: (tile, cin)
    const bool xc_ok = ci0 + xc < g.cin;
    float rx[16], ry[2][4];
    // tile coordinates of this thread's three loader items, advanced incrementally (GT tiles per batch): no divisions in the loop
    struct TileIt { int n, ty, tx; };
    auto tile_init = [&](long tg) {
        TileIt t;
        const long tt = tg < g.tiles ? tg : g.tiles - 1;
        t.n = (int)(tt / per);
        const int rem = (int)(tt - (long)t.n * per);
        t.ty = rem / g.TX; t.tx = rem - t.ty * g.TX;
        return t;
    };
    auto tile_next = [&](TileIt& t) {
        t.tx += GT;
        while (t.tx >= g.TX) { t.tx -= g.TX; if (++t.ty == g.TY) { t.ty = 0; ++t.n; } }
    };
    TileIt itx = tile_init(t_begin + xt), ity0 = tile_init(t_begin + (tid >> 6)), ity1 = tile_init(t_begin + ((tid + 256) >> 6));
    auto prefetch = [&](long tb) {
        {   // x patch of tile tb + xt
            const bool tv = tb + xt < t_end && xc_ok;
            const int h0 = 2 * itx.ty - 1, w0 = 2 * itx.tx - 1;
            const float* base = g.x + (((long)itx.n * g.H + h0) * g.W + w0) * g.ldx + ci0 + xc;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const bool ok = tv && (unsigned)(h0 + a) < (unsigned)g.H && (unsigned)(w0 + b) < (unsigned)g.W;
                    rx[a * 4 + b] = ok ? base[a * rowx + b * g.ldx] : 0.f;
                }
            tile_next(itx);
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {   // dy tiles: items (tile, cout) = tid, tid + 256
            const int idx = tid + 256 * v;
            const int yt = idx >> 6, yc = idx & 63;
            TileIt& it = v ? ity1 : ity0;
            const bool tv = tb + yt < t_end && co0 + yc < g.cout;
            const float* base = g.dy + (((long)it.n * g.H + 2 * it.ty) * g.W + 2 * it.tx) * g.ldy + co0 + yc;
            ry[v][0] = tv ? base[0] : 0.f;
            ry[v][1] = tv ? base[g.ldy] : 0.f;
            ry[v][2] = tv ? base[rowy] : 0.f;
            ry[v][3] = tv ? base[rowy + g.ldy] : 0.f;
            tile_next(it);
        }
    };
    auto transform_store = [&]() {
        float tmp[16];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float d0 = rx[b], d1 = rx[4 + b], d2 = rx[8 + b], d3 = rx[12 + b];
            tmp[b] = d0 - d2; tmp[4 + b] = d1 + d2; tmp[8 + b] = d2 - d1; tmp[12 + b] = d1 - d3;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float v0 = tmp[a * 4], v1 = tmp[a * 4 + 1], v2 = tmp[a * 4 + 2], v3 = tmp[a * 4 + 3];
            Vs[((a * 4 + 0) * GT + xt) * GCI + xc] = v0 - v2;
            Vs[((a * 4 + 1) * GT + xt) * GCI + xc] = v1 + v2;
            Vs[((a * 4 + 2) * GT + xt) * GCI + xc] = v2 - v1;
            Vs[((a * 4 + 3) * GT + xt) * GCI + xc] = v1 - v3;
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int idx = tid + 256 * v;
            const int yt = idx >> 6, yc = idx & 63;
            const float y00 = ry[v][0], y01 = ry[v][1], y10 = ry[v][2], y11 = ry[v][3];
            // Z = A Y A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
            const float r[4][2] = {{y00, y01}, {y00 + y10, y01 + y11}, {y00 - y10, y01 - y11}, {-y10, -y11}};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float p = r[a][0], q = r[a][1];
                Zs[((a * 4 + 0) * GT + yt) * GCO + yc] = p;
                Zs[((a * 4 + 1) * GT + yt) * GCO + yc] = p + q;
                Zs[((a * 4 + 2) * GT + yt) * GCO + yc] = p - q;
                Zs[((a * 4 + 3) * GT + yt) * GCO + yc] = -q;
            }
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.f;

    if (t_begin < t_end) prefetch(t_begin);
    for (long tb = t_begin; tb < t_end; tb += GT) {
        __syncthreads();
        transform_store();
        if (tb + GT < t_end) prefetch(tb + GT);     // issued before the barrier: in flight during barrier + MFMAs
        __syncthreads();
        // operands of position xl+1 (12 LDS reads) are fetched in front of position xl's 8 MFMAs; without this every MFMA pair waits
        // for the ds_read issued right before it
        float fa[2][GT / 2], fb[2][GT / 2][2];
        auto read_ops = [&](int xl, float (&a)[GT / 2], float (&b)[GT / 2][2]) {
            const int xi = wid * 4 + xl;
#pragma unroll
            for (int s = 0; s < GT / 2; ++s) {
                a[s] = Vs[(xi * GT + 2 * s + lh) * GCI + li];
                b[s][0] = Zs[(xi * GT + 2 * s + lh) * GCO + li];
                b[s][1] = Zs[(xi * GT + 2 * s + lh) * GCO + 32 + li];
            }
        };
        read_ops(0, fa[0], fb[0]);
#pragma unroll
        for (int xl = 0; xl < 4; ++xl) {
            if (xl + 1 < 4) read_ops(xl + 1, fa[(xl + 1) & 1], fb[(xl + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < GT / 2; ++s) {
                acc[xl][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[xl & 1][s], fb[xl & 1][s][0], acc[xl][0], 0, 0, 0);
                acc[xl][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[xl & 1][s], fb[xl & 1][s][1], acc[xl][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    float* slab = g.slabs + (long)bsplit * 16 * g.cin * g.cout;
#pragma unroll
    for (int xl = 0; xl < 4; ++xl)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int co = co0 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci < g.cin && co < g.cout) slab[((long)(wid * 4 + xl) * g.cin + ci) * g.cout + co] = acc[xl][b][r];
            }
        }
}

// ---- the same weight gradient WITHOUT LDS and without barriers ("register-direct").
// In v_mfma_f32_32x32x2_f32 lane (li, lh) supplies A[m = li][k = lh] and B[k = lh][n = li].  With m = input channel, n = output channel
// and k = TILE, a lane that loads the x patch of (tile t + lh, cin li) and the 2x2 dy tile of (tile t + lh, cout li) can compute, in its
// own registers, B^T d B and A dY A^T - which ARE its A and B operands for the Winograd positions of that k-step.  Nothing is exchanged
// between lanes: no V / Z staging, no barrier.  A wave owns a 32 cin x 32 cout block of EIGHT positions (rows {0,1} or {2,3} of the 4x4
// position grid: 8 accumulator tiles = 128 registers, 3 of the 4 patch rows, half the transform) and streams over its tile range; a
// block is 8 waves = 2 x 2 (or 1 x 4) channel blocks x the two position halves, sharing one tile range (x / dy lines shared through L1).
// Measured on the way here: a wave owning all 16 positions (256 accumulators, one wave per SIMD) runs VALU time PLUS matrix time - a
// wave's own VALU work does not overlap its MFMAs (398 us = 250 MFMA + ~150 VALU at 64->64 @ 16x256x256, loads removed) - so two
// waves per SIMD are needed for the transforms of one to hide under the MFMAs of the other.
struct WinoWgradDirectArgs {
    const float* x; int ldx;
    const float* dy; int ldy;
    float* slabs;                 // [splits][16][cin][cout]
    int cin, cout, Nimg, H, W, TY, TX;
    int wci;                      // channel blocks along cin (2: block = 64 cin x 64 cout, 1: block = 32 cin x 128 cout)
    int ci_chunks, co_chunks, nchunks, nsplits;
    long tiles, tiles_per_split;  // tiles_per_split even
    long x_bytes, dy_bytes;       // extents for the bounds-checked buffer loads (< 2^31)
    int abl;                      // timing ablations (wrong results): 1 = x loads dropped by the range check, 2 = dy loads, 3 = both
};

// PH: position half (rows 2*PH, 2*PH + 1 of the position grid).  CABL: compile-time timing ablations (4: no loads, 8: no MFMA).
template <int PH, int CABL, bool ZIG = (CABL & 16) != 0>
__device__ __forceinline__ void wino_wgrad_direct_body(const WinoWgradDirectArgs& g, int bsplit, int bchunk, int sub, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int wco = 4 / g.wci;
    const int ca = (bchunk / g.co_chunks) * 32 * g.wci + (sub / wco) * 32 + li;      // this lane's input channel (A operand rows)
    const int cb = (bchunk % g.co_chunks) * 32 * wco + (sub % wco) * 32 + li;        // this lane's output channel (B operand columns)
    const bool ca_ok = ca < g.cin, cb_ok = cb < g.cout;
    const long t_begin = (long)bsplit * g.tiles_per_split;
    const long t_end = t_begin + g.tiles_per_split < g.tiles ? t_begin + g.tiles_per_split : g.tiles;
    const int per = g.TY * g.TX;

    // Addressing: bounds-checked buffer loads with 32-bit BYTE offsets (host guarantees both tensors < 2 GB).  Anything that must read as
    // zero (halo outside the image, channels beyond cin / cout, tiles beyond the range) gets an offset >= 2^31, which the hardware range
    // check turns into 0.0 - no select on loaded values.  Walking the tiles in (image, tile row, tile column) order the patch origin
    // moves by a constant per step plus one image row at a tile-row wrap; the wrap into the next image needs nothing, (n*H + 2*ty) being
    // the global pixel row.
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, (int)g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dy), 0, (int)g.dy_bytes, 0x00020000);
    constexpr unsigned BIG = 0x80000000u;
    const unsigned ldx4 = 4u * g.ldx, ldy4 = 4u * g.ldy, rowx4 = 4u * g.W * g.ldx, rowy4 = 4u * g.W * g.ldy;

    const int t_end_i = (int)t_end;                     // tiles < 2^31 (host check)
    int tg = (int)t_begin + lh;                         // this lane's tile: t_begin + lh, then += 2 per step
    int tty, ttx;
    unsigned offx, offy;                                // byte offsets of pixel (2ty, 2tx) (= patch element (1,1) = dy tile origin) + channel
    {
        const long tt = tg < g.tiles ? (long)tg : g.tiles - 1;
        int tn;
        if constexpr (ZIG) {
            // zig-zag order (see load()): position t' -> pair u = t' / 2 (two horizontally adjacent tiles, one per lane half), u even: upper
            // tile row of a row pair, u odd: lower; v = u / 2 = (global row pair, column pair).  t_begin is a multiple of 4: u even here.
            const long v = tt >> 2;
            const int half = g.TX >> 1;
            const long rp = v / half;
            const int cp = (int)(v - rp * half);
            const long grow = 2 * rp + ((tt >> 1) & 1);             // global tile row (image * TY + tile row)
            tn = (int)(grow / g.TY);
            tty = (int)(grow - (long)tn * g.TY);
            ttx = 2 * cp + (int)(tt & 1);
        } else {
            tn = (int)(tt / per);
            const int rem = (int)(tt - (long)tn * per);
            tty = rem / g.TX; ttx = rem - tty * g.TX;
        }
        offx = (unsigned)((((long)tn * g.H + 2 * tty) * g.W + 2 * ttx) * g.ldx + ca) * 4u;             // patch element (1, 1): always inside the image
        offy = (unsigned)((((long)tn * g.H + 2 * tty) * g.W + 2 * ttx) * g.ldy + cb) * 4u;
    }
    // Software pipeline, one k-step (two tiles) per stage:  raw[3]: the loads of step s + 3 are issued during step s;  op[2]: step s + 1's
    // operands are computed during step s.  Patch rows: half 0 needs rows 0,1,2, half 1 rows 1,2,3 -> raw_x[.][r] = patch row PH + r.
    float raw_x[3][12], raw_y[3][4], op_v[2][8], op_z[2][8];
    // All constant parts of an address (patch row a, patch column b, dy tile element) ride in the buffer load's SCALAR offset: the vector
    // offset of every load of a stage is the patch origin itself or BIG.  fp32 MFMAs and VALU instructions compete for the same
    // SIMD cycles (a step costs matrix time PLUS 4 cycles per VALU instruction, measured), so every v_add removed here is matrix time.
    auto load = [&](float (&rx)[12], float (&ry)[4], const int par) {
        const bool tv = tg < t_end_i;
        const unsigned ox = (tv && ca_ok && !(g.abl & 1)) ? offx : BIG, oy = (tv && cb_ok && !(g.abl & 2)) ? offy : BIG;
        // Patch rows PH, PH+1, PH+2 (patch row 1 = pixel row 2ty): the first (PH = 0) or the last (PH = 1) of them may lie outside the
        // image, as may patch columns 0 and 3.  Vector offsets never go below zero (the range check sees voffset alone): rows / columns in
        // front of the origin subtract in the VALU and are only formed where they exist.
        const bool edge_ok = PH == 0 ? tty > 0 : tty < g.TY - 1;
        const bool left = ttx > 0, right = ttx < g.TX - 1;
        const unsigned oe = edge_ok ? (PH == 0 ? ox - rowx4 : ox) : BIG;            // origin of the edge row (PH = 0: row 0, scalar part 0; PH = 1: row 3)
        const unsigned ox_l = left ? ox - ldx4 : BIG, ox_r = right ? ox : BIG;
        const unsigned oe_l = (left && edge_ok) ? oe - ldx4 : BIG, oe_r = right ? oe : BIG;
        if constexpr (CABL & 4) {                       // timing ablation: no load instructions at all
#pragma unroll
            for (int i = 0; i < 12; ++i) rx[i] = __builtin_bit_cast(float, (i & 1 ? ox_l : oe_r) + (i & 2 ? oe_l : ox_r) + (unsigned)i);
#pragma unroll
            for (int i = 0; i < 4; ++i) ry[i] = __builtin_bit_cast(float, oy + i);
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const bool e = (PH == 0 && a == 0) || (PH == 1 && a == 2);
                // scalar part of patch row PH + a relative to its vector origin: PH = 0: rows 0,1,2 -> 0 (own origin), 0, rowx4; PH = 1: rows 1,2,3 -> 0, rowx4, 2 rowx4
                const unsigned so = PH == 0 ? (a == 2 ? rowx4 : 0u) : (unsigned)a * rowx4;
                rx[a * 4 + 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx_, e ? oe_l : ox_l, so, 0));
                rx[a * 4 + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx_, e ? oe : ox, so, 0));
                rx[a * 4 + 2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx_, e ? oe : ox, so + ldx4, 0));
                rx[a * 4 + 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx_, e ? oe_r : ox_r, so + 2 * ldx4, 0));
            }
            ry[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, oy, 0, 0));
            ry[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, oy, ldy4, 0));
            ry[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, oy, rowy4, 0));
            ry[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, oy, rowy4 + ldy4, 0));
        }
        // advance by two tiles (TX >= 2: at most one wrap); branch-free, the loop body stays one basic block
        tg += 2;
        if constexpr (ZIG) {
            // ZIG-ZAG over pairs of tile rows: (ty, tx..tx+1) -> (ty + 1, tx..tx+1) -> (ty, tx+2..tx+3) -> ...  Tile rows ty and ty + 1 share two
            // of their four patch rows: in row-major order the second use came TX tiles later, after the XCD's L2 had turned over (PMC: 2.0x
            // the algorithmic bytes); here it is the wave's next step.  `par` (the step's parity, a compile-time constant at every call
            // site: the stage loop is unrolled six-fold and a tile range starts on an even step) says which move follows this load.
            if (par == 0) {
                tty += 1;
                offx += 2 * rowx4;
                offy += 2 * rowy4;
            } else {
                ttx += 2;
                const bool wx = ttx >= g.TX;                        // end of the row pair: on to the next one (4 pixels right of the last column = one row down)
                ttx -= wx ? g.TX : 0;
                tty += wx ? 1 : -1;
                tty = tty >= g.TY ? 0 : tty;
                offx += 4 * ldx4 + (wx ? rowx4 : 0u - 2 * rowx4);
                offy += 4 * ldy4 + (wx ? rowy4 : 0u - 2 * rowy4);
            }
        } else {
            ttx += 2;
            const bool wx = ttx >= g.TX;
            ttx -= wx ? g.TX : 0;
            tty += wx ? 1 : 0;
            tty = tty >= g.TY ? 0 : tty;
            offx += 4 * ldx4 + (wx ? rowx4 : 0u);
            offy += 4 * ldy4 + (wx ? rowy4 : 0u);
        }
    };
    // B^T d B rows {2PH, 2PH+1}: column pass over the three loaded rows, then the row pass;  A dY A^T rows likewise
    auto transform = [&](const float (&r)[12], const float (&cy)[4], float (&v)[8], float (&z)[8]) {
        float t0[4], t1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if constexpr (PH == 0) { t0[b] = r[b] - r[8 + b]; t1[b] = r[4 + b] + r[8 + b]; }            // d0 - d2, d1 + d2
            else { t0[b] = r[4 + b] - r[b]; t1[b] = r[b] - r[8 + b]; }                                  // d2 - d1, d1 - d3
        }
        v[0] = t0[0] - t0[2]; v[1] = t0[1] + t0[2]; v[2] = t0[2] - t0[1]; v[3] = t0[1] - t0[3];
        v[4] = t1[0] - t1[2]; v[5] = t1[1] + t1[2]; v[6] = t1[2] - t1[1]; v[7] = t1[1] - t1[3];
        const float y00 = cy[0], y01 = cy[1], y10 = cy[2], y11 = cy[3];
        float p0, q0, p1, q1;
        if constexpr (PH == 0) { p0 = y00; q0 = y01; p1 = y00 + y10; q1 = y01 + y11; }
        else { p0 = y00 - y10; q0 = y01 - y11; p1 = -y10; q1 = -y11; }
        z[0] = p0; z[1] = p0 + q0; z[2] = p0 - q0; z[3] = -q0;
        z[4] = p1; z[5] = p1 + q1; z[6] = p1 - q1; z[7] = -q1;
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // one stage: multiply step s (operands op[s & 1]) | transform step s + 1 | load step s + 3
    auto stage = [&](int cur) {
        transform(raw_x[(cur + 1) % 3], raw_y[(cur + 1) % 3], op_v[(cur + 1) & 1], op_z[(cur + 1) & 1]);
        load(raw_x[cur % 3], raw_y[cur % 3], (cur + 1) & 1);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if constexpr (CABL & 8) acc[p][0] += op_v[cur & 1][p] * op_z[cur & 1][p];      // timing ablation: no MFMA
            else acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(op_v[cur & 1][p], op_z[cur & 1][p], acc[p], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    load(raw_x[0], raw_y[0], 0);                        // past the end of the range: out-of-range offsets, zero values
    load(raw_x[1], raw_y[1], 1);
    load(raw_x[2], raw_y[2], 0);
    transform(raw_x[0], raw_y[0], op_v[0], op_z[0]);
    __builtin_amdgcn_sched_barrier(0);
    for (long tb = t_begin; tb < t_end; tb += 12) {     // six stages of two tiles (lcm of the two ring lengths); stages past t_end multiply zeros
        stage(0);
        stage(1);
        stage(2);
        stage(3);
        stage(4);
        stage(5);
    }

    float* slab = g.slabs + (long)bsplit * 16 * g.cin * g.cout;
    const int ci_base = ca - li;
    if (cb_ok) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci_base + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci < g.cin) slab[((long)(8 * PH + p) * g.cin + ci) * g.cout + cb] = acc[p][r];
            }
    }
}

template <int CABL>
__global__ __launch_bounds__(512) void wino_wgrad_direct_kernel(WinoWgradDirectArgs g) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bsplit, bchunk;
    {   // XCD-aware order (see wino_wgrad_kernel): the chunks of one tile range sit on one XCD
        const int L = blockIdx.x, nch = g.nchunks, span = 8 * nch;
        const int grp = L / span, r = L - grp * span;
        bsplit = grp * 8 + (r & 7);
        bchunk = r >> 3;
        if (grp * 8 + 8 > g.nsplits) {
            const int done = grp * 8, rem = g.nsplits - done;
            bsplit = done + r % rem;
            bchunk = r / rem;
        }
    }
    // waves 2k, 2k+1 (position halves of channel block k) sit on different SIMDs; wave w and w + 4 share a SIMD: another channel block
    if (wid & 1) wino_wgrad_direct_body<1, CABL>(g, bsplit, bchunk, wid >> 1, lane);
    else wino_wgrad_direct_body<0, CABL>(g, bsplit, bchunk, wid >> 1, lane);
}

// dw[r][s][ci][co] = (G^T (sum_splits slab) G)[r][s];  block = 32 (ci,co) columns x 8 split-lanes
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ slabs, int nsplit, long kn, float* __restrict__ dw) {
    __shared__ float red[8][16][32];
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + col;
    float m[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) m[xi] = 0.f;
    if (i < kn)
        for (int k = sl; k < nsplit; k += 8) {
            const float* p = slabs + (long)k * 16 * kn + i;
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] += p[(long)xi * kn];
        }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) red[sl][xi][col] = m[xi];
    __syncthreads();
    if (sl == 0 && i < kn) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int j = 1; j < 8; ++j) m[xi] += red[j][xi][col];
        // G^T (3x4) = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
        float t[3][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = m[b] + 0.5f * (m[4 + b] + m[8 + b]);
            t[1][b] = 0.5f * (m[4 + b] - m[8 + b]);
            t[2][b] = 0.5f * (m[4 + b] + m[8 + b]) + m[12 + b];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            dw[(long)(r * 3 + 0) * kn + i] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
            dw[(long)(r * 3 + 1) * kn + i] = 0.5f * (t[r][1] - t[r][2]);
            dw[(long)(r * 3 + 2) * kn + i] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
        }
    }
}

struct WinoWgradPlan { int ci_chunks, co_chunks, splits, wci; long tiles, tps; };
// register-direct form: needs 32-bit byte offsets (both tensors < 2 GB; the pixel strides are not known here, so the test is on the
// channel counts and repeated with the real strides at launch) and at least two tile columns
static bool wino_wgrad_direct(int n_img, int h, int w, long ldx, long ldy) {
    static const bool lds_form = getenv("RUNET_WINO_WGRAD_LDS") && atoi(getenv("RUNET_WINO_WGRAD_LDS")) != 0;
    const long px = (long)n_img * h * w;
    return !lds_form && w >= 4 && px * ldx * 4 < (1L << 31) && px * ldy * 4 < (1L << 31) && px / 4 + 64 < (1L << 31);
}
static bool wino_wgrad_rowmajor() {      // measurement knob: RUNET_WINO_WGRAD_ROWMAJOR=1 -> the row-major tile order of round 2
    static const bool v = getenv("RUNET_WINO_WGRAD_ROWMAJOR") && atoi(getenv("RUNET_WINO_WGRAD_ROWMAJOR")) != 0;
    return v;
}
static WinoWgradPlan wino_wgrad_plan(int n_img, int h, int w, int cin, int cout, bool direct) {
    WinoWgradPlan p{};
    if (direct) {
        // one block (four waves of 512 registers) per CU: ~256 blocks, or two rounds when the tile ranges stay long
        p.wci = cin > 32 ? 2 : 1;
        p.ci_chunks = cdiv(cin, 32 * p.wci); p.co_chunks = cdiv(cout, 32 * (4 / p.wci));
        p.tiles = (long)n_img * (h / 2) * (w / 2);
        long splits = cdiv(256, p.ci_chunks * p.co_chunks);
        const long maxs = cdiv(p.tiles, 64);
        if (splits > maxs) splits = maxs;
        if (splits < 1) splits = 1;
        p.tps = cdiv(cdiv(p.tiles, splits), 4) * 4L;     // multiple of 4: a tile range starts on an even step of the zig-zag order
        p.splits = cdiv(p.tiles, p.tps);
        return p;
    }
    p.ci_chunks = cdiv(cin, GCI); p.co_chunks = cdiv(cout, GCO);
    p.tiles = (long)n_img * (h / 2) * (w / 2);
    long splits = cdiv(640, p.ci_chunks * p.co_chunks);
    const long maxs = cdiv(p.tiles, GT);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    p.tps = cdiv(cdiv(p.tiles, splits), GT) * (long)GT;
    p.splits = cdiv(p.tiles, p.tps);
    return p;
}

}  // namespace

extern "C" int runet_wino_supported(int h, int w, int cin, int cout) {
    return (h % 2 == 0 && w % 2 == 0 && cin % 16 == 0 && cin >= 16 && cout % 2 == 0) ? 1 : 0;
}

extern "C" int runet_wino_fits(int n_img, int h, int w, int ldx, int ldy, int cin, int cout) {
    if (!runet_wino_supported(h, w, cin, cout)) return 0;
    const long px = (long)n_img * h * w;
    return (px * ldx < (1L << 29) && px * ldy < (1L << 29) && 16L * cin * cout < (1L << 29)) ? 1 : 0;
}

extern "C" int runet_wino_weights(const float* w_hwio, float* U, int cin, int cout, int dgrad, void* stream) {
    RUNET_REQUIRE(w_hwio && U && cin > 0 && cout > 0, "bad arguments");
    const long total = (long)cin * cout;
    hipLaunchKernelGGL(wino_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_hwio, U, cin, cout, dgrad);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_wino_conv(const float* x, int ldx, const float* U, const float* bias, float* y, int ldy, int n_img, int h, int w,
                               int k, int n, int accumulate, void* stream) {
    RUNET_REQUIRE(x && U && y, "null pointer");
    RUNET_REQUIRE(runet_wino_supported(h, w, k, n), "shape not supported by the Winograd kernel (H, W even; K multiple of 16; N even)");
    RUNET_REQUIRE(ldx >= k && ldx % 2 == 0 && ldy >= n && ldy % 2 == 0, "pixel strides must be even and cover the channels");
    RUNET_REQUIRE(((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)U % 16) == 0 && (!bias || ((uintptr_t)bias % 8) == 0), "alignment");
    RUNET_REQUIRE(runet_wino_fits(n_img, h, w, ldx, ldy, k, n), "tensor too large for 32-bit buffer offsets (runet_wino_fits)");
    WinoArgs a{};
    a.x = x; a.ldx = ldx; a.U = U; a.bias = bias; a.y = y; a.ldy = ldy; a.K = k; a.N = n;
    a.Nimg = n_img; a.H = h; a.W = w; a.TY = h / 2; a.TX = w / 2; a.tiles = (long)n_img * a.TY * a.TX; a.accumulate = accumulate;
    a.npatches = n_img * cdiv(a.TY, 4) * cdiv(a.TX, 8);
    a.nchunks = cdiv(n, WBN);
    dim3 grid(a.npatches * a.nchunks);
    static const int abl = getenv("RUNET_WINO_ABL") ? atoi(getenv("RUNET_WINO_ABL")) : 0;      // timing ablations only (wrong results)
    switch (abl) {
    case 1: hipLaunchKernelGGL((wino_conv_kernel<1, false>), grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL((wino_conv_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 4: hipLaunchKernelGGL((wino_conv_kernel<4, false>), grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 8: hipLaunchKernelGGL((wino_conv_kernel<8, false>), grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 7: hipLaunchKernelGGL((wino_conv_kernel<7, false>), grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 15: hipLaunchKernelGGL((wino_conv_kernel<15, false>), grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 32: hipLaunchKernelGGL((wino_conv_kernel<0, false>), grid, dim3(256), 0, (hipStream_t)stream, a); break;     // 32x32x2 form with the LDS exchange
    default: hipLaunchKernelGGL((wino_conv_kernel<0, true>), grid, dim3(256), 0, (hipStream_t)stream, a);             // 16x16x4 form: 6-10 % faster on the 64/128-channel layers
    }
    RUNET_CHECK_LAUNCH();
}

extern "C" long runet_wino_wgrad_workspace_floats(int n_img, int h, int w, int cin, int cout) {
    const WinoWgradPlan p = wino_wgrad_plan(n_img, h, w, cin, cout, false), q = wino_wgrad_plan(n_img, h, w, cin, cout, true);
    return (long)(p.splits > q.splits ? p.splits : q.splits) * 16 * cin * cout;      // either form (chosen at launch from the pixel strides)
}

extern "C" int runet_wino_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats,
                                int n_img, int h, int w, int cin, int cout, void* stream) {
    RUNET_REQUIRE(x && dy && dw && workspace, "null pointer");
    RUNET_REQUIRE(h % 2 == 0 && w % 2 == 0 && cin > 0 && cout > 0, "H and W must be even");
    RUNET_REQUIRE(ldx >= cin && ldy >= cout, "bad pixel strides");
    const bool direct = wino_wgrad_direct(n_img, h, w, ldx, ldy);
    const WinoWgradPlan p = wino_wgrad_plan(n_img, h, w, cin, cout, direct);
    RUNET_REQUIRE(workspace_floats >= (long)p.splits * 16 * cin * cout, "workspace too small (runet_wino_wgrad_workspace_floats)");
    hipStream_t st = (hipStream_t)stream;
    WinoWgradArgs a{};
    a.x = x; a.ldx = ldx; a.dy = dy; a.ldy = ldy; a.slabs = workspace; a.cin = cin; a.cout = cout; a.Nimg = n_img; a.H = h; a.W = w;
    a.TY = h / 2; a.TX = w / 2; a.co_chunks = p.co_chunks; a.tiles = p.tiles; a.tiles_per_split = p.tps;
    a.nchunks = p.ci_chunks * p.co_chunks; a.nsplits = p.splits;
    if (direct) {
        WinoWgradDirectArgs d{};
        d.x_bytes = (long)n_img * h * w * ldx * 4; d.dy_bytes = (long)n_img * h * w * ldy * 4;
        static const int abl = getenv("RUNET_WINO_WGRAD_ABL") ? atoi(getenv("RUNET_WINO_WGRAD_ABL")) : 0;
        d.abl = abl;
        d.x = x; d.ldx = ldx; d.dy = dy; d.ldy = ldy; d.slabs = workspace; d.cin = cin; d.cout = cout; d.Nimg = n_img; d.H = h; d.W = w;
        d.TY = h / 2; d.TX = w / 2; d.wci = p.wci; d.ci_chunks = p.ci_chunks; d.co_chunks = p.co_chunks; d.nchunks = p.ci_chunks * p.co_chunks;
        d.nsplits = p.splits; d.tiles = p.tiles; d.tiles_per_split = p.tps;
        if (abl & 4) hipLaunchKernelGGL(wino_wgrad_direct_kernel<4>, dim3(d.nchunks * p.splits), dim3(512), 0, st, d);
        else if (abl & 8) hipLaunchKernelGGL(wino_wgrad_direct_kernel<8>, dim3(d.nchunks * p.splits), dim3(512), 0, st, d);
        else if ((h / 2) % 2 == 0 && !wino_wgrad_rowmajor()) hipLaunchKernelGGL(wino_wgrad_direct_kernel<16>, dim3(d.nchunks * p.splits), dim3(512), 0, st, d);   // zig-zag tile order
        else hipLaunchKernelGGL(wino_wgrad_direct_kernel<0>, dim3(d.nchunks * p.splits), dim3(512), 0, st, d);
    } else {
        hipLaunchKernelGGL(wino_wgrad_kernel, dim3(a.nchunks * p.splits), dim3(256), 0, st, a);
    }
    const long kn = (long)cin * cout;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(cdiv(kn, 32)), dim3(256), 0, st, workspace, p.splits, kn, dw);
    RUNET_CHECK_LAUNCH();
}
