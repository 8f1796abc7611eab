// bf16-operand convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation): the reduced-precision path of
// BASELINE.json configs 3 and 5.  The reference (/root/reference/Main_Final.py) is fp32 only; this path keeps fp32 master weights,
// fp32 activations in HBM, fp32 BatchNorm / attention / loss / Adam, and rounds ONLY the two operands of each convolution's
// multiply-adds to bf16 on their way into LDS (round-to-nearest-even, v_cvt_pk_bf16_f32).  16x the fp32 MFMA rate turns every
// convolution of the step from matrix-pipe-bound into HBM-bound.
//
//   forward / data gradient   igemm_bf16_kernel   out[p][n] = sum_tap sum_k src[pix(p,tap)][k] * B_tap[k][n]
//     GEMM M = pixels (tile 128), N = output channels (tile 32/64/128), K = taps x channels in steps of 64 channels.
//     A: fp32 NHWC rows gathered per tap (predicated zero padding), converted to bf16, LDS image [pixel][64 + 8 pad] so that a lane's
//     8 consecutive k are one conflict-free ds_read_b128.  B: weights PRE-PACKED once per optimizer step by bf16_pack_kernel into
//     [tap][k/8][n][8] bf16 - a lane's B fragment (8 consecutive k of one column) is one 16-byte unit, the tile is copied to LDS verbatim.
//     Register-staged double buffering, one barrier per 64-deep k-step (16 MFMAs per wave).
//   weight gradient           wgrad_bf16_kernel   dW[tap][ci][co] = sum_p x[pix(p,tap)][ci] * dy[p][co]
//     GEMM M = cin (tile 64), N = cout (tile 64), K = pixels.  Both operands have the contraction index (pixels) as the SLOW memory index
//     (NHWC), so both tiles sit in LDS as [pixel][64 channels] exactly as they come from HBM and the fragments are fetched with the
//     hardware transposing read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group -> each lane gets 4 consecutive pixels of
//     its channel).  A block owns a 4 x 16 pixel tile of dy and the x tile WITH ITS HALO, and accumulates all taps from that one tile
//     (tap shifts move whole LDS rows, so alignment is unaffected); fixed-order split-K slabs + slab reduce as in the fp32 path.
#include "runet_common.h"
#include <stdlib.h>
#include "../../include/runet_hip.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

// -------------------------------------------------------------------------------------------------- weight packing
// src: HWIO fp32 w[tap][cin][cout].  transpose == 0: B(k = cin, n = cout); != 0: B(k = cout, n = cin) (data gradients).
// dst[tap][k/8][n][8] bf16, k padded with zeros to a multiple of 8.
__global__ __launch_bounds__(256) void bf16_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int taps, int cin, int cout,
                                                        int transpose) {
    const int K = transpose ? cout : cin, N = transpose ? cin : cout;
    const int K8 = (K + 7) / 8;
    const long total = (long)taps * K8 * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i % N);
        const long t2 = i / N;
        const int ko = (int)(t2 % K8), tap = (int)(t2 / K8);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ko * 8 + j;
            float f = 0.f;
            if (k < K) f = transpose ? w[((long)tap * cin + n) * cout + k] : w[((long)tap * cin + k) * cout + n];
            v[j] = (__bf16)f;
        }
        *reinterpret_cast<bf16x8*>(dst + i * 8) = v;
    }
}

// -------------------------------------------------------------------------------------------------- forward / data gradient
struct BGemmArgs {
    const float* x; int ldx;      // A source, fp32 [Nimg, Hin, Win, ldx]
    const __bf16* w;              // packed [taps][K8][Ncols][8]
    const float* bias;
    float* y; int ldy;            // destination fp32 [Nimg, Hout, Wout, ldy]
    int K, K8, Ncols;             // K channels read from x (multiple of 4); K8 = ceil(K / 8) octets in the packed weight
    int Nimg, H, W;               // iteration space
    int Hin, Win, a_scale;        // source pixel = (h*a_scale + bh + r*tdh, w*a_scale + bw + s*tdw)
    int KH, KW, tdh, tdw, bh, bw;
    int Hout, Wout, o_scale, o_dh, o_dw;
    int z_taps;                   // >0: blockIdx.z selects one weight tap AND the destination offset (k2-s2 transposed conv forward)
    int accumulate;
};

constexpr int BK = 64;            // channels per k-step
constexpr int LDA = BK + 8;       // bf16 elements per A row (144 B: ds_read_b128 of 32 rows hits 64 distinct banks)

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm_bf16_kernel(BGemmArgs g) {
    constexpr int BM = 128;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    constexpr int A_ELEMS = BM * LDA;              // bf16
    constexpr int B_ELEMS = 8 * BN * 8;            // 8 octets x BN columns x 8
    constexpr int STAGE = A_ELEMS + B_ELEMS;
    constexpr int AROWS = BM / 16;                 // rows staged per thread (16 threads cover the 64 channels of one pixel)
    constexpr int BITEMS = (8 * BN + 255) / 256;   // 16-byte weight units per thread

    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    const long P = (long)g.Nimg * g.H * g.W;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int HW = g.H * g.W;

    const __bf16* wbase = g.w;
    int o_dh = g.o_dh, o_dw = g.o_dw;
    const long tap_stride = (long)g.K8 * g.Ncols * 8;
    if (g.z_taps > 0) {
        wbase += (long)blockIdx.z * tap_stride;
        o_dh = blockIdx.z / g.z_taps;
        o_dw = blockIdx.z % g.z_taps;
    }

    const int aq = tid & 15;
    long a_img[AROWS];
    int a_h[AROWS], a_w[AROWS];
    bool a_ok[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const long p = m0 + (tid >> 4) + 16 * i;
        a_ok[i] = p < P;
        const long pp = a_ok[i] ? p : 0;
        const int n = (int)(pp / HW);
        const int rem = (int)(pp - (long)n * HW);
        const int h = rem / g.W;
        a_img[i] = (long)n * g.Hin * g.Win;
        a_h[i] = h * g.a_scale + g.bh;
        a_w[i] = (rem - h * g.W) * g.a_scale + g.bw;
    }

    const int KC = (g.K + BK - 1) / BK;
    const int ntaps = (g.z_taps > 0) ? 1 : g.KH * g.KW;
    const int nks = ntaps * KC;

    f32x4 ra[AROWS];
    f32x4 rb[BITEMS];
    int ld_tr = 0, ld_ts = 0, ld_kc = 0, ld_tap = 0;

    auto load_tile = [&]() {
        const int dh = ld_tr * g.tdh, dw = ld_ts * g.tdw;
        const int kofs = ld_kc * BK;
        const bool kok = kofs + aq * 4 < g.K;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int ih = a_h[i] + dh, iw = a_w[i] + dw;
            const bool ok = a_ok[i] && kok && (unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(g.x + ((a_img[i] + (long)ih * g.Win + iw) * g.ldx + kofs + aq * 4));
            ra[i] = v;
        }
        const __bf16* wt = wbase + (long)ld_tap * tap_stride;
#pragma unroll
        for (int i = 0; i < BITEMS; ++i) {
            const int id = tid + 256 * i;
            const int oc = id / BN, n = id - oc * BN;
            const int ko = ld_kc * 8 + oc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (id < 8 * BN && ko < g.K8 && n0 + n < g.Ncols) v = *reinterpret_cast<const f32x4*>(wt + ((long)ko * g.Ncols + n0 + n) * 8);
            rb[i] = v;
        }
        ++ld_tap;
        if (++ld_ts == g.KW) {
            ld_ts = 0;
            if (++ld_tr == g.KH) { ld_tr = 0; ld_tap = 0; ++ld_kc; }
        }
    };
    auto store_tile = [&](int buf) {
        __bf16* As = smem + buf * STAGE;
        __bf16* Bs = As + A_ELEMS;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            bf16x4 v;
            v[0] = (__bf16)ra[i][0]; v[1] = (__bf16)ra[i][1]; v[2] = (__bf16)ra[i][2]; v[3] = (__bf16)ra[i][3];
            *reinterpret_cast<bf16x4*>(As + ((tid >> 4) + 16 * i) * LDA + aq * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < BITEMS; ++i) {
            const int id = tid + 256 * i;
            if (id < 8 * BN) *reinterpret_cast<f32x4*>(Bs + id * 8) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    load_tile();
    store_tile(0);
    __syncthreads();

    int cur = 0;
    for (int ks = 0; ks < nks; ++ks) {
        const bool more = ks + 1 < nks;
        if (more) load_tile();
        const __bf16* As = smem + cur * STAGE;
        const __bf16* Bs = As + A_ELEMS;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const bf16x8*>(As + (wm0 + a * 32 + li) * LDA + s * 16 + lh * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const bf16x8*>(Bs + ((2 * s + lh) * BN + wn0 + b * 32 + li) * 8);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: D[row = pixel][col = n]; lane holds col (lane&31), rows (r&3)+8*(r>>2)+4*(lane>>5) ----
    const bool same_pix = (g.o_scale == 1 && g.Hout == g.H && g.Wout == g.W);
    int ncol[TN];
    float bv[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        ncol[b] = n0 + wn0 + b * 32 + li;
        bv[b] = (g.bias != nullptr && ncol[b] < g.Ncols) ? g.bias[ncol[b]] : 0.f;
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float* drow[8];
            float old[8][TN];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = half * 8 + i;
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const long p = m0 + wm0 + a * 32 + row;
                long op = p < P ? p : 0;
                if (!same_pix) {
                    const int nimg = (int)(op / HW);
                    const int rem = (int)(op - (long)nimg * HW);
                    const int h = rem / g.W, w = rem - h * g.W;
                    op = ((long)nimg * g.Hout + (h * g.o_scale + o_dh)) * g.Wout + (w * g.o_scale + o_dw);
                }
                drow[i] = p < P ? g.y + op * g.ldy : nullptr;
#pragma unroll
                for (int b = 0; b < TN; ++b) old[i][b] = (g.accumulate && drow[i] && ncol[b] < g.Ncols) ? drow[i][ncol[b]] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (drow[i]) {
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        if (ncol[b] < g.Ncols) drow[i][ncol[b]] = acc[a][b][half * 8 + i] + bv[b] + old[i][b];
                }
            }
        }
    }
}

template <int BN, int WM, int WN>
void launch_bgemm(const BGemmArgs& a, int gz, hipStream_t st) {
    const long P = (long)a.Nimg * a.H * a.W;
    dim3 grid(cdiv(P, 128), cdiv(a.Ncols, BN), gz);
    const size_t lds = 2 * (128 * LDA + 8 * BN * 8) * sizeof(__bf16);
    hipLaunchKernelGGL((igemm_bf16_kernel<BN, WM, WN>), grid, dim3(256), lds, st, a);
}

// -------------------------------------------------------------------------------------------------- 3x3 (dilation 1) forward / data gradient
// The generic kernel above gathers its A rows per tap from global memory: nine shifted reads of the same pixels, served by L2.  At the
// bf16 matrix rate that gather IS the bound (7-9 TB/s of L2 traffic for a 64 -> 64 layer at 16 x 256^2, 3.7x its HBM time).  This kernel
// reads the input ONCE: a block owns a 16 x 16 output patch x 64 output channels, stages the 18 x 18 halo patch of a 64-channel chunk in
// LDS (fp32 -> bf16 on the way) and runs all nine taps from it - a tap shift moves the fragment's LDS ROW, so every ds_read_b128 stays
// aligned.  Weights stream through a double-buffered 8 KB LDS tile per tap.  Wave w owns patch rows 4w .. 4w+3 (64 pixels x 64 channels:
// 2 x 2 MFMA tiles, 64 accumulator registers).
struct C3Args {
    const float* x; int ldx;
    const __bf16* w;              // packed [9][K8][Ncols][8]
    const float* bias;
    float* y; int ldy;
    int K, K8, Ncols;
    int Nimg, H, W, tiles_h, tiles_w, nchunks_n;
    int flip;                     // 1: data gradient (tap (r, s) reads the source at offset (1 - r, 1 - s))
    int accumulate;
};
constexpr int PT = 16, HP = PT + 2;                 // patch width, halo patch width

// PH = patch height (16: 256 pixels per block, 2 blocks per CU; 8: 128 pixels, 3 blocks per CU).
// Instruction count matters here: with no memory traffic and no MFMAs at all the first version of this kernel still took half its time
// (timing ablations) - 2000+ address / predicate / exec-mask instructions per wave against 144 MFMAs.  Hence: halo row offsets
// and validity are computed ONCE per block; interior patches (the vast majority) take branch-free load and store paths with 32-bit
// offsets from per-block base pointers; LDS fragment addresses are one add per tap plus immediates.
template <int PH>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(C3Args g) {
    constexpr int HROWS = (PH + 2) * HP;            // halo pixels
    constexpr int TMA = PH / 8;                     // 32-pixel M-tiles per wave (wave w owns patch rows (PH/4)*w ..)
    constexpr int HPASS = (HROWS + 15) / 16;        // halo staging passes (16 rows per pass: 16 threads cover the 64 channels of a row)
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    __bf16* halo = smem;                            // [HROWS][LDA]
    __bf16* Bs = smem + HROWS * LDA;                // [2][8 octets][64][8]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nb = blockIdx.x % g.nchunks_n;
    const int patch = blockIdx.x / g.nchunks_n;
    const int tx = patch % g.tiles_w;
    const int t2 = patch / g.tiles_w;
    const int ty = t2 % g.tiles_h, n = t2 / g.tiles_h;
    const int h0 = ty * PH, w0 = tx * PT, n0 = nb * 64;
    const long tap_stride = (long)g.K8 * g.Ncols * 8;
    const int W = g.W, H = g.H;

    f32x16 acc[TMA][2];
#pragma unroll
    for (int a = 0; a < TMA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- halo staging plan, once per block: element offset of each of this thread's rows inside the image, validity bit mask
    const int aq = tid & 15, ar = tid >> 4;
    const float* xb = g.x + (long)n * H * W * g.ldx + aq * 4;
    int roff[HPASS];
    unsigned rmask = 0;
#pragma unroll
    for (int j = 0; j < HPASS; ++j) {
        const int row = ar + 16 * j;
        const int ry = row / HP, rx = row - ry * HP;
        const int ih = h0 + ry - 1, iw = w0 + rx - 1;
        const bool ok = row < HROWS && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        roff[j] = ok ? (ih * W + iw) * g.ldx : 0;
        rmask |= (ok ? 1u : 0u) << j;
    }
    const bool interior = h0 >= 1 && w0 >= 1 && h0 + PH + 1 <= H && w0 + PT + 1 <= W;      // block-uniform: every halo pixel inside the image

    // ---- LDS fragment bases (bytes), once per block
    int a_base[TMA];
#pragma unroll
    for (int a = 0; a < TMA; ++a) a_base[a] = ((((PH / 4) * wid + 2 * a + (li >> 4)) * HP + (li & 15)) * LDA + lh * 8) * 2;
    const int b_base = ((lh * 64 + li) * 8) * 2;
    const char* halo_c = reinterpret_cast<const char*>(halo);
    const char* Bs_c = reinterpret_cast<const char*>(Bs);

    const int KC = (g.K + BK - 1) / BK;
    const bool n_full = n0 + 64 <= g.Ncols;
    f32x4 rb[2];
    const int b_oc = tid >> 6, b_nn = tid & 63;       // weight unit (octet, column) of this thread; second unit: octet + 4
    auto load_b = [&](int kc, int tap) {
        const __bf16* wt = g.w + (long)tap * tap_stride + ((long)(kc * 8 + b_oc) * g.Ncols + n0 + b_nn) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kc * 8 + b_oc + 4 * i < g.K8 && (n_full || n0 + b_nn < g.Ncols)) v = *reinterpret_cast<const f32x4*>(wt + (long)4 * i * g.Ncols * 8);
            rb[i] = v;
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(Bs + buf * 4096 + (tid + 256 * i) * 8) = rb[i];
    };

    for (int kc = 0; kc < KC; ++kc) {
        __syncthreads();                            // every wave is done reading the previous chunk's halo and B tiles
        load_b(kc, 0);
        const int kofs = kc * BK;
        const bool kfull = kofs + BK <= g.K;        // block-uniform
        const bool kok = kofs + aq * 4 < g.K;
        const float* xk = xb + kofs;
#pragma unroll
        for (int grp = 0; grp < 3; ++grp) {
            constexpr int HG = (HPASS + 2) / 3;
            f32x4 v[HG];
            if (interior && kfull) {                // branch-free: every row of every pass but (possibly) the last is a halo pixel inside the image
#pragma unroll
                for (int j = 0; j < HG; ++j) {
                    const int jj = grp * HG + j;
                    if (jj < HPASS) {
                        if ((jj + 1) * 16 <= HROWS) v[j] = *reinterpret_cast<const f32x4*>(xk + roff[jj]);
                        else {
                            f32x4 t = {0.f, 0.f, 0.f, 0.f};
                            if (ar + 16 * jj < HROWS) t = *reinterpret_cast<const f32x4*>(xk + roff[jj]);
                            v[j] = t;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < HG; ++j) {
                    const int jj = grp * HG + j;
                    f32x4 t = {0.f, 0.f, 0.f, 0.f};
                    if (jj < HPASS && ((rmask >> jj) & 1u) && kok) t = *reinterpret_cast<const f32x4*>(xk + roff[jj]);
                    v[j] = t;
                }
            }
#pragma unroll
            for (int j = 0; j < HG; ++j) {
                const int jj = grp * HG + j;
                if (jj < HPASS && (((jj + 1) * 16 <= HROWS) || ar + 16 * jj < HROWS)) {
                    bf16x4 b;
                    b[0] = (__bf16)v[j][0]; b[1] = (__bf16)v[j][1]; b[2] = (__bf16)v[j][2]; b[3] = (__bf16)v[j][3];
                    *reinterpret_cast<bf16x4*>(halo + (ar + 16 * jj) * LDA + aq * 4) = b;
                }
            }
        }
        store_b(0);
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 8) load_b(kc, tap + 1);
            const int r = tap / 3, s = tap - 3 * r;
            const int dr = g.flip ? 2 - r : r, ds = g.flip ? 2 - s : s;
            const int a_tap = (dr * HP + ds) * LDA * 2;                       // scalar byte offset of this tap's shifted halo window
            const char* Bt = Bs_c + (tap & 1) * 8192 + b_base;
#pragma unroll
            for (int s16 = 0; s16 < BK / 16; ++s16) {
                bf16x8 af[TMA], bf[2];
#pragma unroll
                for (int a = 0; a < TMA; ++a) af[a] = *reinterpret_cast<const bf16x8*>(halo_c + a_base[a] + a_tap + s16 * 32);
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const bf16x8*>(Bt + s16 * 2048 + b * 512);
#pragma unroll
                for (int a = 0; a < TMA; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
            if (tap < 8) {
                store_b((tap + 1) & 1);
                __syncthreads();
            }
        }
    }

    // ---- epilogue.  Register rr of tile a: pixel (dy, dx) = ((PH/4)*wid + 2a + (rr >> 3), 8*((rr >> 2) & 1) + 4*lh + (rr & 3)); column n0 + b*32 + li
    const int oy0 = h0 + (PH / 4) * wid;
    float* yb = g.y + ((long)n * H * W + (long)oy0 * W + w0) * g.ldy + n0 + li;      // 64-bit once; everything below is a 32-bit offset from it
    const int ldy = g.ldy;
    const bool fast = h0 + PH <= H && w0 + PT <= W && n_full && !g.accumulate;        // block-uniform
    float bv[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) bv[b] = (g.bias != nullptr && n0 + b * 32 + li < g.Ncols) ? g.bias[n0 + b * 32 + li] : 0.f;
    if (fast) {
#pragma unroll
        for (int a = 0; a < TMA; ++a)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int off = ((2 * a + (rr >> 3)) * W + 8 * ((rr >> 2) & 1) + 4 * lh + (rr & 3)) * ldy;
#pragma unroll
                for (int b = 0; b < 2; ++b) yb[off + b * 32] = acc[a][b][rr] + bv[b];
            }
    } else {
#pragma unroll
        for (int a = 0; a < TMA; ++a)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int dy = 2 * a + (rr >> 3), dx = 8 * ((rr >> 2) & 1) + 4 * lh + (rr & 3);
                if (oy0 + dy < H && w0 + dx < W) {
                    const int off = (dy * W + dx) * ldy;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        if (n0 + b * 32 + li < g.Ncols) {
                            float val = acc[a][b][rr] + bv[b];
                            if (g.accumulate) val += yb[off + b * 32];
                            yb[off + b * 32] = val;
                        }
                }
            }
    }
}

// -------------------------------------------------------------------------------------------------- weight gradient
struct BWGradArgs {
    const float* x; int ldx;     // [Nimg, H*xs, W*xs... see below]
    const float* dy; int ldy;
    float* out;                  // slabs [split][ntaps][Ci][Co]
    int Ci, Co;                  // channel counts (multiples of 4)
    int Nimg, H, W;              // pixel space of the TILES (the lower-resolution side for the transposed convolution)
    int KH, KW, dil;             // taps; x pixel of tap (r, s) = (h + (r - KH/2) * dil, w + (s - KW/2) * dil)     (dy_up == 0)
    int dy_up;                   // 1: k2-s2 transposed convolution: x [N,H,W,Ci] unshifted, dy [N,2H,2W,Co] read at (2h + r, 2w + s), taps = 4
    int tiles_h, tiles_w;        // 4 x 16 pixel tiles per image
    long total_tiles;
    int tiles_per_split;
};

constexpr int WT_H = 4, WT_W = 16;        // pixel tile
constexpr int WLD = 64 + 8;               // bf16 per LDS row ([pixel][64 channels + pad]); 144 B keeps every tr-read address 8-byte aligned

// Measured and NOT adopted (64 -> 64 @ 16 x 256^2: 0.335 ms as written): per-block precomputed staging offsets + branch-free interior path +
// immediate-offset transposing reads (0.377 ms) - unlike the forward kernel this one is not bound by its instruction count but by LDS:
// a 32 x 32 wave tile with nine taps reads ~1.1 KB of fragments per MFMA (rocprofv3: SQ_LDS_BANK_CONFLICT = 40 % of SQ_LDS_IDX_ACTIVE on
// the transposing reads, MFMA busy 10 %).  Next step: 64-row wave tiles per tap group and a conflict-free (XOR-swizzled) tile image.
// transposing fragment read: lane l of the wave gets, for the 32 channels c0 .. c0+31 (its channel = c0 + (l & 31)) and the 8 pixels
// rows[0..7] (LDS row index of k = 8*(l>>5) + j given by the caller through `row_of`), the 8 values [pixel j][channel].
// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of block row q, columns 4p..4p+3; lane i receives column i of the 4 rows.
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int c0, int row_q_lo, int row_q_hi) {
    // row_q_lo / row_q_hi: LDS row (pixel) this lane must ADDRESS for the first / second 4-pixel block
    const int lane = threadIdx.x & 63;
    const int p = lane & 3, g = (lane >> 4) & 1;            // g: which 16-channel half of the 32 channels
    const int col = c0 + 16 * g + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + row_q_lo * WLD + col));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + row_q_hi * WLD + col));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo; u.s[1] = hi;
    return u.b;
}

// NT = taps accumulated per block (9: 3x3 dilation 1 from one halo tile; 1: one tap per blockIdx.y with its own shifted tile)
template <int NT>
__global__ __launch_bounds__(256, 1) void wgrad_bf16_kernel(BWGradArgs g) {
    constexpr int HALO = NT == 9 ? 1 : 0;
    constexpr int XH = WT_H + 2 * HALO, XW = WT_W + 2 * HALO;      // x tile with halo
    constexpr int XROWS = XH * XW, YROWS = WT_H * WT_W;
    __shared__ __attribute__((aligned(16))) __bf16 xs[XROWS * WLD];
    __shared__ __attribute__((aligned(16))) __bf16 ys[YROWS * WLD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;               // wave tile: 32 ci x 32 co of the block's 64 x 64
    const int co_chunks = (g.Co + 63) / 64;
    const int ci0 = (blockIdx.x / co_chunks) * 64, co0 = (blockIdx.x % co_chunks) * 64;
    const int tap0 = NT == 9 ? 0 : blockIdx.y;
    const int tr = tap0 / g.KW, ts = tap0 % g.KW;
    const long t_begin = (long)blockIdx.z * g.tiles_per_split;
    long t_end = t_begin + g.tiles_per_split;
    if (t_end > g.total_tiles) t_end = g.total_tiles;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // staging: 16 threads cover the 64 channels (256 B fp32) of one pixel row of a tile
    const int sq = tid & 15, sr = tid >> 4;              // channel quad, first staged row
    constexpr int XPASS = (XROWS + 15) / 16, YPASS = YROWS / 16;
    const int q4 = (lane >> 2) & 3, kh8 = lane >> 5;     // tr-read address role of this lane: block row q4; k half (pixels 8*kh8 ..)
    const int Hy = g.dy_up ? 2 * g.H : g.H, Wy = g.dy_up ? 2 * g.W : g.W;

    // register prefetch: the global loads of tile t+1 are in flight while tile t multiplies
    f32x4 rx[XPASS], ry[YPASS];
    auto issue_loads = [&](long t) {
        const int tw = (int)(t % g.tiles_w);
        const long t2 = t / g.tiles_w;
        const int th = (int)(t2 % g.tiles_h);
        const int n = (int)(t2 / g.tiles_h);
        const int h0 = th * WT_H, w0 = tw * WT_W;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int row = sr + 16 * i;
            const int ry_ = row / XW, rx_ = row - ry_ * XW;
            int ih = h0 + ry_ - HALO, iw = w0 + rx_ - HALO;
            if (NT == 1 && !g.dy_up) { ih += (tr - g.KH / 2) * g.dil; iw += (ts - g.KW / 2) * g.dil; }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < XROWS && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W && ci0 + sq * 4 < g.Ci)
                v = *reinterpret_cast<const f32x4*>(g.x + (((long)n * g.H + ih) * g.W + iw) * g.ldx + ci0 + sq * 4);
            rx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {
            const int row = sr + 16 * i;
            const int ry_ = row / WT_W, rx_ = row - ry_ * WT_W;
            int oh = h0 + ry_, ow = w0 + rx_;
            const bool in = oh < g.H && ow < g.W;
            if (g.dy_up) { oh = 2 * oh + tr; ow = 2 * ow + ts; }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (in && co0 + sq * 4 < g.Co) v = *reinterpret_cast<const f32x4*>(g.dy + (((long)n * Hy + oh) * Wy + ow) * g.ldy + co0 + sq * 4);
            ry[i] = v;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < XPASS; ++i) {
            const int row = sr + 16 * i;
            if (row < XROWS) {
                bf16x4 b;
                b[0] = (__bf16)rx[i][0]; b[1] = (__bf16)rx[i][1]; b[2] = (__bf16)rx[i][2]; b[3] = (__bf16)rx[i][3];
                *reinterpret_cast<bf16x4*>(xs + row * WLD + sq * 4) = b;
            }
        }
#pragma unroll
        for (int i = 0; i < YPASS; ++i) {
            bf16x4 b;
            b[0] = (__bf16)ry[i][0]; b[1] = (__bf16)ry[i][1]; b[2] = (__bf16)ry[i][2]; b[3] = (__bf16)ry[i][3];
            *reinterpret_cast<bf16x4*>(ys + (sr + 16 * i) * WLD + sq * 4) = b;
        }
    };

    if (t_begin < t_end) issue_loads(t_begin);
    for (long t = t_begin; t < t_end; ++t) {
        __syncthreads();          // every wave has fetched the previous tile's fragments
        store_lds();
        __syncthreads();
        if (t + 1 < t_end) issue_loads(t + 1);
        // ---- 4 k-steps of 16 pixels (one tile row of 16 pixels each); B = dy fragments, shared by all taps
#pragma unroll
        for (int ky = 0; ky < WT_H; ++ky) {
            // k = ky*16 + 8*kh8 + j  ->  pixel (ky, 8*kh8 + j);  this lane addresses block rows q4 (first 4 pixels) and 4 + q4
            const int px = 8 * kh8 + q4;
            const bf16x8 bfrag = tr_frag(ys, wn * 32, ky * WT_W + px, ky * WT_W + px + 4);
#pragma unroll
            for (int t9 = 0; t9 < NT; ++t9) {
                const int r = NT == 9 ? t9 / 3 : HALO, s = NT == 9 ? t9 % 3 : HALO;
                const int xr = (ky + r) * XW + px + s;
                const bf16x8 afrag = tr_frag(xs, wm * 32, xr, xr + 4);
                acc[t9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[t9], 0, 0, 0);
            }
        }
    }
    // ---- write the slab: D[row = ci][col = co]
    const int li = lane & 31, lh = lane >> 5;
    const int ntaps_total = g.KH * g.KW;
    float* slab = g.out + (long)blockIdx.z * ntaps_total * g.Ci * g.Co;
    const int co = co0 + wn * 32 + li;
#pragma unroll
    for (int t9 = 0; t9 < NT; ++t9) {
        const int tap = NT == 9 ? t9 : tap0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (ci < g.Ci && co < g.Co) slab[((long)tap * g.Ci + ci) * g.Co + co] = acc[t9][r];
        }
    }
}

// out[i] = sum_k slabs[k][i]: block = 32 float4 columns x 8 split-lanes, partial sums combined through LDS in a fixed order (bitwise
// reproducible, no atomics).  (One thread per column walking all splits left 36 blocks on the chip for a 64 x 64 filter: 100 us.)
__global__ __launch_bounds__(256) void slab_reduce_bf16path(const float* __restrict__ slabs, float* __restrict__ out, long n, int nsplit) {
    __shared__ f32x4 red[256];
    const int col = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const long i = ((long)blockIdx.x * 32 + col) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n)
        for (int k = kl; k < nsplit; k += 8) s += *reinterpret_cast<const f32x4*>(slabs + (long)k * n + i);
    red[threadIdx.x] = s;
    __syncthreads();
    if (kl == 0 && i < n) {
#pragma unroll
        for (int j = 1; j < 8; ++j) s += red[j * 32 + col];
        *reinterpret_cast<f32x4*>(out + i) = s;
    }
}

struct WPlan { int tiles_h, tiles_w; long total; int splits, tps; };
WPlan wgrad_bf16_plan(int n_img, int h, int w, int cin, int cout, int ntaps_grid) {
    WPlan p;
    p.tiles_h = cdiv(h, WT_H); p.tiles_w = cdiv(w, WT_W);
    p.total = (long)n_img * p.tiles_h * p.tiles_w;
    const long blocks = (long)cdiv(cin, 64) * cdiv(cout, 64) * ntaps_grid;
    long want = blocks >= 256 ? 1 : (512 + blocks - 1) / blocks;      // ~2 blocks per CU; enough (ci, co) tiles: no split at all
    if (want > p.total / 8) want = p.total / 8;         // at least 8 tiles (512 pixels) per split
    if (want > 512) want = 512;
    if (want < 1) want = 1;
    p.tps = (int)cdiv(p.total, want);
    p.splits = (int)cdiv(p.total, p.tps);
    return p;
}

}  // namespace

extern "C" long runet_bf16_pack_elems(int taps, int k, int n) { return (long)taps * ((k + 7) / 8) * n * 8; }

extern "C" int runet_bf16_pack_weights(const float* w_hwio, void* packed, int taps, int cin, int cout, int transpose, void* stream) {
    RUNET_REQUIRE(w_hwio && packed && taps > 0 && cin > 0 && cout > 0, "bad arguments");
    RUNET_REQUIRE(((uintptr_t)packed % 16) == 0, "packed buffer must be 16-byte aligned");
    const long total = (long)taps * (((transpose ? cout : cin) + 7) / 8) * (transpose ? cin : cout);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bf16_pack_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, (__bf16*)packed, taps, cin, cout, transpose);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_conv_igemm_bf16(const float* x, int ldx, const void* wpacked, const float* bias, float* y, int ldy, int n_img, int h, int w_,
                                     int cin, int cout, int kh, int kw, int dil, int mode, int accumulate, void* stream) {
    RUNET_REQUIRE(x && wpacked && y, "null pointer");
    RUNET_REQUIRE(cin > 0 && cin % 4 == 0 && cout > 0 && cout % 4 == 0, "channel counts must be positive multiples of 4");
    RUNET_REQUIRE(ldx >= cin && ldx % 4 == 0 && ldy % 4 == 0, "pixel strides must be multiples of 4 floats and cover the channels");
    RUNET_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)wpacked % 16) == 0 && ((uintptr_t)y % 16) == 0, "pointers must be 16-byte aligned");
    RUNET_REQUIRE(n_img > 0 && h > 0 && w_ > 0, "empty iteration space");
    RUNET_REQUIRE((kh == 1 && kw == 1) || (kh == 3 && kw == 3) || (kh == 2 && kw == 2), "kernel must be 1x1, 3x3 or 2x2(transposed)");
    BGemmArgs a{};
    a.x = x; a.ldx = ldx; a.w = (const __bf16*)wpacked; a.bias = bias; a.y = y; a.ldy = ldy;
    a.Nimg = n_img; a.accumulate = accumulate; a.KH = kh; a.KW = kw;
    a.K = cin; a.K8 = (cin + 7) / 8; a.Ncols = cout;
    a.H = h; a.W = w_; a.Hin = h; a.Win = w_; a.a_scale = 1; a.Hout = h; a.Wout = w_; a.o_scale = 1;
    int gz = 1;
    switch (mode) {
    case RUNET_CONV_FWD:
        RUNET_REQUIRE(kh != 2 && ldy >= cout, "2x2 kernels are transposed-conv only; ldy must cover cout");
        a.tdh = dil; a.tdw = dil; a.bh = -dil * (kh / 2); a.bw = -dil * (kw / 2);
        break;
    case RUNET_CONV_DGRAD:       // x := dy [.., cin = conv Cout], y := dx [.., cout = conv Cin]; weights packed with transpose = 1
        RUNET_REQUIRE(kh != 2, "use RUNET_CONVT_DGRAD for 2x2");
        a.tdh = -dil; a.tdw = -dil; a.bh = dil * (kh / 2); a.bw = dil * (kw / 2);
        break;
    case RUNET_CONVT_FWD:        // y[N,2h,2w,cout] = convT_k2s2(x[N,h,w,cin]); weights packed with transpose = 0 (4 taps)
        RUNET_REQUIRE(kh == 2 && kw == 2, "transposed conv is 2x2 stride 2");
        a.KH = 1; a.KW = 1; a.Hout = 2 * h; a.Wout = 2 * w_; a.o_scale = 2; a.z_taps = 2; gz = 4;
        break;
    case RUNET_CONVT_DGRAD:      // dx[N,h,w,cout] from dy[N,2h,2w,cin]; weights packed with transpose = 1 (4 taps)
        RUNET_REQUIRE(kh == 2 && kw == 2, "transposed conv is 2x2 stride 2");
        a.Hin = 2 * h; a.Win = 2 * w_; a.a_scale = 2; a.tdh = 1; a.tdw = 1;
        break;
    default:
        RUNET_REQUIRE(false, "unknown mode");
    }
    hipStream_t st = (hipStream_t)stream;
    if (kh == 3 && dil == 1 && (mode == RUNET_CONV_FWD || mode == RUNET_CONV_DGRAD)) {
        C3Args c{};
        c.x = x; c.ldx = ldx; c.w = (const __bf16*)wpacked; c.bias = bias; c.y = y; c.ldy = ldy; c.K = cin; c.K8 = (cin + 7) / 8; c.Ncols = cout;
        // patch height 16 unless that leaves fewer than two blocks per CU (the deepest levels): then 8 doubles the block count
        // (measured: 1024 -> 1024 @ 16 x 16^2 0.119 -> 0.105 ms; everywhere else 16 is 5-25 % faster).  RUNET_C3_PH=8|16 overrides.
        static const int force_ph = getenv("RUNET_C3_PH") ? atoi(getenv("RUNET_C3_PH")) : 0;
        const long blocks16 = (long)n_img * cdiv(h, 16) * cdiv(w_, PT) * cdiv(cout, 64);
        const int ph = force_ph ? force_ph : (blocks16 < 512 ? 8 : 16);
        c.Nimg = n_img; c.H = h; c.W = w_; c.tiles_h = cdiv(h, ph); c.tiles_w = cdiv(w_, PT); c.nchunks_n = cdiv(cout, 64);
        c.flip = mode == RUNET_CONV_DGRAD ? 1 : 0; c.accumulate = accumulate;
        const size_t lds = ((ph + 2) * HP * LDA + 2 * 4096) * sizeof(__bf16);
        const long blocks = (long)n_img * c.tiles_h * c.tiles_w * c.nchunks_n;
        RUNET_REQUIRE(blocks < (1L << 31), "grid too large");
        if (ph == 8) hipLaunchKernelGGL(conv3x3_bf16_kernel<8>, dim3((unsigned)blocks), dim3(256), lds, st, c);
        else hipLaunchKernelGGL(conv3x3_bf16_kernel<16>, dim3((unsigned)blocks), dim3(256), lds, st, c);
        RUNET_CHECK_LAUNCH();
    }
    if (cout <= 32) launch_bgemm<32, 32, 32>(a, gz, st);
    else if (cout <= 64) launch_bgemm<64, 64, 32>(a, gz, st);
    else launch_bgemm<128, 64, 64>(a, gz, st);
    RUNET_CHECK_LAUNCH();
}

extern "C" long runet_conv_wgrad_bf16_workspace_floats(int n_img, int h, int w_, int cin, int cout, int kh, int kw, int dil, int transposed) {
    const bool halo9 = (!transposed && kh == 3 && kw == 3 && dil == 1);
    const WPlan p = wgrad_bf16_plan(n_img, h, w_, cin, cout, halo9 ? 1 : kh * kw);
    return p.splits > 1 ? (long)p.splits * kh * kw * cin * cout : 0;
}

extern "C" int runet_conv_wgrad_bf16(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats,
                                     int n_img, int h, int w_, int cin, int cout, int kh, int kw, int dil, int transposed, void* stream) {
    RUNET_REQUIRE(x && dy && dw, "null pointer");
    RUNET_REQUIRE(cin > 0 && cin % 4 == 0 && cout > 0 && cout % 4 == 0, "channel counts must be positive multiples of 4");
    RUNET_REQUIRE(ldx >= cin && ldx % 4 == 0 && ldy >= cout && ldy % 4 == 0, "bad pixel strides");
    RUNET_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dw % 16) == 0, "pointers must be 16-byte aligned");
    RUNET_REQUIRE((kh == 1 && kw == 1) || (kh == 3 && kw == 3) || (kh == 2 && kw == 2 && transposed), "unsupported kernel size");
    hipStream_t st = (hipStream_t)stream;
    const bool halo9 = (!transposed && kh == 3 && kw == 3 && dil == 1);
    const int ntaps = kh * kw;
    WPlan p = wgrad_bf16_plan(n_img, h, w_, cin, cout, halo9 ? 1 : ntaps);
    const long wsize = (long)ntaps * cin * cout;
    if (p.splits > 1 && (workspace == nullptr || workspace_floats < p.splits * wsize)) {
        long s2 = workspace ? workspace_floats / wsize : 1;
        if (s2 < 1) s2 = 1;
        p.tps = (int)cdiv(p.total, s2);
        p.splits = (int)cdiv(p.total, p.tps);
    }
    BWGradArgs a{};
    a.x = x; a.ldx = ldx; a.dy = dy; a.ldy = ldy; a.out = p.splits > 1 ? workspace : dw;
    a.Ci = cin; a.Co = cout; a.Nimg = n_img; a.H = h; a.W = w_; a.KH = kh; a.KW = kw; a.dil = dil; a.dy_up = transposed ? 1 : 0;
    a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.total_tiles = p.total; a.tiles_per_split = p.tps;
    const int tiles_c = cdiv(cin, 64) * cdiv(cout, 64);
    if (halo9) hipLaunchKernelGGL((wgrad_bf16_kernel<9>), dim3(tiles_c, 1, p.splits), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_bf16_kernel<1>), dim3(tiles_c, ntaps, p.splits), dim3(256), 0, st, a);
    if (p.splits > 1) hipLaunchKernelGGL(slab_reduce_bf16path, dim3(cdiv(wsize, 128)), dim3(256), 0, st, workspace, dw, wsize, p.splits);
    RUNET_CHECK_LAUNCH();
}
