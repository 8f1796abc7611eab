// Implicit-GEMM convolutions on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces the ATen/oneDNN kernels behind every nn.Conv2d / nn.ConvTranspose2d of the reference's
// RobustUNet (/root/reference/Main_Final.py:157,159,172,126,131,205-208,261-270) for forward, data
// gradient and weight gradient.  One kernel template serves 1x1 / 3x3 (dilation 1,2,4) convolutions
// and the k2-s2 transposed convolution; the geometry struct says how a GEMM row (a pixel of the
// iteration space) maps to source and destination pixels.
//
// Data layout in HBM: activations NHWC fp32 (pixel stride `ld` floats, so a tensor may be a channel
// slice of a wider concat buffer); weights "HWIO": w[tap][cin][cout], cout contiguous.
//
// fwd/dgrad kernel:  out[p][n] = sum_tap sum_k  src[pix(p,tap)][k] * B_tap[k][n]
//   GEMM M = pixels (tile BM), N = output channels (tile BN), K = taps x channels (step 16 channels).
//   A tile (BM x 16) and B tile (16 x BN) are register-staged into double-buffered LDS
//   (global_load_dwordx4 issued one k-step ahead, ds_write after the MFMAs of the current step),
//   one barrier per k-step.  A rows are padded to 20 floats so the ds_read_b128 fragment reads are
//   bank-conflict free; lane (i, h) of a wave reads 4 consecutive k for row i and feeds them to 4
//   MFMAs (the k order inside a group of 8 is permuted identically for A and B).
// wgrad kernel:      dW[tap][ci][co] = sum_p  x[pix(p,tap)][ci] * dy[p][co]
//   GEMM M = cin, N = cout, K = pixels (step 16 pixels); split over pixel ranges (grid.z) into
//   partial slabs that a second kernel sums in a fixed order (bitwise reproducible, no atomics).
#include "runet_common.h"
#include <stdlib.h>
#include "../../include/runet_hip.h"

namespace {

struct IGemmArgs {
    const float* x; int ldx;      // A source, [Nimg, Hin, Win, ldx]
    const float* w; long w_tap_stride; int w_sk, w_sn;   // B(tap,k,n) = w[tap*w_tap_stride + k*w_sk + n*w_sn]
    const float* bias;            // [Ncols] or nullptr
    float* y; int ldy;            // destination [Nimg, Hout, Wout, ldy]
    int K, Kx, Kvalid, Ncols;     // K: k-loop extent (multiple of 16); x holds Kx channels; rows >= Kvalid of B are zero
    int Nimg, H, W;               // iteration space
    int Hin, Win, a_scale;        // source pixel = (h*a_scale + bh + r*tdh, w*a_scale + bw + s*tdw)
    int KH, KW, tdh, tdw, bh, bw;
    int Hout, Wout, o_scale, o_dh, o_dw;   // destination pixel = (h*o_scale + o_dh, w*o_scale + o_dw)
    int z_taps;                   // >0: blockIdx.z selects one weight tap AND the destination offset (convT k2 fwd)
    int accumulate;               // y += result
    int a_div;                    // >1: source coordinate must be divisible by a_div and is divided by it (data gradient of a strided conv)
    int zmode4;                   // 1: ConvTranspose2d(k4,s2,p1) forward: blockIdx.z = output parity class (ph,pw); taps (a,b) in 2x2 read
                                  //    weight tap (1-ph+2a, 1-pw+2b) of the 4x4 filter at source offset (ph-a, pw-b)
    long zs_x, zs_w, zs_y;        // zbatch: blockIdx.z selects one GEMM of a batch (element strides of x, w and y)
    int zbatch;
};

// SIMPLE: every tap of every row lies inside the source image (plain 1x1 convolutions, the k2-s2 transposed convolution both ways), K a
// multiple of 16 and fully backed by x and w: every address is a per-thread pointer set up once plus a wave-uniform offset per k-step,
// nothing is predicated (rows / columns beyond the problem are clamped
// to valid addresses: their products land in output rows / columns that are never stored).  The general loader recomputes tap geometry,
// bounds and 64-bit addresses every k-step: ~220 VALU instructions (20 of them quarter-rate integer multiplies) per 32 MFMAs, and fp32
// MFMAs share SIMD cycles with VALU work (conv_winograd.hip) - the 1x1 forward / data-gradient launches ran at 52-69 TFLOP/s with it.
template <int BM, int BN, int WM, int WN, bool KCONTIG, bool SIMPLE = false>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IGemmArgs g) {
    constexpr int LDA = 20;
    constexpr int LDB = BN + 4;
    constexpr int A_ELEMS = BM * LDA;
    constexpr int B_ELEMS = 16 * LDB;
    constexpr int STAGE = A_ELEMS + B_ELEMS;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    constexpr int AROWS = BM / 64;
    constexpr int BREGS = (BN >= 64) ? BN / 64 : 1;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const long P = (long)g.Nimg * g.H * g.W;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int HW = g.H * g.W;

    const float* wbase = g.w;
    const float* xb = g.x;
    float* yb = g.y;
    if (g.zbatch) { xb += (long)blockIdx.z * g.zs_x; wbase += (long)blockIdx.z * g.zs_w; yb += (long)blockIdx.z * g.zs_y; }
    int o_dh = g.o_dh, o_dw = g.o_dw;
    const int ph4 = (blockIdx.z >> 1) & 1, pw4 = blockIdx.z & 1;
    if (g.zmode4) { o_dh = ph4; o_dw = pw4; }
    if (g.z_taps > 0) {
        wbase += (long)blockIdx.z * g.w_tap_stride;
        o_dh = blockIdx.z / g.z_taps;   // z_taps = taps per row (2 for the 2x2 transposed conv)
        o_dw = blockIdx.z % g.z_taps;
    }

    // ---- per-thread A staging rows ----
    const int akq = tid & 3;
    long a_img[AROWS];
    int a_h[AROWS], a_w[AROWS];
    bool a_ok[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const long p = m0 + (tid >> 2) + 64 * i;
        a_ok[i] = p < P;
        const long pp = a_ok[i] ? p : 0;
        const int n = (int)(pp / HW);
        const int rem = (int)(pp - (long)n * HW);
        const int h = rem / g.W;
        a_img[i] = (long)n * g.Hin * g.Win;
        a_h[i] = h * g.a_scale + g.bh;
        a_w[i] = (rem - h * g.W) * g.a_scale + g.bw;
    }

    const int KC = g.K >> 4;
    const int ntaps = (g.z_taps > 0) ? 1 : g.KH * g.KW;
    const int nks = ntaps * KC;

    f32x4 ra[AROWS];
    f32x4 rb[BREGS];
    int ld_tr = 0, ld_ts = 0, ld_kc = 0, ld_tap = 0;   // counters of the NEXT tile to load

    // SIMPLE: running pointers of this thread's A rows and B float4s
    const float* ap[AROWS];
    const float* bp[BREGS];
    long bstep = 0;
    if constexpr (SIMPLE) {
#pragma unroll
        for (int i = 0; i < AROWS; ++i)                  // a_ok false: pixel 0 (a_img = a_h = a_w = 0 from pp = 0 above)
            ap[i] = xb + (a_img[i] + (long)a_h[i] * g.Win + a_w[i]) * g.ldx + akq * 4;
        if constexpr (!KCONTIG) {
            constexpr int NQ = BN / 4;
            constexpr int KSTEP = 256 / NQ;
#pragma unroll
            for (int i = 0; i < BREGS; ++i) {
                const int nq = tid % NQ;
                int kr = tid / NQ + KSTEP * i, n = n0 + nq * 4;
                kr = kr < 16 ? kr : 15;
                n = n < g.Ncols ? n : g.Ncols - 4;
                bp[i] = wbase + (long)kr * g.w_sk + n;
            }
            bstep = 16L * g.w_sk;
        } else {
#pragma unroll
            for (int i = 0; i < BREGS; ++i) {
                int n = n0 + (tid >> 2) + 64 * i;
                n = n < g.Ncols ? n : g.Ncols - 1;
                bp[i] = wbase + (long)n * g.w_sn + akq * 4;
            }
            bstep = 16;
        }
    }

    auto load_tile = [&]() {
        if constexpr (SIMPLE) {
            // wave-uniform offsets of the tile (tap shift of the source pixel, channel chunk, weight tap): scalar arithmetic only
            const long offa = ((long)(ld_tr * g.tdh) * g.Win + ld_ts * g.tdw) * g.ldx + ld_kc * 16;
            const long offb = (long)ld_tap * g.w_tap_stride + (long)ld_kc * bstep;
#pragma unroll
            for (int i = 0; i < AROWS; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + offa);
#pragma unroll
            for (int i = 0; i < BREGS; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i] + offb);
            ++ld_tap;
            if (++ld_ts == g.KW) {
                ld_ts = 0;
                if (++ld_tr == g.KH) { ld_tr = 0; ld_tap = 0; ++ld_kc; }
            }
            return;
        }
        int dh = ld_tr * g.tdh, dw = ld_ts * g.tdw;
        int wtap = ld_tap;
        if (g.zmode4) { dh = ph4 - ld_tr; dw = pw4 - ld_ts; wtap = (1 - ph4 + 2 * ld_tr) * 4 + (1 - pw4 + 2 * ld_ts); }
        const int kofs = ld_kc * 16;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            int ih = a_h[i] + dh, iw = a_w[i] + dw;
            bool ok = a_ok[i] && kofs + akq * 4 < g.Kx;
            if (g.a_div > 1) {
                ok = ok && ih >= 0 && iw >= 0 && (ih % g.a_div) == 0 && (iw % g.a_div) == 0;
                ih /= g.a_div; iw /= g.a_div;
            }
            ok = ok && (unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(xb + ((a_img[i] + (long)ih * g.Win + iw) * g.ldx + kofs + akq * 4));
            ra[i] = v;
        }
        const float* wt = wbase + (long)wtap * g.w_tap_stride;
        if constexpr (!KCONTIG) {   // n contiguous: 16 k-rows x BN/4 float4
            constexpr int NQ = BN / 4;
            constexpr int KSTEP = 256 / NQ;
#pragma unroll
            for (int i = 0; i < BREGS; ++i) {
                const int nq = tid % NQ, kr = tid / NQ + KSTEP * i;
                const int k = kofs + kr, n = n0 + nq * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kr < 16 && k < g.Kvalid && n < g.Ncols) v = *reinterpret_cast<const f32x4*>(wt + (long)k * g.w_sk + n);
                rb[i] = v;
            }
        } else {   // k contiguous: BN n-rows x 4 float4
#pragma unroll
            for (int i = 0; i < BREGS; ++i) {
                const int nr = (tid >> 2) + 64 * i;
                const int n = n0 + nr, k = kofs + akq * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (nr < BN && n < g.Ncols && k < g.Kvalid) v = *reinterpret_cast<const f32x4*>(wt + (long)n * g.w_sn + k);
                rb[i] = v;
            }
        }
        // advance counters: tap fastest (column, then row), then the channel chunk - the 9 shifted reads of one 16-channel
        // chunk touch the same cache lines back to back, so 8 of them are served by L1/L2 instead of the fabric
        ++ld_tap;
        if (++ld_ts == g.KW) {
            ld_ts = 0;
            if (++ld_tr == g.KH) { ld_tr = 0; ld_tap = 0; ++ld_kc; }
        }
    };

    auto store_tile = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + A_ELEMS;
#pragma unroll
        for (int i = 0; i < AROWS; ++i)
            *reinterpret_cast<f32x4*>(As + ((tid >> 2) + 64 * i) * LDA + akq * 4) = ra[i];
        if constexpr (!KCONTIG) {
            constexpr int NQ = BN / 4;
            constexpr int KSTEP = 256 / NQ;
#pragma unroll
            for (int i = 0; i < BREGS; ++i) {
                const int nq = tid % NQ, kr = tid / NQ + KSTEP * i;
                if (kr < 16) *reinterpret_cast<f32x4*>(Bs + kr * LDB + nq * 4) = rb[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < BREGS; ++i) {
                const int nr = (tid >> 2) + 64 * i;
                if (nr < BN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Bs[(akq * 4 + e) * LDB + nr] = rb[i][e];
                }
            }
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
        const float* As = smem + cur * STAGE;
        const float* Bs = As + A_ELEMS;
        // all fragments of the k-step are read up front (2*TM ds_read_b128 + 4*TN ds_read2_b32), then the MFMA chain runs
        // without LDS waits inside it; the scheduler is told to keep that order (it otherwise sinks each B read to its use)
        f32x4 af[2][TM];
        float bf[2][TN][4];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[kh][a] = *reinterpret_cast<const f32x4*>(As + (wm0 + a * 32 + li) * LDA + kh * 8 + 4 * lh);
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) bf[kh][b][q] = Bs[(kh * 8 + 4 * lh + q) * LDB + wn0 + b * 32 + li];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * TM + 4 * TN, 0);     // DS reads first
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kh][a][q], bf[kh][b][q], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 8 * TM * TN, 0);            // then the whole MFMA chain
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: D[row = pixel][col = n]; lane holds col (lane&31), rows (r&3)+8*(r>>2)+4*(lane>>5) ----
    const bool same_pix = (g.o_scale == 1 && g.Hout == g.H && g.Wout == g.W);   // plain conv: destination pixel == GEMM row
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
            // accumulate mode: the old values of 8 rows are all requested before the first add, so the loads overlap
            // instead of one load -> wait -> store per element (8 at a time keeps the register count where it was)
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
                drow[i] = p < P ? yb + op * g.ldy : nullptr;
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

// ------------------------------------------------------------------------------------------ wgrad
struct WGradArgs {
    const float* x; int ldx;     // [Nimg, Hin, Win, ldx], channels [0, Kci)
    const float* dy; int ldy;    // [Nimg, Hout, Wout, ldy], channels [0, Nco)
    float* out;                  // slabs: [split][ntaps][Kvalid][Nco]
    int Kci, Kvalid, Nco;        // Kci multiple of 16 (padded read width of x); only ci < Kvalid are written
    int Nimg, H, W;              // iteration space (pixels summed over)
    int Hin, Win, a_scale, KH, KW, tdh, tdw, bh, bw;   // x pixel = (h*a_scale + bh + r*tdh, ...)
    int Hout, Wout, o_scale;     // dy pixel = (h*o_scale + o_dh, w*o_scale + o_dw)
    int tap_on_output;           // 1: tap (r,s) offsets the dy pixel (transposed conv), x pixel unshifted
    int o_bh, o_bw;              // base offset of the dy pixel when tap_on_output (k4-s2-p1 transposed conv: -1)
    long pix_per_split;          // multiple of 16
    long tap_bs_x, tap_bs_dy;    // batched form: tap t reads x + t*tap_bs_x and dy + t*tap_bs_dy (no spatial shift)
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WGradArgs g) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int A_ELEMS = 16 * LDA, B_ELEMS = 16 * LDB, STAGE = A_ELEMS + B_ELEMS;
    constexpr int TM = WM / 32, TN = WN / 32, WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    constexpr int AQ = BM / 4, BQ = BN / 4;              // float4 per pixel row
    constexpr int AREGS = (16 * AQ + 255) / 256, BREGS = (16 * BQ + 255) / 256;
    constexpr int APSTEP = 256 / AQ, BPSTEP = 256 / BQ;  // pixel rows covered per pass (AQ,BQ <= 64... see launch)

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int n_tiles = (g.Nco + BN - 1) / BN;
    const int ci0 = (blockIdx.x / n_tiles) * BM, co0 = (blockIdx.x % n_tiles) * BN;
    const int tap = blockIdx.y, tr = tap / g.KW, ts = tap % g.KW;
    const long P = (long)g.Nimg * g.H * g.W;
    const long p_begin = (long)blockIdx.z * g.pix_per_split;
    long p_end = p_begin + g.pix_per_split;
    if (p_end > P) p_end = P;
    const int HW = g.H * g.W;

    int xdh = g.bh, xdw = g.bw, ydh = 0, ydw = 0;
    if (g.tap_on_output) { ydh = tr + g.o_bh; ydw = ts + g.o_bw; } else { xdh += tr * g.tdh; xdw += ts * g.tdw; }

    const float* xb = g.x + (long)tap * g.tap_bs_x;
    const float* dyb = g.dy + (long)tap * g.tap_bs_dy;
    f32x4 ra[AREGS], rb[BREGS];
    auto load_tile = [&](long pk) {   // pk: first pixel of the 16-pixel k-step
#pragma unroll
        for (int i = 0; i < AREGS; ++i) {
            const int cq = tid % AQ, pr = tid / AQ + APSTEP * i;
            const long p = pk + pr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pr < 16 && p < p_end && ci0 + cq * 4 < g.Kci) {
                const int n = (int)(p / HW);
                const int rem = (int)(p - (long)n * HW);
                const int h = rem / g.W, w = rem - h * g.W;
                const int ih = h * g.a_scale + xdh, iw = w * g.a_scale + xdw;
                if ((unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win)
                    v = *reinterpret_cast<const f32x4*>(xb + (((long)n * g.Hin + ih) * g.Win + iw) * g.ldx + ci0 + cq * 4);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BREGS; ++i) {
            const int cq = tid % BQ, pr = tid / BQ + BPSTEP * i;
            const long p = pk + pr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pr < 16 && p < p_end && co0 + cq * 4 < g.Nco) {
                const int n = (int)(p / HW);
                const int rem = (int)(p - (long)n * HW);
                const int h = rem / g.W, w = rem - h * g.W;
                const int oh = h * g.o_scale + ydh, ow = w * g.o_scale + ydw;
                if ((unsigned)oh < (unsigned)g.Hout && (unsigned)ow < (unsigned)g.Wout)
                    v = *reinterpret_cast<const f32x4*>(dyb + (((long)n * g.Hout + oh) * g.Wout + ow) * g.ldy + co0 + cq * 4);
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + A_ELEMS;
#pragma unroll
        for (int i = 0; i < AREGS; ++i) {
            const int cq = tid % AQ, pr = tid / AQ + APSTEP * i;
            if (pr < 16) *reinterpret_cast<f32x4*>(As + pr * LDA + cq * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BREGS; ++i) {
            const int cq = tid % BQ, pr = tid / BQ + BPSTEP * i;
            if (pr < 16) *reinterpret_cast<f32x4*>(Bs + pr * LDB + cq * 4) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nks = (int)((p_end - p_begin + 15) / 16);
    if (nks > 0) {
        load_tile(p_begin);
        store_tile(0);
    }
    __syncthreads();
    int cur = 0;
    for (int ks = 0; ks < nks; ++ks) {
        const bool more = ks + 1 < nks;
        if (more) load_tile(p_begin + (long)(ks + 1) * 16);
        const float* As = smem + cur * STAGE;
        const float* Bs = As + A_ELEMS;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = As[(2 * s + lh) * LDA + wm0 + a * 32 + li];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = Bs[(2 * s + lh) * LDB + wn0 + b * 32 + li];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    const int ntaps = g.KH * g.KW;
    float* slab = g.out + ((long)blockIdx.z * ntaps + tap) * g.Kvalid * g.Nco;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int co = co0 + wn0 + b * 32 + li;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < g.Nco && ci < g.Kvalid) slab[(long)ci * g.Nco + co] = acc[a][b][r];
            }
    }
}


// ------------------------------------------------------------------------------------------ wgrad, tile-resident form
// For 3x3 (dilation 1) and 1x1 convolutions: a block keeps a 4x16-pixel tile of dy and the matching x tile (with its
// 1-pixel halo for 3x3) in LDS and accumulates ALL taps from it, so x and dy are read from HBM/L2 once per
// (cin-chunk, cout-chunk) pair instead of once per tap; it then walks `tiles_per_split` tiles with the accumulators in
// registers (9 taps x one 32x32 tile per wave = 144 accumulator registers for 3x3) and writes one partial slab.
// The next tile's global loads are issued before the current tile's MFMAs (register prefetch).
struct WGradTileArgs {
    const float* x; int ldx;
    const float* dy; int ldy;
    float* out;                 // slabs [split][KS*KS][Kvalid][Nco]
    int Kci, Kvalid, Nco;
    int Nimg, H, W;
    int tiles_h, tiles_w, total_tiles, tiles_per_split, co_chunks;
    int dy_up;                  // 1: transposed conv (KS == 1 only): dy pixel of x pixel (h, w) is (2h + z/2, 2w + z%2), z = blockIdx.z
};

template <int KS, int TM, int TN>
__global__ __launch_bounds__(256, 2) void wgrad_tile_kernel(WGradTileArgs g) {
    constexpr int CI = 64 * TM, CO = 64 * TN;
    constexpr int TH = 4, TW = 16, NPX = TH * TW;
    constexpr int HALO = (KS == 3) ? 1 : 0;
    constexpr int HTW = TW + 2 * HALO, HTH = TH + 2 * HALO, NHPX = HTW * HTH;
    constexpr int XQ = CI / 4, YQ = CO / 4;
    constexpr int NXV = (NHPX * XQ + 255) / 256, NYV = (NPX * YQ + 255) / 256;
    constexpr int NT = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* dys = smem + NHPX * CI;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, lh = lane >> 5;
    const int ci0 = (blockIdx.x / g.co_chunks) * CI, co0 = (blockIdx.x % g.co_chunks) * CO;
    const int t_begin = blockIdx.y * g.tiles_per_split;
    const int t_end = min(g.total_tiles, t_begin + g.tiles_per_split);
    const long P = (long)g.Nimg * g.H * g.W;
    const int tiles_per_img = g.tiles_h * g.tiles_w;

    f32x4 rx[NXV], ry[NYV];
    auto prefetch = [&](int t) {
        int n = 0, h0 = 0, w0 = 0;
        if constexpr (KS == 3) {
            n = t / tiles_per_img;
            const int rem = t - n * tiles_per_img;
            const int th = rem / g.tiles_w;
            h0 = th * TH; w0 = (rem - th * g.tiles_w) * TW;
        }
#pragma unroll
        for (int v = 0; v < NXV; ++v) {
            const int idx = tid + 256 * v;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (idx < NHPX * XQ) {
                const int px = idx / XQ, cq = idx % XQ;
                if (ci0 + cq * 4 < g.Kci) {
                    if constexpr (KS == 3) {
                        const int hy = px / HTW, hx = px - hy * HTW;
                        const int ih = h0 - 1 + hy, iw = w0 - 1 + hx;
                        if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
                            val = *reinterpret_cast<const f32x4*>(g.x + (((long)n * g.H + ih) * g.W + iw) * g.ldx + ci0 + cq * 4);
                    } else {
                        const long p = (long)t * NPX + px;
                        if (p < P) val = *reinterpret_cast<const f32x4*>(g.x + p * g.ldx + ci0 + cq * 4);
                    }
                }
            }
            rx[v] = val;
        }
#pragma unroll
        for (int v = 0; v < NYV; ++v) {
            const int idx = tid + 256 * v;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (idx < NPX * YQ) {
                const int px = idx / YQ, cq = idx % YQ;
                if (co0 + cq * 4 < g.Nco) {
                    if constexpr (KS == 3) {
                        const int oh = h0 + (px >> 4), ow = w0 + (px & 15);
                        if (oh < g.H && ow < g.W)
                            val = *reinterpret_cast<const f32x4*>(g.dy + (((long)n * g.H + oh) * g.W + ow) * g.ldy + co0 + cq * 4);
                    } else {
                        const long p = (long)t * NPX + px;
                        if (p < P) {
                            long q = p;
                            if (g.dy_up) {
                                const int hw = g.H * g.W;
                                const int ni = (int)(p / hw);
                                const int rem = (int)(p - (long)ni * hw);
                                const int hh = rem / g.W, ww = rem - hh * g.W;
                                q = ((long)ni * 2 * g.H + 2 * hh + (blockIdx.z >> 1)) * (2 * g.W) + 2 * ww + (blockIdx.z & 1);
                            }
                            val = *reinterpret_cast<const f32x4*>(g.dy + q * g.ldy + co0 + cq * 4);
                        }
                    }
                }
            }
            ry[v] = val;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int v = 0; v < NXV; ++v) {
            const int idx = tid + 256 * v;
            if (idx < NHPX * XQ) *reinterpret_cast<f32x4*>(xs + idx * 4) = rx[v];
        }
#pragma unroll
        for (int v = 0; v < NYV; ++v) {
            const int idx = tid + 256 * v;
            if (idx < NPX * YQ) *reinterpret_cast<f32x4*>(dys + idx * 4) = ry[v];
        }
    };

    f32x16 acc[NT][TM][TN];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.f;

    if (t_begin < t_end) prefetch(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        stash();
        __syncthreads();
        if (t + 1 < t_end) prefetch(t + 1);
#pragma unroll 2
        for (int m = 0; m < NPX / 2; ++m) {
            const int p = 2 * m + lh;
            const int ty = p >> 4, tx = p & 15;
            float bf[TN];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = dys[p * CO + (wc * TN + b) * 32 + li];
#pragma unroll
            for (int r = 0; r < KS; ++r)
#pragma unroll
                for (int q = 0; q < KS; ++q) {
                    float af[TM];
#pragma unroll
                    for (int a = 0; a < TM; ++a) af[a] = xs[((ty + r) * HTW + tx + q) * CI + (wr * TM + a) * 32 + li];
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[r * KS + q][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[r * KS + q][a][b], 0, 0, 0);
                }
        }
    }

#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* slab = g.out + (((long)blockIdx.y * gridDim.z + blockIdx.z) * NT + t) * g.Kvalid * g.Nco;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int co = co0 + (wc * TN + b) * 32 + li;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = ci0 + (wr * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (co < g.Nco && ci < g.Kvalid) slab[(long)ci * g.Nco + co] = acc[t][a][b][r];
                }
        }
    }
}

struct TilePlan { int tm, tn, ci_chunks, co_chunks, total_tiles, tiles_h, tiles_w, splits, tps; };

static TilePlan wgrad_tile_plan(int n_img, int h, int w_, int cin_w, int cout, int ks) {
    TilePlan p{};
    if (ks == 3) { p.tm = 1; p.tn = 1; }
    else { p.tm = cin_w > 64 ? 2 : 1; p.tn = cout > 64 ? 2 : 1; }
    p.ci_chunks = cdiv(cin_w, 64 * p.tm);
    p.co_chunks = cdiv(cout, 64 * p.tn);
    if (ks == 3) { p.tiles_h = cdiv(h, 4); p.tiles_w = cdiv(w_, 16); p.total_tiles = n_img * p.tiles_h * p.tiles_w; }
    else { p.tiles_h = p.tiles_w = 0; p.total_tiles = cdiv((long)n_img * h * w_, 64); }
    const int chunks = p.ci_chunks * p.co_chunks;
    int splits = cdiv(640, chunks);                 // ~2.5 blocks per CU in total
    if (splits > p.total_tiles) splits = p.total_tiles;
    if (splits < 1) splits = 1;
    p.tps = cdiv(p.total_tiles, splits);
    p.splits = cdiv(p.total_tiles, p.tps);
    return p;
}

template <int KS, int TM, int TN>
static void launch_wgrad_tile(const WGradTileArgs& a, const TilePlan& p, hipStream_t st) {
    constexpr int HALO = (KS == 3) ? 1 : 0;
    const size_t lds = ((size_t)(16 + 2 * HALO) * (4 + 2 * HALO) * 64 * TM + 64 * 64 * TN) * sizeof(float);
    hipLaunchKernelGGL((wgrad_tile_kernel<KS, TM, TN>), dim3(p.ci_chunks * p.co_chunks, p.splits, a.dy_up ? 4 : 1), dim3(256), lds, st, a);
}


// ------------------------------------------------------------------------------------------ wgrad of the RGB stem (cin <= 4)
// dW[3][3][cin_w][cout] = sum_p x4[p + tap][0:cin_w] * dy[p][co].  A 27-deep "GEMM" wastes the matrix cores (the tile kernel pads
// cin to 64), and the op is bound by reading dy once, so this is a plain VALU kernel: a thread owns 4 output channels and every
// 16th pixel of a strip of image rows, keeps the 9 x 4 x 4 partial products in registers, the block reduces them through LDS and
// writes one partial slab; slab_reduce_kernel sums the slabs.
constexpr int STEM_ROWS = 4;     // image rows per block
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int ldy,
                                                         float* __restrict__ slabs, int H, int W, int cin_w, int cout) {
    __shared__ float red[256 * 16];                       // one tap at a time: [thread][ci 0..3][co 0..3]
    const int cq = cout / 4;                              // column threads
    const int rows = 256 / cq;                            // pixel lanes
    const int tid = threadIdx.x, col = tid % cq, pl = tid / cq;
    const int n = blockIdx.y, r0 = blockIdx.x * STEM_ROWS;
    const int npx = min(STEM_ROWS, H - r0) * W;
    float acc[9][4][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][c][q] = 0.f;
    if (pl < rows) {
        for (int i = pl; i < npx; i += rows) {
            const int h = r0 + i / W, w = i % W;
            const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (((long)n * H + h) * W + w) * ldy + col * 4);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    const int ih = h + r - 1, iw = w + s2 - 1;
                    f32x4 xv = {0.f, 0.f, 0.f, 0.f};
                    if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                        xv = *reinterpret_cast<const f32x4*>(x + (((long)n * H + ih) * W + iw) * ldx);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[r * 3 + s2][c][q] += xv[c] * g[q];
                }
        }
    }
    float* slab = slabs + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 9 * cin_w * cout;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[tid * 16 + c * 4 + q] = (pl < rows) ? acc[t][c][q] : 0.f;
        __syncthreads();
        // 16*cq outputs of this tap: (ci, co) = (c, col*4+q); thread u sums over the pixel lanes
        for (int u = tid; u < 16 * cq; u += 256) {
            const int cc = u / 16, e = u % 16, c = e >> 2, q = e & 3;
            float sacc = 0.f;
            for (int p = 0; p < rows; ++p) sacc += red[(p * cq + cc) * 16 + e];
            if (c < cin_w) slab[((long)t * cin_w + c) * cout + cc * 4 + q] = sacc;
        }
    }
}

// out[i] = sum_k slabs[k][i]: block = 32 float4 columns x 8 split-lanes (fixed summation order: reproducible)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n, int nsplit) {
    __shared__ f32x4 red[256];
    const int col = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const long i = ((long)blockIdx.x * 32 + col) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i + 4 <= n) {
        for (int k = kl; k < nsplit; k += 8) s += *reinterpret_cast<const f32x4*>(slabs + (long)k * n + i);
    } else if (i < n) {
        for (int k = kl; k < nsplit; k += 8)
            for (int e = 0; e < 4 && i + e < n; ++e) s[e] += slabs[(long)k * n + i + e];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (kl == 0 && i < n) {
#pragma unroll
        for (int j = 1; j < 8; ++j) s += red[j * 32 + col];
        if (i + 4 <= n) *reinterpret_cast<f32x4*>(out + i) = s;
        else for (int e = 0; e < 4 && i + e < n; ++e) out[i + e] = s[e];
    }
}

template <int BM, int BN, int WM, int WN, bool KC>
void launch_igemm(const IGemmArgs& a, int gz, hipStream_t st) {
    const long P = (long)a.Nimg * a.H * a.W;
    dim3 grid(cdiv(P, BM), cdiv(a.Ncols, BN), gz);
    const size_t lds = 2 * (BM * 20 + 16 * (BN + 4)) * sizeof(float);
    static const bool no_simple = getenv("RUNET_IGEMM_GENERAL") && atoi(getenv("RUNET_IGEMM_GENERAL")) != 0;      // measurement knob
    // every tap of every row inside the source image, no coordinate division: the plain 1x1 convolution (also the four 1x1 GEMMs of the
    // k2-s2 transposed forward, z_taps > 0) and the k2-s2 transposed convolution's data gradient (source pixel (2h + r, 2w + s))
    const bool plain1 = a.KH == 1 && a.KW == 1 && a.a_scale == 1 && a.Hin == a.H && a.Win == a.W;
    const bool up2 = a.KH == 2 && a.KW == 2 && a.z_taps == 0 && a.a_scale == 2 && a.tdh == 1 && a.tdw == 1 && a.Hin == 2 * a.H && a.Win == 2 * a.W;
    const bool simple = !no_simple && (plain1 || up2) && a.a_div <= 1 && !a.zmode4 && a.bh == 0 && a.bw == 0 && a.K % 16 == 0 && a.Kx >= a.K &&
                        a.Kvalid >= a.K && a.Ncols >= 4 && P > 0;
    if (simple) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, KC, true>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, KC, false>), grid, dim3(256), lds, st, a);
}

enum IGemmVariant { V_128x32 = 0, V_256x64 = 1, V_128x64 = 2, V_128x128 = 3, V_64x64 = 4 };

int g_forced_variant = -1;      // tools/igemm_variants.py: time every tile variant on one shape (runet_igemm_force_variant)

// k1: channels read by a plain 1x1 convolution (0 for every other geometry); kc: k-contiguous weights (the data gradients);
// up2: the k2-s2 transposed convolution's data gradient
IGemmVariant pick_variant(long P, int ncols, int k1, bool kc, bool up2 = false) {
    if (g_forced_variant >= 0) return (IGemmVariant)g_forced_variant;
    if (ncols <= 32) return V_128x32;
    // 1x1 convolutions are short (K / 16 steps per tile) and near the HBM / MFMA balance point: many small tiles hide the prologue and
    // the `+=` epilogue behind other tiles better than few large ones.  Measured on the 32 1x1 launches of the 16 x 256^2 step
    // (tools/igemm_variants.py, profiles/round2_igemm_variants.txt): 64x64 wins for every data gradient and for every forward with
    // K <= 128 (10-30 %); from K = 256 on the forward prefers the large tiles below (weights re-read per tile start to count).
    static const bool old_rules = getenv("RUNET_IGEMM_OLD_TILES") && atoi(getenv("RUNET_IGEMM_OLD_TILES")) != 0;      // A/B knob
    if (!old_rules && k1 > 0 && (kc || k1 <= 128)) return V_64x64;
    if (!old_rules && up2 && kc && ncols <= 256) return V_64x64;      // same table: 199 -> 175 us (128 -> 256 @ 64^2), 198 -> 184 us (64 -> 128 @ 128^2)
    if (ncols <= 64) return (P >= 256L * 512) ? V_256x64 : V_128x64;
    const long blocks128 = (long)cdiv(P, 128) * cdiv(ncols, 128);
    if (blocks128 >= 512 || (ncols % 128 == 0 && blocks128 >= 256)) return V_128x128;
    if ((long)cdiv(P, 128) * cdiv(ncols, 64) < 256) return V_64x64;      // few pixels (deep levels): smaller tiles keep every CU busy
    return V_128x64;
}

template <bool KC>
void dispatch_igemm(const IGemmArgs& a, int gz, hipStream_t st) {
    const bool plain1 = a.KH == 1 && a.KW == 1 && a.a_scale == 1 && a.Hin == a.H && a.Win == a.W && a.z_taps == 0;
    const bool up2 = a.KH == 2 && a.KW == 2 && a.z_taps == 0 && a.a_scale == 2;
    switch (pick_variant((long)a.Nimg * a.H * a.W, a.Ncols, plain1 ? a.K : 0, KC, up2)) {
    case V_128x32: launch_igemm<128, 32, 32, 32, KC>(a, gz, st); break;
    case V_256x64: launch_igemm<256, 64, 64, 64, KC>(a, gz, st); break;
    case V_128x64: launch_igemm<128, 64, 64, 32, KC>(a, gz, st); break;
    case V_128x128: launch_igemm<128, 128, 64, 64, KC>(a, gz, st); break;
    case V_64x64: launch_igemm<64, 64, 32, 32, KC>(a, gz, st); break;
    }
}

}  // namespace

namespace {
// wt[tap][co][ci] = w[tap][ci][co]: 32x32 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void transpose_taps_kernel(const float* __restrict__ w, float* __restrict__ wt, int cin, int cout) {
    __shared__ float tile[32][33];
    const long base = (long)blockIdx.z * cin * cout;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (ci0 + r < cin && co0 + tx < cout) ? w[base + (long)(ci0 + r) * cout + co0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (co0 + r < cout && ci0 + tx < cin) wt[base + (long)(co0 + r) * cin + ci0 + tx] = tile[tx][r];
}
}  // namespace

extern "C" int runet_transpose_taps(const float* w, float* wt, int taps, int cin, int cout, void* stream) {
    RUNET_REQUIRE(w && wt && taps > 0 && cin > 0 && cout > 0, "bad arguments");
    hipLaunchKernelGGL(transpose_taps_kernel, dim3(cdiv(cout, 32), cdiv(cin, 32), taps), dim3(256), 0, (hipStream_t)stream, w, wt, cin, cout);
    RUNET_CHECK_LAUNCH();
}

int g_lowp_forced_variant = -1;      // shared by conv_bf16.hip / conv_fp16.hip (conv_lowp.inc)

extern "C" int runet_igemm_lowp_force_variant(int variant) {
    RUNET_REQUIRE(variant >= -1 && variant <= 2, "variant: -1 (automatic) or 0..2");
    g_lowp_forced_variant = variant;
    return 0;
}

extern "C" int runet_igemm_force_variant(int variant) {
    RUNET_REQUIRE(variant >= -1 && variant <= 4, "variant: -1 (automatic) or 0..4");
    g_forced_variant = variant;
    return 0;
}

extern "C" const char* runet_conv_igemm_kernel_name(int n_img, int h, int w_, int cin, int cout, int kh, int mode) {
    // [k-contiguous weights][SIMPLE loader][tile variant]: the names rocprofv3 prints for the instantiations runet_conv_igemm launches
    static const char* names[2][2][5] = {
        {{"igemm_kernel<128, 32, 32, 32, false, false>", "igemm_kernel<256, 64, 64, 64, false, false>", "igemm_kernel<128, 64, 64, 32, false, false>",
          "igemm_kernel<128, 128, 64, 64, false, false>", "igemm_kernel<64, 64, 32, 32, false, false>"},
         {"igemm_kernel<128, 32, 32, 32, false, true>", "igemm_kernel<256, 64, 64, 64, false, true>", "igemm_kernel<128, 64, 64, 32, false, true>",
          "igemm_kernel<128, 128, 64, 64, false, true>", "igemm_kernel<64, 64, 32, 32, false, true>"}},
        {{"igemm_kernel<128, 32, 32, 32, true, false>", "igemm_kernel<256, 64, 64, 64, true, false>", "igemm_kernel<128, 64, 64, 32, true, false>",
          "igemm_kernel<128, 128, 64, 64, true, false>", "igemm_kernel<64, 64, 32, 32, true, false>"},
         {"igemm_kernel<128, 32, 32, 32, true, true>", "igemm_kernel<256, 64, 64, 64, true, true>", "igemm_kernel<128, 64, 64, 32, true, true>",
          "igemm_kernel<128, 128, 64, 64, true, true>", "igemm_kernel<64, 64, 32, 32, true, true>"}}};
    const bool kc = (mode == RUNET_CONV_DGRAD || mode == RUNET_CONVT_DGRAD);      // the _T modes read pre-transposed weights with the n-contiguous kernels
    static const bool no_simple = getenv("RUNET_IGEMM_GENERAL") && atoi(getenv("RUNET_IGEMM_GENERAL")) != 0;
    // SIMPLE loader (launch_igemm): 1x1 convolutions and the k2-s2 transposed convolution, K a multiple of 16
    const bool simple = !no_simple && (kh == 1 || kh == 2) && cin % 16 == 0 && cout >= 4;
    const bool plain1 = kh == 1 && (mode == RUNET_CONV_FWD || mode == RUNET_CONV_DGRAD || mode == RUNET_CONV_DGRAD_T);
    return names[kc ? 1 : 0][simple ? 1 : 0][pick_variant((long)n_img * h * w_, cout, plain1 ? cin : 0, kc, mode == RUNET_CONVT_DGRAD)];
}

extern "C" int runet_conv_igemm(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                                 int n_img, int h, int w_, int cin, int cin_w, int cout, int kh, int kw, int dil,
                                 int mode, int accumulate, void* stream) {
    RUNET_REQUIRE(x && w && y, "null pointer");
    RUNET_REQUIRE(cin > 0 && cin % 4 == 0, "cin (channels read from x) must be a positive multiple of 4");
    RUNET_REQUIRE(cin_w > 0 && cin_w <= cin, "cin_w (rows present in the weight) must be in (0, cin]");
    RUNET_REQUIRE(cout > 0 && cout % 4 == 0, "cout must be a positive multiple of 4");
    RUNET_REQUIRE(ldx >= cin && ldx % 4 == 0 && ldy % 4 == 0, "pixel strides must be multiples of 4 floats and cover the channels");
    RUNET_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0, "pointers must be 16-byte aligned");
    RUNET_REQUIRE(n_img > 0 && h > 0 && w_ > 0, "empty iteration space");
    RUNET_REQUIRE((kh == 1 && kw == 1) || (kh == 3 && kw == 3) || (kh == 2 && kw == 2), "kernel must be 1x1, 3x3 or 2x2(transposed)");
    hipStream_t st = (hipStream_t)stream;
    IGemmArgs a{};
    a.x = x; a.ldx = ldx; a.w = w; a.bias = bias; a.y = y; a.ldy = ldy;
    a.Nimg = n_img; a.accumulate = accumulate; a.KH = kh; a.KW = kw;
    int gz = 1;
    switch (mode) {
    case RUNET_CONV_FWD:      // y[N,h,w,cout] = conv(x[N,h,w,cin], w[kh,kw,cin_w,cout]), 'same' padding = dil*(k-1)/2
        RUNET_REQUIRE(kh != 2, "2x2 kernels are transposed-conv only");
        RUNET_REQUIRE(ldy >= cout, "ldy < cout");
        a.K = (cin + 15) / 16 * 16; a.Kx = cin; a.Kvalid = cin_w; a.Ncols = cout;
        a.w_tap_stride = (long)cin_w * cout; a.w_sk = cout; a.w_sn = 1;
        a.H = h; a.W = w_; a.Hin = h; a.Win = w_; a.a_scale = 1;
        a.tdh = dil; a.tdw = dil; a.bh = -dil * (kh / 2); a.bw = -dil * (kw / 2);
        a.Hout = h; a.Wout = w_; a.o_scale = 1;
        dispatch_igemm<false>(a, gz, st);
        break;
    case RUNET_CONV_DGRAD:    // dx[N,h,w,cout_w... ] : here `cin` = channels of dy (=conv Cout), `cout` = conv Cin
        // x := dy [N,h,w,cin], y := dx [N,h,w,cout];  w is the forward weight [kh,kw,cout(conv Cin),cin(conv Cout)]
        RUNET_REQUIRE(kh != 2, "use RUNET_CONVT_DGRAD for 2x2");
        RUNET_REQUIRE(cin_w == cin, "dgrad reads every output channel");
        a.K = (cin + 15) / 16 * 16; a.Kx = cin; a.Kvalid = cin; a.Ncols = cout;
        a.w_tap_stride = (long)cout * cin; a.w_sk = 1; a.w_sn = cin;
        a.H = h; a.W = w_; a.Hin = h; a.Win = w_; a.a_scale = 1;
        a.tdh = -dil; a.tdw = -dil; a.bh = dil * (kh / 2); a.bw = dil * (kw / 2);
        a.Hout = h; a.Wout = w_; a.o_scale = 1;
        dispatch_igemm<true>(a, gz, st);
        break;
    case RUNET_CONV_DGRAD_T:  // as RUNET_CONV_DGRAD with the weight already transposed per tap: w [kh,kw,cin(conv Cout),cout(conv Cin)]
        RUNET_REQUIRE(kh != 2 && cin_w == cin, "dgrad reads every output channel; use RUNET_CONVT_DGRAD_T for 2x2");
        a.K = (cin + 15) / 16 * 16; a.Kx = cin; a.Kvalid = cin; a.Ncols = cout;
        a.w_tap_stride = (long)cout * cin; a.w_sk = cout; a.w_sn = 1;
        a.H = h; a.W = w_; a.Hin = h; a.Win = w_; a.a_scale = 1;
        a.tdh = -dil; a.tdw = -dil; a.bh = dil * (kh / 2); a.bw = dil * (kw / 2);
        a.Hout = h; a.Wout = w_; a.o_scale = 1;
        dispatch_igemm<false>(a, gz, st);
        break;
    case RUNET_CONVT_DGRAD_T: // as RUNET_CONVT_DGRAD with w [2,2,cin(convT Cout),cout(convT Cin)]
        RUNET_REQUIRE(kh == 2 && kw == 2 && cin_w == cin, "transposed conv is 2x2 stride 2");
        a.K = (cin + 15) / 16 * 16; a.Kx = cin; a.Kvalid = cin; a.Ncols = cout;
        a.w_tap_stride = (long)cout * cin; a.w_sk = cout; a.w_sn = 1;
        a.H = h; a.W = w_; a.Hin = 2 * h; a.Win = 2 * w_; a.a_scale = 2; a.tdh = 1; a.tdw = 1;
        a.Hout = h; a.Wout = w_; a.o_scale = 1;
        dispatch_igemm<false>(a, gz, st);
        break;
    case RUNET_CONVT_FWD:     // y[N,2h,2w,cout] = convT_k2s2(x[N,h,w,cin]); w [2,2,cin,cout]
        RUNET_REQUIRE(kh == 2 && kw == 2 && cin_w == cin, "transposed conv is 2x2 stride 2");
        a.K = (cin + 15) / 16 * 16; a.Kx = cin; a.Kvalid = cin; a.Ncols = cout;
        a.w_tap_stride = (long)cin * cout; a.w_sk = cout; a.w_sn = 1;
        a.H = h; a.W = w_; a.Hin = h; a.Win = w_; a.a_scale = 1; a.KH = 1; a.KW = 1;
        a.Hout = 2 * h; a.Wout = 2 * w_; a.o_scale = 2; a.z_taps = 2; gz = 4;
        dispatch_igemm<false>(a, gz, st);
        break;
    case RUNET_CONVT_DGRAD:   // dx[N,h,w,cout(=convT Cin)] from dy[N,2h,2w,cin(=convT Cout)]; w [2,2,cout,cin]
        RUNET_REQUIRE(kh == 2 && kw == 2 && cin_w == cin, "transposed conv is 2x2 stride 2");
        a.K = (cin + 15) / 16 * 16; a.Kx = cin; a.Kvalid = cin; a.Ncols = cout;
        a.w_tap_stride = (long)cout * cin; a.w_sk = 1; a.w_sn = cin;
        a.H = h; a.W = w_; a.Hin = 2 * h; a.Win = 2 * w_; a.a_scale = 2; a.tdh = 1; a.tdw = 1;
        a.Hout = h; a.Wout = w_; a.o_scale = 1;
        dispatch_igemm<true>(a, gz, st);
        break;
    default:
        RUNET_REQUIRE(false, "unknown mode");
    }
    RUNET_CHECK_LAUNCH();
}

static int pick_splits(long P, int tiles, int ntaps) {
    const long base = (long)tiles * ntaps;
    long want = (768 + base - 1) / base;            // aim for >= 3 blocks per CU
    long maxs = (P + 255) / 256;                     // at least 256 pixels per split
    if (want > maxs) want = maxs;
    if (want > 64) want = 64;
    if (want < 1) want = 1;
    return (int)want;
}

static void wgrad_plan(long P, int cin_w, int cout, int ntaps, bool& big, int& tiles, int& splits) {
    big = (cin_w > 64 && cout > 64);
    const int BMt = big ? 128 : 64, BNt = big ? 128 : 64;
    tiles = cdiv(cin_w, BMt) * cdiv(cout, BNt);
    splits = pick_splits(P, tiles, ntaps);
}

static bool use_tile_kernel(int kh, int kw, int dil, int transposed) {
    if (transposed) return kh == 2 && kw == 2;
    return (kh == 1 && kw == 1) || (kh == 3 && kw == 3 && dil == 1);
}

// 1x1 weight gradient = one TN GEMM over the pixels (dW[cin][cout] = x^T dy): rows per split so that ~512 blocks exist, 16-aligned
static int wgrad1x1_rows_per_split(long P, int cin, int cout) {
    // gemm.hip picks one-, two- or four-wave blocks by the output's size: aim at ~2048 waves on the chip
    const int wm = cin > 64 ? 2 : 1, wn = cout > 64 ? 2 : 1;
    const long tiles = (long)cdiv(cin, 64 * wm) * cdiv(cout, 64 * wn);
    long splits = cdiv(2048, tiles * wm * wn);
    const long maxs = P / 256 > 0 ? P / 256 : 1;
    if (splits > maxs) splits = maxs;
    return (int)((cdiv(P, splits) + 15) / 16 * 16);
}
// The 1x1 weight gradient as one TN GEMM over the pixels (gemm.hip: 128 x 128 LDS-ring tiles, lean loader) where both channel counts
// fill such a tile; narrower layers stay on wgrad_tile_kernel<1,..>, whose per-tile integer address arithmetic costs it 30-77 TFLOP/s
// on the wide ones.  Measured (tools/conv_launches.py, 16 x 256^2 step): 256->128 @ 128^2 231 -> 159 us, 512->256 @ 64^2 223 -> 155,
// 1024->512 @ 32^2 220 -> 153, the ten wide launches 1.13 -> 0.83 ms; 128->64 / 64->128 @ 128^2 would go 89 -> 131 us (half-empty tiles).
// The step time does not move (weight gradients run on the side stream); the GPU does 0.3 ms less work.  RUNET_WGRAD1X1_TILE=1: tile kernel
// everywhere.  (With RUNET_GEMM_TN_DIRECT=1 the GEMM is the register-direct one, which also serves narrow outputs.)
static bool wgrad1x1_direct(long P, int cin, int cout) {
    static const bool off = getenv("RUNET_WGRAD1X1_TILE") && atoi(getenv("RUNET_WGRAD1X1_TILE")) != 0;
    static const bool direct = getenv("RUNET_GEMM_TN_DIRECT") && atoi(getenv("RUNET_GEMM_TN_DIRECT")) != 0;
    if (off || P * 4 >= (1L << 31)) return false;
    if (direct) return !((long)cin * cout <= 8192 && P >= (1L << 19));
    // RUNET_WGRAD1X1_MIN_NARROW (measurement knob, default 128): channels the NARROWER side needs (the wider one always >= 128)
    static const int narrow = getenv("RUNET_WGRAD1X1_MIN_NARROW") ? atoi(getenv("RUNET_WGRAD1X1_MIN_NARROW")) : 128;
    const int lo = cin < cout ? cin : cout, hi = cin < cout ? cout : cin;
    return hi >= 128 && lo >= narrow;
}
// ... and of those the ones the split-operand TN GEMM (gemm_split.hip, BF16 matrix cores, fp32-accurate) takes: pixel count a multiple of 16
static bool wgrad1x1_x3(long P, int cin, int cout) {
    static const bool off = (getenv("RUNET_NO_X3") && atoi(getenv("RUNET_NO_X3")) != 0) || (getenv("RUNET_GEMM_TN_DIRECT") && atoi(getenv("RUNET_GEMM_TN_DIRECT")) != 0);
    return !off && P % 16 == 0;
}

// the transposed convolution's weight gradient on the split-operand TN GEMM: image width a multiple of 16 (a k-step's rows share an image row),
// both channel counts >= RUNET_CONVT_WGRAD_X3_MIN [64]
static bool convt_wgrad_x3(long P, int w_, int cin, int cout) {
    static const bool off = (getenv("RUNET_NO_X3") && atoi(getenv("RUNET_NO_X3")) != 0) || (getenv("RUNET_NO_CONVT_WGRAD_X3") && atoi(getenv("RUNET_NO_CONVT_WGRAD_X3")) != 0);
    static const int lo = getenv("RUNET_CONVT_WGRAD_X3_MIN") ? atoi(getenv("RUNET_CONVT_WGRAD_X3_MIN")) : 64;
    return !off && w_ % 16 == 0 && P % 16 == 0 && P < (1L << 31) && cin >= lo && cout >= lo;
}

extern "C" long runet_conv_wgrad_workspace_floats(int n_img, int h, int w_, int cin_w, int cout, int kh, int kw) {
    if (kh == 1 && kw == 1) {
        const long P = (long)n_img * h * w_;
        const long direct = (long)cdiv(P, wgrad1x1_rows_per_split(P, cin_w, cout)) * cin_w * cout;
        TilePlan p = wgrad_tile_plan(n_img, h, w_, cin_w, cout, 1);
        const long tile = p.splits > 1 ? (long)p.splits * cin_w * cout : 0;
        return direct > tile ? direct : tile;
    }
    if (kh == 3 && kw == 3 && cin_w <= 4) return (long)cdiv(h, 4) * n_img * 9 * cin_w * cout + 64L * 9 * cin_w * cout;   // stem kernel slabs
    if (kh == 2 && kw == 2) {                                // transposed: the split-operand TN plan (up to 512 / tiles splits) or the tile plan below
        const long P = (long)n_img * h * w_;
        const long tiles = (long)cdiv(cin_w, 128) * cdiv(cout, 128) * 4;
        long splits = cdiv(512, tiles);
        const long maxs = P / 256 > 0 ? P / 256 : 1;
        if (splits > maxs) splits = maxs;
        const long x3 = (long)cdiv(P, (cdiv(P, splits) + 15) / 16 * 16) * 4 * cin_w * cout;
        TilePlan p = wgrad_tile_plan(n_img, h, w_, cin_w, cout, 1);
        p.splits = cdiv(p.splits, 4); p.tps = cdiv(p.total_tiles, p.splits); p.splits = cdiv(p.total_tiles, p.tps);
        bool big; int tl, sp;
        wgrad_plan(P, cin_w, cout, 4, big, tl, sp);
        const long a = p.splits > 1 ? (long)p.splits * 4 * cin_w * cout : 0;
        const long b = sp > 1 ? (long)sp * 4 * cin_w * cout : 0;
        const long m = a > b ? a : b;
        return x3 > m ? x3 : m;
    }
    if (kh == kw && (kh == 1 || kh == 3 || kh == 2)) {       // dilation unknown here: take the larger of the two plans
        TilePlan p = wgrad_tile_plan(n_img, h, w_, cin_w, cout, kh == 2 ? 1 : kh);
        if (kh == 2) { p.splits = cdiv(p.splits, 4); p.tps = cdiv(p.total_tiles, p.splits); p.splits = cdiv(p.total_tiles, p.tps); }
        bool big; int tiles, splits;
        wgrad_plan((long)n_img * h * w_, cin_w, cout, kh * kw, big, tiles, splits);
        const long a = p.splits > 1 ? (long)p.splits * kh * kw * cin_w * cout : 0;
        const long b = splits > 1 ? (long)splits * kh * kw * cin_w * cout : 0;
        return a > b ? a : b;
    }
    bool big; int tiles, splits;
    wgrad_plan((long)n_img * h * w_, cin_w, cout, kh * kw, big, tiles, splits);
    return splits > 1 ? (long)splits * kh * kw * cin_w * cout : 0;
}

extern "C" int runet_conv_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace,
                                 long workspace_floats, int n_img, int h, int w_, int cin, int cin_w, int cout,
                                 int kh, int kw, int dil, int transposed, void* stream) {
    RUNET_REQUIRE(x && dy && dw, "null pointer");
    RUNET_REQUIRE(cin > 0 && cin % 4 == 0 && cin_w > 0 && cin_w <= cin, "cin must be a multiple of 4, cin_w in (0, cin]");
    RUNET_REQUIRE(cout > 0 && cout % 4 == 0, "cout must be a positive multiple of 4");
    RUNET_REQUIRE(ldx >= cin && ldx % 4 == 0 && ldy >= cout && ldy % 4 == 0, "bad pixel strides");
    RUNET_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dw % 16) == 0, "pointers must be 16-byte aligned");
    RUNET_REQUIRE((kh == 1 && kw == 1) || (kh == 3 && kw == 3) || (kh == 2 && kw == 2 && transposed), "unsupported kernel size");
    hipStream_t st = (hipStream_t)stream;
    if (!transposed && kh == 3 && kw == 3 && dil == 1 && cin == 4 && cout <= 1024 && 256 % (cout / 4) == 0) {      // RGB stem
        const int nblk = cdiv(h, STEM_ROWS) * n_img;
        const long wsize = 9L * cin_w * cout;
        RUNET_REQUIRE(workspace && workspace_floats >= nblk * wsize, "workspace too small for the stem weight gradient");
        hipLaunchKernelGGL(stem_wgrad_kernel, dim3(cdiv(h, STEM_ROWS), n_img), dim3(256), 0, st, x, ldx, dy, ldy, workspace, h, w_, cin_w, cout);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(wsize, 32 * 4)), dim3(256), 0, st, workspace, dw, wsize, nblk);
        RUNET_CHECK_LAUNCH();
    }
    if (!transposed && kh == 1 && kw == 1 && cin_w == cin && wgrad1x1_direct((long)n_img * h * w_, cin, cout)) {
        // register-direct TN GEMM (gemm.hip): both operands are read from HBM in MFMA operand order, no LDS, no VALU
        const long P = (long)n_img * h * w_;
        const int rps = wgrad1x1_rows_per_split(P, cin, cout);
        const int splits = cdiv(P, rps);
        const long wsize = (long)cin * cout;
        RUNET_REQUIRE(splits == 1 || (workspace && workspace_floats >= splits * wsize), "workspace too small (runet_conv_wgrad_workspace_floats)");
        const int rc = wgrad1x1_x3(P, cin, cout)
                           ? runet_gemm_x3_tn_batched(x, ldx, 0, dy, ldy, 0, splits > 1 ? workspace : dw, 1, (int)P, cin, cout, rps, stream)
                           : runet_gemm_tn_launch(x, ldx, 0, dy, ldy, 0, splits > 1 ? workspace : dw, 1, (int)P, cin, cout, rps, st);
        if (rc) return rc;
        if (splits > 1) hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(wsize, 32 * 4)), dim3(256), 0, st, workspace, dw, wsize, splits);
        RUNET_CHECK_LAUNCH();
    }
    if (transposed && cin_w == cin && convt_wgrad_x3((long)n_img * h * w_, w_, cin, cout)) {
        // ConvTranspose2d weight gradient on the split-operand TN GEMM (gemm_split.hip): 4 taps = 4 GEMMs over the low-resolution pixels
        const long P = (long)n_img * h * w_;
        const long tiles = (long)cdiv(cin, 128) * cdiv(cout, 128) * 4;
        long splits = cdiv(512, tiles);
        const long maxs = P / 256 > 0 ? P / 256 : 1;
        if (splits > maxs) splits = maxs;
        const int rps = (int)((cdiv(P, splits) + 15) / 16 * 16);
        const int ns = cdiv(P, rps);
        const long wsize = 4L * cin * cout;
        if (ns == 1 || (workspace && workspace_floats >= ns * wsize)) {
            const int rc = runet_gemm_x3_tn_convt(x, ldx, dy, ldy, ns > 1 ? workspace : dw, n_img, h, w_, cin, cout, rps, stream);
            if (rc) return rc;
            if (ns > 1) hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(wsize, 32 * 4)), dim3(256), 0, st, workspace, dw, wsize, ns);
            RUNET_CHECK_LAUNCH();
        }
    }
    if (use_tile_kernel(kh, kw, dil, transposed)) {
        TilePlan p = wgrad_tile_plan(n_img, h, w_, cin_w, cout, transposed ? 1 : kh);
        if (transposed) { p.splits = cdiv(p.splits, 4); p.tps = cdiv(p.total_tiles, p.splits); p.splits = cdiv(p.total_tiles, p.tps); }   // 4 taps ride on grid.z
        const long wsize = (long)kh * kw * cin_w * cout;
        if (p.splits > 1 && (workspace == nullptr || workspace_floats < p.splits * wsize)) {
            int s2 = workspace ? (int)(workspace_floats / wsize) : 1;
            if (s2 < 1) s2 = 1;
            p.tps = cdiv(p.total_tiles, s2);
            p.splits = cdiv(p.total_tiles, p.tps);
        }
        WGradTileArgs t{};
        t.x = x; t.ldx = ldx; t.dy = dy; t.ldy = ldy; t.out = (p.splits > 1) ? workspace : dw;
        t.Kci = cin; t.Kvalid = cin_w; t.Nco = cout; t.Nimg = n_img; t.H = h; t.W = w_;
        t.tiles_h = p.tiles_h; t.tiles_w = p.tiles_w; t.total_tiles = p.total_tiles; t.tiles_per_split = p.tps; t.co_chunks = p.co_chunks;
        t.dy_up = transposed ? 1 : 0;
        if (kh == 3) launch_wgrad_tile<3, 1, 1>(t, p, st);
        else if (p.tm == 2 && p.tn == 2) launch_wgrad_tile<1, 2, 2>(t, p, st);
        else if (p.tm == 2) launch_wgrad_tile<1, 2, 1>(t, p, st);
        else if (p.tn == 2) launch_wgrad_tile<1, 1, 2>(t, p, st);
        else launch_wgrad_tile<1, 1, 1>(t, p, st);
        if (p.splits > 1)
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(wsize, 32 * 4)), dim3(256), 0, st, workspace, dw, wsize, p.splits);
        RUNET_CHECK_LAUNCH();
    }
    WGradArgs a{};
    a.x = x; a.ldx = ldx; a.dy = dy; a.ldy = ldy;
    a.Kci = cin; a.Kvalid = cin_w; a.Nco = cout;
    a.Nimg = n_img; a.H = h; a.W = w_; a.Hin = h; a.Win = w_; a.a_scale = 1; a.KH = kh; a.KW = kw;
    if (transposed) {
        a.tap_on_output = 1; a.Hout = 2 * h; a.Wout = 2 * w_; a.o_scale = 2;
    } else {
        a.tdh = dil; a.tdw = dil; a.bh = -dil * (kh / 2); a.bw = -dil * (kw / 2);
        a.Hout = h; a.Wout = w_; a.o_scale = 1;
    }
    const long P = (long)n_img * h * w_;
    const int ntaps = kh * kw;
    bool big; int tiles, splits;
    wgrad_plan(P, cin_w, cout, ntaps, big, tiles, splits);
    const long wsize = (long)ntaps * cin_w * cout;
    if (splits > 1 && (workspace == nullptr || workspace_floats < splits * wsize)) {
        splits = workspace ? (int)(workspace_floats / wsize) : 1;
        if (splits < 1) splits = 1;
    }
    long pps = (P + splits - 1) / splits;
    pps = (pps + 15) / 16 * 16;
    splits = cdiv(P, pps);
    a.pix_per_split = pps;
    a.out = (splits > 1) ? workspace : dw;
    dim3 grid(tiles, ntaps, splits);
    if (big) {
        const size_t lds = 2 * (16 * (128 + 4) * 2) * sizeof(float);
        hipLaunchKernelGGL((wgrad_kernel<128, 128, 64, 64>), grid, dim3(256), lds, st, a);
    } else {
        const size_t lds = 2 * (16 * (64 + 4) * 2) * sizeof(float);
        hipLaunchKernelGGL((wgrad_kernel<64, 64, 32, 32>), grid, dim3(256), lds, st, a);
    }
    if (splits > 1) {
        const int thr = 256;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(wsize, 32 * 4)), dim3(256), 0, st, workspace, dw, wsize, splits);
    }
    RUNET_CHECK_LAUNCH();
}


// ---------------------------------------------------------------------------------------------------------------------------
// General (strided / large-kernel / k4 transposed) convolutions for the DeepLabV3+ baseline (/root/reference/Main_Final.py:325-433:
// Conv2d 7x7 s2 p3, 3x3 s2 p1, 3x3 dilation 6/12/18, ConvTranspose2d k4 s2 p1).  Same kernels, other geometry.
extern "C" int runet_conv2d_general(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int n_img, int hin,
                                    int win, int cin, int cin_w, int cout, int kh, int kw, int stride, int pad, int dil, int mode,
                                    int accumulate, void* stream) {
    RUNET_REQUIRE(x && w && y, "null pointer");
    RUNET_REQUIRE(cin > 0 && cin % 4 == 0 && cin_w > 0 && cin_w <= cin && cout > 0 && cout % 4 == 0, "channel counts must be multiples of 4");
    RUNET_REQUIRE(kh >= 1 && kh <= 7 && kw >= 1 && kw <= 7 && stride >= 1 && stride <= 2 && pad >= 0 && dil >= 1, "unsupported geometry");
    RUNET_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0, "alignment");
    const int ho = (hin + 2 * pad - dil * (kh - 1) - 1) / stride + 1, wo = (win + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
    RUNET_REQUIRE(ho > 0 && wo > 0, "empty output");
    hipStream_t st = (hipStream_t)stream;
    IGemmArgs a{};
    a.x = x; a.ldx = ldx; a.w = w; a.bias = bias; a.y = y; a.ldy = ldy; a.Nimg = n_img; a.accumulate = accumulate; a.KH = kh; a.KW = kw;
    a.a_div = 1; a.o_scale = 1;
    if (mode == RUNET_CONV_FWD) {          // x [n,hin,win,cin] -> y [n,ho,wo,cout]
        RUNET_REQUIRE(ldx >= cin && ldy >= cout, "bad pixel strides");
        a.K = (cin + 15) / 16 * 16; a.Kx = cin; a.Kvalid = cin_w; a.Ncols = cout;
        a.w_tap_stride = (long)cin_w * cout; a.w_sk = cout; a.w_sn = 1;
        a.H = ho; a.W = wo; a.Hin = hin; a.Win = win; a.a_scale = stride; a.tdh = dil; a.tdw = dil; a.bh = -pad; a.bw = -pad;
        a.Hout = ho; a.Wout = wo;
        dispatch_igemm<false>(a, 1, st);
    } else if (mode == RUNET_CONV_DGRAD) { // x := dy [n,ho,wo,cin(=conv Cout)] -> y := dx [n,hin,win,cout(=conv Cin)]; w [kh,kw,cout,cin]
        RUNET_REQUIRE(cin_w == cin && ldx >= cin && ldy >= cout, "bad arguments for the data gradient");
        a.K = (cin + 15) / 16 * 16; a.Kx = cin; a.Kvalid = cin; a.Ncols = cout;
        a.w_tap_stride = (long)cout * cin; a.w_sk = 1; a.w_sn = cin;
        a.H = hin; a.W = win; a.Hin = ho; a.Win = wo; a.a_scale = 1; a.tdh = -dil; a.tdw = -dil; a.bh = pad; a.bw = pad; a.a_div = stride;
        a.Hout = hin; a.Wout = win;
        dispatch_igemm<true>(a, 1, st);
    } else {
        RUNET_REQUIRE(false, "mode must be RUNET_CONV_FWD or RUNET_CONV_DGRAD");
    }
    RUNET_CHECK_LAUNCH();
}

// ConvTranspose2d(k4, s2, p1): w [4][4][cin][cout].  FWD: x [n,h,w,cin] -> y [n,2h,2w,cout];  DGRAD: x := dy [n,2h,2w,cin(=Cout)] -> y := dx [n,h,w,cout(=Cin)]
extern "C" int runet_convt4_igemm(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int n_img, int h, int w_,
                                  int cin, int cout, int mode, int accumulate, void* stream) {
    RUNET_REQUIRE(x && w && y && cin % 4 == 0 && cout % 4 == 0 && cin > 0 && cout > 0, "bad arguments");
    RUNET_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0, "alignment");
    hipStream_t st = (hipStream_t)stream;
    IGemmArgs a{};
    a.x = x; a.ldx = ldx; a.w = w; a.bias = bias; a.y = y; a.ldy = ldy; a.Nimg = n_img; a.accumulate = accumulate; a.a_div = 1;
    a.K = (cin + 15) / 16 * 16; a.Kx = cin; a.Kvalid = cin; a.Ncols = cout; a.H = h; a.W = w_;
    if (mode == RUNET_CONVT_FWD) {
        a.w_tap_stride = (long)cin * cout; a.w_sk = cout; a.w_sn = 1;
        a.Hin = h; a.Win = w_; a.a_scale = 1; a.KH = 2; a.KW = 2; a.zmode4 = 1;
        a.Hout = 2 * h; a.Wout = 2 * w_; a.o_scale = 2;
        dispatch_igemm<false>(a, 4, st);
    } else if (mode == RUNET_CONVT_DGRAD) {
        a.w_tap_stride = (long)cout * cin; a.w_sk = 1; a.w_sn = cin;
        a.Hin = 2 * h; a.Win = 2 * w_; a.a_scale = 2; a.KH = 4; a.KW = 4; a.tdh = 1; a.tdw = 1; a.bh = -1; a.bw = -1;
        a.Hout = h; a.Wout = w_; a.o_scale = 1;
        dispatch_igemm<true>(a, 1, st);
    } else {
        RUNET_REQUIRE(false, "mode must be RUNET_CONVT_FWD or RUNET_CONVT_DGRAD");
    }
    RUNET_CHECK_LAUNCH();
}

// weight gradient for the general geometries (per-tap split-K kernel).  transposed4 != 0: ConvTranspose2d(k4,s2,p1), x [n,h,w,cin], dy [n,2h,2w,cout].
extern "C" long runet_conv_wgrad_general_workspace_floats(int pixels, int cin_w, int cout, int kh, int kw) {
    bool big; int tiles, splits;
    wgrad_plan(pixels, cin_w, cout, kh * kw, big, tiles, splits);
    return splits > 1 ? (long)splits * kh * kw * cin_w * cout : 0;
}

extern "C" int runet_conv_wgrad_general(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats,
                                        int n_img, int hin, int win, int cin, int cin_w, int cout, int kh, int kw, int stride, int pad, int dil,
                                        int transposed4, void* stream) {
    RUNET_REQUIRE(x && dy && dw, "null pointer");
    RUNET_REQUIRE(cin % 4 == 0 && cin_w > 0 && cin_w <= cin && cout % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "channel counts / strides must be multiples of 4");
    hipStream_t st = (hipStream_t)stream;
    WGradArgs a{};
    a.x = x; a.ldx = ldx; a.dy = dy; a.ldy = ldy; a.Kci = cin; a.Kvalid = cin_w; a.Nco = cout; a.Nimg = n_img; a.KH = kh; a.KW = kw;
    long P;
    if (transposed4) {
        RUNET_REQUIRE(kh == 4 && kw == 4, "transposed4 is the k4-s2-p1 transposed convolution");
        a.H = hin; a.W = win; a.Hin = hin; a.Win = win; a.a_scale = 1; a.tap_on_output = 1; a.o_bh = -1; a.o_bw = -1;
        a.Hout = 2 * hin; a.Wout = 2 * win; a.o_scale = 2;
        P = (long)n_img * hin * win;
    } else {
        const int ho = (hin + 2 * pad - dil * (kh - 1) - 1) / stride + 1, wo = (win + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
        a.H = ho; a.W = wo; a.Hin = hin; a.Win = win; a.a_scale = stride; a.tdh = dil; a.tdw = dil; a.bh = -pad; a.bw = -pad;
        a.Hout = ho; a.Wout = wo; a.o_scale = 1;
        P = (long)n_img * ho * wo;
    }
    const int ntaps = kh * kw;
    bool big; int tiles, splits;
    wgrad_plan(P, cin_w, cout, ntaps, big, tiles, splits);
    const long wsize = (long)ntaps * cin_w * cout;
    if (splits > 1 && (workspace == nullptr || workspace_floats < splits * wsize)) {
        splits = workspace ? (int)(workspace_floats / wsize) : 1;
        if (splits < 1) splits = 1;
    }
    long pps = (P + splits - 1) / splits;
    pps = (pps + 15) / 16 * 16;
    splits = cdiv(P, pps);
    a.pix_per_split = pps;
    a.out = (splits > 1) ? workspace : dw;
    dim3 grid(tiles, ntaps, splits);
    if (big) hipLaunchKernelGGL((wgrad_kernel<128, 128, 64, 64>), grid, dim3(256), 2 * (16 * (128 + 4) * 2) * sizeof(float), st, a);
    else hipLaunchKernelGGL((wgrad_kernel<64, 64, 32, 32>), grid, dim3(256), 2 * (16 * (64 + 4) * 2) * sizeof(float), st, a);
    if (splits > 1) hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(wsize, 32 * 4)), dim3(256), 0, st, workspace, dw, wsize, splits);
    RUNET_CHECK_LAUNCH();
}


// ---------------------------------------------------------------------------------------------------------------------------
// Batched plain GEMMs on the same kernels (the 36 position-GEMMs of the unfused Winograd F(4x4,3x3) path, conv_winograd4.hip).
//   runet_gemm_batched   : C[z][rows][n] = A[z][rows][k] . B[z][k][n]                (blockIdx.z = z)
//   runet_gemm_tn_batched: C[split][z][k][n] = sum over the split's rows of A[z][row][k] * B[z][row][n];  splits = ceil(rows / rows_per_split)
// which device kernel runet_gemm_batched launches for a shape (live profiling labels must match the rocprofv3 kernel names)
extern "C" const char* runet_gemm_batched_kernel_name(int batch, int rows, int k, int n) {
    static const bool old_path = getenv("RUNET_GEMM_OLD") != nullptr;
    const long b128 = (long)cdiv(rows, 128) * cdiv(n, 128) * batch;
    const double per_cu = b128 / 256.0, bal = per_cu / (double)((b128 + 255) / 256);
    const bool fast = k % 16 == 0 && k >= 16 && n >= 4;       // the FAST / SIMPLE instantiations (last template argument)
    if (!old_path && bal < 0.8 && n % 64 == 0) return fast ? "igemm_kernel<128, 64, 64, 32, false, true>" : "igemm_kernel<128, 64, 64, 32, false, false>";
    if (!old_path && n >= 96) return fast ? "gemm_nn_kernel<16, true>" : "gemm_nn_kernel<16, false>";
    if (n % 128 == 0 && b128 >= 256) return fast ? "igemm_kernel<128, 128, 64, 64, false, true>" : "igemm_kernel<128, 128, 64, 64, false, false>";
    if (n > 32 && (long)cdiv(rows, 128) * cdiv(n, 64) * batch >= 256) return fast ? "igemm_kernel<128, 64, 64, 32, false, true>" : "igemm_kernel<128, 64, 64, 32, false, false>";
    if (n > 32) return fast ? "igemm_kernel<64, 64, 32, 32, false, true>" : "igemm_kernel<64, 64, 32, 32, false, false>";
    return fast ? "igemm_kernel<128, 32, 32, 32, false, true>" : "igemm_kernel<128, 32, 32, 32, false, false>";
}

extern "C" int runet_gemm_batched(const float* a, int lda, long stride_a, const float* b, long stride_b, float* c, int ldc, long stride_c,
                                  int batch, int rows, int k, int n, void* stream) {
    RUNET_REQUIRE(a && b && c && batch > 0 && batch <= 65535 && rows > 0, "bad arguments");
    RUNET_REQUIRE(k > 0 && k % 4 == 0 && n > 0 && n % 4 == 0 && lda >= k && lda % 4 == 0 && ldc >= n && ldc % 4 == 0, "k, n and the row strides must be multiples of 4");
    RUNET_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)c % 16) == 0 && stride_a % 4 == 0 && stride_b % 4 == 0 && stride_c % 4 == 0,
                  "alignment");
    IGemmArgs g{};
    g.x = a; g.ldx = lda; g.w = b; g.y = c; g.ldy = ldc; g.K = (k + 15) / 16 * 16; g.Kx = k; g.Kvalid = k; g.Ncols = n;
    g.w_sk = n; g.w_sn = 1; g.Nimg = 1; g.H = 1; g.W = rows; g.Hin = 1; g.Win = rows; g.a_scale = 1; g.KH = 1; g.KW = 1;
    g.Hout = 1; g.Wout = rows; g.o_scale = 1; g.zbatch = 1; g.zs_x = stride_a; g.zs_w = stride_b; g.zs_y = stride_c;
    hipStream_t st = (hipStream_t)stream;
    static const bool old_path = getenv("RUNET_GEMM_OLD") != nullptr;      // A/B switch for tools/bench_gemm.py
    {   // 128x64 tiles where 128x128 tiles would fill the 256 CUs unevenly (e.g. 576 blocks = 2.25 per CU -> 3 rounds for 2.25 of work)
        const long b128 = (long)cdiv(rows, 128) * cdiv(n, 128) * batch;
        const double per_cu = b128 / 256.0, bal = per_cu / (double)((b128 + 255) / 256);
        if (!old_path && bal < 0.8 && n % 64 == 0) {
            launch_igemm<128, 64, 64, 32, false>(g, batch, st);
            RUNET_CHECK_LAUNCH();
        }
    }
    if (!old_path && n >= 96) {
        runet_gemm_nn_launch(a, lda, stride_a, b, stride_b, c, ldc, stride_c, batch, rows, k, n, st);
        RUNET_CHECK_LAUNCH();
    }
    const long b128 = (long)cdiv(rows, 128) * cdiv(n, 128) * batch;
    if (n % 128 == 0 && b128 >= 256) launch_igemm<128, 128, 64, 64, false>(g, batch, st);
    else if (n > 32 && (long)cdiv(rows, 128) * cdiv(n, 64) * batch >= 256) launch_igemm<128, 64, 64, 32, false>(g, batch, st);
    else if (n > 32) launch_igemm<64, 64, 32, 32, false>(g, batch, st);
    else launch_igemm<128, 32, 32, 32, false>(g, batch, st);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_gemm_tn_batched(const float* a, int lda, long stride_a, const float* b, int ldb, long stride_b, float* c, int batch, int rows,
                                     int k, int n, int rows_per_split, void* stream) {
    RUNET_REQUIRE(a && b && c && batch > 0 && batch <= 65535 && rows > 0 && rows_per_split > 0 && rows_per_split % 16 == 0, "bad arguments (rows_per_split: multiple of 16)");
    const int splits = cdiv(rows, rows_per_split);
    RUNET_REQUIRE(splits <= 65535, "too many splits");
    RUNET_REQUIRE(k > 0 && k % 4 == 0 && n > 0 && n % 4 == 0 && lda >= k && lda % 4 == 0 && ldb >= n && ldb % 4 == 0, "k, n and the row strides must be multiples of 4");
    RUNET_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && ((uintptr_t)c % 16) == 0 && stride_a % 4 == 0 && stride_b % 4 == 0, "alignment");
    WGradArgs g{};
    g.x = a; g.ldx = lda; g.dy = b; g.ldy = ldb; g.out = c; g.Kci = k; g.Kvalid = k; g.Nco = n;
    g.Nimg = 1; g.H = 1; g.W = rows; g.Hin = 1; g.Win = rows; g.a_scale = 1; g.KH = batch; g.KW = 1; g.Hout = 1; g.Wout = rows; g.o_scale = 1;
    g.tap_bs_x = stride_a; g.tap_bs_dy = stride_b;
    g.pix_per_split = rows_per_split;
    hipStream_t st = (hipStream_t)stream;
    const bool big = k > 64 && n > 64;
    static const bool old_path = getenv("RUNET_GEMM_OLD") != nullptr;
    if (!old_path && big) {
        runet_gemm_tn_launch(a, lda, stride_a, b, ldb, stride_b, c, batch, rows, k, n, rows_per_split, st);
        RUNET_CHECK_LAUNCH();
    }
    const int tiles = big ? cdiv(k, 128) * cdiv(n, 128) : cdiv(k, 64) * cdiv(n, 64);
    dim3 grid(tiles, batch, splits);
    if (big) hipLaunchKernelGGL((wgrad_kernel<128, 128, 64, 64>), grid, dim3(256), 2 * (16 * (128 + 4) * 2) * sizeof(float), st, g);
    else hipLaunchKernelGGL((wgrad_kernel<64, 64, 32, 32>), grid, dim3(256), 2 * (16 * (64 + 4) * 2) * sizeof(float), st, g);
    RUNET_CHECK_LAUNCH();
}
