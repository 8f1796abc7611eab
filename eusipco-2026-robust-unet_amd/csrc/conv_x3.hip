// 1x1 convolutions and the k2-s2 transposed convolution (forward and data gradient) as fp32-accurate GEMMs on the gfx950 BF16 matrix cores:
// the split-operand ("bf16x6") scheme of gemm_split.hip behind the geometry of conv_igemm.hip's SIMPLE loader.
//
// Replaces igemm_kernel<.., SIMPLE> (v_mfma_f32_32x32x2_f32, the FP32 vector rate) for nn.Conv2d(k=1) - the ResidualBlock shortcuts, the
// attention gates' W_g / W_x, the DilatedBlock's conv1 (/root/reference/Main_Final.py:126,131,172,205) - and nn.ConvTranspose2d(k=2, s=2)
// (:261-270), forward and data gradient, wherever the contraction is a multiple of 16 channels.
//
//   out[dst(p)][n] (+)= bias[n] + sum_tap sum_k  src[pix(p) + tap][k] * B_tap[k][n]
//     1x1 (forward, data gradient):  one tap, pix(p) = dst(p) = p
//     convT forward:  blockIdx z = tap (a, b): one tap per GEMM, pix(p) = p over the H x W input, dst(p) = (2h + a, 2w + b) of the 2H x 2W output
//     convT data gradient:  four taps in the k-loop, pix(p) = (2h, 2w) of the 2H x 2W gradient + (a * 2W + b), dst(p) = p
// A (activations, fp32) is split x = h + m + l on its way into LDS exactly as gemm_nn_x3_kernel does; B comes PRE-SPLIT from
// runet_conv_x3_pack ([tap][plane 3][K/8][N][8] bf16, once per optimizer step).  Every A address is a per-thread row pointer set up once plus
// a wave-uniform (tap, channel-chunk) offset; the destination row offsets of a block sit in a 1-KB LDS table (one integer division per row
// per block instead of one per stored element).  Same block shape, LDS ring and XCD-aware block order as gemm_nn_x3_kernel; the blocks that
// share an A tile (all column tiles, and for the transposed forward all four taps) are adjacent in that order.
#include "x3_common.h"
#include "../../include/runet_hip.h"
#include "derive_weights.h"
#include <stdlib.h>

namespace {

using namespace x3;

struct X3ConvArgs {
    const float* a; int lda;             // source activations, pixel stride (floats)
    const __bf16* b; long sb;            // packed planes; sb = elements per tap (3 * k * n)
    const float* bias;                   // [n] or nullptr
    float* c; int ldc;
    int rows, k, n;                      // rows = pixels of the iteration space (Nimg * H * W), k = channels per tap, n = output channels
    int H, W;                            // iteration-space image
    int a_scale, c_scale;                // 1 or 2: source / destination pixel of row (img, h, w) = (img, h * s [+ zh], w * s [+ zw]) in an (H s) x (W s) image
    int ntaps;                           // taps in the k-loop (1, or 4 = the transposed data gradient: source offset (a * W a_scale + b) pixels)
    int accumulate;
    float* stats;                        // nullptr, or [gm][n][3]: (count, mean, M2) of every output channel over the block's rows (BatchNorm statistics)
    int gm, gn, gz;                      // row tiles, column tiles, z (4 = transposed forward: z is the tap AND the destination offset)
};

template <int BN>
__global__ __launch_bounds__(256, 2) void conv_nn_x3_kernel(X3ConvArgs g) {
    using C = NNX3<BN>;
    constexpr int TN = C::TN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long* dst_tab = reinterpret_cast<long*>(smem + 3 * C::STAGE);       // [128] destination element offset of the block's rows
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * (BN / 2);
    const int li = lane & 31, lh = lane >> 5;
    // column tile fastest, then z (tap of the transposed forward), then row tile: everything that reads one A tile is contiguous
    const long total = (long)g.gm * g.gn * g.gz;
    long id = xcd_remap(blockIdx.x, total);
    const int nt = (int)(id % g.gn); id /= g.gn;
    const int z = (int)(id % g.gz);
    const int mt = (int)(id / g.gz);
    const int m0 = mt * 128, n0 = nt * BN;
    const __bf16* B = g.b + (g.gz > 1 ? (long)z * g.sb : 0);
    const int K8 = g.k >> 3;
    const int HW = g.H * g.W;

    // pixel of row r in an image scaled by s (element offsets are formed by the callers)
    auto pixel = [&](int r, int s, int dh, int dw) -> long {
        if (s == 1) return r;
        const int img = r / HW, rem = r - img * HW;
        const int h = rem / g.W, w = rem - h * g.W;
        return ((long)(img * g.H + h) * s + dh) * (g.W * s) + w * s + dw;
    };
    if (tid < 128) {
        int r = m0 + tid;
        r = r < g.rows ? r : g.rows - 1;
        dst_tab[tid] = pixel(r, g.c_scale, g.gz > 1 ? (z >> 1) : 0, g.gz > 1 ? (z & 1) : 0) * g.ldc;
    }

    // ---- loaders.  A item (row, k-quad) = (tid >> 2 (+64), tid & 3): rows beyond `rows` clamped (their products are never stored).
    const int akq = tid & 3, arow = tid >> 2;
    const float* ap[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        int r = m0 + arow + 64 * v;
        r = r < g.rows ? r : g.rows - 1;
        ap[v] = g.a + pixel(r, g.a_scale, 0, 0) * g.lda + akq * 4;
    }
    const __bf16* bp[C::BITEMS];
#pragma unroll
    for (int j = 0; j < C::BITEMS; ++j) {
        int i = tid + 256 * j;
        i = i < 3 * 2 * BN ? i : 3 * 2 * BN - 1;
        const int pl = i / (2 * BN), oc = (i / BN) & 1;
        int col = n0 + i % BN;
        col = col < g.n ? col : g.n - 1;
        bp[j] = B + (((long)pl * K8 + oc) * g.n + col) * 8;
    }
    const long bstep = (long)2 * g.n * 8;              // packed elements per 16-deep k-step
    const int KC = g.k >> 4;
    const int nks = g.ntaps * KC;
    const long a_row = (long)g.W * g.a_scale * g.lda;  // tap a = 1: one source row down
    struct Raw { f32x4 ra[2]; f32x4 rb[C::BITEMS]; };
    // tiles are loaded strictly in order: running (tap, chunk) counters, all scalar; past the end the last tile is read again into a
    // stage nobody multiplies
    int ld_tap = 0, ld_kc = 0;
    auto load_tile = [&](Raw& R) {
        const long offa = (long)(ld_tap >> 1) * a_row + (long)(ld_tap & 1) * g.lda + ld_kc * 16;
        const long offb = (long)ld_tap * g.sb + (long)ld_kc * bstep;
#pragma unroll
        for (int v = 0; v < 2; ++v) R.ra[v] = *reinterpret_cast<const f32x4*>(ap[v] + offa);
#pragma unroll
        for (int j = 0; j < C::BITEMS; ++j) R.rb[j] = *reinterpret_cast<const f32x4*>(bp[j] + offb);
        if (ld_kc + 1 < KC) ++ld_kc;
        else if (ld_tap + 1 < g.ntaps) { ld_kc = 0; ++ld_tap; }
    };
    int a_st[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int row = arow + 64 * v;
        a_st[v] = (row * 2 + ((akq >> 1) ^ ((row >> 3) & 1))) * 16 + (akq & 1) * 8;
    }
    auto store_tile = [&](int buf, const Raw& R) {
        unsigned char* As = smem + buf * C::STAGE;
        unsigned char* Bs = As + C::A_BYTES;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            bf16x4 h, m, l;
            split3(R.ra[v], h, m, l);
            *reinterpret_cast<bf16x4*>(As + a_st[v]) = h;
            *reinterpret_cast<bf16x4*>(As + 4096 + a_st[v]) = m;
            *reinterpret_cast<bf16x4*>(As + 8192 + a_st[v]) = l;
        }
#pragma unroll
        for (int j = 0; j < C::BITEMS; ++j) {
            const int i = tid + 256 * j;
            if (3 * 2 * BN % 256 == 0 || i < 3 * 2 * BN) *reinterpret_cast<f32x4*>(Bs + i * 16) = R.rb[j];
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    int a_rd[2], b_rd[TN];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = wm0 + a * 32 + li;
        a_rd[a] = (row * 2 + (lh ^ ((row >> 3) & 1))) * 16;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) b_rd[b] = C::A_BYTES + (lh * BN + wn0 + b * 32 + li) * 16;

    // four register sets, loads four steps ahead of their LDS store (gemm_nn_x3_kernel)
    Raw R0, R1, R2, R3;
    load_tile(R0);
    load_tile(R1);
    store_tile(0, R0);
    store_tile(1, R1);
    load_tile(R2);
    load_tile(R3);
    load_tile(R0);
    lds_barrier();
    int cur = 0;
    auto step = [&](const Raw& cur_set, Raw& nxt_set) {
        const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
        load_tile(nxt_set);
        const unsigned char* St = smem + cur * C::STAGE;
        bf16x8 af[2][3], bf[TN][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a][p] = *reinterpret_cast<const bf16x8*>(St + p * 4096 + a_rd[a]);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b][p] = *reinterpret_cast<const bf16x8*>(St + p * (2 * BN * 16) + b_rd[b]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) X3_MMA(acc[a][b], af[a], bf[b]);
        store_tile(nx2, cur_set);
        lds_barrier();
        __builtin_amdgcn_sched_barrier(0);      // keep the next step's split / stores from being hoisted over this one (they would pull its vmcnt wait forward)
        cur = nxt;
    };
    // four unconditional steps per trip (see gemm_split.hip: a conditional step inside the loop makes the compiler drain vmcnt at the header)
    int s = 0;
    for (; s + 4 <= nks; s += 4) {
        step(R2, R1);
        step(R3, R2);
        step(R0, R3);
        step(R1, R0);
    }
    if (s < nks) {
        step(R2, R1);
        if (s + 1 < nks) {
            step(R3, R2);
            if (s + 2 < nks) step(R0, R3);
        }
    }

    // ---- epilogue: lane holds column li of each 32-wide tile, rows (r&3) + 8*(r>>2) + 4*lh
    float bv[TN];
    int colv[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        colv[b] = n0 + wn0 + b * 32 + li;
        bv[b] = (g.bias && colv[b] < g.n) ? g.bias[colv[b]] : 0.f;
    }
    // accumulate: ALL the old values are requested before the first store (a load behind a store to a pointer the compiler cannot tell
    // apart waits for it: read-modify-write row by row ran 2.7x slower than the f32-MFMA kernel it replaces)
    if (g.accumulate) {
        // unconditional loads from clamped (always valid) addresses, consumed only after the last one is issued: a predicated
        // `acc += *p` puts load, wait and add into one exec-masked block each - 64 serial round trips per lane
        float old[2][TN][16];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* crow = g.c + dst_tab[wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];      // rows past the end: the last row's offset
#pragma unroll
                for (int b = 0; b < TN; ++b) old[a][b][r] = crow[colv[b] < g.n ? colv[b] : g.n - 1];
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] += old[a][b][r];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] += bv[b];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m0 + rl < g.rows) {
                float* crow = g.c + dst_tab[rl];
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    if (colv[b] < g.n) crow[colv[b]] = acc[a][b][r];
            }
        }
    if (g.stats) {
        // BatchNorm statistics of what was just stored, per output channel over this block's 128 rows (the pass over the tensor that
        // chan_stats_partial would make): a lane holds 32 rows of its column(s) -> two-pass (mean, M2) in registers, Chan-combined with the
        // other lane half (xor 32) and, through LDS, with the wave that holds the other 64 rows.  Fixed order: bitwise reproducible.
        float* xch = reinterpret_cast<float*>(smem);                       // [2 row halves][BN columns][3]: the stages are dead by now
        lds_barrier();
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            float cnt = 0.f, s1 = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < g.rows;
                    cnt += ok ? 1.f : 0.f;
                    s1 += ok ? acc[a][b][r] : 0.f;
                }
            float mean = cnt > 0.f ? s1 / cnt : 0.f, m2 = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < g.rows;
                    const float d = acc[a][b][r] - mean;
                    m2 += ok ? d * d : 0.f;
                }
            // the other lane half (rows + 4)
            const float cnt_o = __shfl_xor(cnt, 32, 64), mean_o = __shfl_xor(mean, 32, 64), m2_o = __shfl_xor(m2, 32, 64);
            {
                const float nt = cnt + cnt_o, dlt = mean_o - mean;
                if (nt > 0.f) { m2 = m2 + m2_o + dlt * dlt * (cnt * cnt_o / nt); mean = mean + dlt * (cnt_o / nt); }
                cnt = nt;
            }
            if (lh == 0) {
                float* o = xch + (((wid >> 1) * BN) + wn0 + b * 32 + li) * 3;
                o[0] = cnt; o[1] = mean; o[2] = m2;
            }
        }
        lds_barrier();
        for (int cl = tid; cl < BN; cl += 256) {
            const int col = n0 + cl;
            if (col < g.n) {
                const float* a0 = xch + cl * 3;
                const float* a1 = xch + (BN + cl) * 3;
                float cnt = a0[0], mean = a0[1], m2 = a0[2];
                const float nt = cnt + a1[0], dlt = a1[1] - mean;
                if (nt > 0.f) { m2 = m2 + a1[2] + dlt * dlt * (cnt * a1[0] / nt); mean = mean + dlt * (a1[0] / nt); }
                float* o = g.stats + ((long)mt * g.n + col) * 3;
                o[0] = nt; o[1] = mean; o[2] = m2;
            }
        }
    }
}

// w (fp32) -> split planes dst[z][plane 3][k/8][n][8] bf16 with B_z[kk][col] = w[z * stride_z + kk * sk + col * sn]; thread = (z, octet, column)
__global__ __launch_bounds__(256) void conv_x3_pack_kernel(const float* __restrict__ w, long stride_z, long sk, long sn, __bf16* __restrict__ dst, int batch,
                                                           int k, int n) {
    derive::pack_x3_body(w, stride_z, sk, sn, dst, batch, k, n, blockIdx.x);
}

bool mode_ok(int mode) { return mode == RUNET_CONV_FWD || mode == RUNET_CONV_DGRAD || mode == RUNET_CONVT_FWD || mode == RUNET_CONVT_DGRAD; }

// 128 x 64 tiles where 128 x 128 would leave the 256 CUs (two resident blocks each) unevenly filled, or the output is narrow
bool narrow_tile(long rows, int n, int gz) {
    static const int force = getenv("RUNET_CONV_X3_BN") ? atoi(getenv("RUNET_CONV_X3_BN")) : 0;      // measurement knob: 64 / 128
    if (force == 64) return true;
    if (force == 128) return false;
    if (n <= 64) return true;
    const long b128 = (long)cdiv(rows, 128) * cdiv(n, 128) * gz;
    const double rounds = b128 / 512.0;
    return rounds / (double)((b128 + 511) / 512) < 0.8;
}

}  // namespace

extern "C" int runet_conv_x3_supported(int cin, int cout, int mode) {
    return (mode_ok(mode) && cin >= 16 && cin % 16 == 0 && cout >= 4 && cout % 4 == 0) ? 1 : 0;
}

extern "C" long runet_conv_x3_pack_elems(int cin, int cout, int mode) {
    return 3L * cin * cout * ((mode == RUNET_CONVT_FWD || mode == RUNET_CONVT_DGRAD) ? 4 : 1);
}

// cin = channels of the tensor the convolution READS in this mode (the contraction), cout = channels it WRITES.  The weight is always the
// module's forward weight in HWIO: 1x1 [Ci][Co], transposed [2][2][Ci][Co]; the data-gradient modes read it transposed (cin = Co, cout = Ci).
extern "C" int runet_conv_x3_pack(const float* w, void* packed, int cin, int cout, int mode, void* stream) {
    RUNET_REQUIRE(w && packed && runet_conv_x3_supported(cin, cout, mode), "bad arguments (cin: multiple of 16, cout: multiple of 4)");
    RUNET_REQUIRE(((uintptr_t)packed % 16) == 0, "alignment");
    int taps;
    long stride_z, sk, sn;
    derive::conv_x3_pack_strides(cin, cout, mode, taps, stride_z, sk, sn);
    const long total = (long)taps * (cin / 8) * cout;
    hipLaunchKernelGGL(conv_x3_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, stride_z, sk, sn, (__bf16*)packed, taps, cin, cout);
    RUNET_CHECK_LAUNCH();
}

extern "C" const char* runet_conv_x3_kernel_name(int n_img, int h, int w_, int cout, int mode) {
    return narrow_tile((long)n_img * h * w_, cout, mode == RUNET_CONVT_FWD ? 4 : 1) ? "conv_nn_x3_kernel<64>" : "conv_nn_x3_kernel<128>";
}

// h, w_: the iteration space - the image the 1x1 convolution runs over; for both transposed modes the LOW-resolution image (the
// forward's input / the data gradient's output), the other side being 2h x 2w_.
extern "C" int runet_conv_x3_stats_parts(int n_img, int h, int w_) { return cdiv((long)n_img * h * w_, 128); }

static int conv_x3_launch(const float* x, int ldx, const void* wpacked, const float* bias, float* y, int ldy, int n_img, int h, int w_, int cin,
                          int cout, int mode, int accumulate, float* stats, void* stream);

extern "C" int runet_conv_x3(const float* x, int ldx, const void* wpacked, const float* bias, float* y, int ldy, int n_img, int h, int w_, int cin,
                             int cout, int mode, int accumulate, void* stream) {
    return conv_x3_launch(x, ldx, wpacked, bias, y, ldy, n_img, h, w_, cin, cout, mode, accumulate, nullptr, stream);
}

// runet_conv_x3 (1x1 modes only) that also leaves the BatchNorm statistics partials of its output behind: stats [runet_conv_x3_stats_parts][cout][3]
extern "C" int runet_conv_x3_stats(const float* x, int ldx, const void* wpacked, const float* bias, float* y, int ldy, int n_img, int h, int w_, int cin,
                                   int cout, int mode, int accumulate, float* stats, void* stream) {
    RUNET_REQUIRE(stats && (mode == RUNET_CONV_FWD || mode == RUNET_CONV_DGRAD), "statistics: 1x1 modes only, stats must not be NULL");
    return conv_x3_launch(x, ldx, wpacked, bias, y, ldy, n_img, h, w_, cin, cout, mode, accumulate, stats, stream);
}

static int conv_x3_launch(const float* x, int ldx, const void* wpacked, const float* bias, float* y, int ldy, int n_img, int h, int w_, int cin,
                          int cout, int mode, int accumulate, float* stats, void* stream) {
    RUNET_REQUIRE(x && wpacked && y, "null pointer");
    RUNET_REQUIRE(runet_conv_x3_supported(cin, cout, mode), "shape / mode not supported (cin: multiple of 16, cout: multiple of 4)");
    RUNET_REQUIRE(n_img > 0 && h > 0 && w_ > 0 && (long)n_img * h * w_ * 4 < (1L << 31), "iteration space empty or too large");
    RUNET_REQUIRE(ldx >= cin && ldx % 4 == 0 && ldy >= cout, "pixel strides must cover the channels (ldx: multiple of 4)");
    RUNET_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)wpacked % 16) == 0 && (!bias || ((uintptr_t)bias % 4) == 0), "alignment");
    X3ConvArgs g{};
    g.a = x; g.lda = ldx; g.b = (const __bf16*)wpacked; g.sb = 3L * cin * cout; g.bias = bias; g.c = y; g.ldc = ldy;
    g.rows = n_img * h * w_; g.k = cin; g.n = cout; g.H = h; g.W = w_; g.a_scale = 1; g.c_scale = 1; g.ntaps = 1; g.accumulate = accumulate; g.gz = 1;
    g.stats = stats;
    if (mode == RUNET_CONVT_FWD) { g.c_scale = 2; g.gz = 4; }
    if (mode == RUNET_CONVT_DGRAD) { g.a_scale = 2; g.ntaps = 4; }
    g.gm = cdiv(g.rows, 128);
    hipStream_t st = (hipStream_t)stream;
    if (narrow_tile(g.rows, cout, g.gz)) {
        g.gn = cdiv(cout, 64);
        hipLaunchKernelGGL(conv_nn_x3_kernel<64>, dim3((unsigned)((long)g.gm * g.gn * g.gz)), dim3(256), 3 * NNX3<64>::STAGE + 1024, st, g);
    } else {
        g.gn = cdiv(cout, 128);
        hipLaunchKernelGGL(conv_nn_x3_kernel<128>, dim3((unsigned)((long)g.gm * g.gn * g.gz)), dim3(256), 3 * NNX3<128>::STAGE + 1024, st, g);
    }
    RUNET_CHECK_LAUNCH();
}
