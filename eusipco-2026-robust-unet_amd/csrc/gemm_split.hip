// fp32 GEMMs on the gfx950 BF16 matrix cores by exact three-way operand splitting ("bf16x6").
//
// Why: on MI355X the f32-input MFMA (v_mfma_f32_32x32x2_f32) runs at the FP32 vector rate, 64 FLOP/clk/SIMD = 157 TFLOP/s, 1/16 of the
// BF16 MFMA rate (MI355X_MICROARCH.md, Matrix cores).  An fp32 value has 24 significant bits, a bf16 value 8 with the SAME exponent
// range, so   x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)   holds EXACTLY (both subtractions are exact in fp32,
// and the last residual has at most 8 significant bits).  A product of two fp32 numbers is then the sum of nine bf16 x bf16 products, each
// of which the BF16 MFMA forms exactly and accumulates in fp32.  The three smallest of them (m*l, l*m, l*l <= 2^-23 |a b|, below the
// half-ulp 2^-24 |a b| a rounded fp32 product would carry... they are of the size of ONE rounding of the product) are dropped, six remain:
//     a*b  ~=  h_a h_b + (h_a m_b + m_a h_b) + (h_a l_b + m_a m_b + l_a h_b)            |error| <= 2^-23 |a b| per product, unbiased
// i.e. the result is an fp32-accumulated dot product whose per-term error is of the order of fp32's own rounding - not a reduced-precision
// GEMM.  tests/test_gpu_conv.py measures it against float64 beside the f32-MFMA kernels it replaces.
// Cost: 6 MFMAs of 32x32x16 (32 cycles each) per 16-deep k-step of a 32x32 tile = 192 cycles against 8 x 64 = 512 cycles of
// v_mfma_f32_32x32x2_f32: 2.67x less matrix time, after which these GEMMs are bound by moving their operands, like everything else on this chip.
//
//   gemm_nn_x3_kernel: C[z][rows][n]     = A[z][rows][k] . B[z][k][n]       A fp32, split on its way into LDS;
//                                                                            B PRE-SPLIT and packed once per optimizer step (weights):
//                                                                            Bp[z][plane 3][k/8][n][8] bf16 - a lane's B fragment is one 16-B unit
//   gemm_tn_x3_kernel: C[split][z][k][n] = sum over the split's rows of A[z][row][k] * B[z][row][n]      both fp32, split on the way into LDS,
//                                                                            fragments by the transposing LDS read (ds_read_b64_tr_b16): the
//                                                                            contraction index (rows) is the slow memory index of both
// Block = 128 x BN output tile (BN = 128: 4 waves of 64 x 64; BN = 64: 4 waves of 64 x 32), 16-deep k-steps through a three-stage LDS ring
// (global loads two steps ahead of their LDS store, as gemm.hip), one LDS-only barrier per step, two blocks per CU.
// XCD-aware block order: consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2; the linear id is remapped so that
// every XCD owns a CONTIGUOUS range of (z, row-tile, column-tile) with the column tile fastest - the column tiles that share an A tile
// and the row tiles that share a B matrix meet in one L2 (gemm.hip's order re-fetched A across XCDs: 1.49x its algorithmic bytes).
#include "x3_common.h"
#include "../../include/runet_hip.h"
#include <stdlib.h>

namespace {

using namespace x3;

struct X3Args {
    const float* a; int lda; long sa;
    const void* b; int ldb; long sb;          // NN: packed planes (sb = elements per z);  TN: fp32
    float* c; int ldc; long sc;
    int rows, k, n;
    int rps;                                  // TN: rows per split (multiple of 16)
    int gm, gn, gz;                           // tile grid (row tiles, column tiles, batch)
    int cw;                                   // TN, transposed-convolution form: width W of the low-resolution image (rows = its pixels); B = dy [.., 2H, 2W, n]
};

// ------------------------------------------------------------------------------------------------------------------ NN
template <int BN>
__global__ __launch_bounds__(256, 2) void gemm_nn_x3_kernel(X3Args g) {
    using C = NNX3<BN>;
    constexpr int TN = C::TN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * (BN / 2);
    const int li = lane & 31, lh = lane >> 5;
    // tile coordinates from the remapped linear id: column tile fastest, then row tile, then z
    const long total = (long)g.gm * g.gn * g.gz;
    long id = xcd_remap(blockIdx.x, total);
    const int nt = (int)(id % g.gn); id /= g.gn;
    const int mt = (int)(id % g.gm);
    const int z = (int)(id / g.gm);
    const int m0 = mt * 128, n0 = nt * BN;
    const float* A = g.a + (long)z * g.sa;
    const __bf16* B = reinterpret_cast<const __bf16*>(g.b) + (long)z * g.sb;
    float* Cc = g.c + (long)z * g.sc;
    const int K8 = g.k >> 3;

    // ---- loaders.  A item (row, k-quad) = (tid >> 2 (+64), tid & 3): rows beyond `rows` clamped (their products are never stored).
    const int akq = tid & 3, arow = tid >> 2;
    const float* ap[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        int r = m0 + arow + 64 * v;
        r = r < g.rows ? r : g.rows - 1;
        ap[v] = A + (long)r * g.lda + akq * 4;
    }
    // B item i = tid + 256 j -> (plane, octet, column) = (i / (2 BN), (i / BN) & 1, i % BN); columns beyond n clamped
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
    const long bstep = (long)2 * g.n * 8;              // elements per k-step (two octets)
    struct Raw { f32x4 ra[2]; f32x4 rb[C::BITEMS]; };
    const int nks = g.k >> 4;
    auto load_tile = [&](int ks, Raw& R) {
        const int kc = ks < nks ? ks : nks - 1;        // past the end: re-read the last tile into a stage nobody multiplies
#pragma unroll
        for (int v = 0; v < 2; ++v) R.ra[v] = *reinterpret_cast<const f32x4*>(ap[v] + kc * 16);
#pragma unroll
        for (int j = 0; j < C::BITEMS; ++j) R.rb[j] = *reinterpret_cast<const f32x4*>(bp[j] + kc * bstep);
    };
    // LDS addresses of this thread's stores
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

    // fragment addresses (bytes inside a stage)
    int a_rd[2], b_rd[TN];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = wm0 + a * 32 + li;
        a_rd[a] = (row * 2 + (lh ^ ((row >> 3) & 1))) * 16;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) b_rd[b] = C::A_BYTES + (lh * BN + wn0 + b * 32 + li) * 16;

    // Four register sets (set = tile % 4): a tile's global loads are issued FOUR steps before its LDS store.  A step is 24 MFMAs = 768
    // cycles per wave (the f32-MFMA kernel's: 2048), an L2 / HBM round trip under load 2-3 us: two steps ahead (gemm.hip's depth) left
    // every step waiting on vmcnt.
    Raw R0, R1, R2, R3;
    load_tile(0, R0);
    load_tile(1, R1);
    store_tile(0, R0);
    store_tile(1, R1);
    load_tile(2, R2);
    load_tile(3, R3);
    load_tile(4, R0);
    lds_barrier();
    int cur = 0;
    // step s: issue the loads of tile s+5, multiply tile s, split + store tile s+2 (loaded during step s-3) into the stage of tile s-1
    auto step = [&](int s, const Raw& cur_set, Raw& nxt_set) {
        const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
        load_tile(s + 5, nxt_set);
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
    // Four UNCONDITIONAL steps per trip: with `if (s + 1 < nks) step(...)` inside the loop the CFG has a path header -> step 0 -> latch -> header
    // on which step 0's loads are still in flight when the header's LDS reads reuse their registers, and the compiler's s_waitcnt insertion
    // (conservative over all paths) drained vmcnt to 0 at every fourth step - the four-deep prefetch never got more than one step deep there.
    int s = 0;
    for (; s + 4 <= nks; s += 4) {
        step(s, R2, R1);
        step(s + 1, R3, R2);
        step(s + 2, R0, R3);
        step(s + 3, R1, R0);
    }
    if (s < nks) {                                      // k not a multiple of 64: up to three more
        step(s, R2, R1);
        if (s + 1 < nks) {
            step(s + 1, R3, R2);
            if (s + 2 < nks) step(s + 2, R0, R3);
        }
    }

    // ---- epilogue: lane holds column li of each 32-wide tile, rows (r&3) + 8*(r>>2) + 4*lh
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < g.rows) {
                float* crow = Cc + (long)row * g.ldc;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int col = n0 + wn0 + b * 32 + li;
                    if (col < g.n) crow[col] = acc[a][b][r];
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------ TN
// LDS stage: As[plane 3][16 rows][128 k-columns] bf16 + Bs[plane 3][16 rows][128 n-columns] bf16, 256-B rows; the four 64-B chunks of a
// row are XOR-swizzled with (row & 3): the transposing read addresses 4 consecutive rows x 64 B per 32-lane half - with the swizzle the
// four rows land on four different bank quarters (the linear image of conv_lowp.inc's weight gradient is 2-way conflicted: 40 % of its LDS cycles).
constexpr int TN_PLANE = 16 * 256, TN_OP = 3 * TN_PLANE, TN_STAGE_B = 2 * TN_OP;

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of block row q, columns 4p..4p+3; lane i receives column i of the 4 rows
__device__ __forceinline__ bf16x8 tr_frag8(const unsigned char* plane, int col0, int row_lo) {
    // this lane ADDRESSES rows row_lo and row_lo + 4 at columns col0 .. col0+3 (col0 includes the lane's 16-column half and quad)
    const int ch = col0 >> 5, within = (col0 & 31) * 2;       // 64-B chunk (32 bf16), byte offset inside it
    const int r0 = row_lo, r1 = row_lo + 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(plane + r0 * 256 + ((ch ^ (r0 & 3)) << 6) + within));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(plane + r1 * 256 + ((ch ^ (r1 & 3)) << 6) + within));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo; u.s[1] = hi;
    return u.b;
}

// CONVT: the weight gradient of ConvTranspose2d(k2, s2) - dW[a][b][ci][co] = sum over low-resolution pixels p of x[p][ci] * dy[(2h + a, 2w + b)][co].
// z = tap (a, b); A rows are the low-resolution pixels as they lie in memory; the B row of pixel (img, h, w) is the high-resolution pixel
// ((img * H + h) * 2 + a) * 2W + 2w + b.  W is a multiple of 16 (host check), so the 16 rows of a k-step share (img, h): their B rows are 2 pixels
// apart from a per-step base that advances by 32 pixels, plus 2W more (the skipped odd / even row) when the step wraps to the next image row.
template <bool CONVT>
__global__ __launch_bounds__(256, 2) void gemm_tn_x3_kernel(X3Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    // tile = (z, k-tile, n-tile) from the remapped id of blockIdx.x; blockIdx.y = split.  n-tile fastest: the column tiles of one
    // (z, k-tile) share the A rows, all tiles of one z share both operands' rows of this split
    const long total = (long)g.gm * g.gn * g.gz;
    long id = xcd_remap(blockIdx.x, total);
    const int nt = (int)(id % g.gn); id /= g.gn;
    const int kt = (int)(id % g.gm);
    const int z = (int)(id / g.gm);
    const int k0 = kt * 128, n0 = nt * 128;
    const float* A = g.a + (CONVT ? 0L : (long)z * g.sa);
    const float* B = reinterpret_cast<const float*>(g.b) + (CONVT ? 0L : (long)z * g.sb);
    const int t_begin = blockIdx.y * g.rps;
    const int t_end = min(g.rows, t_begin + g.rps);
    const int nks = (t_end - t_begin) >> 4;            // whole 16-row tiles (host check)

    // loaders: item (row, column quad) = (tid >> 5 (+8), tid & 31) for both operands; columns beyond k / n clamped (never stored)
    const int q4 = (tid & 31) * 4;
    const int acol = k0 + q4 < g.k ? k0 + q4 : g.k - 4, bcol = n0 + q4 < g.n ? n0 + q4 : g.n - 4;
    struct Raw { f32x4 ra[2], rb[2]; };
    // CONVT: base offset (elements) of the NEXT step's B rows and its column in the image row; steps are requested strictly in order
    long cv_boff = 0;
    int cv_w0 = 0, cv_ks = 0;
    if constexpr (CONVT) {
        const int R0 = t_begin / g.cw;
        cv_w0 = t_begin - R0 * g.cw;
        cv_boff = ((long)(2 * R0 + (z >> 1)) * (2 * g.cw) + 2 * cv_w0 + (z & 1)) * g.ldb;
    }
    auto load_tile = [&](int ks, Raw& R) {
        const int kc = ks < nks ? ks : nks - 1;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int rr = (tid >> 5) + 8 * v;
            const long tr = t_begin + kc * 16 + rr;
            R.ra[v] = *reinterpret_cast<const f32x4*>(A + tr * g.lda + acol);
            if constexpr (CONVT) R.rb[v] = *reinterpret_cast<const f32x4*>(B + cv_boff + (long)(2 * rr) * g.ldb + bcol);
            else R.rb[v] = *reinterpret_cast<const f32x4*>(B + tr * g.ldb + bcol);
        }
        if constexpr (CONVT) {
            if (cv_ks + 1 < nks) {                     // past the end the last step is read again
                ++cv_ks;
                cv_w0 += 16;
                cv_boff += 32L * g.ldb;
                if (cv_w0 >= g.cw) { cv_w0 = 0; cv_boff += 2L * g.cw * g.ldb; }
            }
        }
    };
    int st_off[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int row = (tid >> 5) + 8 * v;
        st_off[v] = row * 256 + ((((q4 >> 5) ^ (row & 3))) << 6) + (q4 & 31) * 2;
    }
    auto store_tile = [&](int buf, const Raw& R) {
        unsigned char* As = smem + buf * TN_STAGE_B;
        unsigned char* Bs = As + TN_OP;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            bf16x4 h, m, l;
            split3(R.ra[v], h, m, l);
            *reinterpret_cast<bf16x4*>(As + st_off[v]) = h;
            *reinterpret_cast<bf16x4*>(As + TN_PLANE + st_off[v]) = m;
            *reinterpret_cast<bf16x4*>(As + 2 * TN_PLANE + st_off[v]) = l;
            split3(R.rb[v], h, m, l);
            *reinterpret_cast<bf16x4*>(Bs + st_off[v]) = h;
            *reinterpret_cast<bf16x4*>(Bs + TN_PLANE + st_off[v]) = m;
            *reinterpret_cast<bf16x4*>(Bs + 2 * TN_PLANE + st_off[v]) = l;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposing-read role of this lane: block row q = (lane >> 2) & 3 of the 4-row block, contraction half lh (rows 8 lh ..), 16-column
    // half (lane >> 4) & 1 of the 32 columns, column quad lane & 3
    const int trow = 8 * lh + ((lane >> 2) & 3);
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    Raw R0, R1, R2, R3;                               // four sets: loads four steps ahead of their LDS store (see gemm_nn_x3_kernel)
    load_tile(0, R0);
    load_tile(1, R1);
    store_tile(0, R0);
    store_tile(1, R1);
    load_tile(2, R2);
    load_tile(3, R3);
    load_tile(4, R0);
    lds_barrier();
    int cur = 0;
    auto step = [&](int s, const Raw& cur_set, Raw& nxt_set) {
        const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
        load_tile(s + 5, nxt_set);
        const unsigned char* As = smem + cur * TN_STAGE_B;
        const unsigned char* Bs = As + TN_OP;
        bf16x8 af[2][3], bf[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a][p] = tr_frag8(As + p * TN_PLANE, wm0 + a * 32 + tcol, trow);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b][p] = tr_frag8(Bs + p * TN_PLANE, wn0 + b * 32 + tcol, trow);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) X3_MMA(acc[a][b], af[a], bf[b]);
        store_tile(nx2, cur_set);
        lds_barrier();
        __builtin_amdgcn_sched_barrier(0);      // keep the next step's split / stores from being hoisted over this one (they would pull its vmcnt wait forward)
        cur = nxt;
    };
    // Four UNCONDITIONAL steps per trip: with `if (s + 1 < nks) step(...)` inside the loop the CFG has a path header -> step 0 -> latch -> header
    // on which step 0's loads are still in flight when the header's LDS reads reuse their registers, and the compiler's s_waitcnt insertion
    // (conservative over all paths) drained vmcnt to 0 at every fourth step - the four-deep prefetch never got more than one step deep there.
    int s = 0;
    for (; s + 4 <= nks; s += 4) {
        step(s, R2, R1);
        step(s + 1, R3, R2);
        step(s + 2, R0, R3);
        step(s + 3, R1, R0);
    }
    if (s < nks) {                                      // k not a multiple of 64: up to three more
        step(s, R2, R1);
        if (s + 1 < nks) {
            step(s + 1, R3, R2);
            if (s + 2 < nks) step(s + 2, R0, R3);
        }
    }

    float* Cs = g.c + ((long)blockIdx.y * g.gz + z) * g.k * g.n;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn0 + b * 32 + li;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = k0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (col < g.n && kk < g.k) Cs[(long)kk * g.n + col] = acc[a][b][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------ B packing
// src fp32 [z][k][n] (row stride n) -> dst [z][plane 3][k/8][n][8] bf16; thread = (z, octet, column)
__global__ __launch_bounds__(256) void x3_pack_kernel(const float* __restrict__ src, long stride_src, __bf16* __restrict__ dst, int batch, int k, int n) {
    const int K8 = k >> 3;
    const long per = (long)K8 * n, total = per * batch;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int z = (int)(i / per);
    const long r = i - (long)z * per;
    const int oc = (int)(r / n), col = (int)(r - (long)oc * n);
    const float* s = src + (long)z * stride_src + (long)oc * 8 * n + col;
    bf16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = s[(long)j * n];
        const __bf16 hj = (__bf16)x;
        const float r1 = x - (float)hj;
        const __bf16 mj = (__bf16)r1;
        h[j] = hj; m[j] = mj; l[j] = (__bf16)(r1 - (float)mj);
    }
    __bf16* d = dst + (long)z * 3 * per * 8 + r * 8;
    *reinterpret_cast<bf16x8*>(d) = h;
    *reinterpret_cast<bf16x8*>(d + per * 8) = m;
    *reinterpret_cast<bf16x8*>(d + 2 * per * 8) = l;
}

}  // namespace

extern "C" int runet_gemm_x3_supported(int rows, int k, int n) { return (rows > 0 && k >= 16 && k % 16 == 0 && n >= 4 && n % 4 == 0) ? 1 : 0; }

extern "C" long runet_gemm_x3_pack_elems(int batch, int k, int n) { return 3L * batch * k * n; }

extern "C" int runet_gemm_x3_pack(const float* b, long stride_b, void* packed, int batch, int k, int n, void* stream) {
    RUNET_REQUIRE(b && packed && batch > 0 && k > 0 && k % 8 == 0 && n > 0, "bad arguments (k: multiple of 8)");
    RUNET_REQUIRE(((uintptr_t)packed % 16) == 0, "alignment");
    const long total = (long)batch * (k / 8) * n;
    hipLaunchKernelGGL(x3_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, b, stride_b, (__bf16*)packed, batch, k, n);
    RUNET_CHECK_LAUNCH();
}

// which tile the NN kernel uses for a shape: 128 x 64 where 128 x 128 tiles would fill the 256 CUs (two resident blocks each) unevenly
static bool nn_x3_narrow(int batch, int rows, int n) {
    static const int force = getenv("RUNET_GEMM_X3_BN") ? atoi(getenv("RUNET_GEMM_X3_BN")) : 0;      // measurement knob: 64 / 128
    if (force == 64) return true;
    if (force == 128) return false;
    if (n <= 64) return true;
    // narrower tiles only where 128 x 128 tiles cannot fill the chip once (512 resident blocks): small-batch shards.  The earlier rule also chose
    // them for unevenly filled last waves (balance < 0.8); with the loop's prefetch fixed the two rules measure the same at 16 x 256^2 (586.3 vs 586.4
    // img/s) and the simpler one is 0.3 % ahead on the 2-image shard
    static const int mode = getenv("RUNET_GEMM_X3_BALANCE") ? atoi(getenv("RUNET_GEMM_X3_BALANCE")) : 0;      // measurement knob: 1 = the earlier rule
    const long b128 = (long)cdiv(rows, 128) * cdiv(n, 128) * batch;
    if (mode == 1) {
        const double rounds = b128 / 512.0;
        return rounds / (double)((b128 + 511) / 512) < 0.8;
    }
    return b128 < 512;
}

extern "C" const char* runet_gemm_x3_kernel_name(int batch, int rows, int k, int n) {
    return nn_x3_narrow(batch, rows, n) ? "gemm_nn_x3_kernel<64>" : "gemm_nn_x3_kernel<128>";
}

extern "C" int runet_gemm_x3_batched(const float* a, int lda, long stride_a, const void* packed_b, float* c, int ldc, long stride_c, int batch,
                                     int rows, int k, int n, void* stream) {
    RUNET_REQUIRE(a && packed_b && c && batch > 0 && runet_gemm_x3_supported(rows, k, n), "bad arguments (k: multiple of 16, n: multiple of 4)");
    RUNET_REQUIRE(lda >= k && lda % 4 == 0 && ldc >= n && ((uintptr_t)a % 16) == 0 && ((uintptr_t)packed_b % 16) == 0 && stride_a % 4 == 0, "alignment");
    X3Args g{};
    g.a = a; g.lda = lda; g.sa = stride_a; g.b = packed_b; g.sb = 3L * k * n; g.c = c; g.ldc = ldc; g.sc = stride_c; g.rows = rows; g.k = k; g.n = n;
    g.gm = cdiv(rows, 128); g.gz = batch;
    hipStream_t st = (hipStream_t)stream;
    if (nn_x3_narrow(batch, rows, n)) {
        g.gn = cdiv(n, 64);
        const long total = (long)g.gm * g.gn * g.gz;
        RUNET_REQUIRE(total < (1L << 31), "grid too large");
        hipLaunchKernelGGL(gemm_nn_x3_kernel<64>, dim3((unsigned)total), dim3(256), 3 * NNX3<64>::STAGE, st, g);
    } else {
        g.gn = cdiv(n, 128);
        const long total = (long)g.gm * g.gn * g.gz;
        RUNET_REQUIRE(total < (1L << 31), "grid too large");
        hipLaunchKernelGGL(gemm_nn_x3_kernel<128>, dim3((unsigned)total), dim3(256), 3 * NNX3<128>::STAGE, st, g);
    }
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_gemm_x3_tn_batched(const float* a, int lda, long stride_a, const float* b, int ldb, long stride_b, float* c, int batch, int rows,
                                        int k, int n, int rows_per_split, void* stream) {
    RUNET_REQUIRE(a && b && c && batch > 0 && rows > 0 && rows % 16 == 0 && rows_per_split > 0 && rows_per_split % 16 == 0,
                  "bad arguments (rows, rows_per_split: multiples of 16)");
    RUNET_REQUIRE(k >= 4 && k % 4 == 0 && n >= 4 && n % 4 == 0 && lda >= k && lda % 4 == 0 && ldb >= n && ldb % 4 == 0, "k, n and the row strides must be multiples of 4");
    RUNET_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && stride_a % 4 == 0 && stride_b % 4 == 0, "alignment");
    const int splits = cdiv(rows, rows_per_split);
    RUNET_REQUIRE(splits <= 65535, "too many splits");
    X3Args g{};
    g.a = a; g.lda = lda; g.sa = stride_a; g.b = b; g.ldb = ldb; g.sb = stride_b; g.c = c; g.rows = rows; g.k = k; g.n = n; g.rps = rows_per_split;
    g.gm = cdiv(k, 128); g.gn = cdiv(n, 128); g.gz = batch;
    const long total = (long)g.gm * g.gn * g.gz;
    RUNET_REQUIRE(total < (1L << 31), "grid too large");
    hipLaunchKernelGGL(gemm_tn_x3_kernel<false>, dim3((unsigned)total, splits), dim3(256), 3 * TN_STAGE_B, (hipStream_t)stream, g);
    RUNET_CHECK_LAUNCH();
}

// Weight gradient of ConvTranspose2d(k2, s2) (Main_Final.py:261-270) on the split-operand TN GEMM: x [n_img, h, w, cin] (row stride ldx), dy
// [n_img, 2h, 2w, cout] (ldy) -> c [splits][4 taps][cin][cout], splits = ceil(n_img h w / rows_per_split); w a multiple of 16.
extern "C" int runet_gemm_x3_tn_convt(const float* x, int ldx, const float* dy, int ldy, float* c, int n_img, int h, int w, int cin, int cout,
                                      int rows_per_split, void* stream) {
    const long rows = (long)n_img * h * w;
    RUNET_REQUIRE(x && dy && c && n_img > 0 && h > 0 && w > 0 && w % 16 == 0 && rows < (1L << 31) && rows_per_split > 0 && rows_per_split % 16 == 0,
                  "bad arguments (w and rows_per_split: multiples of 16)");
    RUNET_REQUIRE(cin >= 4 && cin % 4 == 0 && cout >= 4 && cout % 4 == 0 && ldx >= cin && ldx % 4 == 0 && ldy >= cout && ldy % 4 == 0, "channel counts and strides must be multiples of 4");
    RUNET_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "alignment");
    const int splits = cdiv(rows, rows_per_split);
    RUNET_REQUIRE(splits <= 65535, "too many splits");
    X3Args g{};
    g.a = x; g.lda = ldx; g.b = dy; g.ldb = ldy; g.c = c; g.rows = (int)rows; g.k = cin; g.n = cout; g.rps = rows_per_split; g.cw = w;
    g.gm = cdiv(cin, 128); g.gn = cdiv(cout, 128); g.gz = 4;
    const long total = (long)g.gm * g.gn * g.gz;
    hipLaunchKernelGGL(gemm_tn_x3_kernel<true>, dim3((unsigned)total, splits), dim3(256), 3 * TN_STAGE_B, (hipStream_t)stream, g);
    RUNET_CHECK_LAUNCH();
}
