// Batched fp32 GEMMs on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32), the position-GEMMs of the unfused Winograd F(4x4,3x3)
// path (conv_winograd4.hip):
//   gemm_nn_kernel: C[z][rows][n]      = A[z][rows][k] . B[z][k][n]
//   gemm_tn_kernel: C[split][z][k][n]  = sum over the split's rows of A[z][row][k] * B[z][row][n]
// Block = 128 x 128 output tile, 4 waves of 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers... x4 = 128 per lane), 16-deep
// k-steps through a THREE-stage LDS ring: the global loads of step s+2 are in flight while step s multiplies, and the first
// operand fragments of step s+1 are read from LDS during the second half of step s, so the barrier at the end of a step does not
// expose LDS latency (the two-stage form of conv_igemm.hip reads all fragments right after the barrier).  Every global load is
// unconditional (clamped address, value zeroed by a select) so the compiler's vmcnt counting stays exact.
#include "runet_common.h"
#include "../../include/runet_hip.h"
#include <stdlib.h>

namespace {

struct GemmArgs {
    const float* a; int lda; long sa;
    const float* b; int ldb; long sb;
    float* c; int ldc; long sc;
    int rows, k, n;
    int rps;            // TN: rows per split (multiple of 16)
};

constexpr int BK = 16;            // TN kernel k-step (rows per stage)

// NN kernel: KB-deep k-steps (16: three LDS stages, 32: two fragment sets still, half the barriers per FLOP)
template <int KB>
struct NNCfg {
    static constexpr int LDA = KB + 4, LDB = 128 + 4;
    static constexpr int A = 128 * LDA, B = KB * LDB, STAGE = A + B;
    static constexpr int AQ = KB / 4;                 // float4 per A row
    static constexpr int AV = 128 * AQ / 256;         // A float4 loads per thread
    static constexpr int BV = KB * 32 / 256;          // B float4 loads per thread
    static constexpr int NKH = KB / 8;                // 8-deep fragment groups per step
};

// FAST (k a multiple of KB - every layer of the network): loads are unconditional and nothing is zero-filled.  Rows beyond `rows` and
// columns beyond `n` are clamped to valid addresses at set-up (their products land in output rows / columns that are never stored), the
// tiles prefetched past the end re-read the last k-tile.  The general form spends ~55 VALU instructions per 32 MFMAs on predicates,
// address selects and zero selects; fp32 MFMAs share SIMD cycles with VALU work (conv_winograd.hip).
template <int KB, bool FAST = false>
__global__ __launch_bounds__(256, 2) void gemm_nn_kernel(GemmArgs g) {
    using C = NNCfg<KB>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const float* A = g.a + (long)blockIdx.z * g.sa;
    const float* B = g.b + (long)blockIdx.z * g.sb;
    float* Cc = g.c + (long)blockIdx.z * g.sc;

    // ---- loaders: A item (row, k-quad) = (tid / AQ (+256/AQ per pass), tid % AQ);  B item (k-row, n-quad) = (tid >> 5 (+8), tid & 31)
    const int akq = tid % C::AQ, arow = tid / C::AQ, bnq = tid & 31;
    constexpr int ARSTEP = 256 / C::AQ;
    long aoff[C::AV]; bool aok[C::AV];
#pragma unroll
    for (int v = 0; v < C::AV; ++v) {
        const int r = m0 + arow + ARSTEP * v;
        aok[v] = r < g.rows;
        aoff[v] = (long)(aok[v] ? r : 0) * g.lda + akq * 4;
    }
    const bool bn_ok = n0 + bnq * 4 < g.n;
    const int bcol = bn_ok ? n0 + bnq * 4 : 0;
    // One k-tile of raw global loads (validity applied at the LDS store, so the loads stay in flight).  TWO sets: the loads of tile
    // s+3 are issued in step s and stored at the end of step s+1 - two whole steps (~5000 cycles) to arrive.  With one set (issued and
    // stored in the same step) every step ended in a vmcnt wait for loads that take longer than a step under load: 58 % MFMA-busy.
    struct Raw { f32x4 ra[C::AV], rb[C::BV]; bool oka[C::AV], okb[C::BV]; };
    long boff[C::BV];                                  // FAST: B float4 of k-row (tid >> 5) + 8v at k-tile 0
#pragma unroll
    for (int v = 0; v < C::BV; ++v) boff[v] = (long)((tid >> 5) + 8 * v) * g.ldb + (bn_ok ? bcol : (g.n >= 4 ? g.n - 4 : 0));
    auto load_tile = [&](int kofs, Raw& R) {
        if constexpr (FAST) {
            const int kc = kofs < g.k ? kofs : g.k - KB;                   // wave-uniform
            const long kb = (long)kc * g.ldb;
#pragma unroll
            for (int v = 0; v < C::AV; ++v) R.ra[v] = *reinterpret_cast<const f32x4*>(A + aoff[v] + kc);
#pragma unroll
            for (int v = 0; v < C::BV; ++v) R.rb[v] = *reinterpret_cast<const f32x4*>(B + boff[v] + kb);
            return;
        }
#pragma unroll
        for (int v = 0; v < C::AV; ++v) {
            R.oka[v] = aok[v] && kofs + akq * 4 < g.k;
            R.ra[v] = *reinterpret_cast<const f32x4*>(A + (R.oka[v] ? aoff[v] + kofs : 0));
        }
#pragma unroll
        for (int v = 0; v < C::BV; ++v) {
            const int kr = kofs + (tid >> 5) + 8 * v;
            R.okb[v] = bn_ok && kr < g.k;
            R.rb[v] = *reinterpret_cast<const f32x4*>(B + (R.okb[v] ? (long)kr * g.ldb + bcol : 0));
        }
    };
    auto store_tile = [&](int buf, const Raw& R) {
        float* As = smem + buf * C::STAGE;
        float* Bs = As + C::A;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < C::AV; ++v)
            *reinterpret_cast<f32x4*>(As + (arow + ARSTEP * v) * C::LDA + akq * 4) = (FAST || R.oka[v]) ? R.ra[v] : zero;
#pragma unroll
        for (int v = 0; v < C::BV; ++v)
            *reinterpret_cast<f32x4*>(Bs + ((tid >> 5) + 8 * v) * C::LDB + bnq * 4) = (FAST || R.okb[v]) ? R.rb[v] : zero;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    struct Frag { f32x4 a[2]; float b[2][4]; };
    auto read_frag = [&](int buf, int kh, Frag& f) {
        const float* As = smem + buf * C::STAGE;
        const float* Bs = As + C::A;
#pragma unroll
        for (int a = 0; a < 2; ++a) f.a[a] = *reinterpret_cast<const f32x4*>(As + (wm0 + a * 32 + li) * C::LDA + kh * 8 + 4 * lh);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) f.b[b][q] = Bs[(kh * 8 + 4 * lh + q) * C::LDB + wn0 + b * 32 + li];
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[a][q], f.b[b][q], acc[a][b], 0, 0, 0);
    };

    const int nks = (g.k + KB - 1) / KB;
    // prologue: tiles 0 and 1 into stages 0 and 1 (a k extent of one step loads an all-zero second tile: kofs >= k), tile 2 in flight
    Raw R0, R1;
    load_tile(0, R0);
    load_tile(KB, R1);
    store_tile(0, R0);
    store_tile(1, R1);
    load_tile(2 * KB, R0);
    __syncthreads();
    Frag f[2];
    read_frag(0, 0, f[0]);
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): the loop header then merges two states with nothing outstanding
    int cur = 0;                      // stage holding tile s
    // The body is branch-free: past the end it re-loads zeros (kofs >= k), stores them into a stage nobody reads again and
    // prefetches fragments that are never multiplied, which keeps the compiler's waitcnt bookkeeping exact.
    // step s: issue the loads of tile s+3 into `nxt_set`, multiply tile s, store tile s+2 (loaded during step s-1, `cur_set`) into
    // the stage of tile s-1 (dead since the previous barrier).
    auto step = [&](int sidx, const Raw& cur_set, Raw& nxt_set) {
        const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
        load_tile((sidx + 3) * KB, nxt_set);
#pragma unroll
        for (int kh = 0; kh < C::NKH; ++kh) {
            if (kh + 1 < C::NKH) read_frag(cur, kh + 1, f[(kh + 1) & 1]);
            else read_frag(nxt, 0, f[(kh + 1) & 1]);        // tile s+1 became visible at the barrier that ended step s-1
            __builtin_amdgcn_sched_barrier(0);
            mma(f[kh & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        store_tile(nx2, cur_set);
        __syncthreads();
        cur = nxt;
    };
    for (int s = 0; s < nks; s += 2) {
        step(s, R0, R1);
        if (s + 1 < nks) step(s + 1, R1, R0);
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
                for (int b = 0; b < 2; ++b) {
                    const int col = n0 + wn0 + b * 32 + li;
                    if (col < g.n) crow[col] = acc[a][b][r];
                }
            }
        }
}

// ---- TN: contraction over rows.  LDS stage = As[16 rows][128 k] + Bs[16 rows][128 n]; operands are scalar LDS reads (row = t, column = lane)
constexpr int TN_LD = 128 + 4;
constexpr int TN_STAGE = 2 * BK * TN_LD;

// FAST (every split a whole number of 16-row tiles): unconditional loads, nothing zero-filled - columns beyond k / n are clamped to valid
// addresses (their products are never stored), the tiles prefetched past the end re-read the split's last tile into a stage nobody reads.
template <bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    const int n_tiles = (g.n + 127) >> 7;
    const int k0 = (blockIdx.x / n_tiles) * 128, n0 = (blockIdx.x % n_tiles) * 128;
    const int z = blockIdx.y;
    const float* A = g.a + (long)z * g.sa;
    const float* B = g.b + (long)z * g.sb;
    const int t_begin = blockIdx.z * g.rps;
    const int t_end = min(g.rows, t_begin + g.rps);

    // loaders: item (t-row, quad) = (tid >> 5 (+8), tid & 31) for both operands
    const int q4 = (tid & 31) * 4;
    const bool ak_ok = k0 + q4 < g.k, bn_ok = n0 + q4 < g.n;
    const int acol = ak_ok ? k0 + q4 : 0, bcol = bn_ok ? n0 + q4 : 0;
    struct Raw { f32x4 ra[2], rb[2]; bool okt[2]; };      // two sets: loads issued two steps ahead of their LDS store (see gemm_nn_kernel)
    auto load_tile = [&](int t0, Raw& R) {
        if constexpr (FAST) {
            const int tc = t0 < t_end ? t0 : t_end - BK;                    // wave-uniform
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const long tr = tc + (tid >> 5) + 8 * v;
                R.ra[v] = *reinterpret_cast<const f32x4*>(A + tr * g.lda + acol);
                R.rb[v] = *reinterpret_cast<const f32x4*>(B + tr * g.ldb + bcol);
            }
            return;
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int t = t0 + (tid >> 5) + 8 * v;
            R.okt[v] = t < t_end;
            const long tr = R.okt[v] ? t : 0;
            R.ra[v] = *reinterpret_cast<const f32x4*>(A + tr * g.lda + acol);
            R.rb[v] = *reinterpret_cast<const f32x4*>(B + tr * g.ldb + bcol);
        }
    };
    auto store_tile = [&](int buf, const Raw& R) {
        float* As = smem + buf * TN_STAGE;
        float* Bs = As + BK * TN_LD;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            *reinterpret_cast<f32x4*>(As + ((tid >> 5) + 8 * v) * TN_LD + q4) = (FAST || (R.okt[v] && ak_ok)) ? R.ra[v] : zero;
            *reinterpret_cast<f32x4*>(Bs + ((tid >> 5) + 8 * v) * TN_LD + q4) = (FAST || (R.okt[v] && bn_ok)) ? R.rb[v] : zero;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    struct Frag { float a[4][2], b[4][2]; };          // four k2-steps (8 rows of the stage)
    auto read_frag = [&](int buf, int half, Frag& f) {
        const float* As = smem + buf * TN_STAGE;
        const float* Bs = As + BK * TN_LD;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = half * 8 + 2 * s + lh;
#pragma unroll
            for (int a = 0; a < 2; ++a) f.a[s][a] = As[row * TN_LD + wm0 + a * 32 + li];
#pragma unroll
            for (int b = 0; b < 2; ++b) f.b[s][b] = Bs[row * TN_LD + wn0 + b * 32 + li];
        }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s][a], f.b[s][b], acc[a][b], 0, 0, 0);
    };

    const int nks = (t_end - t_begin + BK - 1) / BK;
    Raw R0, R1;
    load_tile(t_begin, R0);
    load_tile(t_begin + BK, R1);
    store_tile(0, R0);
    store_tile(1, R1);
    load_tile(t_begin + 2 * BK, R0);
    __syncthreads();
    Frag f0, f1;
    read_frag(0, 0, f0);
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0), see gemm_nn_kernel
    int cur = 0;
    auto step = [&](int sidx, const Raw& cur_set, Raw& nxt_set) {          // branch-free body, see gemm_nn_kernel
        const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
        load_tile(t_begin + (sidx + 3) * BK, nxt_set);
        read_frag(cur, 1, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(nxt, 0, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(f1);
        __builtin_amdgcn_sched_barrier(0);
        store_tile(nx2, cur_set);
        __syncthreads();
        cur = nxt;
    };
    for (int s = 0; s < nks; s += 2) {
        step(s, R0, R1);
        if (s + 1 < nks) step(s + 1, R1, R0);
    }

    float* C = g.c + ((long)blockIdx.z * gridDim.y + z) * g.k * g.n;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = n0 + wn0 + b * 32 + li;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = k0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (col < g.n && kk < g.k) C[(long)kk * g.n + col] = acc[a][b][r];
            }
    }
}

// ---- TN without LDS ("register-direct"): with the contraction over ROWS both operands are already in MFMA operand order in memory -
// lane (li, lh) of v_mfma_f32_32x32x2_f32 supplies A[m = li][k = lh] and B[k = lh][n = li], i.e. 32 consecutive floats of row t + lh of
// each matrix: one coalesced dword load per operand tile, straight into the register the MFMA reads.  No staging, no barrier, and no
// VALU work at all: the per-lane part of an address is a constant vector offset, the row advance rides in the buffer load's scalar
// offset.  (fp32 MFMAs and VALU instructions share SIMD cycles on gfx950 - see conv_winograd.hip - so "no VALU" is matrix time.)
// Wave = 64 x 64 of C (2 x 2 MFMA tiles), block = 2 x 2 waves; a ring of D k2-steps of loads is in flight per wave.
struct Yes { static constexpr bool value = true; };
struct No { static constexpr bool value = false; };

// TM x TN MFMA tiles per wave, WM x WN waves per block (block tile = 32 TM WM x 32 TN WN of C); D k2-steps of loads in flight.
// Narrow outputs (the 1x1 weight gradients of 32- / 64-channel layers) use one- or two-wave blocks and more row splits instead of
// multiplying zero columns.
template <int TM, int TN, int WM, int WN, int D>
__global__ __launch_bounds__(64 * WM * WN, TN >= 4 ? 2 : 1) void gemm_tn_direct_kernel(GemmArgs g) {
    constexpr int BMT = 32 * TM * WM, BNT = 32 * TN * WN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wid / WN) * 32 * TM, wn0 = (wid % WN) * 32 * TN;
    const int li = lane & 31, lh = lane >> 5;
    const int n_tiles = (g.n + BNT - 1) / BNT;
    const int k0 = (blockIdx.x / n_tiles) * BMT, n0 = (blockIdx.x % n_tiles) * BNT;
    const int z = blockIdx.y;
    const int t_begin = blockIdx.z * g.rps;
    const int t_end = min(g.rows, t_begin + g.rps);
    constexpr unsigned BIG = 0x80000000u;
    // descriptors start at the split's first row: scalar offsets stay small, num_records = the split's bytes (< 2^31, host check)
    const float* A = g.a + (long)z * g.sa + (long)t_begin * g.lda;
    const float* B = g.b + (long)z * g.sb + (long)t_begin * g.ldb;
    const int nrows = t_end - t_begin;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, nrows * g.lda * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, nrows * g.ldb * 4, 0x00020000);
    unsigned va[TM], vb[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = k0 + wm0 + a * 32 + li;
        va[a] = m < g.k ? (unsigned)(lh * g.lda + m) * 4u : BIG;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn0 + b * 32 + li;
        vb[b] = n < g.n ? (unsigned)(lh * g.ldb + n) * 4u : BIG;
    }
    const unsigned sa2 = 8u * g.lda, sb2 = 8u * g.ldb;          // bytes per k2-step (two rows)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float fa[D][TM], fb[D][TN];
    unsigned soa = 0, sob = 0;                                  // scalar offsets of the next step to LOAD
    int ls = 0;                                                 // ... and its index
    // PRED: rows >= nrows (steps beyond the split, the odd last row) get the out-of-range vector offset and read as zero; the main loop's
    // loads are all inside the split and carry no predicate (the scalar offset is NOT relied on for range checking)
    auto load = [&](int d, auto pred) {
        const bool ok = !decltype(pred)::value || 2 * ls + lh < nrows;
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[d][a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra_, ok ? va[a] : BIG, soa, 0));
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[d][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_, ok ? vb[b] : BIG, sob, 0));
        soa += sa2; sob += sb2; ++ls;
    };
    auto mma = [&](int d) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[d][a], fb[d][b], acc[a][b], 0, 0, 0);
    };
    const int nsteps = (nrows + 1) >> 1;
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, Yes{});
    int s = 0;
    for (; ls + D <= nsteps - 1; s += D) {                      // every load of this group targets a full step inside the split
#pragma unroll
        for (int d = 0; d < D; ++d) { mma(d); load(d, No{}); }
    }
    for (; s < nsteps; s += D) {                                // last groups: predicated loads; steps past the end multiply zeros
#pragma unroll
        for (int d = 0; d < D; ++d) { mma(d); load(d, Yes{}); }
    }

    float* C = g.c + ((long)blockIdx.z * gridDim.y + z) * g.k * g.n;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn0 + b * 32 + li;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = k0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (col < g.n && kk < g.k) C[(long)kk * g.n + col] = acc[a][b][r];
            }
    }
}

template <int TM, int TN, int WM, int WN>
void launch_tn_direct(const GemmArgs& g, int batch, hipStream_t st) {
    constexpr int D = (TM + TN) >= 6 ? 6 : ((TM + TN) >= 4 ? 8 : 12);      // 24-36 dword loads in flight per wave (wide tile: <= 256 registers)
    hipLaunchKernelGGL((gemm_tn_direct_kernel<TM, TN, WM, WN, D>), dim3(cdiv(g.k, 32 * TM * WM) * cdiv(g.n, 32 * TN * WN), batch, cdiv(g.rows, g.rps)),
                       dim3(64 * WM * WN), 0, st, g);
}

}  // namespace

int runet_gemm_nn_launch(const float* a, int lda, long sa, const float* b, long sb, float* c, int ldc, long sc, int batch, int rows, int k, int n,
                         hipStream_t st) {
    GemmArgs g{};
    g.a = a; g.lda = lda; g.sa = sa; g.b = b; g.ldb = n; g.sb = sb; g.c = c; g.ldc = ldc; g.sc = sc; g.rows = rows; g.k = k; g.n = n;
    // 16-deep k-steps; the 32-deep instantiation (three 35 KB stages -> one block per CU) measured 5-10 % slower (tools/bench_gemm.py)
    static const bool general = getenv("RUNET_GEMM_NN_GENERAL") && atoi(getenv("RUNET_GEMM_NN_GENERAL")) != 0;      // measurement knob
    if (!general && k % 16 == 0 && k >= 16 && n >= 4 && rows > 0)
        hipLaunchKernelGGL((gemm_nn_kernel<16, true>), dim3(cdiv(rows, 128), cdiv(n, 128), batch), dim3(256), 3 * NNCfg<16>::STAGE * sizeof(float), st, g);
    else
        hipLaunchKernelGGL((gemm_nn_kernel<16, false>), dim3(cdiv(rows, 128), cdiv(n, 128), batch), dim3(256), 3 * NNCfg<16>::STAGE * sizeof(float), st, g);
    return 0;
}

int runet_gemm_tn_launch(const float* a, int lda, long sa, const float* b, int ldb, long sb, float* c, int batch, int rows, int k, int n, int rps,
                         hipStream_t st) {
    GemmArgs g{};
    g.a = a; g.lda = lda; g.sa = sa; g.b = b; g.ldb = ldb; g.sb = sb; g.c = c; g.rows = rows; g.k = k; g.n = n; g.rps = rps;
    // The register-direct form is OPT-IN (RUNET_GEMM_TN_DIRECT=1).  Standalone it is 8-10 % faster than the LDS-ring kernel on the eleven
    // F(4x4) weight-gradient GEMMs of a 16 x 256^2 step (2.53 vs 2.75 ms, 96-117 vs 87-105 TFLOP/s executed) - but these GEMMs run on the
    // weight-gradient side stream, and there the direct form makes the STEP slower (36.55 vs 36.17 ms, two runs each): at 16 FLOP per
    // loaded byte it pulls 4+ TB/s through L2 beside the main chain's streaming kernels, the LDS form (32 FLOP/B) half of that.  A wider
    // wave tile (64 x 128, RUNET_GEMM_TN_WIDE=1) was slower even standalone (272 vs 208 us).
    static const bool lds_form = !(getenv("RUNET_GEMM_TN_DIRECT") && atoi(getenv("RUNET_GEMM_TN_DIRECT")) != 0);
    static const bool wide = getenv("RUNET_GEMM_TN_WIDE") && atoi(getenv("RUNET_GEMM_TN_WIDE")) != 0;
    const long split_rows = rows < rps ? rows : rps;
    // register-direct form: 32-bit byte offsets inside one split, D steps of slack past the end of the offset range
    if (!lds_form && (split_rows + 64) * lda * 4 < (1L << 31) && (split_rows + 64) * ldb * 4 < (1L << 31)) {
        // per dimension: 1 tile (<= 32), 2 tiles in one wave (<= 64), 2 waves of 2 tiles (> 64)
        const int cm = k <= 32 ? 0 : (k <= 64 ? 1 : 2), cn = n <= 32 ? 0 : (n <= 64 ? 1 : 2);
        switch (cm * 3 + cn) {
        case 0: launch_tn_direct<1, 1, 1, 1>(g, batch, st); break;
        case 1: launch_tn_direct<1, 2, 1, 1>(g, batch, st); break;
        case 2: launch_tn_direct<1, 2, 1, 2>(g, batch, st); break;
        case 3: launch_tn_direct<2, 1, 1, 1>(g, batch, st); break;
        case 4: launch_tn_direct<2, 2, 1, 1>(g, batch, st); break;
        case 5: launch_tn_direct<2, 2, 1, 2>(g, batch, st); break;
        case 6: launch_tn_direct<2, 1, 2, 1>(g, batch, st); break;
        case 7: launch_tn_direct<2, 2, 2, 1>(g, batch, st); break;
        default:
            // wide outputs: 64 x 128 per wave (128 accumulators) raises the FLOPs per loaded byte from 16 to 21 - these GEMMs draw
            // 4+ TB/s through L2 with the 64 x 64 wave tile.  RUNET_GEMM_TN_WIDE=0 keeps the square tile (measurement knob).
            if (wide && n >= 256) launch_tn_direct<2, 4, 2, 2>(g, batch, st);
            else if (wide && n > 64 && n <= 128) launch_tn_direct<2, 4, 2, 1>(g, batch, st);
            else launch_tn_direct<2, 2, 2, 2>(g, batch, st);
            break;
        }
        return 0;
    }
    static const bool general = getenv("RUNET_GEMM_TN_GENERAL") && atoi(getenv("RUNET_GEMM_TN_GENERAL")) != 0;      // measurement knob
    if (!general && rows % BK == 0 && rps % BK == 0 && k >= 4 && n >= 4)
        hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3(cdiv(k, 128) * cdiv(n, 128), batch, cdiv(rows, rps)), dim3(256), 3 * TN_STAGE * sizeof(float), st, g);
    else
        hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(cdiv(k, 128) * cdiv(n, 128), batch, cdiv(rows, rps)), dim3(256), 3 * TN_STAGE * sizeof(float), st, g);
    return 0;
}
