// Derived-weight bodies shared by the per-tensor kernels (conv_winograd4.hip, conv_winograd_x3.hip, conv_x3.hip) and the multi-tensor launch
// (derive_multi.hip): every derived weight of the split-operand paths is a pure function of one fp32 weight tensor, thread = (octet of the
// contraction side, column), 256 threads per block, no LDS.  `vb` is the block index WITHIN the tensor's own grid.
#pragma once
#include "runet_common.h"
#include "../../include/runet_hip.h"

namespace derive {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// gm[0..7][r][q] = p[0..7] by two 16-byte loads (p 16-byte aligned)
__device__ __forceinline__ void load8(const float* __restrict__ p, float (&gm)[8][3][3], const int r, const int q) {
    const f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) { gm[j][r][q] = a[j]; gm[4 + j][r][q] = b[j]; }
}

__device__ __forceinline__ void store_split(const float (&u)[8], __bf16* d, const long plane) {
    bf16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = u[j];
        const __bf16 hj = (__bf16)x;
        const float r1 = x - (float)hj;
        const __bf16 mj = (__bf16)r1;
        h[j] = hj; m[j] = mj; l[j] = (__bf16)(r1 - (float)mj);
    }
    *reinterpret_cast<bf16x8*>(d) = h;
    *reinterpret_cast<bf16x8*>(d + plane) = m;
    *reinterpret_cast<bf16x8*>(d + 2 * plane) = l;
}

// F(4x4,3x3): Up[36][plane 3][K/8][N][8] = split(G g G^T).
// dgrad: 0 forward (K = cin);  1 rotated filter for the data gradient computed as a convolution (K = cout);  2 the forward's U TRANSPOSED over
// (k, n) and not rotated, for the data gradient computed as the ADJOINT of the forward algorithm (K = cout; wino4_output_adj_kernel)
__device__ __forceinline__ void wino4_weight_x3_body(const float* __restrict__ w, __bf16* __restrict__ Up, int cin, int cout, int dgrad, long vb) {
    const int K = dgrad ? cout : cin, N = dgrad ? cin : cout;
    const int K8 = K >> 3;
    const long per = (long)K8 * N;
    const long i = vb * 256 + threadIdx.x;
    if (i >= per) return;
    const int oc = (int)(i / N), n = (int)(i - (long)oc * N);
    float gm[8][3][3];
    if (dgrad && aligned16(w)) {
        // transposed modes: the thread's eight k are contiguous in w (k = output channel, innermost) - two 16-byte loads per tap instead of
        // eight 4-byte loads that each touch 64 different 32-byte sectors per wave
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int tap = dgrad == 1 ? (2 - r) * 3 + (2 - q) : r * 3 + q;
                load8(w + ((long)tap * cin + n) * cout + oc * 8, gm, r, q);
            }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = oc * 8 + j;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    gm[j][r][q] = dgrad == 1 ? w[((long)((2 - r) * 3 + (2 - q)) * cin + n) * cout + k]
                                : dgrad == 2 ? w[((long)(r * 3 + q) * cin + n) * cout + k] : w[((long)(r * 3 + q) * cin + k) * cout + n];
        }
    }
    auto grow = [](const float g0, const float g1, const float g2, const int a) -> float {      // row a of G applied to (g0, g1, g2)
        const float e = (g0 + g2) * (1.f / 6.f), f = g0 * (1.f / 24.f) + g2 * (1.f / 6.f);
        switch (a) {
        case 0: return 0.25f * g0;
        case 1: return -e - g1 * (1.f / 6.f);
        case 2: return -e + g1 * (1.f / 6.f);
        case 3: return f + g1 * (1.f / 12.f);
        case 4: return f - g1 * (1.f / 12.f);
        default: return g2;
        }
    };
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        float u[6][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t0 = grow(gm[j][0][0], gm[j][1][0], gm[j][2][0], a), t1 = grow(gm[j][0][1], gm[j][1][1], gm[j][2][1], a),
                        t2 = grow(gm[j][0][2], gm[j][1][2], gm[j][2][2], a);
#pragma unroll
            for (int b = 0; b < 6; ++b) u[b][j] = grow(t0, t1, t2, b);
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) store_split(u[b], Up + (long)(a * 6 + b) * 3 * per * 8 + i * 8, per * 8);
    }
}

// F(2x2,3x3): Up[16][plane 3][K/8][N][8] = split(G g G^T);  forward: g[r][s] = w[r][s][k][n];  dgrad: g[r][s] = w[2-r][2-s][n][k]
__device__ __forceinline__ void wino_weight_x3_body(const float* __restrict__ w, __bf16* __restrict__ Up, int cin, int cout, int dgrad, long vb) {
    const int K = dgrad ? cout : cin, N = dgrad ? cin : cout;
    const int K8 = K >> 3;
    const long per = (long)K8 * N;
    const long i = vb * 256 + threadIdx.x;
    if (i >= per) return;
    const int oc = (int)(i / N), n = (int)(i - (long)oc * N);
    float gm[8][3][3];
    if (dgrad && aligned16(w)) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) load8(w + ((long)((2 - r) * 3 + (2 - q)) * cin + n) * cout + oc * 8, gm, r, q);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = oc * 8 + j;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    gm[j][r][q] = dgrad ? w[((long)((2 - r) * 3 + (2 - q)) * cin + n) * cout + k] : w[((long)(r * 3 + q) * cin + k) * cout + n];
        }
    }
    auto grow = [](const float g0, const float g1, const float g2, const int a) -> float {      // row a of G = [1,0,0; .5,.5,.5; .5,-.5,.5; 0,0,1]
        switch (a) {
        case 0: return g0;
        case 1: return 0.5f * (g0 + g1 + g2);
        case 2: return 0.5f * (g0 - g1 + g2);
        default: return g2;
        }
    };
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float u[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t0 = grow(gm[j][0][0], gm[j][1][0], gm[j][2][0], a), t1 = grow(gm[j][0][1], gm[j][1][1], gm[j][2][1], a),
                        t2 = grow(gm[j][0][2], gm[j][1][2], gm[j][2][2], a);
#pragma unroll
            for (int b = 0; b < 4; ++b) u[b][j] = grow(t0, t1, t2, b);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) store_split(u[b], Up + (long)(a * 4 + b) * 3 * per * 8 + i * 8, per * 8);
    }
}

// w (fp32) -> split planes dst[z][plane 3][k/8][n][8] bf16 with B_z[kk][col] = w[z * stride_z + kk * sk + col * sn]; thread = (z, octet, column)
__device__ __forceinline__ void pack_x3_body(const float* __restrict__ w, long stride_z, long sk, long sn, __bf16* __restrict__ dst, int batch, int k,
                                             int n, long vb) {
    const int K8 = k >> 3;
    const long per = (long)K8 * n, total = per * batch;
    const long i = vb * 256 + threadIdx.x;
    if (i >= total) return;
    const int z = (int)(i / per);
    const long r = i - (long)z * per;
    const int oc = (int)(r / n), col = (int)(r - (long)oc * n);
    const float* s = w + (long)z * stride_z + (long)oc * 8 * sk + (long)col * sn;
    float u[8];
    if (sk == 1 && aligned16(s)) {               // transposed modes: eight contiguous floats
        const f32x4 a = reinterpret_cast<const f32x4*>(s)[0], b = reinterpret_cast<const f32x4*>(s)[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { u[j] = a[j]; u[4 + j] = b[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = s[(long)j * sk];
    }
    store_split(u, dst + (long)z * 3 * per * 8 + r * 8, per * 8);
}

// strides of the packed weight of runet_conv_x3 in `mode` (conv_x3.hip): taps, stride_z, sk, sn for contraction cin / output cout
__host__ __device__ inline void conv_x3_pack_strides(int cin, int cout, int mode, int& taps, long& stride_z, long& sk, long& sn) {
    const bool tr = mode == RUNET_CONV_DGRAD || mode == RUNET_CONVT_DGRAD;
    taps = (mode == RUNET_CONVT_FWD || mode == RUNET_CONVT_DGRAD) ? 4 : 1;
    stride_z = (long)cin * cout;
    // forward: B[k][n] = w[k][n] (row stride cout);  data gradient: B[k = Co][n = Ci] = w[n][k] (row stride = cin of this mode)
    sk = tr ? 1L : (long)cout;
    sn = tr ? (long)cin : 1L;
}

}  // namespace derive
