// Internal pieces shared by the split-operand ("bf16x6") kernels: gemm_split.hip (batched position GEMMs) and conv_x3.hip (1x1 and k2-s2
// transposed convolutions).  See gemm_split.hip for the arithmetic.
#pragma once
#include "runet_common.h"

namespace x3 {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// x = h + m + l exactly (see the header)
__device__ __forceinline__ void split3(const f32x4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 hj = (__bf16)x[j];
        const float r = x[j] - (float)hj;
        const __bf16 mj = (__bf16)r;
        const float r2 = r - (float)mj;
        h[j] = hj; m[j] = mj; l[j] = (__bf16)r2;
    }
}

// contiguous-range-per-XCD remap of the linear workgroup id (bijection for any total)
__device__ __forceinline__ long xcd_remap(long b, long total) {
    const long q = total >> 3, r = total & 7;
    const long xcd = b & 7, idx = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

// six products of one (A tile, B tile) pair, smallest first
#define X3_MMA(ACC, AF, BF)                                                              \
    do {                                                                                 \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[2], BF[0], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[1], BF[1], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[2], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[1], BF[0], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[1], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[0], BF[0], ACC, 0, 0, 0);      \
    } while (0)

// LDS stage (bytes): A image [plane 3][128 rows][16 k] bf16, 32-B rows; the two 16-B k-octets of a row are swapped on rows with bit 3 set
// (ds_read_b128's 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} then hit 16 distinct 16-B slots: conflict-free);
// B image [plane 3][octet 2][BN columns][8 k] bf16 = the packed global layout verbatim.
template <int BN>
struct NNX3 {
    static constexpr int A_BYTES = 3 * 128 * 32, B_BYTES = 3 * 2 * BN * 16, STAGE = A_BYTES + B_BYTES;
    static constexpr int TN = BN / 64;                  // 32-column MFMA tiles per wave
    static constexpr int BITEMS = (3 * 2 * BN + 255) / 256;
};

}  // namespace x3
