// Cycles per MFMA (s_memtime over a back-to-back chain, one wave per SIMD): which bf16 shapes run at the full gfx950 rate.
// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/micro/mfma_rate tools/micro/mfma_rate.hip ; run tools/micro/mfma_rate on the GPU box
// measured (MI355X, ROCm 7.2; rotating over 4 / 16 accumulators): 16x16x16_bf16 (1k) 24.6 / 29.1 | 16x16x32_bf16 24.2 / 28.9 | 32x32x16_bf16 32.8 / 33.6 |
// 16x16x4_f32 33.1 / 33.8 | 32x32x8_bf16 (1k) 33.0 / 33.8 cycles per MFMA: the legacy K = 16 / 8 bf16 forms cost what the K = 32 / 16 forms cost.
// A chain of 32x32x16_bf16 on ONE accumulator (each MFMA's C = the previous one's D): 32.4 - dependent accumulation issues back to back, so the
// six products of X3_MMA need no interleaving across accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int N = 512;
template <int WHICH>
__global__ __launch_bounds__(256) void k(long long* out, float* sink) {
    f32x4 c4[16] = {};
    f32x16 c16[4] = {};
    bf16x8 a8, b8; bf16x4 a4, b4;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(threadIdx.x * 0.01f + j); b8[j] = (__bf16)(j - threadIdx.x * 0.02f); }
    for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
    s16x4 sa = __builtin_bit_cast(s16x4, a4), sb = __builtin_bit_cast(s16x4, b4);
    float fa = threadIdx.x * 0.5f, fb = 1.f - threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        if constexpr (WHICH == 0) c4[i & 15] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, c4[i & 15], 0, 0, 0);
        if constexpr (WHICH == 1) c4[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c4[i & 15], 0, 0, 0);
        if constexpr (WHICH == 2) c16[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, c16[i & 3], 0, 0, 0);
        if constexpr (WHICH == 3) c4[i & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c4[i & 15], 0, 0, 0);
        if constexpr (WHICH == 4) c16[i & 3] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(sa, sb, c16[i & 3], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int q = 0; q < 16; ++q) for (int j = 0; j < 4; ++j) s += c4[q][j];
    for (int q = 0; q < 4; ++q) for (int j = 0; j < 16; ++j) s += c16[q][j];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[WHICH] = t1 - t0;
}
int main() {
    long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 256 * 256 * 4);
    hipMemset(out, 0, 64);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, sink);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, sink);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, sink);
        hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, sink);
        hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, sink);
    }
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    const char* nm[] = {"v_mfma_f32_16x16x16_bf16 (1k)", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x8_bf16 (1k)"};
    for (int i = 0; i < 5; ++i) printf("%-32s %.1f cycles per MFMA (one wave per SIMD, back to back, %d in a row)\n", nm[i], (double)h[i] / N, N);
    return 0;
}
