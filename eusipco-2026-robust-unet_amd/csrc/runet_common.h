// Shared helpers for the gfx950 Robust U-Net kernels (internal; the public C ABI is include/runet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RUNET_OK 0
#define RUNET_EINVAL 1
#define RUNET_ELAUNCH 2

extern "C" void runet_set_error(const char* msg);

#define RUNET_REQUIRE(cond, msg)                                          \
    do {                                                                  \
        if (!(cond)) {                                                    \
            char _b[512];                                                 \
            snprintf(_b, sizeof(_b), "%s: %s (%s)", __func__, msg, #cond); \
            runet_set_error(_b);                                          \
            return RUNET_EINVAL;                                          \
        }                                                                 \
    } while (0)

#define RUNET_CHECK_LAUNCH()                                                   \
    do {                                                                       \
        hipError_t _e = hipGetLastError();                                     \
        if (_e != hipSuccess) {                                                \
            char _b[512];                                                      \
            snprintf(_b, sizeof(_b), "%s: launch failed: %s", __func__, hipGetErrorString(_e)); \
            runet_set_error(_b);                                               \
            return RUNET_ELAUNCH;                                              \
        }                                                                      \
        return RUNET_OK;                                                       \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// gemm.hip (internal launchers behind runet_gemm_batched / runet_gemm_tn_batched)
int runet_gemm_nn_launch(const float* a, int lda, long sa, const float* b, long sb, float* c, int ldc, long sc, int batch, int rows, int k, int n,
                         hipStream_t st);
int runet_gemm_tn_launch(const float* a, int lda, long sa, const float* b, int ldb, long sb, float* c, int batch, int rows, int k, int n, int rps,
                         hipStream_t st);

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a full workgroup fence, which on gfx9 drains vmcnt too:
// every global load still in flight (register prefetches meant to overlap the next phase) is waited for at the barrier.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- wave / block reductions (wave = 64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// BatchNorm pieces that several kernels must evaluate BIT-IDENTICALLY (bn_apply_kernel, the backward kernels that recompute the ReLU mask from
// x, the F(4x4) input transforms that take the BatchNorm in their loads): explicit FMAs, so the roundings do not depend on what the compiler
// contracts in each kernel.
__device__ __forceinline__ float bn_pre(const float x, const float scale, const float shift) { return __builtin_fmaf(x, scale, shift); }
// dx = gg * sc + x * ca + cb  with  ca = -sc * k2 * invstd,  cb = sc * (mean * invstd * k2 - k1),  k1 = sum(g) / m,  k2 = sum(g * xhat) / m
__device__ __forceinline__ void bn_bwd_coef(const float sc, const float mean, const float is, const float sum_gx, const float sum_g, const float inv_m,
                                            float& ca, float& cb) {
#pragma clang fp contract(off)
    const float k1 = sum_g * inv_m, k2 = sum_gx * inv_m;
    const float mi = mean * is;
    ca = -sc * k2 * is;
    cb = sc * __builtin_fmaf(mi, k2, -k1);
}
__device__ __forceinline__ float bn_bwd_dx(const float gg, const float sc, const float x, const float ca, const float cb) {
    return __builtin_fmaf(gg, sc, __builtin_fmaf(x, ca, cb));
}
