// MaxPool2d(2), input layout conversion, BCE loss, fused multi-tensor Adam and segmentation counts.
//
// Reference call sites (/root/reference/Main_Final.py): nn.MaxPool2d(2) :235,239,243,249; nn.BCELoss() :551,580;
// torch.optim.Adam(lr, weight_decay=1e-4) :552,582 (L2-coupled decay, bias-corrected); ModelEvaluator.calculate_metrics
// :519-547 (thresholded tp / predicted-positive / target-positive counts; the float64 ratios stay on the host).
// All HBM-bound: 16-byte accesses per lane, one pass.
#include "runet_common.h"
#include "../../include/runet_hip.h"

namespace {
constexpr int TPB = 256;

inline int ew_grid(long total) {
    long b = (total + TPB - 1) / TPB;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- 2x2 max pool; idx byte = position (0..3) of the first maximum in scan order (ATen tie rule: strict >, NaN wins)
__global__ __launch_bounds__(TPB) void maxpool2_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                           unsigned char* __restrict__ idx, int N, int Ho, int Wo, int C) {
    const int cvec = C / 4;
    const long total = (long)N * Ho * Wo * cvec;
    const int W = Wo * 2;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const long op = i / cvec;
        const int c = (int)(i - op * cvec) * 4;
        const int wo = (int)(op % Wo);
        const long t = op / Wo;
        const int ho = (int)(t % Ho);
        const long n = t / Ho;
        const long ip = (n * (Ho * 2) + ho * 2) * W + wo * 2;
        f32x4 m = *reinterpret_cast<const f32x4*>(x + ip * ldx + c);
        unsigned int sel[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const long q = ip + (k >> 1) * W + (k & 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + q * ldx + c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] > m[e] || v[e] != v[e]) { m[e] = v[e]; sel[e] = k; }
        }
        *reinterpret_cast<f32x4*>(y + op * ldy + c) = m;
        *reinterpret_cast<unsigned int*>(idx + op * C + c) = sel[0] | (sel[1] << 8) | (sel[2] << 16) | (sel[3] << 24);
    }
}
__global__ __launch_bounds__(TPB) void maxpool2_bwd_kernel(const float* __restrict__ dy, int lddy, const unsigned char* __restrict__ idx,
                                                           float* __restrict__ dx, int lddx, int N, int Ho, int Wo, int C, int accumulate) {
    const int cvec = C / 4;
    const long total = (long)N * Ho * Wo * cvec;
    const int W = Wo * 2;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const long op = i / cvec;
        const int c = (int)(i - op * cvec) * 4;
        const int wo = (int)(op % Wo);
        const long t = op / Wo;
        const int ho = (int)(t % Ho);
        const long n = t / Ho;
        const long ip = (n * (Ho * 2) + ho * 2) * W + wo * 2;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + op * lddy + c);
        const unsigned int s = *reinterpret_cast<const unsigned int*>(idx + op * C + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long q = ip + (k >> 1) * W + (k & 1);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (((s >> (8 * e)) & 0xff) == (unsigned)k) ? g[e] : 0.f;
            float* d = dx + q * lddx + c;
            if (accumulate) o += *reinterpret_cast<const f32x4*>(d);
            *reinterpret_cast<f32x4*>(d) = o;
        }
    }
}

// ---- strided [N,C,H,W] (any strides) -> NHWC with channels zero-padded to Cp
__global__ __launch_bounds__(TPB) void to_nhwc_pad_kernel(const float* __restrict__ x, long sn, long sc, long sh, long sw,
                                                          float* __restrict__ y, int N, int C, int H, int W, int Cp) {
    const long total = (long)N * H * W;
    for (long p = (long)blockIdx.x * TPB + threadIdx.x; p < total; p += (long)gridDim.x * TPB) {
        const int w = (int)(p % W);
        const long t = p / W;
        const int h = (int)(t % H);
        const long n = t / H;
        const float* src = x + n * sn + h * sh + w * sw;
        for (int c = 0; c < Cp; ++c) y[p * Cp + c] = c < C ? src[c * sc] : 0.f;
    }
}

// ---- BCE (mean) on probabilities, ATen semantics: logs clamped at -100
__global__ __launch_bounds__(TPB) void bce_partial_kernel(const float* __restrict__ p, const float* __restrict__ y, long n,
                                                          double* __restrict__ part) {
    double acc = 0;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        const float pi = p[i], yi = y[i];
        const float lp = fmaxf(logf(pi), -100.f), l1 = fmaxf(logf(1.f - pi), -100.f);
        acc -= (double)(yi * lp + (1.f - yi) * l1);
    }
    acc = wave_sum_d(acc);
    __shared__ double red[TPB / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void bce_final_kernel(const double* __restrict__ part, int nparts, long n, float* __restrict__ loss) {
    double acc = 0;
    for (int i = threadIdx.x; i < nparts; i += 64) acc += part[i];
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) loss[0] = (float)(acc / (double)n);
}
__global__ __launch_bounds__(TPB) void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                      const float* __restrict__ gout, float* __restrict__ dp, long n) {
    const float g = (gout ? gout[0] : 1.f) / (float)n;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        const float pi = p[i];
        dp[i] = g * (pi - y[i]) / fmaxf(pi * (1.f - pi), 1e-12f);
    }
}

// ---- multi-tensor Adam.  table: int64 [5][T] = p, g, m, v pointers and element counts; chunks: int32 [B][2] = (tensor, chunk#)
constexpr int ADAM_CHUNK = 16384;
__global__ __launch_bounds__(TPB) void adam_multi_kernel(const long long* __restrict__ table, int T, const int* __restrict__ chunks,
                                                         float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                         float bc2_sqrt, float grad_scale, const int* __restrict__ skip) {
    if (skip && *skip) return;      // loss-scaled step whose gradients overflowed: parameters and moments stay as they are
    const int t = chunks[blockIdx.x * 2], ck = chunks[blockIdx.x * 2 + 1];
    float* p = reinterpret_cast<float*>(table[t]);
    const float* g = reinterpret_cast<const float*>(table[T + t]);
    float* m = reinterpret_cast<float*>(table[2 * T + t]);
    float* v = reinterpret_cast<float*>(table[3 * T + t]);
    const long n = table[4 * T + t];
    const long beg = (long)ck * ADAM_CHUNK;
    const long end = beg + ADAM_CHUNK < n ? beg + ADAM_CHUNK : n;
    const float step = lr / bc1;
    auto upd = [&](float& pv, float gv, float& mv, float& vv) {
        gv = gv * grad_scale + wd * pv;
        mv = beta1 * mv + (1.f - beta1) * gv;
        vv = beta2 * vv + (1.f - beta2) * gv * gv;
        pv -= step * mv / (sqrtf(vv) / bc2_sqrt + eps);
    };
    const bool al = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    if (al) {
        const long nv = (end - beg) / 4;
        for (long i = threadIdx.x; i < nv; i += TPB) {
            const long o = beg + i * 4;
            f32x4 pv = *reinterpret_cast<f32x4*>(p + o), mv = *reinterpret_cast<f32x4*>(m + o), vv = *reinterpret_cast<f32x4*>(v + o);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { float a = pv[e], b = mv[e], c = vv[e]; upd(a, gv[e], b, c); pv[e] = a; mv[e] = b; vv[e] = c; }
            *reinterpret_cast<f32x4*>(p + o) = pv; *reinterpret_cast<f32x4*>(m + o) = mv; *reinterpret_cast<f32x4*>(v + o) = vv;
        }
        for (long o = beg + nv * 4 + threadIdx.x; o < end; o += TPB) upd(p[o], g[o], m[o], v[o]);
    } else {
        for (long o = beg + threadIdx.x; o < end; o += TPB) upd(p[o], g[o], m[o], v[o]);
    }
}

// ---- per-image counts: [n][0]=tp, [1]=predicted positives, [2]=target positives, [3]=pixels where (pred>thr) == (target!=0)
__global__ __launch_bounds__(TPB) void seg_counts_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                         long long* __restrict__ counts, long per_img, float thr) {
    const int n = blockIdx.y;
    const float* p = pred + (long)n * per_img;
    const float* t = target + (long)n * per_img;
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < per_img; i += (long)gridDim.x * TPB) {
        const bool pb = p[i] > thr, tb = t[i] != 0.f;
        c0 += pb && tb; c1 += pb; c2 += tb; c3 += pb == tb;
    }
    auto wred = [](long long v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    c0 = wred(c0); c1 = wred(c1); c2 = wred(c2); c3 = wred(c3);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(counts + n * 4 + 0), (unsigned long long)c0);
        atomicAdd(reinterpret_cast<unsigned long long*>(counts + n * 4 + 1), (unsigned long long)c1);
        atomicAdd(reinterpret_cast<unsigned long long*>(counts + n * 4 + 2), (unsigned long long)c2);
        atomicAdd(reinterpret_cast<unsigned long long*>(counts + n * 4 + 3), (unsigned long long)c3);
    }
}
}  // namespace

extern "C" int runet_maxpool2_fwd(const float* x, int ldx, float* y, int ldy, unsigned char* idx, int n_img, int h, int w, int c, void* stream) {
    RUNET_REQUIRE(x && y && idx, "null pointer");
    RUNET_REQUIRE(h % 2 == 0 && w % 2 == 0 && c % 4 == 0 && c > 0, "h, w must be even and c a multiple of 4");
    const long total = (long)n_img * (h / 2) * (w / 2) * (c / 4);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid(total)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, y, ldy, idx, n_img, h / 2, w / 2, c);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_maxpool2_bwd(const float* dy, int lddy, const unsigned char* idx, float* dx, int lddx, int n_img, int h, int w, int c,
                                  int accumulate, void* stream) {
    RUNET_REQUIRE(dy && dx && idx, "null pointer");
    RUNET_REQUIRE(h % 2 == 0 && w % 2 == 0 && c % 4 == 0 && c > 0, "h, w must be even and c a multiple of 4");
    const long total = (long)n_img * (h / 2) * (w / 2) * (c / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total)), dim3(TPB), 0, (hipStream_t)stream, dy, lddy, idx, dx, lddx, n_img, h / 2, w / 2, c, accumulate);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_to_nhwc_pad(const float* x, long sn, long sc, long sh, long sw, float* y, int n_img, int c, int h, int w, int c_pad, void* stream) {
    RUNET_REQUIRE(x && y && c > 0 && c_pad >= c, "bad arguments");
    hipLaunchKernelGGL(to_nhwc_pad_kernel, dim3(ew_grid((long)n_img * h * w)), dim3(TPB), 0, (hipStream_t)stream, x, sn, sc, sh, sw, y, n_img, c, h, w, c_pad);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_bce_fwd(const float* prob, const float* target, long n, double* workspace1024, float* loss, void* stream) {
    RUNET_REQUIRE(prob && target && workspace1024 && loss && n > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    long b = (n + TPB * 8 - 1) / (TPB * 8);
    if (b > 1024) b = 1024;
    hipLaunchKernelGGL(bce_partial_kernel, dim3((int)b), dim3(TPB), 0, st, prob, target, n, workspace1024);
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(64), 0, st, workspace1024, (int)b, n, loss);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_bce_bwd(const float* prob, const float* target, const float* grad_out, float* dprob, long n, void* stream) {
    RUNET_REQUIRE(prob && target && dprob && n > 0, "bad arguments");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(ew_grid(n)), dim3(TPB), 0, (hipStream_t)stream, prob, target, grad_out, dprob, n);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_adam_chunk_elems(void) { return ADAM_CHUNK; }

// ---- graph-capturable form: hyper-parameters and the step counter live in device memory, so a captured launch stays valid when
// the learning rate changes or the step advances.  hyper = {lr, beta1, beta2, eps, weight_decay, grad_scale}.
namespace {
__global__ void adam_tick_kernel(int* step, const int* skip) { if (!(skip && *skip)) *step += 1; }
__global__ __launch_bounds__(TPB) void adam_multi_dev_kernel(const long long* __restrict__ table, int T, const int* __restrict__ chunks,
                                                             const float* __restrict__ hyper, const int* __restrict__ step_dev,
                                                             const int* __restrict__ skip) {
    if (skip && *skip) return;
    const int t = chunks[blockIdx.x * 2], ck = chunks[blockIdx.x * 2 + 1];
    float* p = reinterpret_cast<float*>(table[t]);
    const float* g = reinterpret_cast<const float*>(table[T + t]);
    float* m = reinterpret_cast<float*>(table[2 * T + t]);
    float* v = reinterpret_cast<float*>(table[3 * T + t]);
    const long n = table[4 * T + t];
    const long beg = (long)ck * ADAM_CHUNK;
    const long end = beg + ADAM_CHUNK < n ? beg + ADAM_CHUNK : n;
    const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], wd = hyper[4], grad_scale = hyper[5];
    const int st = *step_dev;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)st));           // same double-precision corrections as the host form
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)st));
    const float step = lr / bc1;
    for (long o = beg + threadIdx.x; o < end; o += TPB) {
        float pv = p[o], mv = m[o], vv = v[o];
        const float gv = g[o] * grad_scale + wd * pv;
        mv = beta1 * mv + (1.f - beta1) * gv;
        vv = beta2 * vv + (1.f - beta2) * gv * gv;
        pv -= step * mv / (sqrtf(vv) / bc2_sqrt + eps);
        p[o] = pv; m[o] = mv; v[o] = vv;
    }
}
}  // namespace

extern "C" int runet_adam_multi_dev(const long long* table, int n_tensors, const int* chunks, int n_chunks, const float* hyper, int* step_dev,
                                    const int* skip_flag, void* stream) {
    RUNET_REQUIRE(table && chunks && hyper && step_dev && n_tensors > 0 && n_chunks > 0, "bad arguments");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, skip_flag);
    hipLaunchKernelGGL(adam_multi_dev_kernel, dim3(n_chunks), dim3(TPB), 0, (hipStream_t)stream, table, n_tensors, chunks, hyper, step_dev, skip_flag);
    RUNET_CHECK_LAUNCH();
}

// ---- loss scaling (fp16 operands): flag[0] = 1 if any element of buf is Inf / NaN (flag must be zeroed by the caller), flag[1] += flag[0]
namespace {
__global__ __launch_bounds__(TPB) void nonfinite_kernel(const float* __restrict__ buf, long n, int* __restrict__ flag) {
    bool bad = false;
    const long nv = n / 4;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nv; i += (long)gridDim.x * TPB) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(buf + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= !(fabsf(v[e]) <= 3.4e38f);
    }
    if (blockIdx.x == 0)
        for (long i = nv * 4 + threadIdx.x; i < n; i += TPB) bad |= !(fabsf(buf[i]) <= 3.4e38f);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ void nonfinite_count_kernel(int* flag) { flag[1] += flag[0]; }
}  // namespace

// `waiter` will not run anything enqueued after this call before everything enqueued on `waited` so far has finished
// (hipEventRecord + hipStreamWaitEvent on a pooled event): the fork / join of the weight-gradient stream without going through
// torch.cuda's Python stream objects (~25 us of host time per fork there, 50 forks per step).
extern "C" int runet_stream_wait(void* waiter, void* waited) {
    constexpr int POOL = 64;
    static thread_local hipEvent_t pool[POOL];
    static thread_local int next = -1;
    if (next < 0) {
        for (int i = 0; i < POOL; ++i)
            if (hipEventCreateWithFlags(&pool[i], hipEventDisableTiming) != hipSuccess) { runet_set_error("runet_stream_wait: hipEventCreate failed"); return RUNET_ELAUNCH; }
        next = 0;
    }
    hipEvent_t ev = pool[next];
    next = (next + 1) % POOL;
    if (hipEventRecord(ev, (hipStream_t)waited) != hipSuccess || hipStreamWaitEvent((hipStream_t)waiter, ev, 0) != hipSuccess) {
        runet_set_error("runet_stream_wait: record / wait failed");
        return RUNET_ELAUNCH;
    }
    return RUNET_OK;
}

extern "C" int runet_nonfinite_flag(const float* buf, long n, int* flag2, void* stream) {
    RUNET_REQUIRE(buf && flag2 && n > 0 && ((uintptr_t)buf % 16) == 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(flag2, 0, sizeof(int), st) != hipSuccess) { runet_set_error("runet_nonfinite_flag: memset failed"); return RUNET_ELAUNCH; }
    hipLaunchKernelGGL(nonfinite_kernel, dim3(ew_grid(n / 4 + 1)), dim3(TPB), 0, st, buf, n, flag2);
    hipLaunchKernelGGL(nonfinite_count_kernel, dim3(1), dim3(1), 0, st, flag2);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_adam_multi(const long long* table, int n_tensors, const int* chunks, int n_chunks, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int step, float grad_scale, const int* skip_flag, void* stream) {
    RUNET_REQUIRE(table && chunks && n_tensors > 0 && n_chunks > 0 && step >= 1, "bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(TPB), 0, (hipStream_t)stream, table, n_tensors, chunks, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, skip_flag);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_seg_counts(const float* pred, const float* target, long long* counts, int n_img, long per_img, float threshold, void* stream) {
    RUNET_REQUIRE(pred && target && counts && n_img > 0 && per_img > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(long long) * 4 * n_img, st) != hipSuccess) { runet_set_error("runet_seg_counts: memset failed"); return RUNET_ELAUNCH; }
    long b = (per_img + TPB * 4 - 1) / (TPB * 4);
    if (b > 256) b = 256;
    hipLaunchKernelGGL(seg_counts_kernel, dim3((int)b, n_img), dim3(TPB), 0, st, pred, target, counts, per_img, threshold);
    RUNET_CHECK_LAUNCH();
}

// ============================================================================================================================
// DeepLabV3+ baseline extras (/root/reference/Main_Final.py:325-433): MaxPool2d(3, stride 2, padding 1), the ASPP image-pooling
// branch (global average -> 1x1 conv -> bilinear upsample of a 1x1 map = broadcast) and the final Conv2d(16, 1, 3, padding 1) + sigmoid.
namespace {
// ---- MaxPool2d(3, s2, p1): idx byte = window position (0..8) of the first maximum in scan order
__global__ __launch_bounds__(TPB) void maxpool3s2_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                             unsigned char* __restrict__ idx, int N, int H, int W, int Ho, int Wo, int C) {
    const int cvec = C / 4;
    const long total = (long)N * Ho * Wo * cvec;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const long op = i / cvec;
        const int c = (int)(i - op * cvec) * 4;
        const int wo = (int)(op % Wo);
        const long t = op / Wo;
        const int ho = (int)(t % Ho);
        const long n = t / Ho;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned int sel[4] = {0, 0, 0, 0};
        bool first = true;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int ih = 2 * ho - 1 + k / 3, iw = 2 * wo - 1 + k % 3;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((n * H + ih) * W + iw) * ldx + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (first || v[e] > m[e] || v[e] != v[e]) { m[e] = v[e]; sel[e] = k; }
                first = false;
            }
        }
        *reinterpret_cast<f32x4*>(y + op * ldy + c) = m;
        *reinterpret_cast<unsigned int*>(idx + op * C + c) = sel[0] | (sel[1] << 8) | (sel[2] << 16) | (sel[3] << 24);
    }
}
// gather form: input pixel (ih, iw) belongs to the windows ho in {(ih+1)/2 - (0|1)} ...
__global__ __launch_bounds__(TPB) void maxpool3s2_bwd_kernel(const float* __restrict__ dy, int lddy, const unsigned char* __restrict__ idx,
                                                             float* __restrict__ dx, int lddx, int N, int H, int W, int Ho, int Wo, int C) {
    const int cvec = C / 4;
    const long total = (long)N * H * W * cvec;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const long ip = i / cvec;
        const int c = (int)(i - ip * cvec) * 4;
        const int iw = (int)(ip % W);
        const long t = ip / W;
        const int ih = (int)(t % H);
        const long n = t / H;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int ho = (ih + 1) / 2 - 1 + ((ih + 1) & 1 ? 0 : 0); ho <= (ih + 1) / 2; ++ho) {
            if (ho < 0 || ho >= Ho) continue;
            const int kr = ih - (2 * ho - 1);
            if (kr < 0 || kr > 2) continue;
            for (int wo = (iw + 1) / 2 - 1; wo <= (iw + 1) / 2; ++wo) {
                if (wo < 0 || wo >= Wo) continue;
                const int kc = iw - (2 * wo - 1);
                if (kc < 0 || kc > 2) continue;
                const long op = (n * Ho + ho) * Wo + wo;
                const unsigned int s = *reinterpret_cast<const unsigned int*>(idx + op * C + c);
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + op * lddy + c);
                const unsigned k = kr * 3 + kc;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((s >> (8 * e)) & 0xff) == k) g[e] += d[e];
            }
        }
        *reinterpret_cast<f32x4*>(dx + ip * lddx + c) = g;
    }
}
// y[n, p, c] = v[n, c]
__global__ __launch_bounds__(TPB) void broadcast_nc_kernel(const float* __restrict__ v, float* __restrict__ y, int ldy, int HW, int C, long total) {
    const int cvec = C / 4;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const long p = i / cvec;
        const int c = (int)(i - p * cvec) * 4;
        const long n = p / HW;
        *reinterpret_cast<f32x4*>(y + p * ldy + c) = *reinterpret_cast<const f32x4*>(v + n * C + c);
    }
}
// final head: logit[p] = b + sum_{tap,c} x[p+tap][c] * w[tap][c];  prob = sigmoid
__global__ __launch_bounds__(TPB) void head3x3_fwd_kernel(const float* __restrict__ x, int ld, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ prob, int N, int H, int W, int C) {
    extern __shared__ float ws[];
    for (int i = threadIdx.x; i < 9 * C; i += TPB) ws[i] = w[i];
    __syncthreads();
    const long total = (long)N * H * W;
    for (long p = (long)blockIdx.x * TPB + threadIdx.x; p < total; p += (long)gridDim.x * TPB) {
        const int wq = (int)(p % W);
        const long t = p / W;
        const int h = (int)(t % H);
        const long n = t / H;
        float acc = b[0];
        for (int k = 0; k < 9; ++k) {
            const int ih = h - 1 + k / 3, iw = wq - 1 + k % 3;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                const float* xp = x + ((n * H + ih) * W + iw) * ld;
                for (int c = 0; c < C; c += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xp + c);
                    acc += v[0] * ws[k * C + c] + v[1] * ws[k * C + c + 1] + v[2] * ws[k * C + c + 2] + v[3] * ws[k * C + c + 3];
                }
            }
        }
        prob[p] = sigmoidf_(acc);
    }
}
// dl = dprob*p*(1-p);  dx[p][c] = sum_tap dl[p - tap] * w[tap][c];  partial dw[tap][c] = sum_p x[p+tap][c]*dl[p], db = sum dl
// The weight / bias sums are reduced in a FIXED order (butterfly over the wave, one private LDS row per wave, rows added in wave order,
// blocks summed by sum_rows_kernel): an earlier version used LDS float atomics and two runs of the same step differed in the last bit of
// decoder.12.weight (found by tests/test_gpu_deeplab.py::test_deeplab_at_the_benchmarked_size).
__global__ __launch_bounds__(TPB) void head3x3_bwd_kernel(const float* __restrict__ dprob, const float* __restrict__ prob,
                                                          const float* __restrict__ x, int ld, const float* __restrict__ w,
                                                          float* __restrict__ dx, int lddx, float* __restrict__ part, int N, int H, int W, int C) {
    extern __shared__ float sm[];
    constexpr int NW = TPB / 64;
    const int R = 9 * C + 1;
    float* ws = sm;                    // [9*C]
    float* red = sm + 9 * C;           // [NW][9*C + 1] per-wave partials
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 9 * C; i += TPB) ws[i] = w[i];
    for (int i = threadIdx.x; i < NW * R; i += TPB) red[i] = 0.f;
    __syncthreads();
    float* mine = red + wave * R;
    const long total = (long)N * H * W;
    for (long base = (long)blockIdx.x * TPB; base < total; base += (long)gridDim.x * TPB) {      // uniform trip count: every lane joins the wave sums
        const long p = base + threadIdx.x;
        const bool valid = p < total;
        const long pc = valid ? p : total - 1;
        const int wq = (int)(pc % W);
        const long t = pc / W;
        const int h = (int)(t % H);
        const long n = t / H;
        const float pr = prob[pc];
        const float dl = valid ? dprob[pc] * pr * (1.f - pr) : 0.f;
        const float sdl = wave_sum(dl);
        if (lane == 0) mine[9 * C] += sdl;
        if (valid) {
            for (int c = 0; c < C; ++c) {
                float g = 0.f;
                for (int k = 0; k < 9; ++k) {
                    const int oh = h + 1 - k / 3, ow = wq + 1 - k % 3;      // output pixel whose tap k reads this input pixel
                    if ((unsigned)oh < (unsigned)H && (unsigned)ow < (unsigned)W) {
                        const long q = (n * H + oh) * W + ow;
                        const float pq = prob[q];
                        g += dprob[q] * pq * (1.f - pq) * ws[k * C + c];
                    }
                }
                dx[p * lddx + c] = g;
            }
        }
        for (int k = 0; k < 9; ++k) {
            const int ih = h - 1 + k / 3, iw = wq - 1 + k % 3;
            const bool in = valid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            const float* xp = x + ((n * H + (in ? ih : h)) * W + (in ? iw : wq)) * ld;
            for (int c = 0; c < C; ++c) {
                const float s = wave_sum(in ? xp[c] * dl : 0.f);
                if (lane == 0) mine[k * C + c] += s;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += TPB) {
        float a = red[i];
        for (int v = 1; v < NW; ++v) a += red[v * R + i];
        part[(long)blockIdx.x * R + i] = a;
    }
}
__global__ void sum_rows_kernel(const float* __restrict__ part, int nrows, int ncols, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    double a = 0;
    for (int r = 0; r < nrows; ++r) a += part[(long)r * ncols + c];
    out[c] = (float)a;
}
}  // namespace

extern "C" int runet_maxpool3s2_fwd(const float* x, int ldx, float* y, int ldy, unsigned char* idx, int n_img, int h, int w, int c, void* stream) {
    RUNET_REQUIRE(x && y && idx && c % 4 == 0 && c > 0 && h > 0 && w > 0, "bad arguments");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(ew_grid((long)n_img * ho * wo * (c / 4))), dim3(TPB), 0, (hipStream_t)stream, x, ldx, y, ldy, idx, n_img, h, w, ho, wo, c);
    RUNET_CHECK_LAUNCH();
}
extern "C" int runet_maxpool3s2_bwd(const float* dy, int lddy, const unsigned char* idx, float* dx, int lddx, int n_img, int h, int w, int c, void* stream) {
    RUNET_REQUIRE(dy && dx && idx && c % 4 == 0 && c > 0, "bad arguments");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(ew_grid((long)n_img * h * w * (c / 4))), dim3(TPB), 0, (hipStream_t)stream, dy, lddy, idx, dx, lddx, n_img, h, w, ho, wo, c);
    RUNET_CHECK_LAUNCH();
}
extern "C" int runet_broadcast_nc(const float* v_nc, float* y, int ldy, int n_img, int hw, int c, void* stream) {
    RUNET_REQUIRE(v_nc && y && c % 4 == 0 && c > 0, "bad arguments");
    const long total = (long)n_img * hw * (c / 4);
    hipLaunchKernelGGL(broadcast_nc_kernel, dim3(ew_grid(total)), dim3(TPB), 0, (hipStream_t)stream, v_nc, y, ldy, hw, c, total);
    RUNET_CHECK_LAUNCH();
}
extern "C" int runet_head3x3_fwd(const float* x, int ld, const float* w, const float* b, float* prob, int n_img, int h, int w_, int c, void* stream) {
    RUNET_REQUIRE(x && w && b && prob && c % 4 == 0 && c > 0 && c <= 64, "bad arguments (c multiple of 4, <= 64)");
    hipLaunchKernelGGL(head3x3_fwd_kernel, dim3(ew_grid((long)n_img * h * w_)), dim3(TPB), 9 * c * sizeof(float), (hipStream_t)stream, x, ld, w, b, prob, n_img, h, w_, c);
    RUNET_CHECK_LAUNCH();
}
extern "C" long runet_head3x3_bwd_workspace_floats(int n_img, int h, int w_, int c) { return 1024L * (9 * c + 1); }
extern "C" int runet_head3x3_bwd(const float* dprob, const float* prob, const float* x, int ld, const float* w, float* dx, int lddx, float* workspace,
                                 float* dw_db, int n_img, int h, int w_, int c, void* stream) {
    RUNET_REQUIRE(dprob && prob && x && w && dx && workspace && dw_db && c % 4 == 0 && c > 0 && c <= 64, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    long blocks = ((long)n_img * h * w_ + TPB - 1) / TPB;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(head3x3_bwd_kernel, dim3((int)blocks), dim3(TPB), (9 * c + (TPB / 64) * (9 * c + 1)) * sizeof(float), st, dprob, prob, x, ld, w, dx, lddx, workspace, n_img, h, w_, c);
    hipLaunchKernelGGL(sum_rows_kernel, dim3(cdiv(9 * c + 1, 128)), dim3(128), 0, st, workspace, (int)blocks, 9 * c + 1, dw_db);
    RUNET_CHECK_LAUNCH();
}

namespace {
__global__ __launch_bounds__(TPB) void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
    const long nv = n / 4;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nv; i += (long)gridDim.x * TPB)
        reinterpret_cast<f32x4*>(dst)[i] += reinterpret_cast<const f32x4*>(src)[i];
    for (long i = nv * 4 + (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) dst[i] += src[i];
}
}  // namespace
extern "C" int runet_add_inplace(float* dst, const float* src, long n, void* stream) {
    RUNET_REQUIRE(dst && src && n > 0 && ((uintptr_t)dst % 16) == 0 && ((uintptr_t)src % 16) == 0, "bad arguments");
    hipLaunchKernelGGL(add_inplace_kernel, dim3(ew_grid(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, dst, src, n);
    RUNET_CHECK_LAUNCH();
}

// ============================================================================================================================
// Harness helpers: bilinear resize of the probability map when output and mask sizes differ (/root/reference/Main_Final.py:577-578,
// 596-597,648-649; ATen upsample_bilinear2d, align_corners=False) and the per-pixel product of the standalone SpatialAttention (:117).
namespace {
__device__ __forceinline__ void bilin_src(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)o + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}
__global__ __launch_bounds__(TPB) void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int H, int W, int Ho,
                                                           int Wo, float sh, float sw) {
    const long total = planes * Ho * Wo;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const int ox = (int)(i % Wo);
        const long t = i / Wo;
        const int oy = (int)(t % Ho);
        const float* xp = x + (t / Ho) * H * W;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        bilin_src(oy, sh, H, y0, y1, ly0, ly1);
        bilin_src(ox, sw, W, x0, x1, lx0, lx1);
        y[i] = ly0 * (lx0 * xp[(long)y0 * W + x0] + lx1 * xp[(long)y0 * W + x1]) + ly1 * (lx0 * xp[(long)y1 * W + x0] + lx1 * xp[(long)y1 * W + x1]);
    }
}
// adjoint, gather form: input pixel (iy, ix) collects every output pixel whose two source rows / columns include it
__global__ __launch_bounds__(TPB) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long planes, int H, int W, int Ho,
                                                           int Wo, float sh, float sw) {
    const long total = planes * H * W;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const int ix = (int)(i % W);
        const long t = i / W;
        const int iy = (int)(t % H);
        const float* gp = dy + (t / H) * Ho * Wo;
        // candidate outputs: src in (i - 1, i + 1)  <=>  o in ((i - 0.5) / scale - 0.5, (i + 1.5) / scale - 0.5); row 0 also takes the clamped ones
        int oy_lo = iy == 0 ? 0 : max(0, (int)floorf(((float)iy - 0.5f) / sh - 0.5f) - 1);
        int oy_hi = min(Ho - 1, (int)ceilf(((float)iy + 1.5f) / sh - 0.5f) + 1);
        int ox_lo = ix == 0 ? 0 : max(0, (int)floorf(((float)ix - 0.5f) / sw - 0.5f) - 1);
        int ox_hi = min(Wo - 1, (int)ceilf(((float)ix + 1.5f) / sw - 0.5f) + 1);
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1;
            float ly0, ly1;
            bilin_src(oy, sh, H, y0, y1, ly0, ly1);
            const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1;
                float lx0, lx1;
                bilin_src(ox, sw, W, x0, x1, lx0, lx1);
                const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
                if (wx != 0.f) row += wx * gp[(long)oy * Wo + ox];
            }
            acc += wy * row;
        }
        dx[i] = acc;
    }
}
__global__ __launch_bounds__(TPB) void mul_pixel_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ s, float* __restrict__ y,
                                                        int ldy, long P, int C) {
    const int cvec = C / 4;
    const long total = P * cvec;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const long p = i / cvec;
        const int c = (int)(i - p * cvec) * 4;
        *reinterpret_cast<f32x4*>(y + p * ldy + c) = *reinterpret_cast<const f32x4*>(x + p * ldx + c) * s[p];
    }
}
}  // namespace

extern "C" int runet_bilinear_fwd(const float* x, float* y, long planes, int h, int w, int ho, int wo, void* stream) {
    RUNET_REQUIRE(x && y && planes > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "bad arguments");
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(ew_grid(planes * ho * wo)), dim3(TPB), 0, (hipStream_t)stream, x, y, planes, h, w, ho, wo,
                       (float)h / (float)ho, (float)w / (float)wo);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_bilinear_bwd(const float* dy, float* dx, long planes, int h, int w, int ho, int wo, void* stream) {
    RUNET_REQUIRE(dy && dx && planes > 0 && h > 0 && w > 0 && ho > 0 && wo > 0, "bad arguments");
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(ew_grid(planes * h * w)), dim3(TPB), 0, (hipStream_t)stream, dy, dx, planes, h, w, ho, wo,
                       (float)h / (float)ho, (float)w / (float)wo);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_mul_pixel(const float* x, int ldx, const float* s, float* y, int ldy, long pixels, int c, void* stream) {
    RUNET_REQUIRE(x && s && y && pixels > 0 && c > 0 && c % 4 == 0 && ldx >= c && ldy >= c, "bad arguments (c must be a multiple of 4)");
    hipLaunchKernelGGL(mul_pixel_kernel, dim3(ew_grid(pixels * (c / 4))), dim3(TPB), 0, (hipStream_t)stream, x, ldx, s, y, ldy, pixels, c);
    RUNET_CHECK_LAUNCH();
}

// ============================================================================================================================
// Plain 2-class U-Net head + loss of the reference's older trainer (/root/reference/train_water_segmentation.py:209-288 `UNet`,
// :304 `nn.CrossEntropyLoss()`): logits leave the network as [N, classes, H, W]; the loss is the mean over all pixels of
// logsumexp(z) - z[target] with ATen's log_softmax arithmetic (subtract the maximum first).
namespace {
__global__ __launch_bounds__(TPB) void nhwc_to_nchw_kernel(const float* __restrict__ x, int ld, float* __restrict__ y, int C, long HW, long total) {
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {      // i over [n][c][p]
        const long p = i % HW;
        const long t = i / HW;
        const int c = (int)(t % C);
        const long n = t / C;
        y[i] = x[(n * HW + p) * ld + c];
    }
}
constexpr int CE_MAXC = 8;
// per block: partial sum of the per-pixel losses (double); the last kernel sums the partials in order
__global__ __launch_bounds__(TPB) void ce_fwd_partial(const float* __restrict__ z, const long long* __restrict__ tgt, int C, long HW, long P,
                                                      double* __restrict__ part) {
    double acc = 0;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < P; i += (long)gridDim.x * TPB) {
        const long n = i / HW, p = i - n * HW;
        const float* zp = z + n * C * HW + p;
        float v[CE_MAXC], m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) if (c < C) { v[c] = zp[(long)c * HW]; m = fmaxf(m, v[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) if (c < C) se += expf(v[c] - m);
        const float lse = logf(se);
        const int t = (int)tgt[i];
        float zt = 0.f;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) if (c == t) zt = v[c];
        acc += (double)(-(zt - m - lse));
    }
    acc = wave_sum_d(acc);
    __shared__ double red[TPB / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int k = 0; k < TPB / 64; ++k) s += red[k];
        part[blockIdx.x] = s;
    }
}
__global__ void ce_fwd_final(const double* __restrict__ part, int nparts, long P, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0;
        for (int k = 0; k < nparts; ++k) s += part[k];
        *loss = (float)(s / (double)P);
    }
}
__global__ __launch_bounds__(TPB) void ce_bwd_kernel(const float* __restrict__ z, const long long* __restrict__ tgt, const float* __restrict__ gout,
                                                     float* __restrict__ dz, int C, long HW, long P) {
    const float gs = gout[0] / (float)P;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < P; i += (long)gridDim.x * TPB) {
        const long n = i / HW, p = i - n * HW;
        const float* zp = z + n * C * HW + p;
        float* dp = dz + n * C * HW + p;
        float v[CE_MAXC], m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) if (c < C) { v[c] = zp[(long)c * HW]; m = fmaxf(m, v[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c) if (c < C) se += expf(v[c] - m);
        const float lse = logf(se);
        const int t = (int)tgt[i];
#pragma unroll
        for (int c = 0; c < CE_MAXC; ++c)
            if (c < C) dp[(long)c * HW] = (expf(v[c] - m - lse) - (c == t ? 1.f : 0.f)) * gs;
    }
}
}  // namespace

extern "C" int runet_nhwc_to_nchw(const float* x, int ld, float* y, int n_img, int c, long hw, void* stream) {
    RUNET_REQUIRE(x && y && n_img > 0 && c > 0 && hw > 0 && ld >= c, "bad arguments");
    const long total = (long)n_img * c * hw;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid(total)), dim3(TPB), 0, (hipStream_t)stream, x, ld, y, c, hw, total);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_ce_fwd(const float* logits_nchw, const long long* target, int n_img, int classes, long hw, double* partials1024, float* loss,
                            void* stream) {
    RUNET_REQUIRE(logits_nchw && target && partials1024 && loss && n_img > 0 && hw > 0, "bad arguments");
    RUNET_REQUIRE(classes >= 2 && classes <= CE_MAXC, "2..8 classes");
    const long P = (long)n_img * hw;
    long b = (P + TPB * 4 - 1) / (TPB * 4);
    if (b > 1024) b = 1024;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_fwd_partial, dim3((int)b), dim3(TPB), 0, st, logits_nchw, target, classes, hw, P, partials1024);
    hipLaunchKernelGGL(ce_fwd_final, dim3(1), dim3(64), 0, st, partials1024, (int)b, P, loss);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_ce_bwd(const float* logits_nchw, const long long* target, const float* gout, float* dlogits_nchw, int n_img, int classes, long hw,
                            void* stream) {
    RUNET_REQUIRE(logits_nchw && target && gout && dlogits_nchw && n_img > 0 && hw > 0, "bad arguments");
    RUNET_REQUIRE(classes >= 2 && classes <= CE_MAXC, "2..8 classes");
    const long P = (long)n_img * hw;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(ew_grid(P)), dim3(TPB), 0, (hipStream_t)stream, logits_nchw, target, gout, dlogits_nchw, classes, hw, P);
    RUNET_CHECK_LAUNCH();
}
