// BatchNorm2d (train/eval), ReLU and Dropout2d as HBM-bound streaming kernels over NHWC fp32.
//
// Stands in for nn.BatchNorm2d / nn.ReLU / nn.Dropout2d at /root/reference/Main_Final.py:158,160,162,163,
// 173,127,132,137,210,211 and for their autograd.  All kernels read/write 16 B per lane (float4 over the
// channel axis, which is contiguous in NHWC), one pass per tensor; the channel-wise reductions keep a
// shifted (sum, sum-of-squares) pair per thread, combine with Chan's parallel formula in LDS and in a
// tiny second kernel in double precision, so the variance does not suffer from E[x^2]-E[x]^2 cancellation.
//
// Algorithmic HBM bytes per element: chan_stats 4 (read), bn_apply 8 (read+write), bn_bwd_reduce 8-12,
// bn_bwd_apply 12-16.
#include "runet_common.h"
#include "../../include/runet_hip.h"
#include <float.h>

namespace {

constexpr int TPB = 256;
constexpr int STATS_GROUPS = 16;      // chan_stats_partial: threads per channel in the row combine when there are few channels

struct Welf { float n, mean, m2; };

__device__ __forceinline__ void chan_combine(double& n, double& mean, double& m2, double nb, double mb, double m2b) {
    if (nb == 0.0) return;
    const double nt = n + nb;
    const double d = mb - mean;
    mean += d * (nb / nt);
    m2 += m2b + d * d * (n * nb / nt);
    n = nt;
}

// grid (chunks, N); thread owns VEC consecutive channels and every `rows`-th pixel of its chunk.
template <int VEC, bool MINMAX>
__global__ __launch_bounds__(TPB) void chan_stats_partial(const float* __restrict__ x, int ld, int HW, int C,
                                                          int pix_per_chunk, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int cvec = C / VEC;
    const int rows = TPB / cvec;
    const int tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    const int n = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int p0 = chunk * pix_per_chunk;
    const int p1 = min(HW, p0 + pix_per_chunk);
    const bool active = row < rows;

    float K[VEC], s1[VEC], s2[VEC], mx[VEC], mn[VEC];
    int imx[VEC], imn[VEC];
    int cnt = 0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) { K[v] = 0.f; s1[v] = 0.f; s2[v] = 0.f; mx[v] = -FLT_MAX; mn[v] = FLT_MAX; imx[v] = 0; imn[v] = 0; }
    if (active) {
        const float* base = x + ((long)n * HW) * ld + col * VEC;
        auto fetch = [&](int p, float (&v)[VEC]) {
            if constexpr (VEC == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(base + (long)p * ld);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            } else {
                v[0] = base[(long)p * ld];
            }
        };
        auto take = [&](int p, const float (&v)[VEC]) {
            if (cnt == 0) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) K[q] = v[q];
            }
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float d = v[q] - K[q];
                s1[q] += d;
                s2[q] += d * d;
                if constexpr (MINMAX) {
                    if (v[q] > mx[q]) { mx[q] = v[q]; imx[q] = p; }
                    if (v[q] < mn[q]) { mn[q] = v[q]; imn[q] = p; }
                }
            }
            ++cnt;
        };
        int p = p0 + row;
        // four loads in flight per thread (a read-only stream needs them: 3.9 TB/s with one); the pixels are still taken in order
        for (; p + 3 * rows < p1; p += 4 * rows) {
            float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
            fetch(p, v0); fetch(p + rows, v1); fetch(p + 2 * rows, v2); fetch(p + 3 * rows, v3);
            take(p, v0); take(p + rows, v1); take(p + 2 * rows, v2); take(p + 3 * rows, v3);
        }
        for (; p < p1; p += rows) {
            float v[VEC];
            fetch(p, v);
            take(p, v);
        }
    }
    // per-thread (n, mean, M2) -> LDS [row][C][3] (+ minmax [row][C][4])
    float* w3 = sm;
    float* wm = sm + rows * C * 3;
    if (active) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int c = col * VEC + q;
            const float fn = (float)cnt;
            const float mean = cnt ? K[q] + s1[q] / fn : 0.f;
            const float m2 = cnt ? fmaxf(s2[q] - s1[q] * s1[q] / fn, 0.f) : 0.f;
            float* d = w3 + (row * C + c) * 3;
            d[0] = fn; d[1] = mean; d[2] = m2;
            if constexpr (MINMAX) {
                float* e = wm + (row * C + c) * 4;
                e[0] = mx[q]; e[1] = mn[q]; e[2] = __int_as_float(imx[q]); e[3] = __int_as_float(imn[q]);
            }
        }
    }
    __syncthreads();
    // rows r0, r0 + rstep, ... of channel c -> (n, mean, M2[, max, min, first indices])
    auto combine_rows = [&](int c, int r0, int r1, double& n_, double& mean, double& m2, float& bmx, float& bmn, int& bimx, int& bimn) {
        for (int r = r0; r < r1; ++r) {
            const float* d = w3 + (r * C + c) * 3;
            chan_combine(n_, mean, m2, d[0], d[1], d[2]);
            if constexpr (MINMAX) {
                const float* e = wm + (r * C + c) * 4;
                if (d[0] > 0.f) {
                    const int i1 = __float_as_int(e[2]), i2 = __float_as_int(e[3]);
                    if (e[0] > bmx || (e[0] == bmx && i1 < bimx)) { bmx = e[0]; bimx = i1; }
                    if (e[1] < bmn || (e[1] == bmn && i2 < bimn)) { bmn = e[1]; bimn = i2; }
                }
            }
        }
    };
    if (C * STATS_GROUPS <= TPB && rows >= 2 * STATS_GROUPS) {
        // few channels (the attention gates' single-channel maps: rows = 256): one thread combining all the rows in double precision is
        // a 256-step serial chain per workgroup (12 us); STATS_GROUPS threads per channel take contiguous row ranges, then one combines them
        float* g3 = sm + rows * C * (MINMAX ? 7 : 3);            // [group][C][7]
        const int c = tid % C, grp = tid / C;
        if (grp < STATS_GROUPS) {
            const int per = (rows + STATS_GROUPS - 1) / STATS_GROUPS;
            double n_ = 0, mean = 0, m2 = 0;
            float bmx = -FLT_MAX, bmn = FLT_MAX;
            int bimx = 0, bimn = 0;
            combine_rows(c, grp * per, min(rows, grp * per + per), n_, mean, m2, bmx, bmn, bimx, bimn);
            float* o = g3 + (grp * C + c) * 7;
            o[0] = (float)n_; o[1] = (float)mean; o[2] = (float)m2; o[3] = bmx; o[4] = bmn; o[5] = __int_as_float(bimx); o[6] = __int_as_float(bimn);
        }
        __syncthreads();
        if (tid < C) {
            double n_ = 0, mean = 0, m2 = 0;
            float bmx = -FLT_MAX, bmn = FLT_MAX;
            int bimx = 0, bimn = 0;
            for (int q = 0; q < STATS_GROUPS; ++q) {
                const float* d = g3 + (q * C + tid) * 7;
                chan_combine(n_, mean, m2, d[0], d[1], d[2]);
                if constexpr (MINMAX) {
                    if (d[0] > 0.f) {
                        const int i1 = __float_as_int(d[5]), i2 = __float_as_int(d[6]);
                        if (d[3] > bmx || (d[3] == bmx && i1 < bimx)) { bmx = d[3]; bimx = i1; }
                        if (d[4] < bmn || (d[4] == bmn && i2 < bimn)) { bmn = d[4]; bimn = i2; }
                    }
                }
            }
            float* o = part + (((long)n * nchunks + chunk) * C + tid) * (MINMAX ? 7 : 3);
            o[0] = (float)n_; o[1] = (float)mean; o[2] = (float)m2;
            if constexpr (MINMAX) { o[3] = bmx; o[4] = bmn; o[5] = __int_as_float(bimx); o[6] = __int_as_float(bimn); }
        }
        return;
    }
    for (int c = tid; c < C; c += TPB) {
        double n_ = 0, mean = 0, m2 = 0;
        float bmx = -FLT_MAX, bmn = FLT_MAX;
        int bimx = 0, bimn = 0;
        combine_rows(c, 0, rows, n_, mean, m2, bmx, bmn, bimx, bimn);
        float* o = part + (((long)n * nchunks + chunk) * C + c) * (MINMAX ? 7 : 3);
        o[0] = (float)n_; o[1] = (float)mean; o[2] = (float)m2;
        if constexpr (MINMAX) { o[3] = bmx; o[4] = bmn; o[5] = __int_as_float(bimx); o[6] = __int_as_float(bimn); }
    }
}

// block = CW channels x (256/CW) chunk-lanes of ONE image: each thread Chan-combines its share of the chunks (in chunk order,
// so strict comparisons keep the first occurrence of max/min), lane 0 of each channel combines the lanes in lane order
template <bool MINMAX>
__global__ __launch_bounds__(TPB) void chan_stats_combine(const float* __restrict__ part, int C, int nchunks, int CW,
                                                          float* __restrict__ mean_nc, float* __restrict__ m2_nc,
                                                          float* __restrict__ max_nc, float* __restrict__ min_nc,
                                                          int* __restrict__ imax_nc, int* __restrict__ imin_nc) {
    __shared__ double sd[3][TPB];
    __shared__ float sf[2][TPB];
    __shared__ int si[2][TPB];
    constexpr int S = MINMAX ? 7 : 3;
    const int PL = TPB / CW;
    const int cl = threadIdx.x % CW, pl = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl, n = blockIdx.y;
    double n_ = 0, mean = 0, m2 = 0;
    float bmx = -FLT_MAX, bmn = FLT_MAX;
    int bimx = 0x7fffffff, bimn = 0x7fffffff;
    if (c < C && pl < PL) {
        // contiguous share of the chunk list per lane keeps pixel order: lane pl owns chunks [k0, k1)
        const int per = (nchunks + PL - 1) / PL;
        const int k0 = pl * per, k1 = min(nchunks, k0 + per);
        for (int k = k0; k < k1; ++k) {
            const float* o = part + (((long)n * nchunks + k) * C + c) * S;
            chan_combine(n_, mean, m2, o[0], o[1], o[2]);
            if constexpr (MINMAX) {
                if (o[0] > 0.f) {
                    if (o[3] > bmx) { bmx = o[3]; bimx = __float_as_int(o[5]); }
                    if (o[4] < bmn) { bmn = o[4]; bimn = __float_as_int(o[6]); }
                }
            }
        }
    }
    sd[0][threadIdx.x] = n_; sd[1][threadIdx.x] = mean; sd[2][threadIdx.x] = m2;
    if constexpr (MINMAX) { sf[0][threadIdx.x] = bmx; sf[1][threadIdx.x] = bmn; si[0][threadIdx.x] = bimx; si[1][threadIdx.x] = bimn; }
    __syncthreads();
    if (pl == 0 && c < C) {
        for (int j = 1; j < PL; ++j) {
            const int t = j * CW + cl;
            chan_combine(n_, mean, m2, sd[0][t], sd[1][t], sd[2][t]);
            if constexpr (MINMAX) {
                if (sd[0][t] > 0.0) {
                    if (sf[0][t] > bmx) { bmx = sf[0][t]; bimx = si[0][t]; }
                    if (sf[1][t] < bmn) { bmn = sf[1][t]; bimn = si[1][t]; }
                }
            }
        }
        const int i = n * C + c;
        mean_nc[i] = (float)mean;
        m2_nc[i] = (float)m2;
        if constexpr (MINMAX) { max_nc[i] = bmx; min_nc[i] = bmn; imax_nc[i] = bimx; imin_nc[i] = bimn; }
    }
}

__global__ void bn_finalize_kernel(const float* __restrict__ mean_nc, const float* __restrict__ m2_nc, int N, int C, long HW,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* run_mean,
                                   float* run_var, long long* nbt, float momentum, float eps, int training,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && training && nbt) *nbt += 1;
    if (c >= C) return;
    double mean, var;
    if (training) {
        double n_ = 0, m = 0, m2 = 0;
        for (int n = 0; n < N; ++n) chan_combine(n_, m, m2, (double)HW, mean_nc[n * C + c], m2_nc[n * C + c]);
        mean = m;
        var = m2 / n_;
        if (run_mean) {
            run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
            run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * (m2 / (n_ - 1.0)));
        }
    } else {
        mean = run_mean[c];
        var = run_var[c];
    }
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? gamma[c] : 1.0, b = beta ? beta[c] : 0.0;
    scale[c] = (float)(g * invstd);
    shift[c] = (float)(b - mean * g * invstd);
    if (save_mean) { save_mean[c] = (float)mean; save_invstd[c] = (float)invstd; }
}

// Batch statistics straight from the per-chunk partials + the BatchNorm coefficients, in ONE launch (training mode, no per-image outputs
// wanted: every BatchNorm of the network except the nine whose per-image max / min feed the channel attention).  Block = ONE channel:
// each of the 256 threads Chan-combines a contiguous share of the channel's n_img * nchunks partials, a fixed binary tree over the
// threads (8 levels) finishes, thread 0 derives scale / shift / saved statistics / running statistics as bn_finalize_kernel does.
// Replaces chan_stats_combine + bn_finalize (two ~8 us launches, 30 times per step on the forward chain).  (With 8 channels x 32 lanes
// per block and a serial combine of the lanes the kernel took as long as the two it replaced: the chains of double-precision divisions
// are the cost, so they are kept short.)
__global__ __launch_bounds__(TPB) void chan_stats_finalize_kernel(const float* __restrict__ part, int C, int nparts,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* run_mean, float* run_var, long long* nbt, float momentum, float eps,
                                                                   float* __restrict__ scale, float* __restrict__ shift,
                                                                   float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    __shared__ double sd[3][TPB];
    const int c = blockIdx.x, t = threadIdx.x;
    if (c == 0 && t == 0 && nbt) *nbt += 1;
    double n_ = 0, mean = 0, m2 = 0;
    {
        const int per = (nparts + TPB - 1) / TPB;
        const int q0 = t * per, q1 = min(nparts, q0 + per);
        for (int q = q0; q < q1; ++q) {
            const float* o = part + ((long)q * C + c) * 3;
            chan_combine(n_, mean, m2, o[0], o[1], o[2]);
        }
    }
    sd[0][t] = n_; sd[1][t] = mean; sd[2][t] = m2;
    __syncthreads();
    for (int stride = TPB / 2; stride > 0; stride >>= 1) {
        if (t < stride) {
            chan_combine(n_, mean, m2, sd[0][t + stride], sd[1][t + stride], sd[2][t + stride]);
            sd[0][t] = n_; sd[1][t] = mean; sd[2][t] = m2;
        }
        __syncthreads();
    }
    if (t == 0) {
        const double var = m2 / n_;
        if (run_mean) {
            run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
            run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * (m2 / (n_ - 1.0)));
        }
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const double g = gamma ? gamma[c] : 1.0, b = beta ? beta[c] : 0.0;
        scale[c] = (float)(g * invstd);
        shift[c] = (float)(b - mean * g * invstd);
        if (save_mean) { save_mean[c] = (float)mean; save_invstd[c] = (float)invstd; }
    }
}

// Streaming kernels use a division-free layout: grid (chunks, images); a thread owns VEC consecutive channels (its per-channel
// coefficients live in registers) and every `rows`-th pixel of its chunk, so the inner loop is loads, FMAs and one pointer add.
template <int VEC>
__global__ __launch_bounds__(TPB) void bn_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                       int HW, int C, int pix_per_chunk, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ mask, int relu) {
    const int cvec = C / VEC, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    if (row >= rows) return;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    float sc[VEC], sh[VEC], mk[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        const int c = col * VEC + q;
        sc[q] = scale[c]; sh[q] = shift[c]; mk[q] = mask ? mask[(long)n * C + c] : 1.f;
    }
    const long ib = (long)n * HW;
    for (int p = p0 + row; p < p1; p += rows) {
        float v[VEC];
        if constexpr (VEC == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(x + (ib + p) * ldx + col * 4);
            v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        } else v[0] = x[(ib + p) * ldx + col];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float r = bn_pre(v[q], sc[q], sh[q]);
            if (relu) r = fmaxf(r, 0.f);
            v[q] = r * mk[q];
        }
        if constexpr (VEC == 4) {
            f32x4 t = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(y + (ib + p) * ldy + col * 4) = t;
        } else y[(ib + p) * ldy + col] = v[0];
    }
}

// partial sums of g and g*xhat per channel; g = dy [* mask[n,c] * (act > 0)]
template <int VEC>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_partial(const float* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                             int ldx, const float* __restrict__ act, int ldact, int HW, int C,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ mask, int pix_per_chunk,
                                                             float* __restrict__ part, const float* __restrict__ rscale,
                                                             const float* __restrict__ rshift) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int cvec = C / VEC, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    // ReLU(+Dropout2d) backward: either from the saved activation (act > 0) or, when rscale/rshift are given, recomputed from x with
    // the forward's own expression x*scale + shift > 0 - one tensor less to read
    const bool recompute = rscale != nullptr;
    float sg[VEC], sgx[VEC], mu[VEC], is[VEC], mk[VEC], fs[VEC], fh[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        sg[q] = 0.f; sgx[q] = 0.f;
        const int c = col * VEC + q;
        const bool ok = row < rows;
        mu[q] = ok ? mean[c] : 0.f; is[q] = ok ? invstd[c] : 0.f;
        mk[q] = (ok && mask) ? mask[(long)n * C + c] : 1.f;
        fs[q] = (ok && recompute) ? rscale[c] : 0.f; fh[q] = (ok && recompute) ? rshift[c] : 0.f;
    }
    if (row < rows) {
        const long ib = (long)n * HW;
        auto fetch = [&](int p, float (&g)[VEC], float (&xv)[VEC], float (&av)[VEC]) {
            if constexpr (VEC == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(dy + (ib + p) * lddy + col * 4);
                const f32x4 u = *reinterpret_cast<const f32x4*>(x + (ib + p) * ldx + col * 4);
                g[0] = t[0]; g[1] = t[1]; g[2] = t[2]; g[3] = t[3];
                xv[0] = u[0]; xv[1] = u[1]; xv[2] = u[2]; xv[3] = u[3];
                if (act) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(act + (ib + p) * ldact + col * 4);
                    av[0] = a[0]; av[1] = a[1]; av[2] = a[2]; av[3] = a[3];
                }
            } else {
                g[0] = dy[(ib + p) * lddy + col]; xv[0] = x[(ib + p) * ldx + col];
                if (act) av[0] = act[(ib + p) * ldact + col];
            }
        };
        auto take = [&](const float (&g)[VEC], const float (&xv)[VEC], const float (&av)[VEC]) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                float gg = g[q];
                if (act) gg = (av[q] > 0.f) ? gg * mk[q] : 0.f;
                else if (recompute) gg = (bn_pre(xv[q], fs[q], fh[q]) > 0.f) ? gg * mk[q] : 0.f;
                sg[q] += gg;
                sgx[q] += gg * (xv[q] - mu[q]) * is[q];
            }
        };
        int p = p0 + row;
        for (; p + rows < p1; p += 2 * rows) {           // two pixels' loads in flight per thread; same summation order
            float g0[VEC], x0[VEC], a0[VEC], g1[VEC], x1[VEC], a1[VEC];
            fetch(p, g0, x0, a0); fetch(p + rows, g1, x1, a1);
            take(g0, x0, a0); take(g1, x1, a1);
        }
        for (; p < p1; p += rows) {
            float g0[VEC], x0[VEC], a0[VEC];
            fetch(p, g0, x0, a0);
            take(g0, x0, a0);
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            sm[(row * C + col * VEC + q) * 2 + 0] = sg[q];
            sm[(row * C + col * VEC + q) * 2 + 1] = sgx[q];
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += TPB) {
        double a = 0, b = 0;
        for (int r = 0; r < rows; ++r) { a += sm[(r * C + c) * 2]; b += sm[(r * C + c) * 2 + 1]; }
        float* o = part + (((long)n * gridDim.x + chunk) * C + c) * 2;
        o[0] = (float)a; o[1] = (float)b;
    }
}

// block = CW channels x (256/CW) part-lanes; each thread strides over the partials, LDS tree over the part-lanes
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_final(const float* __restrict__ part, int nparts, int C, int CW, float* __restrict__ sums) {
    __shared__ double red[2][TPB];
    const int PL = TPB / CW;
    const int cl = threadIdx.x % CW, pl = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    double a = 0, b = 0;
    if (c < C && pl < PL)
        for (int k = pl; k < nparts; k += PL) { a += part[((long)k * C + c) * 2]; b += part[((long)k * C + c) * 2 + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    if (pl == 0 && c < C) {
        for (int j = 1; j < PL; ++j) { a += red[0][j * CW + cl]; b += red[1][j * CW + cl]; }
        sums[c] = (float)b;        // dgamma = sum g * xhat   (laid out like the parameters: weight, then bias)
        sums[C + c] = (float)a;    // dbeta  = sum g
    }
}

template <int VEC>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x,
                                                           int ldx, const float* __restrict__ act, int ldact, float* __restrict__ dx,
                                                           int lddx, int HW, int C, int pix_per_chunk, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ scale,
                                                           const float* __restrict__ sums, const float* __restrict__ mask, float inv_m,
                                                           const float* __restrict__ rshift) {
    const int cvec = C / VEC, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    if (row >= rows) return;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    // dx = sc*(g - k1 - xhat*k2) = g*sc + x*a + b  with a = -sc*k2*invstd, b = sc*(mean*invstd*k2 - k1)
    const bool recompute = rshift != nullptr;        // ReLU mask from x*scale + shift > 0 (scale IS the forward scale) instead of act > 0
    float sc[VEC], ca[VEC], cb[VEC], mk[VEC], fh[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        const int c = col * VEC + q;
        sc[q] = scale[c];
        bn_bwd_coef(sc[q], mean[c], invstd[c], sums[c], sums[C + c], inv_m, ca[q], cb[q]);
        mk[q] = mask ? mask[(long)n * C + c] : 1.f;
        fh[q] = recompute ? rshift[c] : 0.f;
    }
    const long ib = (long)n * HW;
    for (int p = p0 + row; p < p1; p += rows) {
        float g[VEC], xv[VEC], av[VEC];
        if constexpr (VEC == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(dy + (ib + p) * lddy + col * 4);
            const f32x4 u = *reinterpret_cast<const f32x4*>(x + (ib + p) * ldx + col * 4);
            g[0] = t[0]; g[1] = t[1]; g[2] = t[2]; g[3] = t[3];
            xv[0] = u[0]; xv[1] = u[1]; xv[2] = u[2]; xv[3] = u[3];
            if (act) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(act + (ib + p) * ldact + col * 4);
                av[0] = a[0]; av[1] = a[1]; av[2] = a[2]; av[3] = a[3];
            }
        } else {
            g[0] = dy[(ib + p) * lddy + col]; xv[0] = x[(ib + p) * ldx + col];
            if (act) av[0] = act[(ib + p) * ldact + col];
        }
        float r[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float gg = g[q];
            if (act) gg = (av[q] > 0.f) ? gg * mk[q] : 0.f;
            else if (recompute) gg = (bn_pre(xv[q], sc[q], fh[q]) > 0.f) ? gg * mk[q] : 0.f;
            r[q] = bn_bwd_dx(gg, sc[q], xv[q], ca[q], cb[q]);
        }
        if constexpr (VEC == 4) {
            f32x4 t = {r[0], r[1], r[2], r[3]};
            *reinterpret_cast<f32x4*>(dx + (ib + p) * lddx + col * 4) = t;
        } else dx[(ib + p) * lddx + col] = r[0];
    }
}

// per-channel plain sum over all pixels (bias gradients), reusing the partial->final structure
template <int VEC>
__global__ __launch_bounds__(TPB) void chan_sum_partial(const float* __restrict__ x, int ld, long P, int C, long pix_per_chunk,
                                                        float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int cvec = C / VEC, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    const long p0 = (long)blockIdx.x * pix_per_chunk;
    const long p1 = p0 + pix_per_chunk < P ? p0 + pix_per_chunk : P;
    float s[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) s[q] = 0.f;
    if (row < rows) {
        for (long p = p0 + row; p < p1; p += rows) {
            if constexpr (VEC == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(x + p * ld + col * 4);
                s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
            } else s[0] += x[p * ld + col];
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) sm[row * C + col * VEC + q] = s[q];
    }
    __syncthreads();
    for (int c = tid; c < C; c += TPB) {
        double a = 0;
        for (int r = 0; r < rows; ++r) a += sm[r * C + c];
        part[(long)blockIdx.x * C + c] = (float)a;
    }
}
__global__ __launch_bounds__(TPB) void chan_sum_final(const float* __restrict__ part, int nparts, int C, int CW, float* __restrict__ out, int accumulate) {
    __shared__ double red[TPB];
    const int PL = TPB / CW;
    const int cl = threadIdx.x % CW, pl = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    double a = 0;
    if (c < C && pl < PL)
        for (int k = pl; k < nparts; k += PL) a += part[(long)k * C + c];
    red[threadIdx.x] = a;
    __syncthreads();
    if (pl == 0 && c < C) {
        for (int j = 1; j < PL; ++j) a += red[j * CW + cl];
        out[c] = accumulate ? out[c] + (float)a : (float)a;
    }
}

inline int final_cw(int c, long nparts = 0) { const int w = nparts >= 256 ? 4 : 32; return c >= w ? w : c; }

inline int pick_chunks(int N, int HW, int C, int rows) {
    // ~32 vector loads per thread; at least one pass of `rows` pixels per chunk; and at least ~768 workgroups on the chip: the
    // single-channel maps of the attention gates (16 x 256 x 256 x 1: two chunks per image = 32 workgroups) took 61 us for 4 MB
    long per_img = ((long)HW * C + 32767) / 32768;
    const long want = (768 + N - 1) / N;
    if (per_img < want) per_img = want;
    if (per_img < 1) per_img = 1;
    if (per_img > 1024) per_img = 1024;
    long maxc = (HW + rows - 1) / rows;
    if (per_img > maxc) per_img = maxc;
    return (int)per_img;
}
// chunks per image for the streaming kernels: ~16 vector loads per thread, at least one pass of `rows` pixels per chunk
inline int stream_chunks(int N, int HW, int C, int rows, int& ppc) {
    long per_img = ((long)HW * C + 16383) / 16384;
    const long maxc = (HW + rows - 1) / rows;
    if (per_img > maxc) per_img = maxc;
    if (per_img > 4096) per_img = 4096;
    if (per_img < 1) per_img = 1;
    ppc = (int)((HW + per_img - 1) / per_img);
    return (int)((HW + ppc - 1) / ppc);
}
inline int ew_grid(long total_vec) {
    long b = (total_vec + TPB - 1) / TPB;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define REQ_VEC(C) RUNET_REQUIRE((C) >= 1 && (C) <= 1024 && ((C) % 4 == 0 || (C) == 1), "channels must be 1 or a multiple of 4, at most 1024")

extern "C" long runet_reduce_workspace_floats(int n_img, int hw, int c) {
    // covers every partial-reduction kernel of this library for an [n_img, hw, c] tensor
    const int vec = (c % 4 == 0) ? 4 : 1;
    const int rows = TPB / (c / vec > 0 ? c / vec : 1);
    const long a = (long)n_img * pick_chunks(n_img, hw, c, rows > 0 ? rows : 1) * c * 7;
    const long b = 2048L * (c + 1);
    return (a > b ? a : b) + 64;
}

extern "C" int runet_chan_stats(const float* x, int ld, int n_img, int hw, int c, float* workspace, float* mean_nc,
                                float* m2_nc, float* max_nc, float* min_nc, int* imax_nc, int* imin_nc, int want_minmax,
                                void* stream) {
    RUNET_REQUIRE(x && workspace && mean_nc && m2_nc, "null pointer");
    REQ_VEC(c);
    RUNET_REQUIRE(n_img > 0 && hw > 0 && ld >= c, "bad shape");
    RUNET_REQUIRE(!want_minmax || (max_nc && min_nc && imax_nc && imin_nc), "min/max outputs missing");
    hipStream_t st = (hipStream_t)stream;
    const int vec = (c % 4 == 0) ? 4 : 1;
    const int cvec = c / vec, rows = TPB / cvec;
    const int chunks = pick_chunks(n_img, hw, c, rows);
    const int ppc = (hw + chunks - 1) / chunks;
    const size_t lds = ((size_t)rows * c * (want_minmax ? 7 : 3) + (c * STATS_GROUPS <= TPB ? STATS_GROUPS * c * 7 : 0)) * sizeof(float);
    RUNET_REQUIRE(lds <= 64 * 1024, "LDS budget");
    dim3 grid(chunks, n_img);
#define LAUNCH_STATS(V, MM) hipLaunchKernelGGL((chan_stats_partial<V, MM>), grid, dim3(TPB), lds, st, x, ld, hw, c, ppc, workspace)
    if (vec == 4) { if (want_minmax) LAUNCH_STATS(4, true); else LAUNCH_STATS(4, false); }
    else { if (want_minmax) LAUNCH_STATS(1, true); else LAUNCH_STATS(1, false); }
#undef LAUNCH_STATS
    const int ccw = chunks >= 16 ? (c >= 8 ? 8 : c) : (c >= 64 ? 64 : c);
    dim3 cgrid(cdiv(c, ccw), n_img);
    if (want_minmax)
        hipLaunchKernelGGL((chan_stats_combine<true>), cgrid, dim3(TPB), 0, st, workspace, c, chunks, ccw, mean_nc, m2_nc, max_nc, min_nc, imax_nc, imin_nc);
    else
        hipLaunchKernelGGL((chan_stats_combine<false>), cgrid, dim3(TPB), 0, st, workspace, c, chunks, ccw, mean_nc, m2_nc, nullptr, nullptr, nullptr, nullptr);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_bn_stats(const float* x, int ld, int n_img, int hw, int c, float* workspace, const float* gamma, const float* beta,
                              float* run_mean, float* run_var, long long* num_batches_tracked, float momentum, float eps, float* scale,
                              float* shift, float* save_mean, float* save_invstd, void* stream) {
    RUNET_REQUIRE(x && workspace && scale && shift, "null pointer");
    REQ_VEC(c);
    RUNET_REQUIRE(n_img > 0 && hw > 0 && ld >= c && (long)n_img * hw > 1, "bad shape (training needs > 1 value per channel)");
    hipStream_t st = (hipStream_t)stream;
    const int vec = (c % 4 == 0) ? 4 : 1;
    const int cvec = c / vec, rows = TPB / cvec;
    const int chunks = pick_chunks(n_img, hw, c, rows);
    const int ppc = (hw + chunks - 1) / chunks;
    const size_t lds = ((size_t)rows * c * 3 + (c * STATS_GROUPS <= TPB ? STATS_GROUPS * c * 7 : 0)) * sizeof(float);
    RUNET_REQUIRE(lds <= 64 * 1024, "LDS budget");
    dim3 grid(chunks, n_img);
    if (vec == 4) hipLaunchKernelGGL((chan_stats_partial<4, false>), grid, dim3(TPB), lds, st, x, ld, hw, c, ppc, workspace);
    else hipLaunchKernelGGL((chan_stats_partial<1, false>), grid, dim3(TPB), lds, st, x, ld, hw, c, ppc, workspace);
    hipLaunchKernelGGL(chan_stats_finalize_kernel, dim3(c), dim3(TPB), 0, st, workspace, c, chunks * n_img, gamma, beta,
                       run_mean, run_var, num_batches_tracked, momentum, eps, scale, shift, save_mean, save_invstd);
    RUNET_CHECK_LAUNCH();
}

// Second half of runet_bn_stats on partials a PRODUCING kernel wrote in its epilogue (runet_conv_x3 / runet_wino_conv_x3 with a `stats`
// pointer): part[nparts][c][3] = (count, mean, M2) of disjoint pixel sets that together cover the tensor.
extern "C" int runet_bn_stats_finalize(const float* part, int nparts, int c, const float* gamma, const float* beta, float* run_mean, float* run_var,
                                       long long* num_batches_tracked, float momentum, float eps, float* scale, float* shift, float* save_mean,
                                       float* save_invstd, void* stream) {
    RUNET_REQUIRE(part && scale && shift && nparts > 0 && c > 0, "bad arguments");
    hipLaunchKernelGGL(chan_stats_finalize_kernel, dim3(c), dim3(TPB), 0, (hipStream_t)stream, part, c, nparts, gamma, beta, run_mean, run_var,
                       num_batches_tracked, momentum, eps, scale, shift, save_mean, save_invstd);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_bn_finalize(const float* mean_nc, const float* m2_nc, int n_img, int c, long hw, const float* gamma,
                                 const float* beta, float* run_mean, float* run_var, long long* num_batches_tracked,
                                 float momentum, float eps, int training, float* scale, float* shift, float* save_mean,
                                 float* save_invstd, void* stream) {
    RUNET_REQUIRE(scale && shift && c > 0, "null output");
    RUNET_REQUIRE(training ? (mean_nc && m2_nc && (long)n_img * hw > 1) : (run_mean && run_var), "missing statistics (training needs > 1 value per channel)");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(c, 128)), dim3(128), 0, (hipStream_t)stream, mean_nc, m2_nc, n_img, c, hw,
                       gamma, beta, run_mean, run_var, num_batches_tracked, momentum, eps, training, scale, shift, save_mean, save_invstd);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_bn_apply(const float* x, int ldx, float* y, int ldy, long pixels, int hw, int c, const float* scale,
                              const float* shift, const float* mask_nc, int relu, void* stream) {
    RUNET_REQUIRE(x && y && scale && shift, "null pointer");
    REQ_VEC(c);
    RUNET_REQUIRE(pixels > 0 && hw > 0 && ldx >= c && ldy >= c, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    RUNET_REQUIRE(pixels % hw == 0, "pixels must be a whole number of images");
    const int nimg = (int)(pixels / hw), vec = (c % 4 == 0) ? 4 : 1;
    int ppc;
    const int chunks = stream_chunks(nimg, hw, c, TPB / (c / vec), ppc);
    if (vec == 4) hipLaunchKernelGGL((bn_apply_kernel<4>), dim3(chunks, nimg), dim3(TPB), 0, st, x, ldx, y, ldy, hw, c, ppc, scale, shift, mask_nc, relu);
    else hipLaunchKernelGGL((bn_apply_kernel<1>), dim3(chunks, nimg), dim3(TPB), 0, st, x, ldx, y, ldy, hw, c, ppc, scale, shift, mask_nc, relu);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_bn_bwd_reduce(const float* dy, int lddy, const float* x, int ldx, const float* act, int ldact, int n_img,
                                   int hw, int c, const float* mean, const float* invstd, const float* mask_nc,
                                   float* workspace, float* sums, const float* relu_scale, const float* relu_shift, void* stream) {
    RUNET_REQUIRE(dy && x && mean && invstd && workspace && sums, "null pointer");
    RUNET_REQUIRE((relu_scale == nullptr) == (relu_shift == nullptr) && !(act && relu_scale), "give either act or relu_scale + relu_shift");
    REQ_VEC(c);
    hipStream_t st = (hipStream_t)stream;
    const int vec = (c % 4 == 0) ? 4 : 1, cvec = c / vec, rows = TPB / cvec;
    const int chunks = pick_chunks(n_img, hw, c, rows);
    const int ppc = (hw + chunks - 1) / chunks;
    const size_t lds = (size_t)rows * c * 2 * sizeof(float);
    dim3 grid(chunks, n_img);
    if (vec == 4) hipLaunchKernelGGL((bn_bwd_reduce_partial<4>), grid, dim3(TPB), lds, st, dy, lddy, x, ldx, act, ldact, hw, c, mean, invstd, mask_nc, ppc, workspace, relu_scale, relu_shift);
    else hipLaunchKernelGGL((bn_bwd_reduce_partial<1>), grid, dim3(TPB), lds, st, dy, lddy, x, ldx, act, ldact, hw, c, mean, invstd, mask_nc, ppc, workspace, relu_scale, relu_shift);
    const int cw = final_cw(c, (long)chunks * n_img);
    hipLaunchKernelGGL(bn_bwd_reduce_final, dim3(cdiv(c, cw)), dim3(TPB), 0, st, workspace, chunks * n_img, c, cw, sums);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_bn_bwd_apply(const float* dy, int lddy, const float* x, int ldx, const float* act, int ldact, float* dx,
                                  int lddx, long pixels, int hw, int c, const float* mean, const float* invstd,
                                  const float* scale, const float* sums, const float* mask_nc, long m_total, const float* relu_shift,
                                  void* stream) {
    RUNET_REQUIRE(dy && x && dx && mean && invstd && scale && sums && !(act && relu_shift), "null pointer / both act and relu_shift");
    REQ_VEC(c);
    hipStream_t st = (hipStream_t)stream;
    const float inv_m = 1.0f / (float)(m_total > 0 ? m_total : pixels);
    RUNET_REQUIRE(pixels % hw == 0, "pixels must be a whole number of images");
    const int nimg = (int)(pixels / hw), vec = (c % 4 == 0) ? 4 : 1;
    int ppc;
    const int chunks = stream_chunks(nimg, hw, c, TPB / (c / vec), ppc);
    if (vec == 4) hipLaunchKernelGGL((bn_bwd_apply_kernel<4>), dim3(chunks, nimg), dim3(TPB), 0, st, dy, lddy, x, ldx, act, ldact, dx, lddx, hw, c, ppc, mean, invstd, scale, sums, mask_nc, inv_m, relu_shift);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3(chunks, nimg), dim3(TPB), 0, st, dy, lddy, x, ldx, act, ldact, dx, lddx, hw, c, ppc, mean, invstd, scale, sums, mask_nc, inv_m, relu_shift);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_chan_sum(const float* x, int ld, long pixels, int c, float* workspace, float* out, int accumulate, void* stream) {
    RUNET_REQUIRE(x && workspace && out, "null pointer");
    REQ_VEC(c);
    hipStream_t st = (hipStream_t)stream;
    const int vec = (c % 4 == 0) ? 4 : 1, cvec = c / vec, rows = TPB / cvec;
    long chunks = (pixels * c + 32767) / 32768;
    if (chunks > 2048) chunks = 2048;
    if (chunks < 1) chunks = 1;
    const long ppc = (pixels + chunks - 1) / chunks;
    chunks = (pixels + ppc - 1) / ppc;
    const size_t lds = (size_t)rows * c * sizeof(float);
    if (vec == 4) hipLaunchKernelGGL((chan_sum_partial<4>), dim3((int)chunks), dim3(TPB), lds, st, x, ld, pixels, c, ppc, workspace);
    else hipLaunchKernelGGL((chan_sum_partial<1>), dim3((int)chunks), dim3(TPB), lds, st, x, ld, pixels, c, ppc, workspace);
    const int cw = final_cw(c, chunks);
    hipLaunchKernelGGL(chan_sum_final, dim3(cdiv(c, cw)), dim3(TPB), 0, st, workspace, (int)chunks, c, cw, out, accumulate);
    RUNET_CHECK_LAUNCH();
}
