// Channel attention, spatial attention, the residual-block tail and the attention gate, forward and
// backward, as fused streaming kernels over NHWC fp32.
//
// Reference semantics: ChannelAttention /root/reference/Main_Final.py:82-101, SpatialAttention :104-117,
// ResidualBlock tail :186-194 (bn2 -> ca -> sa -> += residual -> relu), AttentionGate :120-148.
//
// Forward of the tail, given t2 = conv2 output (raw), per-(n,c) statistics of t2 and the bn2 affine:
//   u0 = t2*s2 + h2 (bn2)            avg/max pools of u0 follow analytically from sum/max/min of t2
//   ca[n,c] = sigmoid(fc2(relu(fc1(avg))) + fc2(relu(fc1(max))))          (ca_coeff, one block per image)
//   u  = u0*ca = t2*A[n,c] + B[n,c]                                        (never materialised)
//   sm[p] = (mean_c u, max_c u), amax[p]                                   (sa_reduce, 16 lanes per pixel)
//   sa[p] = sigmoid(conv7x7(sm))                                           (sa_conv7, LDS halo tile)
//   out = relu(u*sa + residual)                                            (rb_out)
// so t2 is read three times and `out` written once; nothing else of tensor size touches HBM.
// Backward recomputes u from t2 instead of storing it (rb_bwd1..3), and obtains the bn2 / channel-attention
// reductions from per-(n,c) sums (ca_bwd_*), see DESIGN.md.
#include "runet_common.h"
#include "../../include/runet_hip.h"
#include <float.h>

namespace {
constexpr int TPB = 256;
constexpr int LPP = 16;              // lanes per pixel in the "pixel-major" kernels
constexpr int PPB = TPB / LPP;       // pixels per block iteration

__device__ __forceinline__ float grp_sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// --------------------------------------------------------------------------------- channel attention
// hid[0:Cr] = W0p^T avg, hid[Cr:2Cr] = W0p^T mx for one image.  Thread (j, part) walks rows c = part, part + P, ... of W0p[c][j]: lanes
// with consecutive j (then consecutive rows) read consecutive addresses; the P partials are summed through LDS in a fixed order.  (One
// wave per output with the lanes striding over c read 64 different cache lines per load: 25-45 us per launch at C = 1024.)
__device__ __forceinline__ void ca_hidden(const float* __restrict__ W0p, const float* avg, const float* mxv, int C, int Cr, float* hid,
                                          float* part /* [2 * blockDim.x] */) {
    const int tid = threadIdx.x, TPB = blockDim.x;           // the channel-attention kernels run CA_TPB threads per image (below)
    const int P = TPB / Cr;
    const int j = tid % Cr, pt = tid / Cr;
    float pa = 0.f, pm = 0.f;
    if (pt < P)
        for (int c = pt; c < C; c += P) {
            const float w = W0p[c * Cr + j];
            pa += w * avg[c];
            pm += w * mxv[c];
        }
    part[tid] = pa; part[TPB + tid] = pm;
    __syncthreads();
    for (int o = tid; o < 2 * Cr; o += TPB) {
        const float* src = part + (o < Cr ? 0 : TPB) + (o % Cr);
        float acc = 0.f;
        for (int q = 0; q < P; ++q) acc += src[q * Cr];
        hid[o] = acc;
    }
}

// One workgroup per image and a chain of dependent phases: the launch is latency-bound, so the workgroup is as wide as it can be (1024
// threads: ca_coeff 24.5 -> 11.5 us, ca_bwd_image 43.0 -> 19.8 us per launch on average over C = 64..1024).
constexpr int CA_TPB = 1024;

// grid N, block CA_TPB.  W0p[c][j] (C x Cr), W2p[j][c] (Cr x C).
__global__ __launch_bounds__(CA_TPB) void ca_coeff_kernel(const float* __restrict__ mean_nc, const float* __restrict__ max_nc,
                                                       const float* __restrict__ min_nc, const int* __restrict__ imax_nc,
                                                       const int* __restrict__ imin_nc, const float* __restrict__ s2,
                                                       const float* __restrict__ h2, const float* __restrict__ W0p,
                                                       const float* __restrict__ W2p, int C, int Cr, float* __restrict__ A,
                                                       float* __restrict__ B, float* __restrict__ ca_out, float* __restrict__ avg_out,
                                                       float* __restrict__ mx_out, int* __restrict__ idx_out, float* __restrict__ tval_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* avg = sm;            // [C]
    float* mxv = sm + C;        // [C]
    float* hid = sm + 2 * C;    // [2*Cr] : pa, pm (pre-activation)
    const int n = blockIdx.x, tid = threadIdx.x, TPB = blockDim.x;
    for (int c = tid; c < C; c += TPB) {
        const float s = s2[c], h = h2[c];
        const bool pos = s >= 0.f;
        const float tv = pos ? max_nc[n * C + c] : min_nc[n * C + c];
        avg[c] = s * mean_nc[n * C + c] + h;
        mxv[c] = s * tv + h;
        if (avg_out) { avg_out[n * C + c] = avg[c]; mx_out[n * C + c] = mxv[c]; idx_out[n * C + c] = pos ? imax_nc[n * C + c] : imin_nc[n * C + c]; tval_out[n * C + c] = tv; }
    }
    __syncthreads();
    ca_hidden(W0p, avg, mxv, C, Cr, hid, sm + 2 * C + 2 * Cr);
    __syncthreads();
    for (int c = tid; c < C; c += TPB) {
        float za = 0.f, zm = 0.f;
        for (int j = 0; j < Cr; ++j) {
            za += W2p[j * C + c] * fmaxf(hid[j], 0.f);
            zm += W2p[j * C + c] * fmaxf(hid[Cr + j], 0.f);
        }
        const float ca = sigmoidf_(za + zm);
        A[n * C + c] = s2[c] * ca;
        B[n * C + c] = h2[c] * ca;
        if (ca_out) ca_out[n * C + c] = ca;
    }
}

// --------------------------------------------------------------------------------- spatial attention
// 16 lanes per pixel; u = t2*A + B; writes sm[p] = (mean, max), amax[p]
__global__ __launch_bounds__(TPB) void sa_reduce_kernel(const float* __restrict__ t2, int ld, const float* __restrict__ A,
                                                        const float* __restrict__ B, long P, int HW, int C,
                                                        float* __restrict__ smap, int* __restrict__ amax) {
    const int sub = threadIdx.x & (LPP - 1);
    for (long p = (long)blockIdx.x * PPB + (threadIdx.x / LPP); p < P; p += (long)gridDim.x * PPB) {
        const int n = (int)(p / HW);
        const float* a = A + (long)n * C;
        const float* b = B + (long)n * C;
        float s = 0.f, m = -FLT_MAX;
        int im = 0;
        for (int c = sub * 4; c < C; c += LPP * 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(t2 + p * ld + c);
            const f32x4 av = *reinterpret_cast<const f32x4*>(a + c);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float u = t[q] * av[q] + bv[q];
                s += u;
                if (u > m) { m = u; im = c + q; }
            }
        }
        s = grp_sum16(s);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, 64);
            const int oi = __shfl_xor(im, o, 64);
            if (om > m || (om == m && oi < im)) { m = om; im = oi; }
        }
        if (sub == 0) {
            smap[p * 2] = s / (float)C;
            smap[p * 2 + 1] = m;
            amax[p] = im;
        }
    }
}

// 7x7 conv over the 2-plane map + sigmoid.  Wp[dy][dx][ch].  block 16x16 pixels, grid (W/16, H/16, N)
__global__ __launch_bounds__(256) void sa_conv7_kernel(const float* __restrict__ smap, const float* __restrict__ Wp,
                                                       float* __restrict__ sa, int H, int W) {
    __shared__ float tile[22][22][2];
    __shared__ float wsm[98];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n = blockIdx.z, h0 = blockIdx.y * 16, w0 = blockIdx.x * 16;
    if (threadIdx.x < 98) wsm[threadIdx.x] = Wp[threadIdx.x];
    for (int i = threadIdx.x; i < 22 * 22; i += 256) {
        const int r = i / 22, c = i % 22;
        const int h = h0 + r - 3, w = w0 + c - 3;
        float a = 0.f, b = 0.f;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
            const long p = ((long)n * H + h) * W + w;
            a = smap[p * 2]; b = smap[p * 2 + 1];
        }
        tile[r][c][0] = a; tile[r][c][1] = b;
    }
    __syncthreads();
    const int h = h0 + ty, w = w0 + tx;
    if (h < H && w < W) {
        float q = 0.f;
#pragma unroll
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx)
                q += wsm[(dy * 7 + dx) * 2] * tile[ty + dy][tx + dx][0] + wsm[(dy * 7 + dx) * 2 + 1] * tile[ty + dy][tx + dx][1];
        sa[((long)n * H + h) * W + w] = sigmoidf_(q);
    }
}

// out = relu((t2*A+B)*sa + res), res = r*rs + rh (conv shortcut) or r (identity, rs == nullptr)
// grid (chunks, images): a thread owns 4 channels (coefficients in registers) and every `rows`-th pixel of its chunk
__global__ __launch_bounds__(TPB) void rb_out_kernel(const float* __restrict__ t2, int ld, const float* __restrict__ A,
                                                     const float* __restrict__ B, const float* __restrict__ sa,
                                                     const float* __restrict__ r, int ldr, const float* __restrict__ rs,
                                                     const float* __restrict__ rh, float* __restrict__ out, int ldo, int HW, int C,
                                                     int pix_per_chunk) {
    const int cvec = C / 4, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    if (row >= rows) return;
    const int n = blockIdx.y, c = col * 4;
    const int p0 = blockIdx.x * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    const f32x4 a = *reinterpret_cast<const f32x4*>(A + (long)n * C + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(B + (long)n * C + c);
    f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, h4 = {0.f, 0.f, 0.f, 0.f};
    if (rs) { s4 = *reinterpret_cast<const f32x4*>(rs + c); h4 = *reinterpret_cast<const f32x4*>(rh + c); }
    const long ib = (long)n * HW;
    for (int p = p0 + row; p < p1; p += rows) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(t2 + (ib + p) * ld + c);
        const f32x4 res = *reinterpret_cast<const f32x4*>(r + (ib + p) * ldr + c) * s4 + h4;
        const float s = sa[ib + p];
        f32x4 o = (t * a + b) * s + res;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = fmaxf(o[q], 0.f);
        *reinterpret_cast<f32x4*>(out + (ib + p) * ldo + c) = o;
    }
}

// ---- backward 1: dv = dout*(out>0) (written), dq[p] = (sum_c dv*u) * sa*(1-sa)
__global__ __launch_bounds__(TPB) void rb_bwd1_kernel(const float* __restrict__ dout, int lddo, const float* __restrict__ out,
                                                      int ldo, const float* __restrict__ t2, int ld, const float* __restrict__ A,
                                                      const float* __restrict__ B, const float* __restrict__ sa,
                                                      float* __restrict__ dv, int lddv, float* __restrict__ dq, long P, int HW, int C) {
    const int sub = threadIdx.x & (LPP - 1);
    for (long p = (long)blockIdx.x * PPB + (threadIdx.x / LPP); p < P; p += (long)gridDim.x * PPB) {
        const int n = (int)(p / HW);
        float s = 0.f;
        for (int c = sub * 4; c < C; c += LPP * 4) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dout + p * lddo + c);
            f32x4 o = {1.f, 1.f, 1.f, 1.f};      // out == NULL: no ReLU behind the attention (standalone SpatialAttention)
            if (out) o = *reinterpret_cast<const f32x4*>(out + p * ldo + c);
            const f32x4 t = *reinterpret_cast<const f32x4*>(t2 + p * ld + c);
            const f32x4 a = *reinterpret_cast<const f32x4*>(A + (long)n * C + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(B + (long)n * C + c);
            f32x4 d;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d[q] = o[q] > 0.f ? g[q] : 0.f;
                s += d[q] * (t[q] * a[q] + b[q]);
            }
            *reinterpret_cast<f32x4*>(dv + p * lddv + c) = d;
        }
        s = grp_sum16(s);
        if (sub == 0) {
            const float v = sa[p];
            dq[p] = s * v * (1.f - v);
        }
    }
}

// ---- backward of the 7x7 conv: dsm[p][ch] (data gradient) and per-block partial weight gradient
__global__ __launch_bounds__(256) void sa_conv7_bwd_kernel(const float* __restrict__ smap, const float* __restrict__ dq,
                                                           const float* __restrict__ Wp, float* __restrict__ dsm,
                                                           float* __restrict__ dw_part, int H, int W) {
    __shared__ float dqt[22][22];
    __shared__ float smt[22][22][2];
    __shared__ float wsm[98];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n = blockIdx.z, h0 = blockIdx.y * 16, w0 = blockIdx.x * 16;
    if (threadIdx.x < 98) wsm[threadIdx.x] = Wp[threadIdx.x];
    for (int i = threadIdx.x; i < 22 * 22; i += 256) {
        const int r = i / 22, c = i % 22;
        const int h = h0 + r - 3, w = w0 + c - 3;
        float d = 0.f, a = 0.f, b = 0.f;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
            const long p = ((long)n * H + h) * W + w;
            d = dq[p]; a = smap[p * 2]; b = smap[p * 2 + 1];
        }
        dqt[r][c] = d; smt[r][c][0] = a; smt[r][c][1] = b;
    }
    __syncthreads();
    const int h = h0 + ty, w = w0 + tx;
    if (h < H && w < W) {
        // dsm[p][ch] = sum_{dy,dx} W[dy][dx][ch] * dq[h - (dy-3), w - (dx-3)]
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const float d = dqt[ty + 6 - dy][tx + 6 - dx];
                g0 += wsm[(dy * 7 + dx) * 2] * d;
                g1 += wsm[(dy * 7 + dx) * 2 + 1] * d;
            }
        const long p = ((long)n * H + h) * W + w;
        dsm[p * 2] = g0; dsm[p * 2 + 1] = g1;
    }
    // weight gradient partial: dW[dy][dx][ch] = sum_{pixels in tile} sm[h+dy-3][w+dx-3][ch] * dq[h][w]
    if (threadIdx.x < 98) {
        const int ch = threadIdx.x & 1, k = threadIdx.x >> 1, dy = k / 7, dx = k % 7;
        float acc = 0.f;
        for (int y = 0; y < 16; ++y)
            for (int x = 0; x < 16; ++x) acc += smt[y + dy][x + dx][ch] * dqt[y + 3][x + 3];
        const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        dw_part[blk * 98 + threadIdx.x] = acc;
    }
}
__global__ void reduce_rows_kernel(const float* __restrict__ part, long nrows, int ncols, float* __restrict__ out) {
    // out[c] = sum_r part[r][c]; one block of 256 threads per column group of 1 (small ncols)
    const int c = blockIdx.x;
    double acc = 0;
    for (long r = threadIdx.x; r < nrows; r += blockDim.x) acc += part[r * ncols + c];
    acc = wave_sum_d(acc);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// ---- backward 2: per-(n,c) sums of du and du*t2;  du = dv*sa + dsm0/C + dsm1*[c == amax]
// BN: also the shortcut BatchNorm's local backward sums - dv IS that BatchNorm's incoming gradient: per (n, chunk, c) sum dv and
// sum dv * rhat (rhat = (r - mean_s) * invstd_s), so the bn_bwd_reduce pass over (dv, r) that used to follow is one read of r here.
template <bool BN>
__global__ __launch_bounds__(TPB) void rb_bwd2_partial(const float* __restrict__ dv, int lddv, const float* __restrict__ t2, int ld,
                                                       const float* __restrict__ sa, const float* __restrict__ dsm,
                                                       const int* __restrict__ amax, int HW, int C, int pix_per_chunk,
                                                       float* __restrict__ part, const float* __restrict__ r, int ldr,
                                                       const float* __restrict__ mean_s, const float* __restrict__ invstd_s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int NV = BN ? 4 : 2;
    const int cvec = C / 4, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    float s[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0}, sv[4] = {0, 0, 0, 0}, svr[4] = {0, 0, 0, 0};
    const float invC = 1.0f / (float)C;
    if (row < rows) {
        float ms[4], is[4];
        if constexpr (BN) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { ms[q] = mean_s[col * 4 + q]; is[q] = invstd_s[col * 4 + q]; }
        }
        const long ib = (long)n * HW;
        for (int p = p0 + row; p < p1; p += rows) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(dv + (ib + p) * lddv + col * 4);
            const f32x4 t = *reinterpret_cast<const f32x4*>(t2 + (ib + p) * ld + col * 4);
            f32x4 rr = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BN) rr = *reinterpret_cast<const f32x4*>(r + (ib + p) * ldr + col * 4);
            const float v = sa[ib + p], g0 = dsm[(ib + p) * 2] * invC, g1 = dsm[(ib + p) * 2 + 1];
            const int am = amax[ib + p];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float du = d[q] * v + g0 + ((col * 4 + q) == am ? g1 : 0.f);
                s[q] += du;
                st[q] += du * t[q];
                if constexpr (BN) { sv[q] += d[q]; svr[q] += d[q] * ((rr[q] - ms[q]) * is[q]); }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* o = sm + (row * C + col * 4 + q) * NV;
            o[0] = s[q]; o[1] = st[q];
            if constexpr (BN) { o[2] = sv[q]; o[3] = svr[q]; }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += TPB) {
        double a[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) a[j] = 0;
        for (int rw = 0; rw < rows; ++rw)
#pragma unroll
            for (int j = 0; j < NV; ++j) a[j] += sm[(rw * C + c) * NV + j];
        float* o = part + (((long)n * gridDim.x + chunk) * C + c) * NV;
#pragma unroll
        for (int j = 0; j < NV; ++j) o[j] = (float)a[j];
    }
}
__global__ void rb_bwd2_final(const float* __restrict__ part, int N, int C, int nchunks, int nv, float* __restrict__ sdu, float* __restrict__ sdut) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    double a = 0, b = 0;
    for (int k = 0; k < nchunks; ++k) {
        const float* o = part + (((long)n * nchunks + k) * C + c) * nv;
        a += o[0]; b += o[1];
    }
    sdu[i] = (float)a; sdut[i] = (float)b;
}
// the shortcut BatchNorm's (dgamma | dbeta) from the partials of rb_bwd2_partial<true>: one block per channel over all (image, chunk) rows
__global__ void rb_bwd2_bn_final(const float* __restrict__ part, long nrows, int C, float* __restrict__ sums_s) {
    const int c = blockIdx.x;
    double a = 0, b = 0;
    for (long rw = threadIdx.x; rw < nrows; rw += blockDim.x) { a += part[(rw * C + c) * 4 + 2]; b += part[(rw * C + c) * 4 + 3]; }
    a = wave_sum_d(a); b = wave_sum_d(b);
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums_s[C + c] = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);      // dbeta  = sum dv
        sums_s[c] = (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);          // dgamma = sum dv * rhat
    }
}

// ---- channel-attention backward, phase 1, per image (grid N): dz, the MLP's hidden gradients, davg/dmx.
// Stores the small per-image vectors (dz[C], dpa/dpm/hs[Cr]) that phase 2 turns into the weight gradients.
__global__ __launch_bounds__(CA_TPB) void ca_bwd_image_kernel(const float* __restrict__ sdu, const float* __restrict__ sdut,
                                                           const float* __restrict__ s2, const float* __restrict__ h2,
                                                           const float* __restrict__ ca, const float* __restrict__ avg,
                                                           const float* __restrict__ mxv, const float* __restrict__ W0p,
                                                           const float* __restrict__ W2p, int C, int Cr,
                                                           float* __restrict__ davg, float* __restrict__ dmx,
                                                           float* __restrict__ dz_out, float* __restrict__ hvec) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dz = sm;               // [C]
    float* hid = sm + C;          // [2*Cr] pre-activations pa, pm
    float* dh = sm + C + 2 * Cr;  // [Cr]
    const int n = blockIdx.x, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, TPB = blockDim.x;
    const float* avg_n = avg + (long)n * C;
    const float* mx_n = mxv + (long)n * C;
    for (int c = tid; c < C; c += TPB) {
        const float k = ca[n * C + c];
        const float dca = s2[c] * sdut[n * C + c] + h2[c] * sdu[n * C + c];
        dz[c] = dca * k * (1.f - k);
        dz_out[n * C + c] = dz[c];
    }
    ca_hidden(W0p, avg_n, mx_n, C, Cr, hid, sm + C + 3 * Cr);
    __syncthreads();
    for (int j = wid; j < Cr; j += TPB / 64) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += W2p[j * C + c] * dz[c];
        acc = wave_sum(acc);
        if (lane == 0) dh[j] = acc;
    }
    __syncthreads();
    for (int j = tid; j < Cr; j += TPB) {   // hvec[n] = (dpa | dpm | relu(pa)+relu(pm))
        hvec[((long)n * 3 + 0) * Cr + j] = hid[j] > 0.f ? dh[j] : 0.f;
        hvec[((long)n * 3 + 1) * Cr + j] = hid[Cr + j] > 0.f ? dh[j] : 0.f;
        hvec[((long)n * 3 + 2) * Cr + j] = fmaxf(hid[j], 0.f) + fmaxf(hid[Cr + j], 0.f);
    }
    for (int c = tid; c < C; c += TPB) {
        float da = 0.f, dm = 0.f;
        for (int j = 0; j < Cr; ++j) {
            const float w0 = W0p[c * Cr + j];
            da += w0 * (hid[j] > 0.f ? dh[j] : 0.f);
            dm += w0 * (hid[Cr + j] > 0.f ? dh[j] : 0.f);
        }
        davg[n * C + c] = da; dmx[n * C + c] = dm;
    }
}
// phase 2 (grid over C*Cr): weight gradients summed over images in image order, and the bn2 backward sums
__global__ void ca_bwd_final_kernel(const float* __restrict__ sdu, const float* __restrict__ sdut, const float* __restrict__ ca,
                                    const float* __restrict__ davg, const float* __restrict__ dmx, const float* __restrict__ mean_nc,
                                    const float* __restrict__ tval, const float* __restrict__ mean2, const float* __restrict__ invstd2,
                                    const float* __restrict__ avg, const float* __restrict__ mxv, const float* __restrict__ dz,
                                    const float* __restrict__ hvec, int N, int C, int Cr,
                                    float* __restrict__ sums2, float* __restrict__ dW0, float* __restrict__ dW2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nw = C * Cr;
    if (i < nw) {
        {   // dW0p[c][j] = sum_n dpa[n][j]*avg[n][c] + dpm[n][j]*mx[n][c]
            const int c = i / Cr, j = i - c * Cr;
            float a = 0.f;
            for (int n = 0; n < N; ++n)
                a += hvec[((long)n * 3 + 0) * Cr + j] * avg[(long)n * C + c] + hvec[((long)n * 3 + 1) * Cr + j] * mxv[(long)n * C + c];
            dW0[i] = a;
        }
        {   // dW2p[j][c] = sum_n dz[n][c] * hs[n][j]
            const int j = i / C, c = i - j * C;
            float b = 0.f;
            for (int n = 0; n < N; ++n) b += dz[(long)n * C + c] * hvec[((long)n * 3 + 2) * Cr + j];
            dW2[i] = b;
        }
    }
    if (i < C) {
        const double mu = mean2[i], is = invstd2[i];
        double s0 = 0, s1 = 0;
        for (int n = 0; n < N; ++n) {
            const int k = n * C + i;
            const double c_ = ca[k], du = sdu[k], dut = sdut[k], da = davg[k], dm = dmx[k];
            s0 += c_ * du + da + dm;
            s1 += (c_ * (dut - mu * du) + da * ((double)mean_nc[k] - mu) + dm * ((double)tval[k] - mu)) * is;
        }
        sums2[i] = (float)s1; sums2[C + i] = (float)s0;   // [dgamma | dbeta]
    }
}

// ---- backward 3: dt2 = s2*(du0 - k1 - xhat*k2);  same thread layout as rb_out
__global__ __launch_bounds__(TPB) void rb_bwd3_kernel(const float* __restrict__ dv, int lddv, const float* __restrict__ t2, int ld,
                                                      const float* __restrict__ sa, const float* __restrict__ dsm,
                                                      const int* __restrict__ amax, const float* __restrict__ ca,
                                                      const float* __restrict__ davg, const float* __restrict__ dmx,
                                                      const int* __restrict__ idx, const float* __restrict__ mean2,
                                                      const float* __restrict__ invstd2, const float* __restrict__ s2,
                                                      const float* __restrict__ sums2, float* __restrict__ dt2, int lddt, int HW, int C,
                                                      int pix_per_chunk, float inv_m) {
    const int cvec = C / 4, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    if (row >= rows) return;
    const int n = blockIdx.y, c = col * 4;
    const int p0 = blockIdx.x * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    const float invC = 1.0f / (float)C, invHW = 1.0f / (float)HW;
    // dt2 = s2*( du*ca + davg/HW + [p==idx]*dmx - k1 - (t2-mean)*invstd*k2 ) = du*(s2*ca) + t2*e + f + [p==idx]*(s2*dmx)
    float cc[4], e[4], f[4], dmxs[4];
    int ix[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = n * C + c + q;
        const float s = s2[c + q], is = invstd2[c + q], k1 = sums2[C + c + q] * inv_m, k2 = sums2[c + q] * inv_m;
        cc[q] = s * ca[k];
        e[q] = -s * is * k2;
        f[q] = s * (davg[k] * invHW - k1 + mean2[c + q] * is * k2);
        dmxs[q] = s * dmx[k];
        ix[q] = idx[k];
    }
    const long ib = (long)n * HW;
    for (int p = p0 + row; p < p1; p += rows) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dv + (ib + p) * lddv + c);
        const f32x4 t = *reinterpret_cast<const f32x4*>(t2 + (ib + p) * ld + c);
        const float v = sa[ib + p], g0 = dsm[(ib + p) * 2] * invC, g1 = dsm[(ib + p) * 2 + 1];
        const int am = amax[ib + p];
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float du = d[q] * v + g0 + ((c + q) == am ? g1 : 0.f);
            r[q] = du * cc[q] + t[q] * e[q] + f[q] + (ix[q] == p ? dmxs[q] : 0.f);
        }
        *reinterpret_cast<f32x4*>(dt2 + (ib + p) * lddt + c) = r;
    }
}

// --------------------------------------------------------------------------------- attention gate
// s[p] = bpsi + sum_f wpsi[f] * relu(g1*sg+hg + x1*sx+hx)
__global__ __launch_bounds__(TPB) void ag_psi_kernel(const float* __restrict__ g1, int ldg, const float* __restrict__ x1, int ldx,
                                                     const float* __restrict__ sg, const float* __restrict__ hg,
                                                     const float* __restrict__ sx, const float* __restrict__ hx,
                                                     const float* __restrict__ wpsi, const float* __restrict__ bpsi,
                                                     float* __restrict__ s, long P, int F) {
    const int sub = threadIdx.x & (LPP - 1);
    for (long p = (long)blockIdx.x * PPB + (threadIdx.x / LPP); p < P; p += (long)gridDim.x * PPB) {
        float acc = 0.f;
        for (int f = sub * 4; f < F; f += LPP * 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(g1 + p * ldg + f);
            const f32x4 b = *reinterpret_cast<const f32x4*>(x1 + p * ldx + f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float pre = a[q] * sg[f + q] + hg[f + q] + b[q] * sx[f + q] + hx[f + q];
                acc += wpsi[f + q] * fmaxf(pre, 0.f);
            }
        }
        acc = grp_sum16(acc);
        if (sub == 0) s[p] = acc + bpsi[0];
    }
}
// att[p][c] = xs[p][c] * sigmoid(s[p]*sp + hp);  thread = 4 channels x strided pixels (flat pixel index, no per-image state)
__global__ __launch_bounds__(TPB) void ag_out_kernel(const float* __restrict__ xs, int ldx, const float* __restrict__ s,
                                                     const float* __restrict__ sp, const float* __restrict__ hp,
                                                     float* __restrict__ out, int ldo, long P, int C, long pix_per_chunk) {
    const int cvec = C / 4, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    if (row >= rows) return;
    const int c = col * 4;
    const long p0 = (long)blockIdx.x * pix_per_chunk;
    const long p1 = p0 + pix_per_chunk < P ? p0 + pix_per_chunk : P;
    const float a = sp[0], b = hp[0];
    for (long p = p0 + row; p < p1; p += rows) {
        const float sig = sigmoidf_(s[p] * a + b);
        *reinterpret_cast<f32x4*>(out + p * ldo + c) = *reinterpret_cast<const f32x4*>(xs + p * ldx + c) * sig;
    }
}

// dsbn[p] = (sum_c datt*xs) * sig*(1-sig);  dxs[p][c] = datt*sig
__global__ __launch_bounds__(TPB) void ag_bwd1_kernel(const float* __restrict__ datt, int ldd, const float* __restrict__ xs, int ldx,
                                                      const float* __restrict__ s, const float* __restrict__ sp,
                                                      const float* __restrict__ hp, float* __restrict__ dxs, int lddx,
                                                      float* __restrict__ dsbn, long P, int C) {
    const int sub = threadIdx.x & (LPP - 1);
    const float a = sp[0], b = hp[0];
    for (long p = (long)blockIdx.x * PPB + (threadIdx.x / LPP); p < P; p += (long)gridDim.x * PPB) {
        const float sig = sigmoidf_(s[p] * a + b);
        float acc = 0.f;
        for (int c = sub * 4; c < C; c += LPP * 4) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(datt + p * ldd + c);
            const f32x4 x = *reinterpret_cast<const f32x4*>(xs + p * ldx + c);
            acc += d[0] * x[0] + d[1] * x[1] + d[2] * x[2] + d[3] * x[3];
            *reinterpret_cast<f32x4*>(dxs + p * lddx + c) = d * sig;
        }
        acc = grp_sum16(acc);
        if (sub == 0) dsbn[p] = acc * sig * (1.f - sig);
    }
}
// dpre[p][f] = ds[p]*wpsi[f]*(pre>0) (written);  partial sums: dwpsi[f] = sum ds*relu(pre), dbpsi = sum ds
// BN: also the local BatchNorm-backward sums of BOTH branches (W_g.1 and W_x.1 see the same incoming gradient dpre): columns
// [F+1, 2F+1) sum dpre * ghat, [2F+1, 3F+1) sum dpre (dbeta of both), [3F+1, 4F+1) sum dpre * xhat - the values are in registers here, so the
// two bn_bwd_reduce passes over (dpre, g1) and (dpre, x1) that used to follow are not needed.
template <bool BN>
__global__ __launch_bounds__(TPB) void ag_bwd2_partial(const float* __restrict__ ds, const float* __restrict__ g1, int ldg,
                                                       const float* __restrict__ x1, int ldx, const float* __restrict__ sg,
                                                       const float* __restrict__ hg, const float* __restrict__ sx,
                                                       const float* __restrict__ hx, const float* __restrict__ wpsi,
                                                       float* __restrict__ dpre, int ldp, long P, int F, long pix_per_chunk,
                                                       float* __restrict__ part, const float* __restrict__ mean_g,
                                                       const float* __restrict__ invstd_g, const float* __restrict__ mean_x,
                                                       const float* __restrict__ invstd_x) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int NC = BN ? 4 * F + 1 : F + 1;
    const int cvec = F / 4, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    const long p0 = (long)blockIdx.x * pix_per_chunk;
    const long p1 = p0 + pix_per_chunk < P ? p0 + pix_per_chunk : P;
    float sw[4] = {0, 0, 0, 0}, sb = 0.f;
    float so[4] = {0, 0, 0, 0}, sog[4] = {0, 0, 0, 0}, sox[4] = {0, 0, 0, 0};
    if (row < rows) {
        float a_g[4], b_g[4], a_x[4], b_x[4], wp[4], mg[4], ig[4], mx[4], ix[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = col * 4 + q;
            a_g[q] = sg[f]; b_g[q] = hg[f]; a_x[q] = sx[f]; b_x[q] = hx[f]; wp[q] = wpsi[f];
            if constexpr (BN) { mg[q] = mean_g[f]; ig[q] = invstd_g[f]; mx[q] = mean_x[f]; ix[q] = invstd_x[f]; }
        }
        for (long p = p0 + row; p < p1; p += rows) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(g1 + p * ldg + col * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(x1 + p * ldx + col * 4);
            const float d = ds[p];
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float pre = a[q] * a_g[q] + b_g[q] + b[q] * a_x[q] + b_x[q];
                o[q] = pre > 0.f ? d * wp[q] : 0.f;
                sw[q] += d * fmaxf(pre, 0.f);
                if constexpr (BN) {
                    so[q] += o[q];
                    sog[q] += o[q] * ((a[q] - mg[q]) * ig[q]);
                    sox[q] += o[q] * ((b[q] - mx[q]) * ix[q]);
                }
            }
            if (col == 0) sb += d;
            *reinterpret_cast<f32x4*>(dpre + p * ldp + col * 4) = o;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sm[row * NC + col * 4 + q] = sw[q];
            if constexpr (BN) {
                sm[row * NC + F + 1 + col * 4 + q] = sog[q];
                sm[row * NC + 2 * F + 1 + col * 4 + q] = so[q];
                sm[row * NC + 3 * F + 1 + col * 4 + q] = sox[q];
            }
        }
        if (col == 0) sm[row * NC + F] = sb;
    }
    __syncthreads();
    for (int c = tid; c < NC; c += TPB) {
        double a = 0;
        for (int r = 0; r < rows; ++r) a += sm[r * NC + c];
        part[(long)blockIdx.x * NC + c] = (float)a;
    }
}
// column sums of ag_bwd2_partial<true>'s partials to their three destinations: (dwpsi | dbpsi), W_g.1's (dgamma | dbeta), W_x.1's (dgamma | dbeta)
__global__ void ag_bwd2_bn_final(const float* __restrict__ part, long nrows, int F, float* __restrict__ dwpsi_db, float* __restrict__ sums_g,
                                 float* __restrict__ sums_x) {
    const int c = blockIdx.x, NC = 4 * F + 1;
    double acc = 0;
    for (long r = threadIdx.x; r < nrows; r += blockDim.x) acc += part[r * NC + c];
    acc = wave_sum_d(acc);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(red[0] + red[1] + red[2] + red[3]);
        if (c <= F) dwpsi_db[c] = v;
        else if (c < 2 * F + 1) sums_g[c - (F + 1)] = v;                                  // dgamma of W_g.1
        else if (c < 3 * F + 1) { sums_g[F + c - (2 * F + 1)] = v; sums_x[F + c - (2 * F + 1)] = v; }      // dbeta of both
        else sums_x[c - (3 * F + 1)] = v;                                                 // dgamma of W_x.1
    }
}

// --------------------------------------------------------------------------------- output conv (64 -> 1) + sigmoid
__global__ __launch_bounds__(TPB) void outc_fwd_kernel(const float* __restrict__ x, int ld, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ logit,
                                                       float* __restrict__ prob, long P, int C) {
    const int sub = threadIdx.x & (LPP - 1);
    for (long p = (long)blockIdx.x * PPB + (threadIdx.x / LPP); p < P; p += (long)gridDim.x * PPB) {
        float acc = 0.f;
        for (int c = sub * 4; c < C; c += LPP * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * ld + c);
            const f32x4 k = *reinterpret_cast<const f32x4*>(w + c);
            acc += v[0] * k[0] + v[1] * k[1] + v[2] * k[2] + v[3] * k[3];
        }
        acc = grp_sum16(acc);
        if (sub == 0) {
            const float l = acc + b[0];
            if (logit) logit[p] = l;
            prob[p] = sigmoidf_(l);
        }
    }
}
// dx[p][c] = dl*w[c];  partial dw[c] = sum dl*x[p][c], db = sum dl;  dl = dprob*prob*(1-prob)
__global__ __launch_bounds__(TPB) void outc_bwd_partial(const float* __restrict__ dprob, const float* __restrict__ prob,
                                                        const float* __restrict__ x, int ld, const float* __restrict__ w,
                                                        float* __restrict__ dx, int lddx, long P, int C, long pix_per_chunk,
                                                        float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int cvec = C / 4, rows = TPB / cvec, tid = threadIdx.x;
    const int col = tid % cvec, row = tid / cvec;
    const long p0 = (long)blockIdx.x * pix_per_chunk;
    const long p1 = p0 + pix_per_chunk < P ? p0 + pix_per_chunk : P;
    float sw[4] = {0, 0, 0, 0}, sb = 0.f;
    if (row < rows) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + col * 4);
        for (long p = p0 + row; p < p1; p += rows) {
            const float pr = prob[p];
            const float dl = dprob[p] * pr * (1.f - pr);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * ld + col * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) sw[q] += dl * v[q];
            if (col == 0) sb += dl;
            *reinterpret_cast<f32x4*>(dx + p * lddx + col * 4) = wv * dl;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sm[row * (C + 1) + col * 4 + q] = sw[q];
        if (col == 0) sm[row * (C + 1) + C] = sb;
    }
    __syncthreads();
    for (int c = tid; c < C + 1; c += TPB) {
        double a = 0;
        for (int r = 0; r < rows; ++r) a += sm[r * (C + 1) + c];
        part[(long)blockIdx.x * (C + 1) + c] = (float)a;
    }
}

inline int px_grid(long P) {
    long b = (P + PPB - 1) / PPB;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}
inline int ew_grid(long total_vec) {
    long b = (total_vec + TPB - 1) / TPB;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}
inline int stream_chunks(int HW, int C, int& ppc) {
    const int rows = TPB / (C / 4);
    long per_img = ((long)HW * C + 16383) / 16384;
    const long maxc = (HW + rows - 1) / rows;
    if (per_img > maxc) per_img = maxc;
    if (per_img > 4096) per_img = 4096;
    if (per_img < 1) per_img = 1;
    ppc = (int)((HW + per_img - 1) / per_img);
    return (int)((HW + ppc - 1) / ppc);
}
inline void chunking(long P, int C, long& chunks, long& ppc) {
    chunks = (P * C + 32767) / 32768;
    if (chunks > 2048) chunks = 2048;
    if (chunks < 1) chunks = 1;
    ppc = (P + chunks - 1) / chunks;
    chunks = (P + ppc - 1) / ppc;
}
}  // namespace

#define REQ_C4(C) RUNET_REQUIRE((C) >= 4 && (C) <= 1024 && (C) % 4 == 0, "channels must be a multiple of 4 in [4, 1024]")

extern "C" int runet_ca_coeff(const float* mean_nc, const float* max_nc, const float* min_nc, const int* imax_nc, const int* imin_nc,
                              const float* s2, const float* h2, const float* w0p, const float* w2p, int n_img, int c, int cr,
                              float* A, float* B, float* ca, float* avg, float* mx, int* idx, float* tval, void* stream) {
    RUNET_REQUIRE(mean_nc && max_nc && min_nc && imax_nc && imin_nc && s2 && h2 && w0p && w2p && A && B, "null pointer");
    REQ_C4(c);
    RUNET_REQUIRE(cr >= 1 && cr <= c, "bad hidden width");
    RUNET_REQUIRE(!avg || (mx && idx && tval), "save buffers must come together");
    const size_t lds = (2 * c + 2 * cr + 2 * CA_TPB) * sizeof(float);
    hipLaunchKernelGGL(ca_coeff_kernel, dim3(n_img), dim3(CA_TPB), lds, (hipStream_t)stream, mean_nc, max_nc, min_nc, imax_nc, imin_nc, s2, h2,
                       w0p, w2p, c, cr, A, B, ca, avg, mx, idx, tval);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_sa_reduce(const float* t2, int ld, const float* A, const float* B, long pixels, int hw, int c, float* smap,
                               int* amax, void* stream) {
    RUNET_REQUIRE(t2 && A && B && smap && amax, "null pointer");
    REQ_C4(c);
    hipLaunchKernelGGL(sa_reduce_kernel, dim3(px_grid(pixels)), dim3(TPB), 0, (hipStream_t)stream, t2, ld, A, B, pixels, hw, c, smap, amax);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_sa_conv7(const float* smap, const float* wp, float* sa, int n_img, int h, int w, void* stream) {
    RUNET_REQUIRE(smap && wp && sa, "null pointer");
    hipLaunchKernelGGL(sa_conv7_kernel, dim3(cdiv(w, 16), cdiv(h, 16), n_img), dim3(256), 0, (hipStream_t)stream, smap, wp, sa, h, w);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_rb_out(const float* t2, int ld, const float* A, const float* B, const float* sa, const float* r, int ldr,
                            const float* rs, const float* rh, float* out, int ldo, long pixels, int hw, int c, void* stream) {
    RUNET_REQUIRE(t2 && A && B && sa && r && out, "null pointer");
    REQ_C4(c);
    RUNET_REQUIRE(pixels % hw == 0, "pixels must be a whole number of images");
    int ppc;
    const int chunks = stream_chunks(hw, c, ppc);
    hipLaunchKernelGGL(rb_out_kernel, dim3(chunks, (int)(pixels / hw)), dim3(TPB), 0, (hipStream_t)stream, t2, ld, A, B, sa, r, ldr, rs, rh, out,
                       ldo, hw, c, ppc);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_rb_bwd1(const float* dout, int lddo, const float* out, int ldo, const float* t2, int ld, const float* A,
                             const float* B, const float* sa, float* dv, int lddv, float* dq, long pixels, int hw, int c, void* stream) {
    RUNET_REQUIRE(dout && t2 && A && B && sa && dv && dq, "null pointer");
    REQ_C4(c);
    hipLaunchKernelGGL(rb_bwd1_kernel, dim3(px_grid(pixels)), dim3(TPB), 0, (hipStream_t)stream, dout, lddo, out, ldo, t2, ld, A, B, sa, dv, lddv,
                       dq, pixels, hw, c);
    RUNET_CHECK_LAUNCH();
}

extern "C" long runet_sa_conv7_bwd_workspace_floats(int n_img, int h, int w) { return (long)n_img * cdiv(h, 16) * cdiv(w, 16) * 98; }

extern "C" int runet_sa_conv7_bwd(const float* smap, const float* dq, const float* wp, float* dsm, float* dwp, float* workspace,
                                  int n_img, int h, int w, void* stream) {
    RUNET_REQUIRE(smap && dq && wp && dsm && dwp && workspace, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(w, 16), cdiv(h, 16), n_img);
    hipLaunchKernelGGL(sa_conv7_bwd_kernel, grid, dim3(256), 0, st, smap, dq, wp, dsm, workspace, h, w);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(98), dim3(256), 0, st, workspace, (long)grid.x * grid.y * grid.z, 98, dwp);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_rb_bwd2(const float* dv, int lddv, const float* t2, int ld, const float* sa, const float* dsm, const int* amax,
                             int n_img, int hw, int c, float* workspace, float* sdu, float* sdut, void* stream) {
    RUNET_REQUIRE(dv && t2 && sa && dsm && amax && workspace && sdu && sdut, "null pointer");
    REQ_C4(c);
    hipStream_t st = (hipStream_t)stream;
    const int rows = TPB / (c / 4);
    long per_img = ((long)hw * c + 32767) / 32768;
    if (per_img > 1024) per_img = 1024;
    const long maxc = (hw + rows - 1) / rows;
    if (per_img > maxc) per_img = maxc;
    if (per_img < 1) per_img = 1;
    const int ppc = (int)((hw + per_img - 1) / per_img);
    const size_t lds = (size_t)rows * c * 2 * sizeof(float);
    hipLaunchKernelGGL(rb_bwd2_partial<false>, dim3((int)per_img, n_img), dim3(TPB), lds, st, dv, lddv, t2, ld, sa, dsm, amax, hw, c, ppc, workspace,
                       nullptr, 0, nullptr, nullptr);
    hipLaunchKernelGGL(rb_bwd2_final, dim3(cdiv((long)n_img * c, 128)), dim3(128), 0, st, workspace, n_img, c, (int)per_img, 2, sdu, sdut);
    RUNET_CHECK_LAUNCH();
}

static long rb_bwd2_chunks(int hw, int c, int& ppc) {
    const int rows = TPB / (c / 4);
    long per_img = ((long)hw * c + 32767) / 32768;
    if (per_img > 1024) per_img = 1024;
    const long maxc = (hw + rows - 1) / rows;
    if (per_img > maxc) per_img = maxc;
    if (per_img < 1) per_img = 1;
    ppc = (int)((hw + per_img - 1) / per_img);
    return per_img;
}
extern "C" long runet_rb_bwd2_bn_workspace_floats(int n_img, int hw, int c) {
    int ppc;
    return (long)n_img * rb_bwd2_chunks(hw, c, ppc) * c * 4 + 64;
}
// runet_rb_bwd2 that also leaves the LOCAL BatchNorm-backward sums of the block's shortcut BatchNorm (Main_Final.py:173) behind: dv is that
// BatchNorm's incoming gradient, r [n, hw, c] its input; sums_s [2c] = (sum dv * rhat | sum dv), what runet_bn_bwd_reduce(dv, r) computes.
extern "C" int runet_rb_bwd2_bn(const float* dv, int lddv, const float* t2, int ld, const float* sa, const float* dsm, const int* amax, const float* r,
                                int ldr, const float* mean_s, const float* invstd_s, int n_img, int hw, int c, float* workspace,
                                long workspace_floats, float* sdu, float* sdut, float* sums_s, void* stream) {
    RUNET_REQUIRE(dv && t2 && sa && dsm && amax && r && mean_s && invstd_s && workspace && sdu && sdut && sums_s, "null pointer");
    REQ_C4(c);
    RUNET_REQUIRE(ldr >= c && ldr % 4 == 0 && workspace_floats >= runet_rb_bwd2_bn_workspace_floats(n_img, hw, c), "bad stride / workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int ppc;
    const long per_img = rb_bwd2_chunks(hw, c, ppc);
    const int rows = TPB / (c / 4);
    const size_t lds = (size_t)rows * c * 4 * sizeof(float);
    RUNET_REQUIRE(lds <= 64 * 1024, "LDS budget");
    hipLaunchKernelGGL(rb_bwd2_partial<true>, dim3((int)per_img, n_img), dim3(TPB), lds, st, dv, lddv, t2, ld, sa, dsm, amax, hw, c, ppc, workspace, r, ldr,
                       mean_s, invstd_s);
    hipLaunchKernelGGL(rb_bwd2_final, dim3(cdiv((long)n_img * c, 128)), dim3(128), 0, st, workspace, n_img, c, (int)per_img, 4, sdu, sdut);
    hipLaunchKernelGGL(rb_bwd2_bn_final, dim3(c), dim3(256), 0, st, workspace, (long)n_img * per_img, c, sums_s);
    RUNET_CHECK_LAUNCH();
}

extern "C" long runet_ca_bwd_workspace_floats(int n_img, int c, int cr) { return (long)n_img * (c + 3L * cr) + 64; }

extern "C" int runet_ca_bwd(const float* sdu, const float* sdut, const float* s2, const float* h2, const float* ca, const float* avg,
                            const float* mx, const float* w0p, const float* w2p, const float* mean_nc, const float* tval,
                            const float* mean2, const float* invstd2, int n_img, int c, int cr, float* workspace, float* davg,
                            float* dmx, float* sums2, float* dw0p, float* dw2p, void* stream) {
    RUNET_REQUIRE(sdu && sdut && s2 && h2 && ca && avg && mx && w0p && w2p && mean_nc && tval && mean2 && invstd2 && workspace && davg && dmx &&
                  sums2 && dw0p && dw2p, "null pointer");
    REQ_C4(c);
    hipStream_t st = (hipStream_t)stream;
    float* dz = workspace;                              // [n_img][c]
    float* hvec = workspace + (long)n_img * c;          // [n_img][3][cr]
    const size_t lds = (c + 3 * cr + 2 * CA_TPB) * sizeof(float);
    hipLaunchKernelGGL(ca_bwd_image_kernel, dim3(n_img), dim3(CA_TPB), lds, st, sdu, sdut, s2, h2, ca, avg, mx, w0p, w2p, c, cr, davg, dmx, dz, hvec);
    const int tot = c * cr > c ? c * cr : c;
    hipLaunchKernelGGL(ca_bwd_final_kernel, dim3(cdiv(tot, 128)), dim3(128), 0, st, sdu, sdut, ca, davg, dmx, mean_nc, tval, mean2, invstd2, avg, mx,
                       dz, hvec, n_img, c, cr, sums2, dw0p, dw2p);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_rb_bwd3(const float* dv, int lddv, const float* t2, int ld, const float* sa, const float* dsm, const int* amax,
                             const float* ca, const float* davg, const float* dmx, const int* idx, const float* mean2,
                             const float* invstd2, const float* s2, const float* sums2, float* dt2, int lddt, long pixels, int hw, int c,
                             long m_total, void* stream) {
    RUNET_REQUIRE(dv && t2 && sa && dsm && amax && ca && davg && dmx && idx && mean2 && invstd2 && s2 && sums2 && dt2, "null pointer");
    REQ_C4(c);
    RUNET_REQUIRE(pixels % hw == 0, "pixels must be a whole number of images");
    int ppc;
    const int chunks = stream_chunks(hw, c, ppc);
    hipLaunchKernelGGL(rb_bwd3_kernel, dim3(chunks, (int)(pixels / hw)), dim3(TPB), 0, (hipStream_t)stream, dv, lddv, t2, ld, sa, dsm, amax, ca, davg,
                       dmx, idx, mean2, invstd2, s2, sums2, dt2, lddt, hw, c, ppc, 1.0f / (float)(m_total > 0 ? m_total : pixels));
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_ag_psi(const float* g1, int ldg, const float* x1, int ldx, const float* sg, const float* hg, const float* sx,
                            const float* hx, const float* wpsi, const float* bpsi, float* s, long pixels, int f, void* stream) {
    RUNET_REQUIRE(g1 && x1 && sg && hg && sx && hx && wpsi && bpsi && s, "null pointer");
    REQ_C4(f);
    hipLaunchKernelGGL(ag_psi_kernel, dim3(px_grid(pixels)), dim3(TPB), 0, (hipStream_t)stream, g1, ldg, x1, ldx, sg, hg, sx, hx, wpsi, bpsi, s, pixels, f);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_ag_out(const float* xs, int ldx, const float* s, const float* sp, const float* hp, float* out, int ldo, long pixels,
                            int c, void* stream) {
    RUNET_REQUIRE(xs && s && sp && hp && out, "null pointer");
    REQ_C4(c);
    long chunks, ppc;
    chunking(pixels, c, chunks, ppc);
    hipLaunchKernelGGL(ag_out_kernel, dim3((int)chunks), dim3(TPB), 0, (hipStream_t)stream, xs, ldx, s, sp, hp, out, ldo, pixels, c, ppc);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_ag_bwd1(const float* datt, int ldd, const float* xs, int ldx, const float* s, const float* sp, const float* hp,
                             float* dxs, int lddx, float* dsbn, long pixels, int c, void* stream) {
    RUNET_REQUIRE(datt && xs && s && sp && hp && dxs && dsbn, "null pointer");
    REQ_C4(c);
    hipLaunchKernelGGL(ag_bwd1_kernel, dim3(px_grid(pixels)), dim3(TPB), 0, (hipStream_t)stream, datt, ldd, xs, ldx, s, sp, hp, dxs, lddx, dsbn, pixels, c);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_ag_bwd2(const float* ds, const float* g1, int ldg, const float* x1, int ldx, const float* sg, const float* hg,
                             const float* sx, const float* hx, const float* wpsi, float* dpre, int ldp, float* workspace,
                             float* dwpsi_db, long pixels, int f, void* stream) {
    RUNET_REQUIRE(ds && g1 && x1 && sg && hg && sx && hx && wpsi && dpre && workspace && dwpsi_db, "null pointer");
    REQ_C4(f);
    hipStream_t st = (hipStream_t)stream;
    long chunks, ppc;
    chunking(pixels, f, chunks, ppc);
    const int rows = TPB / (f / 4);
    const size_t lds = (size_t)rows * (f + 1) * sizeof(float);
    hipLaunchKernelGGL(ag_bwd2_partial<false>, dim3((int)chunks), dim3(TPB), lds, st, ds, g1, ldg, x1, ldx, sg, hg, sx, hx, wpsi, dpre, ldp, pixels, f, ppc,
                       workspace, nullptr, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(f + 1), dim3(256), 0, st, workspace, chunks, f + 1, dwpsi_db);
    RUNET_CHECK_LAUNCH();
}

extern "C" long runet_ag_bwd2_bn_workspace_floats(long pixels, int f) {
    long chunks, ppc;
    chunking(pixels, f, chunks, ppc);
    return chunks * (4L * f + 1) + 64;
}

// runet_ag_bwd2 that also leaves the local BatchNorm-backward sums of W_g.1 and W_x.1 (Main_Final.py:127,132) behind: sums_g / sums_x [2f] =
// (sum dpre * xhat | sum dpre) as runet_bn_bwd_reduce would compute them from (dpre, g1) and (dpre, x1) - two passes over three tensors less.
extern "C" int runet_ag_bwd2_bn(const float* ds, const float* g1, int ldg, const float* x1, int ldx, const float* sg, const float* hg,
                                const float* sx, const float* hx, const float* wpsi, const float* mean_g, const float* invstd_g,
                                const float* mean_x, const float* invstd_x, float* dpre, int ldp, float* workspace, long workspace_floats,
                                float* dwpsi_db, float* sums_g, float* sums_x, long pixels, int f, void* stream) {
    RUNET_REQUIRE(ds && g1 && x1 && sg && hg && sx && hx && wpsi && mean_g && invstd_g && mean_x && invstd_x && dpre && workspace && dwpsi_db &&
                  sums_g && sums_x, "null pointer");
    REQ_C4(f);
    RUNET_REQUIRE(workspace_floats >= runet_ag_bwd2_bn_workspace_floats(pixels, f), "workspace too small (runet_ag_bwd2_bn_workspace_floats)");
    hipStream_t st = (hipStream_t)stream;
    long chunks, ppc;
    chunking(pixels, f, chunks, ppc);
    const int rows = TPB / (f / 4);
    const size_t lds = (size_t)rows * (4 * f + 1) * sizeof(float);
    RUNET_REQUIRE(lds <= 64 * 1024, "LDS budget");
    hipLaunchKernelGGL(ag_bwd2_partial<true>, dim3((int)chunks), dim3(TPB), lds, st, ds, g1, ldg, x1, ldx, sg, hg, sx, hx, wpsi, dpre, ldp, pixels, f, ppc,
                       workspace, mean_g, invstd_g, mean_x, invstd_x);
    hipLaunchKernelGGL(ag_bwd2_bn_final, dim3(4 * f + 1), dim3(256), 0, st, workspace, chunks, f, dwpsi_db, sums_g, sums_x);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_outc_fwd(const float* x, int ld, const float* w, const float* b, float* logit, float* prob, long pixels, int c,
                              void* stream) {
    RUNET_REQUIRE(x && w && b && prob, "null pointer");
    REQ_C4(c);
    hipLaunchKernelGGL(outc_fwd_kernel, dim3(px_grid(pixels)), dim3(TPB), 0, (hipStream_t)stream, x, ld, w, b, logit, prob, pixels, c);
    RUNET_CHECK_LAUNCH();
}

extern "C" int runet_outc_bwd(const float* dprob, const float* prob, const float* x, int ld, const float* w, float* dx, int lddx,
                              float* workspace, float* dw_db, long pixels, int c, void* stream) {
    RUNET_REQUIRE(dprob && prob && x && w && dx && workspace && dw_db, "null pointer");
    REQ_C4(c);
    hipStream_t st = (hipStream_t)stream;
    long chunks, ppc;
    chunking(pixels, c, chunks, ppc);
    const int rows = TPB / (c / 4);
    const size_t lds = (size_t)rows * (c + 1) * sizeof(float);
    hipLaunchKernelGGL(outc_bwd_partial, dim3((int)chunks), dim3(TPB), lds, st, dprob, prob, x, ld, w, dx, lddx, pixels, c, ppc, workspace);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(c + 1), dim3(256), 0, st, workspace, chunks, c + 1, dw_db);
    RUNET_CHECK_LAUNCH();
}
