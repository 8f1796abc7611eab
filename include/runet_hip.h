/* runet_hip.h — C ABI of librunet_hip.so: the gfx950 (MI355X) kernels behind the Robust U-Net
 * training path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes + a HIP stream handle
 * (`void* stream` = hipStream_t; NULL = default stream), launches asynchronously on that stream,
 * allocates nothing, keeps no global state (except the thread-local last-error string) and returns
 * 0 on success / non-zero on error (`runet_last_error()` has the text).  Arguments are validated
 * on the host before any launch: a bad shape is an error code, never an out-of-bounds kernel.
 *
 * The reference (UofgCoastline/EUSIPCO-2026-Robust-Unet) is pure Python on top of torch; it has no
 * FFI of its own.  Each group below names the reference call site(s) (file:line in /root/reference)
 * whose ATen kernels it stands in for; the Python binding that a maintainer of the reference would
 * add is shown in INTEGRATION.md (ctypes) and implemented in eusipco-2026-robust-unet_amd/_lib.py.
 *
 * Layouts: activations are NHWC fp32; `ld*` is the pixel stride in floats, so a tensor may be a
 * channel slice of a wider (concat) buffer.  Convolution weights are "HWIO": w[kh][kw][cin][cout].
 */
#ifndef RUNET_HIP_H
#define RUNET_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* runet_last_error(void);
void runet_set_error(const char* msg);
int runet_abi_version(void);

/* ---- convolutions (Main_Final.py:157,159,172 ResidualBlock; :126,131 AttentionGate 1x1; :205-208
 *      DilatedBlock; :261-270 ConvTranspose2d k2 s2; autograd of the same via :581 loss.backward()) ---- */
enum { RUNET_CONV_FWD = 0, RUNET_CONV_DGRAD = 1, RUNET_CONVT_FWD = 2, RUNET_CONVT_DGRAD = 3, RUNET_CONV_DGRAD_T = 4, RUNET_CONVT_DGRAD_T = 5 };

/* mode FWD   : y[n,h,w,0:cout] (=|+=) bias + conv_{kh x kw, dilation dil, 'same' zero padding}(x[n,h,w,0:cin], w[kh,kw,cin_w,cout])
 *              cin is the channel count READ from x (multiple of 4, zero-padded by the caller);
 *              cin_w <= cin is the number of input channels present in w (3 for the RGB stem).
 * mode DGRAD : x := dy[n,h,w,0:cin] (cin = conv Cout), y := dx[n,h,w,0:cout] (cout = conv Cin),
 *              w = the forward weight [kh,kw,cout,cin]; computes the data gradient.
 * mode CONVT_FWD  : y[n,2h,2w,0:cout] = bias + convT_{2x2,s2}(x[n,h,w,0:cin]),  w[2,2,cin,cout]
 * mode CONVT_DGRAD: x := dy[n,2h,2w,0:cin], y := dx[n,h,w,0:cout],  w[2,2,cout,cin]
 * modes DGRAD_T / CONVT_DGRAD_T: the same data gradients with the weight already transposed per tap, w [taps][cin][cout] in THIS call's
 *              naming (runet_transpose_taps of the forward weight, cached per optimizer step): the weight tile is then staged with the
 *              n-contiguous loader of the forward kernels instead of the k-contiguous one (scalar LDS writes).
 * accumulate != 0 adds into y instead of overwriting.  bias may be NULL. */
int runet_conv_igemm(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                     int n_img, int h, int w_, int cin, int cin_w, int cout, int kh, int kw, int dil,
                     int mode, int accumulate, void* stream);

/* wt[tap][co][ci] = w[tap][ci][co] */
int runet_transpose_taps(const float* w, float* wt, int taps, int cin, int cout, void* stream);

/* name of the kernel instantiation runet_conv_igemm launches for this shape (as rocprofv3 prints it, minus the
 * namespace), so that bench.py's live timings can be matched with the profiler's per-kernel rows */
const char* runet_conv_igemm_kernel_name(int n_img, int h, int w_, int cin, int cout, int kh, int mode);
/* measurement hook (tools/igemm_variants.py): force the tile variant of runet_conv_igemm, 0..4 = 128x32, 256x64, 128x64, 128x128, 64x64; -1 = automatic */
int runet_igemm_force_variant(int variant);
/* the same for runet_conv_igemm_bf16 / _fp16: 0..2 = 128 pixels x 32, 64, 128 output channels; -1 = automatic */
int runet_igemm_lowp_force_variant(int variant);

/* dw[kh,kw,cin_w,cout] = sum_pixels x (x) dy  (weight gradient; transposed != 0: dy is [n,2h,2w,cout]
 * and the 2x2 taps index the dy pixel).  `workspace` (>= runet_conv_wgrad_workspace_floats floats, may be
 * NULL) holds split-K partial slabs that are summed in a fixed order. */
long runet_conv_wgrad_workspace_floats(int n_img, int h, int w_, int cin_w, int cout, int kh, int kw);
int runet_conv_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace,
                     long workspace_floats, int n_img, int h, int w_, int cin, int cin_w, int cout,
                     int kh, int kw, int dil, int transposed, void* stream);


/* ---- Winograd F(2x2,3x3) path for the 3x3 / dilation-1 / 'same' convolutions (Main_Final.py:157,159), forward and data gradient ----
 * runet_wino_weights: U[16][K][N] = G g G^T of w[3][3][cin][cout]; dgrad = 0: K = cin, N = cout;  dgrad != 0: the 180-degree rotated
 * filter with K = cout, N = cin (so that runet_wino_conv(dy, U') is the data gradient).
 * runet_wino_conv: y[n,h,w,0:N] (=|+=) bias + winograd_conv(x[n,h,w,0:K], U).  Needs H, W even, K % 16 == 0, N even. */
int runet_wino_supported(int h, int w, int cin, int cout);
/* shape supported AND the tensors fit the kernel's 32-bit buffer offsets (n_img*h*w*ld < 2^29 floats): callers fall back to
 * runet_conv_igemm (64-bit addressing) when this is 0, e.g. 16 x 512 x 512 with a 128-channel concat input */
int runet_wino_fits(int n_img, int h, int w, int ldx, int ldy, int cin, int cout);
int runet_wino_weights(const float* w_hwio, float* U, int cin, int cout, int dgrad, void* stream);
int runet_wino_conv(const float* x, int ldx, const float* U, const float* bias, float* y, int ldy, int n_img, int h, int w, int k, int n,
                    int accumulate, void* stream);

/* The same convolution with its sixteen position products on the BF16 matrix cores (csrc/conv_winograd_x3.hip: split operands, six bf16
 * MFMAs per fp32 product, fp32-accurate).  runet_wino_weights_x3 writes G g G^T straight into the split planes Up[16][3][K/8][N][8] bf16
 * (runet_wino_x3_pack_elems(K, N) 2-byte elements, once per optimizer step); runet_wino_conv_x3 = runet_wino_conv on them (ldx % 4 == 0,
 * x 16-byte aligned). */
long runet_wino_x3_pack_elems(int k, int n);
int runet_wino_weights_x3(const float* w_hwio, void* Upacked, int cin, int cout, int dgrad, void* stream);
int runet_wino_conv_x3(const float* x, int ldx, const void* Upacked, const float* bias, float* y, int ldy, int n_img, int h, int w, int k, int n,
                       int accumulate, void* stream);
/* ... and with the BatchNorm statistics of its output (Main_Final.py:158,160: the BatchNorm2d behind each convolution) taken in the epilogue
 * instead of by a pass of runet_bn_stats over the tensor: stats [runet_wino_conv_x3_stats_parts(n_img, h, w)][n][3] = (count, mean, M2) per
 * block of pixels; runet_bn_stats_finalize turns them into the coefficients (the second kernel of runet_bn_stats). */
int runet_wino_conv_x3_stats_parts(int n_img, int h, int w);
int runet_wino_conv_x3_stats(const float* x, int ldx, const void* Upacked, const float* bias, float* y, int ldy, int n_img, int h, int w, int k,
                             int n, int accumulate, float* stats, void* stream);

/* Winograd-domain weight gradient of the same convolutions: dw[3][3][cin][cout] = G^T [sum_tiles (B^T x B).*(A dy A^T)] G.
 * workspace: >= runet_wino_wgrad_workspace_floats floats (partial slabs, summed in a fixed order).  H, W even. */
long runet_wino_wgrad_workspace_floats(int n_img, int h, int w, int cin, int cout);
int runet_wino_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats, int n_img,
                     int h, int w, int cin, int cout, void* stream);

/* ---- channel statistics / BatchNorm2d / ReLU / Dropout2d (Main_Final.py:158,160,162,163,173,127,132,137,210,211) ----
 * Scratch buffers ("workspace") are caller-owned; runet_reduce_workspace_floats gives a sufficient size. */
long runet_reduce_workspace_floats(int n_img, int hw, int c);

/* per-(image, channel) mean and M2 (= sum of squared deviations) of x[n, hw, 0:c]; optionally max/min and the
 * pixel index (within the image) of their first occurrence.  c is 1 or a multiple of 4, <= 1024. */
int runet_chan_stats(const float* x, int ld, int n_img, int hw, int c, float* workspace, float* mean_nc, float* m2_nc,
                     float* max_nc, float* min_nc, int* imax_nc, int* imin_nc, int want_minmax, void* stream);

/* training != 0: batch statistics over n_img*hw values from (mean_nc, m2_nc); updates run_mean/run_var (momentum,
 * unbiased variance) and *num_batches_tracked += 1 when those pointers are non-NULL; writes save_mean/save_invstd.
 * training == 0: statistics = run_mean/run_var.  Outputs scale = gamma*invstd, shift = beta - mean*scale. */
/* coefficients from statistics partials part[nparts][c][3] = (count, mean, M2) of disjoint pixel sets (written by runet_conv_x3_stats /
 * runet_wino_conv_x3_stats): the second kernel of runet_bn_stats on its own */
int runet_bn_stats_finalize(const float* part, int nparts, int c, const float* gamma, const float* beta, float* run_mean, float* run_var,
                            long long* num_batches_tracked, float momentum, float eps, float* scale, float* shift, float* save_mean,
                            float* save_invstd, void* stream);
int runet_bn_finalize(const float* mean_nc, const float* m2_nc, int n_img, int c, long hw, const float* gamma, const float* beta,
                      float* run_mean, float* run_var, long long* num_batches_tracked, float momentum, float eps, int training,
                      float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);

/* runet_chan_stats + runet_bn_finalize(training) without the per-image outputs, in two launches instead of three (the forward of
 * nn.BatchNorm2d in training mode, Main_Final.py:158,160, wherever no per-image statistics are needed): partials per chunk, then one
 * kernel that Chan-combines all of a channel's partials and derives scale / shift / saved and running statistics. */
int runet_bn_stats(const float* x, int ld, int n_img, int hw, int c, float* workspace, const float* gamma, const float* beta,
                   float* run_mean, float* run_var, long long* num_batches_tracked, float momentum, float eps, float* scale,
                   float* shift, float* save_mean, float* save_invstd, void* stream);

/* y = [relu](x*scale[c] + shift[c]) [* mask_nc[n, c]]   (mask = Dropout2d keep-mask already divided by 1-p, or NULL) */
int runet_bn_apply(const float* x, int ldx, float* y, int ldy, long pixels, int hw, int c, const float* scale, const float* shift,
                   const float* mask_nc, int relu, void* stream);

/* g = dy, or dy*mask_nc[n,c]*(act > 0) when act != NULL (act = saved output of relu/dropout), or - relu_scale/relu_shift given, act NULL -
 * dy*mask_nc[n,c]*(x*relu_scale[c] + relu_shift[c] > 0): the forward's affine recomputed from x, one tensor less to read.
 * sums[0:c] = sum g*xhat (= dgamma), sums[c:2c] = sum g (= dbeta): the order of (weight, bias). */
int runet_bn_bwd_reduce(const float* dy, int lddy, const float* x, int ldx, const float* act, int ldact, int n_img, int hw, int c,
                        const float* mean, const float* invstd, const float* mask_nc, float* workspace, float* sums, const float* relu_scale,
                        const float* relu_shift, void* stream);
/* dx = scale*(g - sums[C+c]/M - xhat*sums[c]/M), M = m_total if > 0 else pixels (SyncBN: sums all-reduced, M = global count);
 * relu_shift (act NULL): g as above with relu_scale = scale */
int runet_bn_bwd_apply(const float* dy, int lddy, const float* x, int ldx, const float* act, int ldact, float* dx, int lddx,
                       long pixels, int hw, int c, const float* mean, const float* invstd, const float* scale, const float* sums,
                       const float* mask_nc, long m_total, const float* relu_shift, void* stream);
/* out[c] (=|+=) sum over pixels of x[p, c]  (bias gradients) */
int runet_chan_sum(const float* x, int ld, long pixels, int c, float* workspace, float* out, int accumulate, void* stream);

/* ---- ChannelAttention / SpatialAttention / ResidualBlock tail (Main_Final.py:82-117, 186-194) ----
 * w0p = ca.fc.0 weight as [C][Cr], w2p = ca.fc.2 weight as [Cr][C], wp = sa.conv1 weight as [7][7][2]. */
int runet_ca_coeff(const float* mean_nc, const float* max_nc, const float* min_nc, const int* imax_nc, const int* imin_nc,
                   const float* s2, const float* h2, const float* w0p, const float* w2p, int n_img, int c, int cr, float* A, float* B,
                   float* ca, float* avg, float* mx, int* idx, float* tval, void* stream);
int runet_sa_reduce(const float* t2, int ld, const float* A, const float* B, long pixels, int hw, int c, float* smap, int* amax,
                    void* stream);
int runet_sa_conv7(const float* smap, const float* wp, float* sa, int n_img, int h, int w, void* stream);
/* out = relu((t2*A[n,c]+B[n,c])*sa[p] + res),  res = r*rs[c]+rh[c]  (rs == NULL: res = r, identity shortcut) */
int runet_rb_out(const float* t2, int ld, const float* A, const float* B, const float* sa, const float* r, int ldr, const float* rs,
                 const float* rh, float* out, int ldo, long pixels, int hw, int c, void* stream);
/* dv = dout * (out > 0) (out NULL: dv = dout, no ReLU behind the attention), dq[p] = (sum_c dv*u) * sa*(1-sa) with u = t2*A+B */
int runet_rb_bwd1(const float* dout, int lddo, const float* out, int ldo, const float* t2, int ld, const float* A, const float* B,
                  const float* sa, float* dv, int lddv, float* dq, long pixels, int hw, int c, void* stream);
long runet_sa_conv7_bwd_workspace_floats(int n_img, int h, int w);
int runet_sa_conv7_bwd(const float* smap, const float* dq, const float* wp, float* dsm, float* dwp, float* workspace, int n_img, int h,
                       int w, void* stream);
int runet_rb_bwd2(const float* dv, int lddv, const float* t2, int ld, const float* sa, const float* dsm, const int* amax, int n_img,
                  int hw, int c, float* workspace, float* sdu, float* sdut, void* stream);
/* runet_rb_bwd2 that also leaves the LOCAL BatchNorm-backward sums of the block's shortcut BatchNorm (Main_Final.py:173) behind: dv is that
 * BatchNorm's incoming gradient and r its input; sums_s [2c] = (sum dv * rhat | sum dv), what runet_bn_bwd_reduce(dv, r) would compute */
long runet_rb_bwd2_bn_workspace_floats(int n_img, int hw, int c);
int runet_rb_bwd2_bn(const float* dv, int lddv, const float* t2, int ld, const float* sa, const float* dsm, const int* amax, const float* r, int ldr,
                     const float* mean_s, const float* invstd_s, int n_img, int hw, int c, float* workspace, long workspace_floats, float* sdu,
                     float* sdut, float* sums_s, void* stream);
long runet_ca_bwd_workspace_floats(int n_img, int c, int cr);
int runet_ca_bwd(const float* sdu, const float* sdut, const float* s2, const float* h2, const float* ca, const float* avg, const float* mx,
                 const float* w0p, const float* w2p, const float* mean_nc, const float* tval, const float* mean2, const float* invstd2,
                 int n_img, int c, int cr, float* workspace, float* davg, float* dmx, float* sums2, float* dw0p, float* dw2p, void* stream);
int runet_rb_bwd3(const float* dv, int lddv, const float* t2, int ld, const float* sa, const float* dsm, const int* amax, const float* ca,
                  const float* davg, const float* dmx, const int* idx, const float* mean2, const float* invstd2, const float* s2,
                  const float* sums2, float* dt2, int lddt, long pixels, int hw, int c, long m_total, void* stream);

/* ---- AttentionGate (Main_Final.py:120-148): psi conv (F_int -> 1) and the gating multiply ---- */
int runet_ag_psi(const float* g1, int ldg, const float* x1, int ldx, const float* sg, const float* hg, const float* sx, const float* hx,
                 const float* wpsi, const float* bpsi, float* s, long pixels, int f, void* stream);
int runet_ag_out(const float* xs, int ldx, const float* s, const float* sp, const float* hp, float* out, int ldo, long pixels, int c,
                 void* stream);
int runet_ag_bwd1(const float* datt, int ldd, const float* xs, int ldx, const float* s, const float* sp, const float* hp, float* dxs,
                  int lddx, float* dsbn, long pixels, int c, void* stream);
/* dwpsi_db: [f] weight gradient followed by the bias gradient at index f */
int runet_ag_bwd2(const float* ds, const float* g1, int ldg, const float* x1, int ldx, const float* sg, const float* hg, const float* sx,
                  const float* hx, const float* wpsi, float* dpre, int ldp, float* workspace, float* dwpsi_db, long pixels, int f,
                  void* stream);
/* runet_ag_bwd2 that also leaves the LOCAL BatchNorm-backward sums of the gate's two BatchNorms (W_g.1, W_x.1: Main_Final.py:127,132) behind -
 * sums_g / sums_x [2f] = (sum dpre * xhat | sum dpre), what runet_bn_bwd_reduce would compute from (dpre, g1) and (dpre, x1) */
long runet_ag_bwd2_bn_workspace_floats(long pixels, int f);
int runet_ag_bwd2_bn(const float* ds, const float* g1, int ldg, const float* x1, int ldx, const float* sg, const float* hg, const float* sx,
                     const float* hx, const float* wpsi, const float* mean_g, const float* invstd_g, const float* mean_x, const float* invstd_x,
                     float* dpre, int ldp, float* workspace, long workspace_floats, float* dwpsi_db, float* sums_g, float* sums_x, long pixels, int f,
                     void* stream);

/* ---- outc: Conv2d(C, 1, 1) + Sigmoid (Main_Final.py:274-277) ---- */
int runet_outc_fwd(const float* x, int ld, const float* w, const float* b, float* logit, float* prob, long pixels, int c, void* stream);
int runet_outc_bwd(const float* dprob, const float* prob, const float* x, int ld, const float* w, float* dx, int lddx, float* workspace,
                   float* dw_db, long pixels, int c, void* stream);

/* ---- MaxPool2d(2) (Main_Final.py:235,239,243,249); idx: one byte per output element ---- */
int runet_maxpool2_fwd(const float* x, int ldx, float* y, int ldy, unsigned char* idx, int n_img, int h, int w, int c, void* stream);
int runet_maxpool2_bwd(const float* dy, int lddy, const unsigned char* idx, float* dx, int lddx, int n_img, int h, int w, int c,
                       int accumulate, void* stream);
/* strided [N,C,H,W] -> dense NHWC with channels zero-padded to c_pad (module entry: Main_Final.py:290 input x) */
int runet_to_nhwc_pad(const float* x, long sn, long sc, long sh, long sw, float* y, int n_img, int c, int h, int w, int c_pad, void* stream);

/* ---- nn.BCELoss() mean (Main_Final.py:551,580): logs clamped at -100; backward (p-y)/max(p(1-p),1e-12)/n * grad_out[0] ---- */
int runet_bce_fwd(const float* prob, const float* target, long n, double* workspace1024, float* loss, void* stream);
int runet_bce_bwd(const float* prob, const float* target, const float* grad_out, float* dprob, long n, void* stream);

/* ---- torch.optim.Adam(lr, weight_decay) (Main_Final.py:552,582), all tensors in one launch ----
 * table: device int64 [5][n_tensors] = param, grad, exp_avg, exp_avg_sq pointers, element counts;
 * chunks: device int32 [n_chunks][2] = (tensor index, chunk index), chunk = runet_adam_chunk_elems() elements.
 * grad_scale multiplies the gradient first (1/world_size after a sum all-reduce, 1/loss_scale under loss scaling).
 * skip_flag (may be NULL): device int; non-zero -> the whole update is skipped (runet_nonfinite_flag found Inf / NaN gradients). */
int runet_adam_chunk_elems(void);
int runet_adam_multi(const long long* table, int n_tensors, const int* chunks, int n_chunks, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float grad_scale, const int* skip_flag, void* stream);

/* hipGraph-capturable form: hyper = device float[6] {lr, beta1, beta2, eps, weight_decay, grad_scale}; *step_dev is incremented on the
 * device and then used for the bias corrections, so one captured launch serves every step and every learning rate. */
int runet_adam_multi_dev(const long long* table, int n_tensors, const int* chunks, int n_chunks, const float* hyper, int* step_dev,
                         const int* skip_flag, void* stream);
/* loss scaling: flag2[0] = 1 iff buf[0:n] holds an Inf / NaN (reset by this call), flag2[1] += flag2[0] (running count of skipped steps) */
int runet_nonfinite_flag(const float* buf, long n, int* flag2, void* stream);
/* stream `waiter` waits for everything enqueued on stream `waited` so far (event record + stream wait on a pooled event): host-side helper
 * of the weight-gradient side stream (the all-reduce / backward overlap of Main_Final.py:581's loss.backward()). */
int runet_stream_wait(void* waiter, void* waited);


/* ---- ModelEvaluator.calculate_metrics counts (Main_Final.py:519-547): counts[n] = {tp, pred>thr, target!=0, agree} ---- */
int runet_seg_counts(const float* pred, const float* target, long long* counts, int n_img, long per_img, float threshold, void* stream);

/* ---- batched plain GEMMs (fp32 MFMA, LDS-tiled; the position-GEMMs of the F(4x4,3x3) path) ----
 * runet_gemm_batched:    C[z][rows][n] = A[z][rows][k] . B[z][k][n]   (row strides lda / n / ldc, batch strides in floats)
 * runet_gemm_tn_batched: C[split][z][k][n] = sum over rows of split of A[z][row][k] * B[z][row][n];  splits = ceil(rows / rows_per_split) */
int runet_gemm_batched(const float* a, int lda, long stride_a, const float* b, long stride_b, float* c, int ldc, long stride_c, int batch, int rows,
                       int k, int n, void* stream);
int runet_gemm_tn_batched(const float* a, int lda, long stride_a, const float* b, int ldb, long stride_b, float* c, int batch, int rows, int k, int n,
                          int rows_per_split, void* stream);
/* name of the device kernel runet_gemm_batched launches for this shape (128x128 LDS-ring kernel, or 128x64 tiles where those would
 * fill the 256 CUs unevenly) - for matching live timings with rocprofv3 rows */
const char* runet_gemm_batched_kernel_name(int batch, int rows, int k, int n);

/* ---- the same GEMMs in fp32 on the BF16 matrix cores: exact three-way operand splitting, six bf16 MFMAs per fp32 product ("bf16x6") ----
 * The f32-input MFMA runs at 1/16 of the bf16 MFMA rate on gfx950.  x = h + m + l with three bf16 values holds exactly for every fp32 x;
 * six of the nine cross products are kept (the three dropped are <= 2^-23 of the product, the size of one fp32 rounding), all accumulate in
 * fp32: an fp32-accurate GEMM at 6/16 of the matrix time (csrc/gemm_split.hip; measured against float64 in tests/test_gpu_conv.py).
 * These serve the same role as runet_gemm_batched / runet_gemm_tn_batched (the position-GEMMs of nn.Conv2d 3x3, Main_Final.py:157,159).
 * runet_gemm_x3_pack: B fp32 [z][k][n] (the weights side: once per optimizer step) -> packed split planes [z][3][k/8][n][8] bf16,
 *   runet_gemm_x3_pack_elems(batch, k, n) 2-byte elements.  runet_gemm_x3_batched: C[z][rows][n] = A[z][rows][k] . B[z] with A fp32 split on
 *   the fly (k % 16 == 0).  runet_gemm_x3_tn_batched: as runet_gemm_tn_batched, both operands fp32 (rows, rows_per_split multiples of 16). */
int runet_gemm_x3_supported(int rows, int k, int n);
long runet_gemm_x3_pack_elems(int batch, int k, int n);
int runet_gemm_x3_pack(const float* b, long stride_b, void* packed, int batch, int k, int n, void* stream);
int runet_gemm_x3_batched(const float* a, int lda, long stride_a, const void* packed_b, float* c, int ldc, long stride_c, int batch, int rows,
                          int k, int n, void* stream);
int runet_gemm_x3_tn_batched(const float* a, int lda, long stride_a, const float* b, int ldb, long stride_b, float* c, int batch, int rows, int k,
                             int n, int rows_per_split, void* stream);
const char* runet_gemm_x3_kernel_name(int batch, int rows, int k, int n);
/* weight gradient of ConvTranspose2d(k2, s2) (Main_Final.py:261-270) on the same TN kernel: x [n_img,h,w,cin], dy [n_img,2h,2w,cout] ->
 * c [splits][2][2][cin][cout] (sum the splits in order: runet_conv_wgrad does); w % 16 == 0, rows_per_split % 16 == 0 */
int runet_gemm_x3_tn_convt(const float* x, int ldx, const float* dy, int ldy, float* c, int n_img, int h, int w, int cin, int cout,
                           int rows_per_split, void* stream);

/* ---- nn.Conv2d(k=1) (Main_Final.py:126,131,172,205) and nn.ConvTranspose2d(k=2, s=2) (:261-270), forward and data gradient, by the same
 * split-operand scheme (csrc/conv_x3.hip): same modes and argument meaning as runet_conv_igemm (RUNET_CONV_FWD, RUNET_CONV_DGRAD,
 * RUNET_CONVT_FWD, RUNET_CONVT_DGRAD; cin = channels READ in that mode, cout = channels WRITTEN; h, w = the image the 1x1 convolution runs
 * over, for the transposed modes the low-resolution side), but the weight comes pre-split: runet_conv_x3_pack turns the module's forward
 * weight (HWIO: [cin][cout], transposed [2][2][Ci][Co]; the data-gradient modes read it transposed) into runet_conv_x3_pack_elems 2-byte
 * elements, once per optimizer step.  cin % 16 == 0, cout % 4 == 0, ldx % 4 == 0, x 16-byte aligned. */
int runet_conv_x3_supported(int cin, int cout, int mode);
long runet_conv_x3_pack_elems(int cin, int cout, int mode);
int runet_conv_x3_pack(const float* w, void* packed, int cin, int cout, int mode, void* stream);
int runet_conv_x3(const float* x, int ldx, const void* wpacked, const float* bias, float* y, int ldy, int n_img, int h, int w, int cin, int cout,
                  int mode, int accumulate, void* stream);
const char* runet_conv_x3_kernel_name(int n_img, int h, int w, int cout, int mode);

/* ---- every derived weight of a step in one launch (csrc/derive_multi.hip).  The split-operand kernels read their weights as bf16 planes made
 * by runet_wino4_weights_x3 / runet_wino_weights_x3 / runet_conv_x3_pack once per optimizer step (the reference has no such step: its weights
 * are read as they are, Main_Final.py:123-134, 261-270); these three calls batch them: fill a host table with runet_derive_desc (same arguments as
 * the per-tensor call of that kind; returns the entry's block count, < 0 on bad arguments), copy it to the device, launch runet_derive_multi.
 * Results are bit-identical to the per-tensor calls. */
enum { RUNET_DERIVE_WINO4 = 0, RUNET_DERIVE_WINO2 = 1, RUNET_DERIVE_PACK = 2 };
int runet_derive_desc_bytes(void);
int runet_derive_desc(void* host_table, int index, int kind, const float* w, void* dst, int cin, int cout, int mode, int first_block);
int runet_derive_multi(const void* table, int n_desc, int total_blocks, void* stream);
/* the 1x1 modes with the BatchNorm statistics of the output taken in the epilogue: stats [runet_conv_x3_stats_parts(n_img, h, w)][cout][3] */
int runet_conv_x3_stats_parts(int n_img, int h, int w);
int runet_conv_x3_stats(const float* x, int ldx, const void* wpacked, const float* bias, float* y, int ldy, int n_img, int h, int w, int cin, int cout,
                        int mode, int accumulate, float* stats, void* stream);

/* ---- Winograd F(4x4,3x3), unfused, for the deep 3x3 convolutions (Main_Final.py:157,159 at >= 256 channels) and their autograd ----
 * U [36][K][N] from runet_wino4_weights (dgrad != 0: rotated filter, K = cout, N = cin).  conv: x [n,h,w,K] -> y [n,h,w,N] ('same'), H, W % 4 == 0.
 * dil >= 1 (padding = dil: the bottleneck's DilatedBlock, Main_Final.py:207-208): the dilated convolution is run as dil*dil independent
 * dilation-1 convolutions over the (h/dil) x (w/dil) sub-images of every image, so (h/dil) % 4 == 0 and (w/dil) % 4 == 0 are required.
 * workspace: runet_wino4_workspace_floats / runet_wino4_wgrad_workspace_floats floats, 16-byte aligned. */
int runet_wino4_supported(int h, int w, int k, int n);
long runet_wino4_workspace_floats(int n_img, int h, int w, int k, int n);
int runet_wino4_weights(const float* w_hwio, float* U, int cin, int cout, int dgrad, void* stream);
int runet_wino4_conv(const float* x, int ldx, const float* U, const float* bias, float* y, int ldy, int n_img, int h, int w, int k, int n,
                     int dil, int accumulate, float* workspace, long workspace_floats, void* stream);
/* runet_wino4_weights + runet_gemm_x3_pack in one pass: the filter transform written straight into the split planes (36 matrices [k][n] ->
 * [36][3][k/8][n][8] bf16, runet_gemm_x3_pack_elems(36, k, n) elements); k (cin, or cout when dgrad != 0) a multiple of 8 */
int runet_wino4_weights_x3(const float* w_hwio, void* Upacked, int cin, int cout, int dgrad, void* stream);
/* runet_wino4_conv with the position-GEMMs on the bf16 matrix cores (split operands, fp32-accurate): Upacked = runet_gemm_x3_pack of U; k % 16 == 0 */
int runet_wino4_conv_x3(const float* x, int ldx, const void* Upacked, const float* bias, float* y, int ldy, int n_img, int h, int w, int k, int n,
                        int dil, int accumulate, float* workspace, long workspace_floats, void* stream);
/* the stages of the two composites below, one kernel each (T = n_img*(h/4)*(w/4) tiles):
 *   runet_wino4_input  mode 0: V[36][T][c] = B^T d B (6x6 patches of src, stride 4, 1-pixel halo); mode 1: Z[36][T][c] = A dY A^T (4x4 tiles)
 *   runet_gemm_batched / runet_gemm_tn_batched: the 36 position-GEMMs
 *   runet_wino4_output: y = A^T M A (+bias, +y);  runet_wino4_wgrad_output: dw[3][3][cin][cout] = G^T (sum_splits dU[split][36][cin][cout]) G */
int runet_wino4_input(const float* src, int ld, int c, int n_img, int h, int w, int dil, int mode, float* V, void* stream);
/* The same transforms with their input's elementwise producer folded into the loads (the tensor between the two is never written; results
 * bit-identical to the two-step forms).  runet_wino4_input_act: mode 0 of a1 = runet_bn_apply(t, scale, shift, factor_nc, relu = 1), i.e. the
 * BatchNorm + ReLU + Dropout2d in front of a ResidualBlock's conv2 (Main_Final.py:157-160).  runet_wino4_input_bn_bwd: mode 1 of
 * dx = runet_bn_bwd_apply(dy, x, ..., relu_shift = shift), i.e. autograd's backward of that BatchNorm + ReLU feeding conv1's gradients; sums as
 * runet_bn_bwd_reduce leaves them, m_total 0 = the tensor's own pixel count. */
int runet_wino4_input_act(const float* t, int ld, int c, int n_img, int h, int w, int dil, const float* scale, const float* shift,
                          const float* factor_nc, float* V, void* stream);
int runet_wino4_input_bn_bwd(const float* dy, int lddy, const float* x, int ldx, int c, int n_img, int h, int w, int dil, const float* mean,
                             const float* invstd, const float* scale, const float* shift, const float* sums, const float* factor_nc,
                             long m_total, float* Z, void* stream);
int runet_wino4_output(const float* M, int n, int n_img, int h, int w, int dil, const float* bias, float* y, int ldy, int accumulate, void* stream);
/* runet_wino4_output (dil = 1) that also leaves the BatchNorm statistics partials of y behind for runet_bn_stats_finalize: stats
 * [runet_wino4_output_stats_parts(...)][n][3] = (count, mean, M2); _parts returns 0 where the kernel cannot take them (n / 2 not a power of two, dil != 1). */
int runet_wino4_output_stats_parts(int n_img, int h, int w, int n, int dil);
int runet_wino4_output_stats(const float* M, int n, int n_img, int h, int w, const float* bias, float* y, int ldy, int accumulate, float* stats, void* stream);
/* The data gradient as the ADJOINT of the forward algorithm: Z = A dy A^T (runet_wino4_input mode 1 - the weight gradient's transform, shared),
 * M' = Z . U^T by the position GEMMs (runet_wino4_weights_x3 with dgrad = 2: the forward's U transposed, not rotated), and this gather-form
 * output transform dx = overlap-add of B M' B^T: one pass over dy and one filter transform less than the convolution form per layer. */
int runet_wino4_output_adj(const float* M, int n, int n_img, int h, int w, int dil, float* y, int ldy, int accumulate, void* stream);
int runet_wino4_wgrad_output(const float* dU, int splits, int cin, int cout, float* dw, void* stream);
int runet_wino4_wgrad_rows_per_split(int n_img, int h, int w, int cin, int cout);
long runet_wino4_wgrad_workspace_floats(int n_img, int h, int w, int cin, int cout);
int runet_wino4_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats, int n_img, int h, int w,
                      int cin, int cout, int dil, void* stream);

/* ---- the RGB stem (Main_Final.py:157,172 with in_channels = 3; `inc` :235): 1..3 real input channels in an NHWC tensor padded to 4 ----
 * runet_stem_conv: y3 = conv3x3(x, w3 [3][3][cin_w][cout], padding 1) and, if w1 != NULL, y1 = conv1x1(x, w1 [cin_w][cout]) (the first
 * ResidualBlock's shortcut) in ONE launch: x is read once, both filters form one register-resident B matrix.  No bias (the block's
 * convolutions have none).  runet_stem_wgrad: dw [k][k][cin_w][cout] of either convolution (ksize 3 or 1) with the pixels as the MFMA
 * contraction; workspace: runet_stem_wgrad_workspace_floats floats.  cout 32 or 64 (runet_stem_supported). */
int runet_stem_supported(int cin_w, int cout);
int runet_stem_conv(const float* x, int ldx, const float* w3, const float* w1, float* y3, int ldy3, float* y1, int ldy1, int n_img,
                    int h, int w, int cin_w, int cout, void* stream);
/* runet_stem_conv that also leaves the BatchNorm statistics partials of y3 (and y1) behind for runet_bn_stats_finalize:
 * stats3 / stats1 [runet_stem_conv_stats_parts(n_img, h, w)][cout][3] = (count, mean, M2) per 8 x 32 pixel tile */
int runet_stem_conv_stats_parts(int n_img, int h, int w);
int runet_stem_conv_stats(const float* x, int ldx, const float* w3, const float* w1, float* y3, int ldy3, float* y1, int ldy1, int n_img,
                          int h, int w, int cin_w, int cout, float* stats3, float* stats1, void* stream);
long runet_stem_wgrad_workspace_floats(int n_img, int h, int w, int cin_w, int cout, int ksize);
int runet_stem_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats, int n_img,
                     int h, int w, int cin_w, int cout, int ksize, void* stream);

/* ---- DeepLabV3+ baseline (Main_Final.py:325-433): the same implicit-GEMM kernels with general geometry ----
 * runet_conv2d_general: Conv2d(kh x kw <= 7x7, stride 1|2, padding, dilation).  mode RUNET_CONV_FWD: x [n,hin,win,cin] -> y [n,ho,wo,cout],
 * w [kh,kw,cin_w,cout];  mode RUNET_CONV_DGRAD: x := dy [n,ho,wo,cin(=conv Cout)] -> y := dx [n,hin,win,cout(=conv Cin)], w [kh,kw,cout,cin].
 * runet_convt4_igemm: ConvTranspose2d(k4, s2, p1), w [4,4,cin,cout]; RUNET_CONVT_FWD x [n,h,w,cin] -> y [n,2h,2w,cout];
 * RUNET_CONVT_DGRAD x := dy [n,2h,2w,cin(=Cout)] -> y := dx [n,h,w,cout(=Cin)].
 * runet_conv_wgrad_general: dw of either (transposed4 != 0: the k4 transposed conv, x [n,hin,win,cin], dy [n,2hin,2win,cout]). */
int runet_conv2d_general(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int n_img, int hin, int win, int cin,
                         int cin_w, int cout, int kh, int kw, int stride, int pad, int dil, int mode, int accumulate, void* stream);
int runet_convt4_igemm(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int n_img, int h, int w_, int cin, int cout,
                       int mode, int accumulate, void* stream);
long runet_conv_wgrad_general_workspace_floats(int pixels, int cin_w, int cout, int kh, int kw);
int runet_conv_wgrad_general(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats, int n_img,
                             int hin, int win, int cin, int cin_w, int cout, int kh, int kw, int stride, int pad, int dil, int transposed4,
                             void* stream);
/* MaxPool2d(3, stride 2, padding 1) (Main_Final.py:369); idx: one byte per output element (window position of the first maximum) */
int runet_maxpool3s2_fwd(const float* x, int ldx, float* y, int ldy, unsigned char* idx, int n_img, int h, int w, int c, void* stream);
int runet_maxpool3s2_bwd(const float* dy, int lddy, const unsigned char* idx, float* dx, int lddx, int n_img, int h, int w, int c, void* stream);
/* dst[i] += src[i] (dense buffers): joins the ASPP image-pooling gradient with the other four branches' */
int runet_add_inplace(float* dst, const float* src, long n, void* stream);
/* y[n, p, 0:c] = v[n, 0:c]: bilinear upsampling of the ASPP image-pooling branch's 1x1 map (Main_Final.py:350-352) */
int runet_broadcast_nc(const float* v_nc, float* y, int ldy, int n_img, int hw, int c, void* stream);
/* decoder head Conv2d(c, 1, 3, padding 1) + sigmoid (Main_Final.py:414,433); w [3,3,c] ; dw_db = [9*c] weight gradient then the bias gradient */
int runet_head3x3_fwd(const float* x, int ld, const float* w, const float* b, float* prob, int n_img, int h, int w_, int c, void* stream);
long runet_head3x3_bwd_workspace_floats(int n_img, int h, int w_, int c);
int runet_head3x3_bwd(const float* dprob, const float* prob, const float* x, int ld, const float* w, float* dx, int lddx, float* workspace,
                      float* dw_db, int n_img, int h, int w_, int c, void* stream);


/* ---- bf16-operand convolutions (BASELINE.json configs 3 and 5: "bf16" / reduced precision; the reference itself is fp32 only, so the
 *      oracle for this path is the fp32 step and the tolerance is stated in tests/test_gpu_bf16.py) ----
 * Only the two operands of each convolution's multiply-adds are rounded to bf16 (round-to-nearest-even) on their way into LDS;
 * accumulation, activations in HBM, master weights, BatchNorm, attention, loss and Adam stay fp32.
 * runet_bf16_pack_weights: HWIO fp32 w[taps][cin][cout] -> packed bf16 [taps][K/8][N][8] (K padded to a multiple of 8 with zeros);
 *   transpose == 0: K = cin, N = cout (forward, k2-s2 transposed forward);  transpose != 0: K = cout, N = cin (data gradients).
 *   `packed` holds runet_bf16_pack_elems(taps, K, N) 16-bit elements.
 * runet_conv_igemm_bf16: same modes / geometry as runet_conv_igemm (cin_w == cin), weights from runet_bf16_pack_weights.
 * runet_conv_wgrad_bf16: same contract as runet_conv_wgrad (cin_w == cin; dil only for 3x3).
 * The *_fp16 entry points are the same kernels with IEEE half operands (BASELINE config 5 names fp16; csrc/conv_fp16.hip); train with a
 * loss scale (runet_nonfinite_flag + the skip_flag of the fused Adam). */
long runet_bf16_pack_elems(int taps, int k, int n);
int runet_bf16_pack_weights(const float* w_hwio, void* packed, int taps, int cin, int cout, int transpose, void* stream);
int runet_conv_igemm_bf16(const float* x, int ldx, const void* wpacked, const float* bias, float* y, int ldy, int n_img, int h, int w_,
                          int cin, int cout, int kh, int kw, int dil, int mode, int accumulate, void* stream);
long runet_conv_wgrad_bf16_workspace_floats(int n_img, int h, int w_, int cin, int cout, int kh, int kw, int dil, int transposed);
int runet_conv_wgrad_bf16(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats,
                          int n_img, int h, int w_, int cin, int cout, int kh, int kw, int dil, int transposed, void* stream);

long runet_fp16_pack_elems(int taps, int k, int n);
int runet_fp16_pack_weights(const float* w_hwio, void* packed, int taps, int cin, int cout, int transpose, void* stream);
int runet_conv_igemm_fp16(const float* x, int ldx, const void* wpacked, const float* bias, float* y, int ldy, int n_img, int h, int w_,
                          int cin, int cout, int kh, int kw, int dil, int mode, int accumulate, void* stream);
long runet_conv_wgrad_fp16_workspace_floats(int n_img, int h, int w_, int cin, int cout, int kh, int kw, int dil, int transposed);
int runet_conv_wgrad_fp16(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace, long workspace_floats,
                          int n_img, int h, int w_, int cin, int cout, int kh, int kw, int dil, int transposed, void* stream);

/* ---- harness helpers (Main_Final.py:577-578,596-597,648-649: `F.interpolate(outputs, size=masks.shape[-2:], mode='bilinear',
 *      align_corners=False)` when the model output and the mask differ in size; :82-117 standalone attention modules) ----
 * runet_bilinear_fwd: y[planes, ho, wo] = bilinear resize of x[planes, h, w] with ATen's align_corners=False source index
 *   (src = max(0, (dst + 0.5) * in/out - 0.5)).  runet_bilinear_bwd: dx[planes, h, w] = its adjoint applied to dy[planes, ho, wo]
 *   (gather form: fixed summation order, no atomics).
 * runet_mul_pixel: y[p, 0:c] = x[p, 0:c] * s[p]  (NHWC, one factor per pixel: SpatialAttention's output product). */
int runet_bilinear_fwd(const float* x, float* y, long planes, int h, int w, int ho, int wo, void* stream);
int runet_bilinear_bwd(const float* dy, float* dx, long planes, int h, int w, int ho, int wo, void* stream);
int runet_mul_pixel(const float* x, int ldx, const float* s, float* y, int ldy, long pixels, int c, void* stream);

/* ---- plain 2-class U-Net of the reference's older trainer (train_water_segmentation.py:209-288 `UNet`, :304 nn.CrossEntropyLoss;
 *      consumer predict_coastline.py:351): built from the kernels above plus
 * runet_nhwc_to_nchw: y[n][c][p] = x[(n*hw + p)*ld + c]  (the [N, classes, H, W] logits the module returns);
 * runet_ce_fwd / runet_ce_bwd: mean cross-entropy over all pixels of NCHW logits (2..8 classes) against int64 targets [N, H, W],
 *   ATen's log_softmax arithmetic; partials1024: 1024 doubles of scratch; gout: the scalar upstream gradient (device). */
int runet_nhwc_to_nchw(const float* x, int ld, float* y, int n_img, int c, long hw, void* stream);
int runet_ce_fwd(const float* logits_nchw, const long long* target, int n_img, int classes, long hw, double* partials1024, float* loss, void* stream);
int runet_ce_bwd(const float* logits_nchw, const long long* target, const float* gout, float* dlogits_nchw, int n_img, int classes, long hw,
                 void* stream);

#ifdef __cplusplus
}
#endif
#endif
