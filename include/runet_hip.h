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
enum { RUNET_CONV_FWD = 0, RUNET_CONV_DGRAD = 1, RUNET_CONVT_FWD = 2, RUNET_CONVT_DGRAD = 3 };

/* mode FWD   : y[n,h,w,0:cout] (=|+=) bias + conv_{kh x kw, dilation dil, 'same' zero padding}(x[n,h,w,0:cin], w[kh,kw,cin_w,cout])
 *              cin is the channel count READ from x (multiple of 4, zero-padded by the caller);
 *              cin_w <= cin is the number of input channels present in w (3 for the RGB stem).
 * mode DGRAD : x := dy[n,h,w,0:cin] (cin = conv Cout), y := dx[n,h,w,0:cout] (cout = conv Cin),
 *              w = the forward weight [kh,kw,cout,cin]; computes the data gradient.
 * mode CONVT_FWD  : y[n,2h,2w,0:cout] = bias + convT_{2x2,s2}(x[n,h,w,0:cin]),  w[2,2,cin,cout]
 * mode CONVT_DGRAD: x := dy[n,2h,2w,0:cin], y := dx[n,h,w,0:cout],  w[2,2,cout,cin]
 * accumulate != 0 adds into y instead of overwriting.  bias may be NULL. */
int runet_conv_igemm(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                     int n_img, int h, int w_, int cin, int cin_w, int cout, int kh, int kw, int dil,
                     int mode, int accumulate, void* stream);

/* dw[kh,kw,cin_w,cout] = sum_pixels x (x) dy  (weight gradient; transposed != 0: dy is [n,2h,2w,cout]
 * and the 2x2 taps index the dy pixel).  `workspace` (>= runet_conv_wgrad_workspace_floats floats, may be
 * NULL) holds split-K partial slabs that are summed in a fixed order. */
long runet_conv_wgrad_workspace_floats(int n_img, int h, int w_, int cin_w, int cout, int kh, int kw);
int runet_conv_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, float* workspace,
                     long workspace_floats, int n_img, int h, int w_, int cin, int cin_w, int cout,
                     int kh, int kw, int dil, int transposed, void* stream);

#ifdef __cplusplus
}
#endif
#endif
