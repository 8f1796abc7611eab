// bf16-operand convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation): the reduced-precision path of
// BASELINE.json configs 3 and 5.  The reference (/root/reference/Main_Final.py) is fp32 only; this path keeps fp32 master weights,
// fp32 activations in HBM, fp32 BatchNorm / attention / loss / Adam, and rounds ONLY the two operands of each convolution's
// multiply-adds to bf16 on their way into LDS (round-to-nearest-even, v_cvt_pk_bf16_f32).  16x the fp32 MFMA rate turns every
// convolution of the step from matrix-pipe-bound into HBM-bound.
//
//   forward / data gradient   igemm_bf16_kernel   out[p][n] = sum_tap sum_k src[pix(p,tap)][k] * B_tap[k][n]
//     GEMM M = pixels (tile 128), N = output channels (tile 32/64/128), K = taps x channels in steps of 64 channels.
//     A: fp32 NHWC rows gathered per tap (predicated zero padding), converted to bf16, LDS image [pixel][64 + 8 pad] so that a lane's
//     8 consecutive k are one conflict-free ds_read_b128.  B: weights PRE-PACKED once per optimizer step by bf16_pack_kernel into
//     [tap][k/8][n][8] bf16 - a lane's B fragment (8 consecutive k of one column) is one 16-byte unit, the tile is copied to LDS verbatim.
//     Register-staged double buffering, one barrier per 64-deep k-step (16 MFMAs per wave).
//   weight gradient           wgrad_bf16_kernel   dW[tap][ci][co] = sum_p x[pix(p,tap)][ci] * dy[p][co]
//     GEMM M = cin (tile 64), N = cout (tile 64), K = pixels.  Both operands have the contraction index (pixels) as the SLOW memory index
//     (NHWC), so both tiles sit in LDS as [pixel][64 channels] exactly as they come from HBM and the fragments are fetched with the
//     hardware transposing read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group -> each lane gets 4 consecutive pixels of
//     its channel).  A block owns a 4 x 16 pixel tile of dy and the x tile WITH ITS HALO, and accumulates all taps from that one tile
//     (tap shifts move whole LDS rows, so alignment is unaffected); fixed-order split-K slabs + slab reduce as in the fp32 path.
#include "runet_common.h"
#include <stdlib.h>
#include "../../include/runet_hip.h"

#define LP_T __bf16
#define LP_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define LP_PACK_KERNEL bf16_pack_kernel
#define LP_IGEMM_KERNEL igemm_bf16_kernel
#define LP_CONV3_KERNEL conv3x3_bf16_kernel
#define LP_WGRAD_KERNEL wgrad_bf16_kernel
#define LP_SLAB_KERNEL slab_reduce_bf16path
#define LP_SYM_PACK_ELEMS runet_bf16_pack_elems
#define LP_SYM_PACK_WEIGHTS runet_bf16_pack_weights
#define LP_SYM_IGEMM runet_conv_igemm_bf16
#define LP_SYM_WGRAD_WS runet_conv_wgrad_bf16_workspace_floats
#define LP_SYM_WGRAD runet_conv_wgrad_bf16
#include "conv_lowp.inc"
