// fp16-operand convolutions (BASELINE.json config 5 names fp16): the same kernels as conv_bf16.hip with _Float16 operands
// (v_mfma_f32_32x32x16_f16, fp32 accumulation).  fp16 keeps 11 significant bits (bf16: 8) but only 5 exponent bits: gradients below
// 6e-8 vanish and below 6e-5 lose precision when the data-gradient / weight-gradient kernels round dy, so the train step scales the loss
// (trainer.TrainStep(loss_scale=...), un-scaled inside the fused Adam, step skipped on the device when a gradient is not finite).
#include "runet_common.h"
#include <stdlib.h>
#include "../../include/runet_hip.h"

#define LP_T _Float16
#define LP_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define LP_PACK_KERNEL fp16_pack_kernel
#define LP_IGEMM_KERNEL igemm_fp16_kernel
#define LP_CONV3_KERNEL conv3x3_fp16_kernel
#define LP_WGRAD_KERNEL wgrad_fp16_kernel
#define LP_SLAB_KERNEL slab_reduce_fp16path
#define LP_SYM_PACK_ELEMS runet_fp16_pack_elems
#define LP_SYM_PACK_WEIGHTS runet_fp16_pack_weights
#define LP_SYM_IGEMM runet_conv_igemm_fp16
#define LP_SYM_WGRAD_WS runet_conv_wgrad_fp16_workspace_floats
#define LP_SYM_WGRAD runet_conv_wgrad_fp16
#include "conv_lowp.inc"
