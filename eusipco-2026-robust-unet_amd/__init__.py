"""MI355X-native Robust U-Net training path (gfx950 HIP kernels behind a C ABI).

The directory name carries the reference repository's name, so import it with
`importlib.import_module("eusipco-2026-robust-unet_amd")`.

Public surface (mirrors /root/reference/Main_Final.py for the hot path):
  RobustUNet, ResidualBlock, DilatedBlock, AttentionGate, ChannelAttention, SpatialAttention
  CoastalDataset, prepare_dataset, ModelEvaluator;  DeepLabV3Plus (baseline);  UNet (train_water_segmentation.py's 2-class model)
plus the MI355X additions: FusedAdam, sigmoid-free fused BCE loss, GradAllReducer (RCCL).

Sub-modules are imported lazily so that host-only pieces (data, portable_rng) stay usable
on a machine without the HIP library; anything that computes raises if
`csrc/librunet_hip.so` is missing — there is no CPU fallback.
"""
import importlib as _il
import os as _os
import sys as _sys

# The backward pass uses several HIP streams (main chain, weight gradients, gradient all-reduce + RCCL's own): with HIP's default of 4 hardware
# queues they share queues and serialise on each other's event waits (380 vs 414 img/s).  The variable only counts if it is set before the HIP
# runtime comes up, so it is set here, when the package is imported - normally long before the first device call.
_HW_QUEUES_LATE = False
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    _t = _sys.modules.get("torch")
    _HW_QUEUES_LATE = bool(_t is not None and _t.cuda.is_available() and _t.cuda.is_initialized())

_LAZY = {
    "RobustUNet": "model", "ResidualBlock": "model", "DilatedBlock": "model", "AttentionGate": "model",
    "ChannelAttention": "model", "SpatialAttention": "model", "DeepLabV3Plus": "deeplab", "ASPP": "deeplab",
    "CoastalDataset": "data", "prepare_dataset": "data", "synthetic_batch": "data", "DevicePrefetcher": "data",
    "ModelEvaluator": "evaluator", "FusedAdam": "optim", "bce_loss": "ops", "GradAllReducer": "ddp",
    "TrainStep": "trainer", "fit": "trainer", "UNet": "unet", "cross_entropy": "ops", "bilinear_resize": "ops",
}


def __getattr__(name):
    if name in _LAZY:
        return getattr(_il.import_module(f"{__name__}.{_LAZY[name]}"), name)
    raise AttributeError(name)


__all__ = sorted(_LAZY)
