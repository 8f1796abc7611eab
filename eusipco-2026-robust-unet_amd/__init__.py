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
