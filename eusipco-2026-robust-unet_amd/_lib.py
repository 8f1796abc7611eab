"""ctypes binding of csrc/librunet_hip.so (C ABI declared in include/runet_hip.h).

There is no fallback: if the shared library is missing or fails to load, importing this
module raises, and every op in the package is unusable (the product path never routes
through the oracle or any CPU implementation).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librunet_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        f"or `make -C {os.path.join(_HERE, 'csrc')}` (hipcc, --offload-arch=gfx950). "
        "There is no CPU fallback for the Robust U-Net kernels.")

lib = C.CDLL(LIB_PATH)

P, I, L, F, D = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double

_SIGS = {
    "runet_abi_version": (I, []),
    "runet_conv_igemm": (I, [P, I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "runet_conv_wgrad_workspace_floats": (L, [I, I, I, I, I, I, I]),
    "runet_conv_wgrad": (I, [P, I, P, I, P, P, L, I, I, I, I, I, I, I, I, I, I, P]),
    "runet_chan_stats": (I, [P, I, I, I, I, P, P, P, P, P, P, I, P]),
    "runet_bn_finalize": (I, [P, P, I, I, L, P, P, P, P, F, F, I, P, P, P, P, P]),
    "runet_bn_apply": (I, [P, I, P, I, L, I, I, P, P, P, I, P]),
    "runet_bn_bwd_reduce": (I, [P, I, P, I, P, I, P, I, I, I, P, P, P, P, I, P]),
    "runet_bn_bwd_finalize": (I, [P, I, I, L, P, P, P, P, P, P]),
    "runet_bn_bwd_apply": (I, [P, I, P, I, P, I, P, P, I, I, I, P, P, P, P, I, P]),
    "runet_maxpool2_fwd": (I, [P, I, P, I, P, I, I, I, I, P]),
    "runet_maxpool2_bwd": (I, [P, I, P, P, I, I, I, I, I, I, P]),
    "runet_nchw_to_nhwc_pad": (I, [P, P, I, I, I, I, P]),
    "runet_ca_coeff": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P]),
    "runet_sa_reduce": (I, [P, I, P, P, I, I, I, P, P, P]),
    "runet_sa_conv7": (I, [P, P, P, I, I, I, P]),
    "runet_rb_out": (I, [P, I, P, P, P, P, I, P, P, P, I, I, I, I, P]),
    "runet_rb_bwd1": (I, [P, I, P, I, P, I, P, P, P, P, I, P, I, I, I, P]),
    "runet_sa_conv7_bwd": (I, [P, P, P, P, P, P, I, I, I, P]),
    "runet_rb_bwd2": (I, [P, I, P, I, P, P, P, P, I, I, I, P, P, P, I, P]),
    "runet_ca_bwd_coeff": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P]),
    "runet_rb_bwd3": (I, [P, I, P, I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "runet_ag_psi": (I, [P, I, P, I, P, P, P, P, P, P, P, L, I, P]),
    "runet_ag_out": (I, [P, I, P, P, P, P, I, L, I, P]),
    "runet_ag_bwd1": (I, [P, I, P, I, P, P, P, P, I, P, L, I, P]),
    "runet_ag_bwd2": (I, [P, P, I, P, I, P, P, P, P, P, P, I, P, P, L, I, I, P]),
    "runet_outc_fwd": (I, [P, I, P, P, P, P, L, I, P]),
    "runet_outc_bwd": (I, [P, P, P, I, P, P, I, P, P, L, I, I, P]),
    "runet_bce_fwd": (I, [P, P, P, L, I, P]),
    "runet_bce_finalize": (I, [P, I, L, P, P]),
    "runet_bce_bwd": (I, [P, P, P, P, L, P]),
    "runet_adam_multi": (I, [P, I, L, I, F, F, F, F, F, F, F, P]),
    "runet_seg_counts": (I, [P, P, P, I, L, F, P]),
    "runet_fill": (I, [P, F, L, P]),
    "runet_axpy": (I, [P, P, F, L, P]),
    "runet_chan_sum": (I, [P, I, L, I, P, P, I, P]),
}

for _name, (_res, _args) in _SIGS.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        continue  # checked by tests/test_abi.py against include/runet_hip.h
    _fn.restype = _res
    _fn.argtypes = _args

lib.runet_last_error.restype = C.c_char_p


def check(rc: int):
    if rc != 0:
        raise RuntimeError("librunet_hip: " + lib.runet_last_error().decode("utf-8", "replace"))
