"""ctypes binding of csrc/librunet_hip.so; prototypes are read from include/runet_hip.h so the
binding cannot drift from the declared C ABI.

There is no fallback: if the shared library is missing or fails to load, importing this
module raises, and every op in the package is unusable (the product path never routes
through the oracle or any CPU implementation).
"""
from __future__ import annotations

import ctypes as C
import os
import re

# The backward pass uses several HIP streams (main chain, weight gradients, gradient all-reduce + RCCL's own): with HIP's default of
# 4 hardware queues they share queues and serialise on each other's event waits.  Only effective if the HIP runtime is not yet initialised.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # normally already done by the package's __init__ (see there)

import torch  # noqa: F401  (first: librunet_hip.so must bind to the HIP runtime that torch already loaded, not a second copy)

from . import _HW_QUEUES_LATE

if _HW_QUEUES_LATE:
    import warnings
    warnings.warn("the HIP runtime was initialised before eusipco-2026-robust-unet_amd was imported, so GPU_MAX_HW_QUEUES=8 could not take effect: the "
                  "weight-gradient / communication streams will share HIP's default 4 hardware queues (about 8 % slower train steps). "
                  "Import the package (or export GPU_MAX_HW_QUEUES=8) before the first CUDA/HIP call.", RuntimeWarning, stacklevel=2)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RUNET_HIP_LIB: another build of the same library (A/B of two builds inside one GPU call); there is still no fallback if it is missing
LIB_PATH = os.environ.get("RUNET_HIP_LIB") or os.path.join(_HERE, "csrc", "librunet_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "runet_hip.h")

_SCALARS = {"int": C.c_int, "long": C.c_long, "float": C.c_float, "double": C.c_double, "void": None}


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl:
        return C.c_char_p if decl.replace(" ", "") == "constchar*" else C.c_void_p
    base = decl.replace("const", "").split()
    return _SCALARS[base[0]]


def parse_header(path: str = HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?(?:char\*|char \*|int|long|void|float|double))\s+(\w+)\s*\(([^;{]*?)\)\s*;", text, re.M | re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (last identifier) unless the declaration ends with '*'
                t = a if a.endswith("*") else a.rsplit(" ", 1)[0]
                if "*" in a:
                    t = a[: a.rindex("*") + 1]
                argtypes.append(_ctype(t))
        protos[name] = (_ctype(ret), argtypes)
    return protos


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        f"or `make -C {os.path.join(_HERE, 'csrc')}` (hipcc, --offload-arch=gfx950). "
        "There is no CPU fallback for the Robust U-Net kernels.")

lib = C.CDLL(LIB_PATH)
PROTOS = parse_header()
for _name, (_res, _args) in PROTOS.items():
    _fn = getattr(lib, _name)      # AttributeError here = header declares a symbol the library lacks
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc: int):
    if rc != 0:
        raise RuntimeError("librunet_hip: " + lib.runet_last_error().decode("utf-8", "replace"))
