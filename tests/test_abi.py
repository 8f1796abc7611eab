"""CPU: the C-ABI shared library loads and exports every symbol include/runet_hip.h declares
(no compute calls here: there is no GPU in the build container)."""
import ctypes
import importlib
import os
import re

from conftest import ROOT


def test_header_symbols_are_exported():
    lib_mod = importlib.import_module("eusipco-2026-robust-unet_amd._lib")
    protos = lib_mod.parse_header(os.path.join(ROOT, "include", "runet_hip.h"))
    assert len(protos) >= 39
    raw = ctypes.CDLL(lib_mod.LIB_PATH)
    for name in protos:
        assert hasattr(raw, name), f"{name} declared in include/runet_hip.h but not exported by librunet_hip.so"
    assert raw.runet_abi_version() == 1


def test_header_has_no_torch_types_and_cites_reference():
    text = open(os.path.join(ROOT, "include", "runet_hip.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S).lower()
    assert "at::" not in text and "Tensor" not in text
    assert text.count("Main_Final.py:") >= 8            # every group names the reference lines it replaces


def test_host_side_validation_returns_error_codes_without_a_gpu():
    lib_mod = importlib.import_module("eusipco-2026-robust-unet_amd._lib")
    lib = lib_mod.lib
    # null pointers / bad shapes are rejected on the host before any launch
    assert lib.runet_conv_igemm(None, 16, None, None, None, 16, 1, 4, 4, 16, 16, 16, 3, 3, 1, 0, 0, None) != 0
    assert b"null pointer" in lib.runet_last_error()
    assert lib.runet_maxpool2_fwd(ctypes.c_void_p(16), 4, ctypes.c_void_p(16), 4, ctypes.c_void_p(16), 1, 3, 4, 4, None) != 0
    assert b"even" in lib.runet_last_error()
    assert lib.runet_adam_chunk_elems() > 0
    assert lib.runet_conv_wgrad_workspace_floats(16, 256, 256, 64, 64, 3, 3) > 0
    name = lib.runet_conv_igemm_kernel_name(16, 256, 256, 64, 64, 3, 0)
    assert name.startswith(b"igemm_kernel<")
