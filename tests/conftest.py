import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG_NAME = "eusipco-2026-robust-unet_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def oracle():
    return importlib.import_module("oracle.robust_unet_ref")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def sampled(t, meta):
    """Re-take the strided sample stored by make_golden.summary()."""
    stride, numel = int(meta[0]), int(meta[1])
    flat = t.detach().double().reshape(-1).cpu()
    assert flat.numel() == numel
    return flat[::stride][:2048].float().numpy()


def tstat(t):
    t = t.detach().double().reshape(-1).cpu()
    return np.array([t.mean().item(), t.std().item(), t.min().item(), t.max().item(), t.abs().sum().item()])
