"""CPU: pin oracle/deeplab_ref.py (DeepLabV3+ baseline restatement) against goldens captured from the reference's own
DeepLabV3Plus class (tests/golden/make_golden.py deeplab)."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz, sampled

TAPS = {"conv1": "conv1", "conv2": "conv2", "conv3": "conv3", "conv4": "conv4", "aspp": "aspp", "decoder.2": "decoder.0",
        "decoder.5": "decoder.3", "decoder.8": "decoder.6", "decoder.11": "decoder.9", "decoder.12": "logit"}


@pytest.fixture(scope="module")
def dl():
    return importlib.import_module("oracle.deeplab_ref")


@pytest.mark.parametrize("tag", ["n2_s64", "n2_s128"])
def test_deeplab_oracle_matches_reference(dl, pkg, tag):
    with open(os.path.join(GOLDEN, f"deeplab_{tag}.json")) as f:
        meta = json.load(f)
    gold = load_npz(f"deeplab_{tag}.npz")
    P = dl.init_state(seed=meta["seed"], perturb_bn=True)
    assert [[k, list(v.shape), str(v.dtype)] for k, v in P.items()] == meta["state_dict"]
    names = dl.param_names()
    assert names == meta["param_names"]
    assert sum(P[k].numel() for k in names) == meta["n_params"]
    for k in names:
        P[k].requires_grad_(True)
    x, y = pkg.synthetic_batch(meta["n"], meta["size"], seed=meta["seed"])
    taps = {}
    prob = dl.forward(P, x, True, taps)
    loss = torch.nn.functional.binary_cross_entropy(prob, y)
    loss.backward()
    np.testing.assert_allclose(prob.detach().numpy(), gold["prob"], rtol=0, atol=2e-5)
    assert abs(loss.item() - float(gold["loss"])) <= 1e-5
    for rk, ok in TAPS.items():
        np.testing.assert_allclose(sampled(taps[ok], gold[f"tap/{rk}/meta"]), gold[f"tap/{rk}/sample"], rtol=2e-4, atol=2e-4, err_msg=rk)
    gn = np.array([P[k].grad.double().norm().item() for k in names])
    np.testing.assert_allclose(gn, gold["grad_norm"], rtol=2e-3, atol=1e-6 * gold["grad_norm"].max())
    for k, v in P.items():
        if k.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(v.numpy(), gold[f"buf/{k}"], rtol=1e-4, atol=1e-5, err_msg=k)
