"""CPU: pin the oracle (oracle/robust_unet_ref.py) against the golden vectors that
tests/golden/make_golden.py produced from the reference itself."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz, sampled, tstat

TAP_MAP = {"inc": "x1", "down1": "x2", "down2": "x3", "down3": "x4", "bottleneck.1": "xd", "bottleneck": "x5",
           **{f"{k}{l}": f"{k}{l}" for k in ("up", "att", "dec") for l in (1, 2, 3, 4)}}


def test_state_dict_keys_match_reference(oracle):
    with open(os.path.join(GOLDEN, "state_dict_keys_base64.json")) as f:
        gold = json.load(f)
    spec = oracle.state_spec(3, 1, 64)
    mine = [[k, list(s), str(dt).replace("torch.", "")] for k, s, dt, _ in spec]
    assert mine == gold["entries"]
    assert len(mine) == 290
    assert oracle.param_names(3, 1, 64) == gold["param_order"]
    n_params = sum(int(np.prod(s)) for k, s, dt, kind in spec if not str(kind).startswith("buf"))
    assert n_params == gold["n_params"] == 40872223


def test_init_distribution_matches_reference(oracle):
    with open(os.path.join(GOLDEN, "init_stats_base64.json")) as f:
        gold = json.load(f)
    st = oracle.init_state(3, 1, 64, seed=1, perturb_bn=False)
    for k, (mean, std, lo, hi) in gold.items():
        t = st[k].float()
        if t.numel() < 2048 or k.endswith(("running_mean", "running_var")) or ".bn" in k or k.endswith(".1.weight") \
                or k.endswith(".1.bias"):
            continue
        assert abs(t.std().item() - std) <= 0.06 * std + 1e-6, k
        assert abs(t.mean().item() - mean) <= 0.1 * std, k


@pytest.mark.parametrize("tag", ["b16_n2_s64", "b64_n2_s64"])
def test_full_train_step_matches_reference(oracle, pkg, tag):
    with open(os.path.join(GOLDEN, f"model_{tag}.json")) as f:
        meta = json.load(f)
    gold = load_npz(f"model_{tag}.npz")
    base, n, size, seed = meta["base"], meta["n"], meta["size"], meta["seed"]
    P = oracle.init_state(3, 1, base, seed=seed, perturb_bn=True)
    names = oracle.param_names(3, 1, base)
    assert names == meta["param_names"]
    for k in names:
        P[k].requires_grad_(True)
    x, y = pkg.synthetic_batch(n, size, seed=seed)
    taps = {}
    prob, logit = oracle.forward(P, x, True, oracle.dropout_masks(n, base, seed=seed), taps)
    loss = oracle.bce_mean(prob, y)
    loss.backward()
    np.testing.assert_allclose(prob.detach().numpy(), gold["prob"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(logit.detach().numpy(), gold["logit"], rtol=2e-4, atol=2e-4)
    assert abs(loss.item() - float(gold["loss"])) <= 1e-5 * max(1.0, abs(float(gold["loss"])))
    for rk, ok in TAP_MAP.items():
        np.testing.assert_allclose(sampled(taps[ok], gold[f"tap/{rk}/meta"]), gold[f"tap/{rk}/sample"], rtol=2e-4,
                                   atol=2e-4, err_msg=rk)
        np.testing.assert_allclose(tstat(taps[ok]), gold[f"tap/{rk}/stat"], rtol=1e-4, atol=1e-4)
    gn = np.array([P[k].grad.double().norm().item() for k in names])
    np.testing.assert_allclose(gn, gold["grad_norm"], rtol=2e-3, atol=1e-7)
    for k in names:
        if f"grad/{k}" in gold:
            np.testing.assert_allclose(P[k].grad.numpy(), gold[f"grad/{k}"], rtol=2e-3,
                                       atol=2e-3 * float(np.abs(gold[f"grad/{k}"]).max()) + 1e-9, err_msg=k)
    for k, v in P.items():
        if k.endswith(("running_mean", "running_var")) and f"buf/{k}" in gold:
            np.testing.assert_allclose(v.numpy(), gold[f"buf/{k}"], rtol=1e-4, atol=1e-5, err_msg=k)
    nbt = np.array([v.item() for k, v in P.items() if k.endswith("num_batches_tracked")])
    np.testing.assert_array_equal(nbt, gold["num_batches_tracked"])
    # one Adam step
    params = [P[k] for k in names]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    before = [p.detach().clone() for p in params]
    oracle.adam_step(params, [p.grad for p in params], m, v, 1, lr=meta["lr"], weight_decay=meta["weight_decay"])
    delta = np.array([(p.detach().double() - b.double()).abs().sum().item() for p, b in zip(params, before)])
    np.testing.assert_allclose(delta, gold["param_delta_abs_sum"], rtol=2e-3, atol=1e-9)
    # eval-mode forward with updated weights/buffers + metrics
    with torch.no_grad():
        pe, _ = oracle.forward(P, x, False)
    np.testing.assert_allclose(pe.numpy(), gold["eval_prob"], rtol=0, atol=5e-5)
    for i in range(n):
        mets = oracle.seg_metrics(pe[i, 0].numpy(), y[i, 0].numpy())
        for key, val in mets.items():
            assert abs(val - gold[f"eval_metric/{key}"][i]) <= 1e-3, (key, i)


def test_metrics_match_reference(oracle):
    g = load_npz("metrics.npz")
    for i in range(g["pred"].shape[0]):
        m = oracle.seg_metrics(g["pred"][i], g["target"][i])
        for k, v in m.items():
            assert abs(v - g[k][i]) < 1e-12, (k, i)
    assert g["iou"][4] == 0.0  # empty union -> 0, not 1


def test_labelme_masks_match_reference(oracle, pkg):
    g = load_npz("labelme_masks.npz")
    ds = pkg.CoastalDataset([], [])
    for name, mask in g.items():
        path = os.path.join(GOLDEN, "labelme", name + ".json")
        np.testing.assert_array_equal(oracle.labelme_mask(path, (64, 48)), mask, err_msg=name)
        np.testing.assert_array_equal(ds.create_mask_from_labelme(path, (64, 48)), mask, err_msg=name)
    assert g["two_labels"].sum() > 0 and g["bad_json"].sum() == 0 and g["missing_file"].sum() == 0
