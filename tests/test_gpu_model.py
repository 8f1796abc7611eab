"""GPU parity of the full Robust U-Net train step (forward, BCE, backward, Adam, BN buffers, eval IoU)
through the C ABI against golden vectors captured from the reference (tests/golden/model_*.npz)."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

import decisions as D
from conftest import GOLDEN, load_npz

pytestmark = pytest.mark.gpu


def build(pkg, oracle, meta, dev):
    base, seed = meta["base"], meta["seed"]
    model = pkg.RobustUNet(3, 1, base)
    st = oracle.init_state(3, 1, base, seed=seed, perturb_bn=True)
    res = model.load_state_dict(st, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return model.to(dev), st


@pytest.mark.parametrize("tag", ["b16_n2_s64", "b64_n2_s64"])
def test_train_step_matches_reference(pkg, oracle, tag):
    dev = torch.device("cuda:0")
    with open(os.path.join(GOLDEN, f"model_{tag}.json")) as f:
        meta = json.load(f)
    gold = load_npz(f"model_{tag}.npz")
    model, st = build(pkg, oracle, meta, dev)
    n, size, base, seed = meta["n"], meta["size"], meta["base"], meta["seed"]
    assert [k for k, _ in model.named_parameters()] == meta["param_names"]
    assert model.inc.conv1.weight.stride() == (1, base, 3 * 3 * base, 3 * base)   # still HWIO after load + .to()
    x, y = pkg.synthetic_batch(n, size, seed=seed)
    x, y = x.to(dev), y.to(dev)
    model.train()
    model.set_dropout_masks(oracle.dropout_masks(n, base, seed=seed))
    opt = pkg.FusedAdam(model.parameters(), lr=meta["lr"], weight_decay=meta["weight_decay"])
    opt.zero_grad()
    prob, logit = model(x, return_logits=True)
    loss = pkg.bce_loss(prob, y)
    loss.backward()
    # north_star tolerance: logits (pre-sigmoid) and probabilities within 1e-3 fp32
    np.testing.assert_allclose(prob.detach().cpu().numpy(), gold["prob"], rtol=0, atol=1e-3)
    lg = gold["logit"]
    np.testing.assert_allclose(logit.detach().cpu().numpy(), lg, rtol=1e-3, atol=1e-3 * max(1.0, float(np.abs(lg).max()) / 10))
    assert abs(loss.item() - float(gold["loss"])) <= 1e-3 * max(1.0, abs(float(gold["loss"])))
    names = meta["param_names"]
    gn = np.array([p.grad.double().norm().item() for _, p in model.named_parameters()])
    ref = gold["grad_norm"]
    rel = np.abs(gn - ref) / (ref + 1e-3 * ref.max())
    assert rel.max() < 2e-2, f"grad-norm mismatch at {names[int(rel.argmax())]}: {gn[int(rel.argmax())]} vs {ref[int(rel.argmax())]}"
    for k, p in model.named_parameters():
        if f"grad/{k}" in gold:
            gref = gold[f"grad/{k}"]
            if k.endswith(".bias") and (k.startswith("bottleneck.1.conv") or ".W_g.0." in k or ".W_x.0." in k or ".psi.0." in k):
                # conv bias in front of a train-mode BatchNorm: analytically zero gradient, rounding noise on both sides
                assert np.abs(gref).max() <= 1e-5 * ref.max() and p.grad.abs().max().item() <= 1e-5 * ref.max(), k
                continue
            # element-wise against the FIXED golden gradients.  The b64 golden input carries three ReLU near-ties (dec3 out (1,34,0,1):
            # -1.3e-7, dec1 bn1 (1,44,63,20): -1.2e-6, dec1 out (1,22,47,63): -7.7e-7 on tensors of scale 6-9) that the HIP step
            # takes the other way, exactly as the reference does to itself with oneDNN on / off (tests/diagnostics/decision_flips.py:
            # same median 6e-4..1.2e-3, same worst tensors).  The tight gradient check, free of that lottery, is
            # test_gradients_match_oracle_under_the_same_decisions below (1e-4 of scale); this one keeps the loose band.
            scale = float(np.abs(gref).max()) + 1e-7 * float(ref.max())
            err = np.abs(p.grad.cpu().numpy() - gref)
            assert err.max() <= 3e-2 * scale, (k, err.max(), scale)
            allowed = max(1, int(0.05 * err.size))
            assert int((err > 5e-3 * scale).sum()) <= allowed, (k, int((err > 5e-3 * scale).sum()), err.size)
    for k, b in model.named_buffers():
        if f"buf/{k}" in gold:
            np.testing.assert_allclose(b.cpu().numpy(), gold[f"buf/{k}"], rtol=1e-3, atol=1e-4, err_msg=k)
    nbt = np.array([b.item() for k, b in model.named_buffers() if k.endswith("num_batches_tracked")])
    np.testing.assert_array_equal(nbt, gold["num_batches_tracked"])
    opt.step()
    delta = np.array([(p.detach().cpu().double() - st[k].double()).abs().sum().item() for k, p in model.named_parameters()])
    np.testing.assert_allclose(delta, gold["param_delta_abs_sum"], rtol=2e-2, atol=1e-9)
    # eval forward with updated weights/buffers -> IoU within +-0.001 of the reference
    model.eval()
    with torch.no_grad():
        pe = model(x)
    np.testing.assert_allclose(pe.cpu().numpy(), gold["eval_prob"], rtol=0, atol=1e-3)
    counts = pkg.ModelEvaluator(dev).segmentation_counts(pe, y).cpu().numpy()
    for i in range(n):
        m = pkg.ModelEvaluator(dev).calculate_metrics(pe[i, 0], y[i, 0])
        for key in ("accuracy", "iou", "precision", "recall", "f1_score"):
            assert abs(m[key] - gold[f"eval_metric/{key}"][i]) <= 1e-3, (key, i, m[key])
    assert counts.shape == (n, 4)


def test_state_dict_roundtrip_and_foreign_layout(pkg, oracle):
    dev = torch.device("cuda:0")
    model = pkg.RobustUNet(3, 1, 16).to(dev)
    sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}      # plain OIHW-contiguous copies
    m2 = pkg.RobustUNet(3, 1, 16)
    m2.load_state_dict(sd)
    m2 = m2.to(dev).eval()
    model.eval()
    x, _ = pkg.synthetic_batch(1, 32, seed=9)
    with torch.no_grad():
        a, b = model(x.to(dev)), m2(x.to(dev))
    assert torch.equal(a, b)


def test_input_validation(pkg):
    dev = torch.device("cuda:0")
    model = pkg.RobustUNet(3, 1, 16).to(dev)
    with pytest.raises(ValueError):
        model(torch.zeros((1, 3, 40, 40), device=dev))           # not a multiple of 16
    with pytest.raises(RuntimeError):
        model(torch.zeros((1, 3, 32, 32)))                       # CPU tensor: no fallback


def test_reference_style_loop_with_stock_loss_and_optimizer(pkg):
    """the reference's own loop (nn.BCELoss + torch.optim.Adam) runs against the drop-in module,
    and gives the same update as the fused loss/optimizer."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    a = pkg.RobustUNet(3, 1, 16).to(dev)
    b = pkg.RobustUNet(3, 1, 16).to(dev)
    b.load_state_dict(a.state_dict())
    x, y = pkg.synthetic_batch(2, 32, seed=11)
    x, y = x.to(dev), y.to(dev)
    masks = {k: torch.ones(2, rb.out_channels) for k, rb in a._rbs().items()}
    for m in (a, b):
        m.train()
        m.set_dropout_masks(masks)
    opt_a = torch.optim.Adam(a.parameters(), lr=1e-4, weight_decay=1e-4)
    opt_a.zero_grad()
    loss_a = torch.nn.BCELoss()(a(x), y)
    loss_a.backward()
    opt_a.step()
    opt_b = pkg.FusedAdam(b.parameters(), lr=1e-4, weight_decay=1e-4)
    opt_b.zero_grad()
    loss_b = pkg.bce_loss(b(x), y)
    loss_b.backward()
    opt_b.step()
    assert abs(loss_a.item() - loss_b.item()) < 1e-6
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(pa, pb, rtol=0, atol=2e-6), k
    # gradient accumulation: a second backward without zero_grad adds into p.grad
    g1 = b.inc.conv2.weight.grad.clone()
    pkg.bce_loss(b(x), y).backward()
    g2 = b.inc.conv2.weight.grad
    assert not torch.equal(g1, g2) and float(g2.abs().sum()) > float(g1.abs().sum())


@pytest.mark.parametrize("n,h,w,base", [(2, 48, 80, 16), (2, 16, 16, 16), (3, 32, 96, 32)])
def test_rectangular_and_minimum_sizes_against_the_oracle(pkg, oracle, n, h, w, base):
    """H != W, odd deep-level sizes (48x80 -> 3x5 bottleneck: neither Winograd form applies there, the 24x40 / 12x20 / 6x10 levels mix
    F(4x4), F(2x2) and direct kernels) and the smallest legal tile (16x16 -> 1x1 bottleneck): train step against the CPU oracle."""
    dev = torch.device("cuda:0")
    seed = 100 + h + w
    model = pkg.RobustUNet(3, 1, base)
    model.load_state_dict(oracle.init_state(3, 1, base, seed=seed, perturb_bn=True))
    model = model.to(dev).train()
    masks = oracle.dropout_masks(n, base, seed=seed)
    model.set_dropout_masks({k: v.to(dev) for k, v in masks.items()})
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, h, w, generator=g)
    y = (torch.rand(n, 1, h, w, generator=g) > 0.5).float()
    prob, logit = model(x.to(dev), return_logits=True)
    loss = pkg.bce_loss(prob, y.to(dev))
    loss.backward()
    P = oracle.init_state(3, 1, base, seed=seed, perturb_bn=True)
    names = oracle.param_names(3, 1, base)
    for k in names:
        P[k].requires_grad_(True)
    rp, rl = oracle.forward(P, x, True, masks)
    rloss = oracle.bce_mean(rp, y)
    rloss.backward()
    np.testing.assert_allclose(prob.detach().cpu().numpy(), rp.detach().numpy(), rtol=0, atol=1e-3)
    lg = rl.detach().numpy()
    np.testing.assert_allclose(logit.detach().cpu().numpy(), lg, rtol=1e-3, atol=1e-3 * max(1.0, float(np.abs(lg).max()) / 10))
    assert abs(float(loss.detach()) - float(rloss.detach())) <= 1e-3 * max(1.0, abs(float(rloss.detach())))
    # gradients: decision-aware (near-tie decisions forced into the oracle), 1e-4 of each tensor's scale instead of 3e-2 on norms.  The
    # 16x16 tile keeps the old band: its bottleneck BatchNorms normalise TWO values per channel (N=2, 1x1 pixels), xhat = +-1 and
    # 1/sigma amplifies every rounding difference of the encoder by orders of magnitude on both sides
    tight = h * w > 256
    D.check_step(pkg, oracle, base, n, seed, x, y, tol=1e-4 if tight else 5e-2, median_tol=2e-5 if tight else 5e-3)


def test_single_value_per_channel_in_training_raises_like_torch(pkg):
    """1 x 16 x 16 in train mode reaches the bottleneck BatchNorm with ONE value per channel: torch raises ("Expected more than 1 value
    per channel when training"), so does the HIP path; eval mode works."""
    dev = torch.device("cuda:0")
    model = pkg.RobustUNet(3, 1, 16).to(dev)
    x = torch.zeros(1, 3, 16, 16, device=dev)
    model.train()
    with pytest.raises(RuntimeError):
        model(x)
    model.eval()
    with torch.no_grad():
        assert model(x).shape == (1, 1, 16, 16)


@pytest.mark.parametrize("base,n,size,seed", [(16, 2, 64, 3), (64, 2, 64, 5), (64, 2, 64, 12), (32, 3, 32, 21)])
def test_gradients_match_oracle_under_the_same_decisions(pkg, oracle, base, n, size, seed):
    """All 173 gradient tensors within 1e-4 of their scale (median 2e-5) of the fp32 oracle evaluated under the same discrete decisions
    (ReLU masks, pool winners, attention maxima); decisions differ only at near-ties (see tests/decisions.py).  (64, 2, 64, 5) is the golden case of
    test_train_step_matches_reference, whose loose band this test replaces."""
    x, y = pkg.synthetic_batch(n, size, seed=seed)
    D.check_step(pkg, oracle, base, n, seed, x, y, tol=1e-4, median_tol=2e-5)


def test_eval_mode_backward_uses_the_running_statistics(pkg, oracle):
    """model.eval(); loss.backward() (frozen-BN fine-tuning, saliency maps): BatchNorm is a per-channel affine map there, so dx has no
    batch-statistics terms; gradients against the oracle's autograd with training=False."""
    base, n, size, seed = 16, 2, 64, 3
    x, y = pkg.synthetic_batch(n, size, seed=seed)
    D.check_step(pkg, oracle, base, n, seed, x, y, tol=1e-4, median_tol=2e-5, training=False)
