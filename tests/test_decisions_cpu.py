"""CPU: the decision-forcing machinery of tests/decisions.py is itself correct - forcing the oracle's OWN decisions reproduces its
gradients bit for bit (ReLU masks, pool winners, attention maxima, gate ReLU), and forcing a different winner changes them."""
import torch

import decisions as D


def _own_decisions(named, n):
    dec = {"gate": {}, "pool": {}, "sa": {}, "ca": {}, "act": {}}
    for pre in D.RB:
        relus = [e for e in named[pre] if e[0] == "relu"]
        # stand-ins with the right sign pattern (forced_from_hip only looks at > 0)
        dec["act"][pre] = dict(a1=relus[0][1].float(), out=relus[-1][1].float())
        dec["sa"][pre] = [e for e in named[pre] if e[0] == "sa_max"][0][1][:, 0]
        dec["ca"][pre] = [e for e in named[pre] if e[0] == "ca_max"][0][1]
    dec["act"]["bottleneck.1"] = dict(out=[e for e in named["bottleneck.1"] if e[0] == "relu"][-1][1].float())
    for g in D.GATES:
        dec["gate"][g] = [e for e in named[g] if e[0] == "relu"][-1][1]
    for lvl, e in enumerate(named["pool"], 1):
        idx, vals = e[1], e[2]
        w = vals.shape[3]
        dec["pool"][lvl] = ((idx // w) % 2) * 2 + (idx % w) % 2
    return dec


def test_forcing_own_decisions_is_the_identity_and_a_flip_is_not(pkg, oracle):
    base, n, size, seed = 16, 2, 32, 3
    st = oracle.init_state(3, 1, base, seed=seed, perturb_bn=True)
    masks = oracle.dropout_masks(n, base, seed=seed)
    x, y = pkg.synthetic_batch(n, size, seed=seed)
    g0, named, _, _ = D.oracle_step(oracle, st, masks, x, y)
    dec = _own_decisions(named, n)
    assert D.differing_decisions(dec, named, None) == []
    g1, _, _, _ = D.oracle_step(oracle, st, masks, x, y, forced=D.forced_from_hip(dec, None))
    assert max(float((g0[k] - g1[k]).abs().max()) for k in g0) == 0.0
    for kind, key, pos in (("pool", 1, (0, 0, 0, 0)), ("sa", "dec1", (0, 0, 0)), ("ca", "dec4", (0, 0))):
        d2 = {k: (dict(v) if isinstance(v, dict) else v) for k, v in dec.items()}
        t = d2[kind][key].clone()
        t[pos] = (t[pos] + 1) % (4 if kind == "pool" else 8)
        d2[kind] = dict(d2[kind])
        d2[kind][key] = t
        flips = D.differing_decisions(d2, named, None)
        assert len(flips) == 1 and flips[0][3] >= 0.0
        g2, _, _, _ = D.oracle_step(oracle, st, masks, x, y, forced=D.forced_from_hip(d2, None))
        assert max(float((g0[k] - g2[k]).abs().max()) for k in g0) > 0.0, kind


def test_sigmoid_saturation_is_a_decision(pkg, oracle):
    """Forcing the oracle's own probabilities changes nothing; un-saturating ONE saturated pixel (p = 1.0 -> 1 - 2^-24 against label 0) is
    listed as a near-tie flip and moves the gradients by that pixel's 1 / n."""
    base, n, size, seed = 16, 2, 32, 3
    st = oracle.init_state(3, 1, base, seed=seed, perturb_bn=True)
    for k in st:                       # push the logits into saturation: a large output bias
        if k == "outc.0.bias":
            st[k] = st[k] + 30.0
    masks = oracle.dropout_masks(n, base, seed=seed)
    x, y = pkg.synthetic_batch(n, size, seed=seed)
    g0, named, prob, _ = D.oracle_step(oracle, st, masks, x, y)
    sat = ((prob == 1.0) & (y == 0)).nonzero()
    assert len(sat) > 0, "no saturated pixel with label 0: the case needs a larger bias"
    dec = _own_decisions(named, n)
    dec["prob"] = prob.clone()
    assert D.differing_decisions(dec, named, None) == []
    g1, _, _, _ = D.oracle_step(oracle, st, masks, x, y, forced=D.forced_from_hip(dec, None))
    assert max(float((g0[k] - g1[k]).abs().max()) for k in g0) == 0.0
    pos = tuple(int(v) for v in sat[0])
    dec["prob"][pos] = 1.0 - 2.0 ** -24
    flips = D.differing_decisions(dec, named, None)
    assert len(flips) == 1 and flips[0][1] == "sigmoid saturation" and flips[0][3] <= D.NEAR_TIE * flips[0][4]
    g2, _, _, _ = D.oracle_step(oracle, st, masks, x, y, forced=D.forced_from_hip(dec, None))
    db = float((g2["outc.0.bias"] - g0["outc.0.bias"]).abs().max())
    assert abs(db - 1.0 / prob.numel()) <= 1e-3 / prob.numel(), db      # d loss / d logit of that pixel went from 0 to (1 - 2^-24) / n


def test_sequential_recorder_forcing_own_decisions_is_the_identity(pkg):
    """tests/decisions_seq.py (plain U-Net / DeepLabV3+): forcing the oracle's own ReLU masks and pool winners reproduces its gradients
    bit for bit; a moved pool winner is listed with its margin and changes them."""
    import importlib

    import decisions_seq as DS
    pu = importlib.import_module("oracle.plain_unet_ref")
    st = pu.init_state(3, 2, seed=4, perturb_bn=True)
    names = pu.param_names(3, 2)
    x, y = pkg.synthetic_batch(2, 32, seed=4)
    target = y[:, 0].long()

    def run(forced):
        P = {k: v.clone() for k, v in st.items()}
        for k in names:
            P[k].requires_grad_(True)

        def step(rec):
            logits = pu.forward(P, x, True)
            return (lambda _: pu.ce_mean(logits, target)), None, None
        log, _ = DS.run_oracle(pu, step, forced)
        return log, {k: P[k].grad for k in names}

    log, g0 = run(None)
    assert [k for k, _, _ in log].count("pool") == 4 and [k for k, _, _ in log].count("relu") == 18
    own = [d for _, d, _ in log]
    assert DS.differing(own, log) == []
    _, g1 = run(own)
    assert max(float((g0[k] - g1[k]).abs().max()) for k in names) == 0.0
    moved = [d.clone() for d in own]
    i = [j for j, (k, _, _) in enumerate(log) if k == "pool"][0]
    w_in = log[i][2].shape[3]
    cur = int(moved[i][0, 0, 0, 0])
    moved[i][0, 0, 0, 0] = cur + 1 if cur % w_in == 0 else cur - 1          # the other column of the same 2x2 window
    flips = DS.differing(moved, log)
    assert len(flips) == 1 and flips[0][1] == "pool" and flips[0][3] >= 0.0
    _, g2 = run(moved)
    assert max(float((g0[k] - g2[k]).abs().max()) for k in names) > 0.0
