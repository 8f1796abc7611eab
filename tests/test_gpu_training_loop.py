"""GPU: the drop-in training / evaluation loops on a tiny synthetic Labelme dataset: ModelEvaluator.train_model +
evaluate_model (Main_Final.py:549-668 semantics), trainer.fit (checkpoint + early stop, train_water_segmentation.py:514-645
semantics) and the checkpoint's interchange with a plain OIHW state_dict (what the reference's RobustUNet loads)."""
import importlib
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _dataset(d, n=10, size=64):
    os.makedirs(os.path.join(d, "img")); os.makedirs(os.path.join(d, "ann"))
    rng = np.random.RandomState(0)
    for i in range(n):
        img = rng.randint(0, 60, (size, size, 3), dtype=np.uint8)
        x0 = 8 + 3 * (i % 5)
        img[:, x0:x0 + 30] += 150                                   # bright "water" band = the labelled polygon
        Image.fromarray(img).save(os.path.join(d, "img", f"t{i:02d}.png"))
        with open(os.path.join(d, "ann", f"t{i:02d}.json"), "w") as f:
            json.dump({"shapes": [{"label": "water", "points": [[x0, 0], [x0 + 30, 0], [x0 + 30, size], [x0, size]]}]}, f)


def test_train_evaluate_and_checkpoint_loops(pkg, oracle, tmp_path):
    dev = torch.device("cuda:0")
    d = str(tmp_path)
    _dataset(d)
    train, val = pkg.prepare_dataset(os.path.join(d, "img"), os.path.join(d, "ann"), batch_size=4, image_size=(64, 64))
    torch.manual_seed(0)
    model = pkg.RobustUNet(3, 1, 16).to(dev)
    ev = pkg.ModelEvaluator(dev)
    out = ev.train_model(model, train, val, epochs=3, lr=1e-3)
    h = out["history"]
    assert len(h["train_loss"]) == 3 and all(np.isfinite(h["train_loss"])) and h["train_loss"][-1] < h["train_loss"][0]
    res = ev.evaluate_model(model, val)
    assert res["total_samples"] == 2 and 0.0 <= res["mean_iou"] <= 1.0 and res["avg_inference_time"] > 0
    hist = pkg.fit(model, train, val, dev, epochs=4, lr=1e-3, save_dir=os.path.join(d, "models"), stop_patience=1, log=lambda s: None)
    assert 1 <= len(hist["iou_scores"]) <= 4 and len(hist["learning_rates"]) == len(hist["val_losses"])
    ck = torch.load(os.path.join(d, "models", "best_water_segmentation_model.pth"), weights_only=True)
    spec = oracle.state_spec(3, 1, 16)
    assert list(ck.keys()) == [k for k, _, _, _ in spec]
    assert all(v.is_contiguous() and tuple(v.shape) == tuple(s) for (k, s, _, _), v in zip(spec, ck.values()))
    # the checkpoint evaluated by the oracle (CPU restatement of the reference) gives the device's probabilities
    x, _ = next(iter(val))
    m2 = pkg.RobustUNet(3, 1, 16)
    m2.load_state_dict(ck)
    m2 = m2.to(dev).eval()
    with torch.no_grad():
        p_dev = m2(x.to(dev)).cpu()
        p_ref, _ = oracle.forward({k: v.clone() for k, v in ck.items()}, x, training=False)
    np.testing.assert_allclose(p_dev.numpy(), p_ref.numpy(), rtol=0, atol=1e-3)


def test_deeplab_baseline_trains_through_the_same_evaluator(pkg, tmp_path):
    """Main_Final.py:841-871 runs every model of its dict through the same ModelEvaluator; config 4 of BASELINE.json."""
    dev = torch.device("cuda:0")
    d = str(tmp_path)
    _dataset(d)
    train, val = pkg.prepare_dataset(os.path.join(d, "img"), os.path.join(d, "ann"), batch_size=4, image_size=(64, 64))
    torch.manual_seed(0)
    model = pkg.DeepLabV3Plus(n_classes=1).to(dev)
    ev = pkg.ModelEvaluator(dev)
    out = ev.train_model(model, train, val, epochs=4, lr=1e-3)
    h = out["history"]
    assert len(h["train_loss"]) == 4 and all(np.isfinite(h["train_loss"])) and h["train_loss"][-1] < h["train_loss"][0]
    res = ev.evaluate_model(model, val)
    assert res["total_samples"] == 2 and 0.0 <= res["mean_iou"] <= 1.0
