"""`ModelEvaluator`: drop-in for /root/reference/Main_Final.py:513-668 (calculate_metrics,
train_model, evaluate_model) on top of the HIP path.

Differences from the reference, all invisible in the returned numbers:
* metrics: the thresholded tp / predicted / target / agreement counts are integer reductions on the
  device (runet_seg_counts); the ratios are formed on the host in float64 exactly as the reference
  does (strict `>` threshold, +1e-8 denominators, empty union -> IoU 0);
* loss/optimizer: fused BCE kernel and FusedAdam (same semantics as nn.BCELoss / torch.optim.Adam
  with L2-coupled weight_decay=1e-4); ReduceLROnPlateau(patience=5, factor=0.5) stepped on the TRAIN
  loss as the reference does (:622);
* `evaluate_model` synchronises the device around the timed forward (the reference does not, so its
  GPU timings are launch times only).
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import ops
from .optim import FusedAdam


def metrics_from_counts(tp, pp, tpos, agree, total):
    """float64 host arithmetic of Main_Final.py:527-547 from integer counts."""
    tp, pp, tpos = float(tp), float(pp), float(tpos)
    union = pp + tpos - tp
    fp, fn = pp - tp, tpos - tp
    precision = tp / (tp + fp + 1e-8)
    recall = tp / (tp + fn + 1e-8)
    return {"accuracy": float(agree) / float(total), "iou": tp / (union + 1e-8), "precision": precision, "recall": recall,
            "f1_score": 2 * precision * recall / (precision + recall + 1e-8)}


class ModelEvaluator:
    def __init__(self, device):
        self.device = device

    # -- metrics ---------------------------------------------------------------------------
    def segmentation_counts(self, pred, target, threshold=0.5):
        """[N, ...] probability / target tensors on the device -> int64 [N, 4] (tp, pred+, target+, agree)."""
        return ops.seg_counts(pred, target, threshold)

    def calculate_metrics(self, pred, target, threshold=0.5):
        """One image (any shape); returns the reference's dict of python floats."""
        c = ops.seg_counts(pred.reshape(1, -1), target.reshape(1, -1), threshold)[0].tolist()
        return metrics_from_counts(c[0], c[1], c[2], c[3], pred.numel())

    def batch_metrics(self, pred, target, threshold=0.5):
        """Per-image metric dicts for a batch with ONE device->host transfer."""
        n = pred.shape[0]
        c = ops.seg_counts(pred, target, threshold).tolist()
        per = pred.numel() // n
        return [metrics_from_counts(c[i][0], c[i][1], c[i][2], c[i][3], per) for i in range(n)]

    # -- training --------------------------------------------------------------------------
    def train_model(self, model, train_loader, val_loader, epochs=25, lr=1e-4, grad_sync=None):
        optimizer = FusedAdam(model.parameters(), lr=lr, weight_decay=1e-4)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, patience=5, factor=0.5)
        if grad_sync is not None:
            grad_sync.attach(optimizer)
        history = {"train_loss": [], "val_loss": [], "val_iou": [], "val_f1": [], "val_accuracy": []}
        best_iou = 0
        for epoch in range(epochs):
            model.train()
            train_loss = 0.0
            for images, masks in train_loader:
                images, masks = images.to(self.device, non_blocking=True), masks.to(self.device, non_blocking=True)
                optimizer.zero_grad()
                outputs = model(images)
                outputs = ops.match_size(outputs, masks)       # Main_Final.py:577-578
                loss = ops.bce_loss(outputs, masks)
                loss.backward()
                if grad_sync is not None:
                    grad_sync.finish()
                optimizer.step()
                train_loss += loss.item()
            model.eval()
            val_loss, val_metrics = 0.0, []
            with torch.no_grad():
                for images, masks in val_loader:
                    images, masks = images.to(self.device), masks.to(self.device)
                    outputs = ops.match_size(model(images), masks)     # Main_Final.py:596-597
                    val_loss += ops.bce_loss(outputs, masks).item()
                    val_metrics += self.batch_metrics(outputs, masks)
            avg_train_loss = train_loss / len(train_loader)
            history["train_loss"].append(avg_train_loss)
            history["val_loss"].append(val_loss / len(val_loader))
            history["val_iou"].append(float(np.mean([m["iou"] for m in val_metrics])))
            history["val_f1"].append(float(np.mean([m["f1_score"] for m in val_metrics])))
            history["val_accuracy"].append(float(np.mean([m["accuracy"] for m in val_metrics])))
            scheduler.step(avg_train_loss)
            best_iou = max(best_iou, history["val_iou"][-1])
            if epoch % 5 == 0:
                print(f"Epoch {epoch:2d}: Train Loss: {avg_train_loss:.4f}, Val Loss: {history['val_loss'][-1]:.4f}, "
                      f"IoU: {history['val_iou'][-1]:.4f}, F1: {history['val_f1'][-1]:.4f}")
        return {"best_iou": best_iou, "history": history}

    def evaluate_model(self, model, test_loader):
        model.eval()
        all_metrics, inference_times = [], []
        with torch.no_grad():
            for images, masks in test_loader:
                images, masks = images.to(self.device), masks.to(self.device)
                torch.cuda.synchronize()
                t0 = time.time()
                outputs = ops.match_size(model(images), masks)     # Main_Final.py:648-649 (inside the timed span, as in the reference)
                torch.cuda.synchronize()
                inference_times.append((time.time() - t0) / images.shape[0])
                all_metrics += self.batch_metrics(outputs, masks)
        results = {}
        for key in all_metrics[0].keys():
            vals = [m[key] for m in all_metrics]
            results[f"mean_{key}"] = np.mean(vals)
            results[f"std_{key}"] = np.std(vals)
        results["avg_inference_time"] = np.mean(inference_times)
        results["total_samples"] = len(all_metrics)
        return results
