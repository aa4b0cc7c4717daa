"""Checkpoint loading for the drop-in models: harness counterpart of ``load_saved_model`` (tools/train_utils.py:30-116).

The reference SAVES a wrapper dict ``{"epoch", "model_state_dict", "optimizer_state_dict", ...}`` (tools/train.py:250-260)
but its loader feeds the file straight to ``load_state_dict`` as if it were a raw state_dict, so a
checkpoint written by its own train.py loses every parameter on reload.  This loader accepts BOTH layouts, strips the
``module.`` prefix of DataParallel / DDP wrappers, keeps the model's value for missing or shape-mismatched entries (as the
reference does, :93-113) and reports what it did."""
from __future__ import annotations

import glob
import os
import re

import torch


def find_last_checkpoint(save_dir):
    """Highest N among ``*epochN.pth`` in ``save_dir`` (0 if none) -- what the reference's findLastCheckpoint intends."""
    epochs = []
    for f in glob.glob(os.path.join(save_dir, "*epoch*.pth")):
        m = re.findall(r".*epoch(\d+)\.pth.*", os.path.basename(f))
        if m:
            epochs.append(int(m[0]))
    return max(epochs) if epochs else 0


def extract_state_dict(obj):
    """Raw state_dict from either checkpoint layout; ``module.`` prefixes removed."""
    if isinstance(obj, dict) and "model_state_dict" in obj and not torch.is_tensor(obj["model_state_dict"]):
        obj = obj["model_state_dict"]
    if not isinstance(obj, dict):
        raise TypeError(f"checkpoint holds a {type(obj).__name__}, not a state_dict")
    out = {}
    for k, v in obj.items():
        out[k[7:] if k.startswith("module.") and not k.startswith("module_list") else k] = v
    return out


def load_state_into(model, state_dict, verbose=False):
    """Reference semantics (:93-113): entries with a matching name and shape are loaded, everything else keeps the model's
    current value.  Returns {"loaded": n, "dropped": [...], "missing": [...], "shape_mismatch": [...]}."""
    sd = extract_state_dict(state_dict)
    own = model.state_dict()
    use, report = {}, {"loaded": 0, "dropped": [], "missing": [], "shape_mismatch": []}
    for k, v in sd.items():
        if k not in own:
            report["dropped"].append(k)
        elif tuple(v.shape) != tuple(own[k].shape):
            report["shape_mismatch"].append(k)
        else:
            use[k] = v
            report["loaded"] += 1
    report["missing"] = [k for k in own if k not in sd]
    model.load_state_dict(use, strict=False)
    if verbose:
        for key in ("dropped", "missing", "shape_mismatch"):
            for k in report[key]:
                print(f"{key.replace('_', ' ')}: {k}")
    return report


def load_saved_model(saved_path, model, epoch=None, verbose=False):
    """``(initial_epoch, model)`` like the reference; ``saved_path`` is the run directory holding ``net_epochN.pth``."""
    assert os.path.exists(saved_path), "{} not found".format(saved_path)
    initial_epoch = find_last_checkpoint(saved_path) if epoch is None else int(epoch)
    if initial_epoch > 0:
        ckpt = torch.load(os.path.join(saved_path, "net_epoch%d.pth" % initial_epoch), map_location="cpu")
        load_state_into(model, ckpt, verbose)
    return initial_epoch, model
