"""Checkpoint ingest for `tools/test_net.py`-style callers.

Mirrors the behaviour of the reference's loader for the DiffusionVID path:
  mega_core/utils/checkpoint.py:50-85, :117-155   Checkpointer.load / DetectronCheckpointer._load_file
  mega_core/utils/model_serialization.py:12-73    align_and_update_state_dicts (longest-suffix matching)
                                        :76-85    strip_prefix_if_present ("module." of DataParallel checkpoints)
                                        :88-138   remove_modules (DiffusionDet -> DiffusionVID head renaming)
                                        :140-156  load_state_dict (strict load of the merged dict)
  mega_core/utils/c2_model_loading.py:150-165, :199-214   pickled weights; the "R-101-torchvision" branch the shipped
                                        YAML selects (configs/vid_R_101_DiffusionVID.yaml:3-7): `pickle["model"]`, numpy arrays
                                        under detectron2 names (`stem.conv1.weight`, `res2.0.conv1.norm.weight`, ...)

What happens after the merge is this package's own: `DiffusionDet.load_state_dict` drops the repacked engine, and the
next forward folds FrozenBN, rounds to fp16 and repacks into the MFMA operand layouts (csrc/model.hip,
dvid_model_finalize).  Caffe2 blob renaming (c2_model_loading.py:12-129) targets torchvision-style module names that
the DiffusionVID backbone (detectron2 names) never matches; it is not built and asking for it raises.
"""
import logging
import os
import pickle
import re
from collections import OrderedDict
from itertools import accumulate

import numpy as np
import torch

logger = logging.getLogger(__name__)


def strip_prefix_if_present(state_dict, prefix):
    """model_serialization.py:76-85: the prefix goes only when EVERY key carries it."""
    if not all(k.startswith(prefix) for k in state_dict):
        return state_dict
    return OrderedDict((k[len(prefix):], v) for k, v in state_dict.items())


def remap_diffusiondet_heads(model_keys, state_dict, skip_names=None):
    """model_serialization.py:88-138 (`remove_modules`).  A DiffusionDet checkpoint numbers all heads in ONE
    `head_series.{i}` list; DiffusionVID keeps the first NUM_HEADS there and moves the rest to `head_series_cond.{j}`
    (`head_series_local.*` is an older spelling of the same module).  The counts come from the MODEL's keys.  The
    reference rewrites the first digit run of the key in place, one character wide; so does this (single-digit head
    indices, which is all the model family has)."""
    names = ["head_series", "head_series_cond"]
    counts = []
    for kwd in names:
        nums = [int(k.split(kwd + ".")[1][0]) for k in model_keys if kwd + "." in k]
        counts.append(max(nums) + 1 if nums else 0)       # (the reference assumes both lists exist; a head-less model has neither)
    lo, hi = list(accumulate(counts))
    change = ["head_series." + str(i) for i in range(lo, hi)]
    if skip_names is not None:        # training-time partial loads (checkpoint.py:62-65); inference passes None
        skip = list(skip_names) + [f"{i}.block_time_mlp.1.{p}" for p in ("weight", "bias") for i in range(lo, hi)]
    else:
        skip = []
    out = OrderedDict()
    for key, value in state_dict.items():
        if any(n in key for n in skip):
            continue
        if any(n in key for n in change):
            m = re.search(r"\d+", key)
            if m:
                chars = list(key)
                chars[m.start()] = str(int(m.group()) - counts[0])
                out["".join(chars).replace(names[0], names[1])] = value
        elif "head_series_local" in key:
            out[key.replace("head_series_local", "head_series_cond")] = value
        else:
            out[key] = value
    return out


def match_keys(model_keys, loaded_keys):
    """model_serialization.py:26-47 with flownet=None: every model key takes the loaded key that is its LONGEST
    suffix (plain `str.endswith`, not aligned to dots), or None.  -> {model key: loaded key | None}."""
    loaded = set(loaded_keys)
    lens = sorted({len(k) for k in loaded}, reverse=True)
    out = {}
    for key in model_keys:
        hit = None
        for n in lens:
            if n <= len(key) and key[len(key) - n:] in loaded:
                hit = key[len(key) - n:]
                break
        out[key] = hit
    return out


def align_and_update_state_dicts(model_state_dict, loaded_state_dict):
    """model_serialization.py:12-73: in-place update of `model_state_dict`; returns the keys left untouched."""
    mapping = match_keys(sorted(model_state_dict.keys()), sorted(loaded_state_dict.keys()))
    missed = []
    for key, src in mapping.items():
        if src is None:
            missed.append(key)
            continue
        model_state_dict[key] = loaded_state_dict[src]
        logger.info("%s loaded from %s of shape %s", key, src, tuple(np.shape(loaded_state_dict[src])))
    if missed:
        print("{} keys are not updated: {}".format(len(missed), missed))
    return missed


def load_state_dict(model, loaded_state_dict, skip_modules=None):
    """model_serialization.py:140-156 for a DiffusionDet model: strip `module.`, rename heads, suffix-match, strict load."""
    model_state_dict = OrderedDict(model.state_dict())
    loaded = strip_prefix_if_present(loaded_state_dict, "module.")
    loaded = remap_diffusiondet_heads(list(model_state_dict.keys()), loaded, skip_modules)
    missed = align_and_update_state_dicts(model_state_dict, loaded)
    for name, value in model_state_dict.items():
        if isinstance(value, np.ndarray):
            model_state_dict[name] = torch.from_numpy(value)
    model.load_state_dict(model_state_dict)
    return missed


def load_pickled_weights(path, conv_body):
    """c2_model_loading.py:150-165 + :199-214 for `*-torchvision` bodies."""
    with open(path, "rb") as f:
        data = pickle.load(f, encoding="latin1")
    if "torchvision" not in conv_body:
        raise NotImplementedError(
            "MODEL.BACKBONE.CONV_BODY %r: Caffe2 blob renaming (c2_model_loading.py:12-129) maps to torchvision-style names "
            "that no DiffusionVID parameter carries; only the '*-torchvision' pickles (detectron2 names) are supported" % conv_body)
    weights = data["blobs"] if "blobs" in data else data
    return dict(model=weights["model"])


class DetectronCheckpointer:
    """`DetectronCheckpointer(cfg, model, save_dir=...).load(f)` as tools/test_net.py:102-104 calls it."""

    def __init__(self, cfg, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        self.cfg, self.model, self.save_dir = cfg, model, save_dir

    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
                return f.read().strip()
        except IOError:
            return ""

    def _load_file(self, f):
        if f.startswith(("catalog://", "http")):
            raise NotImplementedError("no network / model catalog here: pass a local .pth or .pkl path (MODEL.WEIGHT)")
        if f.endswith(".pkl"):
            return load_pickled_weights(f, self.cfg.MODEL.BACKBONE.CONV_BODY)
        loaded = torch.load(f, map_location=torch.device("cpu"), weights_only=False)
        if "model" not in loaded:
            loaded = dict(model=loaded)
        return loaded

    def load(self, f=None, use_latest=True, ignore=False, flownet=None, skip_modules=None):
        if self.has_checkpoint() and use_latest:
            f = self.get_checkpoint_file()
        if not f:
            logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        checkpoint = self._load_file(f)
        self.missed_keys = load_state_dict(self.model, checkpoint.pop("model"),
                                           skip_modules=skip_modules if "models/" in f else None)
        checkpoint.pop("optimizer", None)
        checkpoint.pop("scheduler", None)
        return checkpoint
