"""Shared test helpers: golden fixtures + seeded weights (no reference code, no /root/reference)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ARGS = ("x", "pre_img", "pre_hm", "repro_hm", "pre_hm_cls", "repro_hm_cls")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def model_keys(name="model_keys.json"):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)["keys"]


_SD_CACHE = {}


def golden_state_dict(seed=0, keys="model_keys.json", peaked=False):
    """The reference model's state_dict entries (560 for dlapawdl3new_34), rebuilt from (seed, name, shape) alone."""
    from sgtapose_amd.synthetic import fill_state_dict_
    if (seed, keys, peaked) not in _SD_CACHE:
        sd = {}
        for k, shape in model_keys(keys):
            dt = torch.long if k.endswith("num_batches_tracked") else torch.float32
            sd[k] = torch.zeros(shape, dtype=dt)
        _SD_CACHE[(seed, keys, peaked)] = fill_state_dict_(sd, seed, peaked=peaked)
    return _SD_CACHE[(seed, keys, peaked)]
