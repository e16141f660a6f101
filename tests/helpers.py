"""Shared helpers for the parity tests (tests/ only; may import the oracle)."""
import json
import os

import numpy as np
import torch

from oracle import wae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_model(name):
    """-> (cfg, sd, inputs dict, golden dict) for model_<name>.npz"""
    z = load_npz("model_" + name)
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(cfg, int(z["salt"]))
    c = torch.from_numpy(z["c"])
    g = torch.from_numpy(z["g"])
    if cfg.get("scalar_input"):
        x = torch.from_numpy(z["x"])
        xin = x
    else:
        x = torch.from_numpy(z["x"]).long()
        xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
    return cfg, sd, dict(c=c, x=x, xin=xin, g=g), z, ocfg


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
