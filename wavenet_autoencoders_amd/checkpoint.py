"""Checkpoints in the reference's layout, both directions (vqwae_train.py:878-999).

A reference checkpoint is ``{"state_dict", "optimizer", "global_step", "global_epoch", "global_test_step"}`` with
``state_dict`` = the VQVAE's named tensors (weight_g / weight_v form) and ``optimizer`` = ``torch.optim.Adam.state_dict()``
(:881-889), next to an ``_ema`` twin holding the shadow weights (:891-910).  The engine keeps parameters and both Adam moments
in flat arenas whose order IS the reference model's ``named_parameters()`` order (packing.param_specs), so parameter index i
of the optimizer state is the i-th key of ``eng.lay.offsets``.
"""
import os
import shutil
from typing import Dict, Optional

import torch


def adam_state_dict(eng, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, amsgrad: bool = False) -> dict:
    """The engine's optimizer state as ``torch.optim.Adam.state_dict()`` would return it (vqwae_train.py:881):
    ``state[i] = {step, exp_avg, exp_avg_sq}`` per parameter in registration order, one param group.  Before the first step
    the state is empty, as torch's is."""
    lay = eng.lay
    keys = list(lay.offsets)
    state = {}
    if getattr(eng, "opt_step", 0) > 0:
        m, v = eng.exp_avg.detach().cpu(), eng.exp_avg_sq.detach().cpu()
        for i, k in enumerate(keys):
            off, n = lay.off(k), lay.numel(k)
            state[i] = {"step": torch.tensor(float(eng.opt_step)),
                        "exp_avg": m[off:off + n].view(lay.shapes[k]).clone(),
                        "exp_avg_sq": v[off:off + n].view(lay.shapes[k]).clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": amsgrad, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(keys)))}
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(eng, opt: Optional[dict]) -> dict:
    """Restore Adam moments and the step count from a torch-format state dict (a reference checkpoint's ``optimizer``,
    vqwae_train.py:967-970) or from this repository's round-1 ``layout == "flat-arena"`` form.  Raises on anything it cannot
    map -- a state that is silently dropped would restart the moments at zero.  -> the first param group (lr, betas, ...)."""
    if not hasattr(eng, "exp_avg"):
        eng.init_optimizer()
    if opt is None:
        return {}
    lay = eng.lay
    if opt.get("layout") == "flat-arena":
        eng.exp_avg.copy_(opt["exp_avg"])
        eng.exp_avg_sq.copy_(opt["exp_avg_sq"])
        eng.opt_step = int(opt["step"])
        return {}
    if "state" not in opt or "param_groups" not in opt:
        raise ValueError("optimizer state is neither a torch.optim state_dict nor a flat-arena dict: keys %s" % sorted(opt))
    keys = list(lay.offsets)
    params = [p for gr in opt["param_groups"] for p in gr["params"]]
    if len(params) != len(keys):
        raise ValueError(f"optimizer state covers {len(params)} parameters, the model has {len(keys)}")
    if any(gr.get("amsgrad") for gr in opt["param_groups"]):
        raise NotImplementedError("amsgrad optimizer state (hparams amsgrad=True) is not supported by the fused update")
    m = torch.zeros(lay.total, dtype=torch.float32)
    v = torch.zeros(lay.total, dtype=torch.float32)
    steps = set()
    state = opt["state"]
    for pos, pid in enumerate(params):
        st = state.get(pid, state.get(str(pid)))
        if st is None:
            continue                                   # a parameter that never received a gradient
        k = keys[pos]
        off, n = lay.off(k), lay.numel(k)
        for name, dst in (("exp_avg", m), ("exp_avg_sq", v)):
            t = st[name]
            if t.numel() != n:
                raise ValueError(f"optimizer state of parameter {pos} ({k}): {tuple(t.shape)} does not match {lay.shapes[k]}")
            dst[off:off + n] = t.detach().to(torch.float32).reshape(-1).cpu()
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): the fused update keeps one step for the arena")
    eng.exp_avg.copy_(m.to(eng.exp_avg.device))
    eng.exp_avg_sq.copy_(v.to(eng.exp_avg.device))
    eng.opt_step = steps.pop() if steps else 0
    return dict(opt["param_groups"][0])


def save_checkpoint(eng, step: int, epoch: int, checkpoint_dir: str, hp, rank: int = 0, test_step: int = 0, lr: Optional[float] = None):
    """vqwae_train.py:878-910: checkpoint_step{:09d}.pth + checkpoint_latest.pth, and the _ema twin with the shadow weights."""
    if rank != 0:
        return None
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = os.path.join(checkpoint_dir, "checkpoint_step{:09d}.pth".format(step))
    sd = {k: v.cpu() for k, v in eng.state_dict().items()}
    opt = None
    if hp.save_optimizer_state and hasattr(eng, "exp_avg"):
        from .hparams import adam_settings
        op = adam_settings(hp)
        opt = adam_state_dict(eng, lr if lr is not None else op["lr"], betas=op["betas"], eps=op["eps"], weight_decay=op["weight_decay"])
    torch.save({"state_dict": sd, "optimizer": opt, "global_step": step, "global_epoch": epoch, "global_test_step": test_step}, path)
    shutil.copyfile(path, os.path.join(checkpoint_dir, "checkpoint_latest.pth"))
    if getattr(eng, "shadow", None) is not None:
        ema_sd = {}
        for k in eng.lay.offsets:
            off, n = eng.lay.off(k), eng.lay.numel(k)
            ema_sd[k] = eng.shadow[off:off + n].view(eng.lay.shapes[k]).cpu()
        epath = os.path.join(checkpoint_dir, "checkpoint_step{:09d}_ema.pth".format(step))
        torch.save({"state_dict": ema_sd, "optimizer": opt, "global_step": step, "global_epoch": epoch,
                    "global_test_step": test_step}, epath)
        shutil.copyfile(epath, os.path.join(checkpoint_dir, "checkpoint_latest_ema.pth"))
    print("Saved checkpoint:", path)
    return path


def load_checkpoint(path: str, eng, reset_optimizer: bool, ema: bool = True):
    """vqwae_train.py:959-976 -> (global_step, global_epoch, global_test_step).  The EMA shadow restarts from the loaded
    weights (the reference registers it after loading, :822-826), unless exponential_moving_average is off."""
    ck = torch.load(path, map_location="cpu")
    eng.load_state_dict(ck["state_dict"])
    eng.init_optimizer(ema=ema)
    if not reset_optimizer:
        opt = ck.get("optimizer")
        if opt is not None:
            print("Load optimizer state from {}".format(path))
            load_adam_state_dict(eng, opt)
    return int(ck.get("global_step", 0)), int(ck.get("global_epoch", 0)), int(ck.get("global_test_step", 0))


def restore_parts(path: str, eng) -> Dict[str, str]:
    """load the tensors whose name and shape match, warn about the others only (vqwae_train.py:980-999)"""
    sd = torch.load(path, map_location="cpu")["state_dict"]
    cur = eng.state_dict()
    skipped = {}
    for k, v in sd.items():
        if k in cur and tuple(cur[k].shape) == tuple(v.shape):
            cur[k] = v
        else:
            skipped[k] = "not in the model" if k not in cur else f"shape {tuple(v.shape)} != {tuple(cur[k].shape)}"
            print("warn: skip", k, "--", skipped[k])
    eng.load_state_dict({k: v.cpu() for k, v in cur.items()})
    return skipped
