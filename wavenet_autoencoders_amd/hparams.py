"""Hyper-parameter registry with the reference's names and defaults (hparams.py:8-135) and a small HParams type with
the part of the TF-1.12 ``HParams`` API the reference uses (tfcompat/hparam.py: ``parse_json`` :594-607, ``parse``
:523, ``values`` :609-616, typed ``set_hparam`` :487-511).

Difference on purpose: only ``hps/vqwae.json`` and ``hps/hp.json`` load against the reference's
registry (unknown keys raise KeyError there, and ``wv_vqvae_hp.json`` has a trailing comma).  ``parse_json`` here is
tolerant by default -- unknown keys are added, trailing commas are accepted -- so every shipped preset loads;
``strict=True`` restores the reference's KeyError.
"""
import json
import re

_DEFAULTS = dict(
    name="wavenet_vocoder",
    input_type="raw", quantize_channels=65536,
    preprocess="", postprocess="", global_gain_scale=1.0,
    sample_rate=22050, silence_threshold=2, num_mels=80, n_mfcc=13, fmin=125, fmax=7600, fft_size=1024, hop_size=256,
    frame_shift_ms=None, win_length=1024, win_length_ms=-1.0, window="hann", min_level_db=-100, highpass_cutoff=70.0,
    output_distribution="Logistic", log_scale_min=-16.0,
    out_channels=30, layers=24, stacks=4, residual_channels=128, gate_channels=256, skip_out_channels=128, dropout=0.0,
    kernel_size=3,
    cin_channels=80, cin_pad=2, upsample_conditional_features=True, upsample_net="ConvInUpsampleNetwork",
    upsample_params={"upsample_scales": [4, 4, 4, 4]},
    gin_channels=-1, n_speakers=7,
    pin_memory=True, num_workers=2,
    batch_size=8, dev_batch_size=1, optimizer="Adam", optimizer_params={"lr": 1e-3, "eps": 1e-8, "weight_decay": 0.0},
    lr_schedule="step_learning_rate_decay", lr_schedule_kwargs={"anneal_rate": 0.5, "anneal_interval": 200000},
    max_train_steps=1000000, nepochs=2000, clip_thresh=-1, max_time_sec=None, max_time_steps=10240,
    exponential_moving_average=True, ema_decay=0.9999,
    checkpoint_interval=100000, train_eval_interval=100000, test_eval_epoch_interval=50, save_optimizer_state=True,
    dim_in=39, encoder_hid=384, language="english", K=256, ema=False,
)


class HParams(object):
    def __init__(self, **kwargs):
        object.__setattr__(self, "_values", {})
        object.__setattr__(self, "_types", {})
        for k, v in kwargs.items():
            self.add_hparam(k, v)

    def add_hparam(self, name, value):
        if name in self._values:
            raise ValueError("Hyperparameter name is reserved: %s" % name)
        self._values[name] = value
        self._types[name] = type(value) if value is not None else None

    def __getattr__(self, name):
        try:
            return object.__getattribute__(self, "_values")[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self.set_hparam(name, value)

    def __contains__(self, name):
        return name in self._values

    def set_hparam(self, name, value):
        """typed override; unknown name -> KeyError like the reference (hparam.py:500)."""
        tp = self._types[name]
        if tp is None or value is None or isinstance(value, (dict, list)):
            self._values[name] = value
        elif tp is bool:
            if isinstance(value, str):
                if value.lower() not in ("true", "false", "1", "0"):
                    raise ValueError("Could not parse %s=%s as bool" % (name, value))
                value = value.lower() in ("true", "1")
            self._values[name] = bool(value)
        elif tp is int and isinstance(value, float) and value != int(value):
            raise ValueError("Must pass an int for %s, got %r" % (name, value))
        else:
            self._values[name] = tp(value)

    def parse_json(self, text, strict=False):
        text = re.sub(r",(\s*[}\]])", r"\1", text)            # tolerate trailing commas (hps/wv_vqvae_hp.json:74-75)
        for k, v in json.loads(text).items():
            if k not in self._values:
                if strict:
                    raise KeyError(k)
                self.add_hparam(k, v)
            else:
                self.set_hparam(k, v)
        return self

    def parse(self, spec):
        """"k=v,k2=v2" overrides (vqwae_train.py:1092)."""
        for item in filter(None, (s.strip() for s in (spec or "").split(","))):
            if "=" not in item:
                raise ValueError("Could not parse hparam %r" % item)
            k, v = item.split("=", 1)
            k = k.strip()
            if k not in self._values:
                raise KeyError(k)
            tp = self._types[k]
            v = v.strip()
            if tp in (dict, list):
                v = json.loads(v)
            elif tp is None:
                try:
                    v = json.loads(v)
                except ValueError:
                    pass
            elif tp is float or tp is int:
                v = float(v) if tp is float else (int(v) if re.fullmatch(r"[-+]?\d+", v) else float(v))
            self.set_hparam(k, v)
        return self

    def values(self):
        return dict(self._values)

    def to_json(self, **kw):
        return json.dumps(self._values, **kw)


hparams = HParams(**_DEFAULTS)


def hparams_debug_string():
    v = hparams.values()
    return "Hyperparameters:\n" + "\n".join("  %s: %s" % (k, v[k]) for k in sorted(v))


def adam_settings(hp):
    """The reference builds ``getattr(torch.optim, hparams.optimizer)(model.parameters(), **hparams.optimizer_params)``
    (vqwae_train.py:1119-1120).  The engine's update is ONE fused kernel that is torch.optim.Adam's arithmetic (plain L2 weight decay,
    no amsgrad; csrc/loss.hip: clip_adam_ema_kernel), so: any other optimizer, amsgrad / maximize, and any key torch.optim.Adam would
    reject RAISE here instead of training silently with something else; lr, betas, eps and weight_decay are forwarded.
    Returns dict(lr, betas, eps, weight_decay)."""
    name = hp.optimizer
    if name != "Adam":
        raise NotImplementedError(f"hparams.optimizer={name!r}: the engine's fused update implements torch.optim.Adam only "
                                  "(the reference would build torch.optim.%s)" % name)
    op = dict(hp.optimizer_params or {})
    # arguments of torch.optim.Adam that only choose an implementation of the same arithmetic
    for k in ("foreach", "capturable", "differentiable", "fused"):
        op.pop(k, None)
    for k in ("amsgrad", "maximize"):
        if op.pop(k, False):
            raise NotImplementedError(f"optimizer_params.{k}=True is not implemented by the fused Adam update")
    out = dict(lr=float(op.pop("lr", 1e-3)), betas=tuple(float(b) for b in op.pop("betas", (0.9, 0.999))),
               eps=float(op.pop("eps", 1e-8)), weight_decay=float(op.pop("weight_decay", 0.0)))
    if op:
        raise TypeError("Adam.__init__() got an unexpected keyword argument %r" % sorted(op)[0])     # what torch.optim.Adam raises
    # torch.optim.Adam's own range checks, same messages
    if not 0.0 <= out["lr"]:
        raise ValueError(f"Invalid learning rate: {out['lr']}")
    if not 0.0 <= out["eps"]:
        raise ValueError(f"Invalid epsilon value: {out['eps']}")
    if len(out["betas"]) != 2:
        raise ValueError(f"betas must be a pair, got {out['betas']}")
    for i, b in enumerate(out["betas"]):
        if not 0.0 <= b < 1.0:
            raise ValueError(f"Invalid beta parameter at index {i}: {b}")
    if not 0.0 <= out["weight_decay"]:
        raise ValueError(f"Invalid weight_decay value: {out['weight_decay']}")
    return out
