"""Config surface on CPU: HParams (defaults, json presets incl. tolerant parsing, typed overrides), lr schedules."""
import json
import os

import pytest

from wavenet_autoencoders_amd import lrschedule
from wavenet_autoencoders_amd.hparams import HParams, _DEFAULTS, hparams, hparams_debug_string

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "misc.json")


def test_defaults_match_reference_registry():
    with open(GOLDEN) as fh:
        m = json.load(fh)
    assert hparams.values() == m["hparams_defaults"]
    assert "layers: 24" in hparams_debug_string()


def test_parse_json_and_overrides_match_reference():
    with open(GOLDEN) as fh:
        m = json.load(fh)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = HParams(**_DEFAULTS)
    with open(os.path.join(root, "hps", "vqwae.json")) as fh:
        h.parse_json(fh.read(), strict=True)            # loads against the reference registry too
    h.parse("layers=24,batch_size=8,ema_decay=0.99")
    assert h.values() == m["vqwae_parsed_with_overrides"]
    assert h.layers == 24 and isinstance(h.ema_decay, float) and h.upsample_params == {"upsample_scales": [4, 4, 8, 5]}


def test_tolerant_json_and_strict_keyerror():
    h = HParams(**_DEFAULTS)
    txt = '{"layers": 20, "beta": 0.25, "frame_rate": 100,}'      # unknown keys + trailing comma
    with pytest.raises(KeyError):
        HParams(**_DEFAULTS).parse_json(txt, strict=True)
    h.parse_json(txt)
    assert h.layers == 20 and h.beta == 0.25 and h.frame_rate == 100
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h2 = HParams(**_DEFAULTS)
    with open(os.path.join(root, "hps", "inae_hp.json")) as fh:
        h2.parse_json(fh.read())                                   # KeyError('beta') in the reference; loads here
    assert h2.gate_channels == 368 and h2.upsample_params["upsample_scales"] == [4, 4, 4, 5]
    with pytest.raises(KeyError):
        h.parse("no_such_key=1")
    with pytest.raises(ValueError):
        h.parse("layers=2.5")


def test_lr_schedules():
    with open(GOLDEN) as fh:
        m = json.load(fh)
    for s, lr, no, cy in zip(m["lr_steps"], m["step_lr"], m["noam"], m["cyclic"]):
        assert lrschedule.step_learning_rate_decay(4e-4, s, anneal_rate=0.5, anneal_interval=400000) == lr
        assert abs(lrschedule.noam_learning_rate_decay(1e-3, s) - no) < 1e-12
        assert abs(lrschedule.cyclic_cosine_annealing(1e-3, s, 1000, 5) - cy) < 1e-12
