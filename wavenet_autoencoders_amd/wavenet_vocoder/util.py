"""input-type predicates (reference: wavenet_vocoder/util.py:9-25)."""
_VALID = ("mulaw-quantize", "mulaw", "raw")


def _check(s):
    assert s in _VALID, s
    return s


def is_mulaw_quantize(s):
    return _check(s) == "mulaw-quantize"


def is_mulaw(s):
    return _check(s) == "mulaw"


def is_raw(s):
    return _check(s) == "raw"


def is_scalar_input(s):
    return is_raw(s) or is_mulaw(s)
