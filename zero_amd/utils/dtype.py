# coding: utf-8
"""Process-wide numeric settings of the hot path (reference: utils/dtype.py:10-43, set from the
HParams ``default_dtype`` / ``dtype_epsilon`` / ``dtype_inf`` at run.py:397-399).

``epsilon`` is the LayerNorm variance floor and ``inf`` the magnitude of the additive attention
mask -- finite on purpose: a fully masked row softmaxes to uniform instead of NaN.  Both are
parity-critical and are passed to the kernels per call.  ``floatx`` names the reference's
storage/compute type (float32 default, float16 optional); the HIP path always computes in bf16
with fp32 accumulation over fp32 master weights (the storage contract of dtype.py:55-69), so the
value is kept for configuration compatibility only.
"""

_KNOWN_FLOATX = ("float16", "float32", "float64", "bfloat16")
_settings = {"floatx": "float32", "epsilon": 1e-8, "inf": 1e8}


def floatx():
    return _settings["floatx"]


def epsilon():
    return _settings["epsilon"]


def inf():
    return _settings["inf"]


def set_floatx(name):
    if name not in _KNOWN_FLOATX:
        raise ValueError('Unknown floatx type: ' + str(name))
    _settings["floatx"] = str(name)


def set_epsilon(value):
    _settings["epsilon"] = float(value)


def set_inf(value):
    _settings["inf"] = float(value)
