# coding: utf-8
"""Global numeric constants of the hot path.

Mirror of the reference's utils/dtype.py:10-43 (floatx / epsilon=1e-8 /
inf=1e8, each overridable through run.py:397-399).  ``floatx`` names the
*storage/compute* contract of the reference (float32 default, float16
optional); the HIP path always computes in bf16 with fp32 accumulation and
fp32 master weights (reference dtype.py:55-69 contract), so ``floatx`` is kept
for config compatibility only.  ``epsilon`` (LayerNorm) and ``inf`` (attention
mask magnitude, finite on purpose) are parity-critical.
"""

_FLOATX = 'float32'
_EPSILON = 1e-8
_INF = 1e8


def epsilon():
    return _EPSILON


def set_epsilon(e):
    global _EPSILON
    _EPSILON = float(e)


def inf():
    return _INF


def set_inf(e):
    global _INF
    _INF = float(e)


def floatx():
    return _FLOATX


def set_floatx(floatx):
    global _FLOATX
    if floatx not in {'float16', 'float32', 'float64', 'bfloat16'}:
        raise ValueError('Unknown floatx type: ' + str(floatx))
    _FLOATX = str(floatx)
