# coding: utf-8
"""Plain-Python stand-in for ``tf.contrib.training.HParams``.

The reference keeps its whole configuration in one HParams object
(run.py:24-239) and relies on: attribute access, ``parse("k=v,k2=[1,2]")``
(run.py:367,376), ``override_from_dict`` (run.py:369-375), ``to_json`` /
``parse_json`` for ``param.json`` (run.py:250-272), ``values()``
(utils/util.py:108) and ``add_hparam`` (run.py:295).  Value types are fixed
by the default; ``parse`` casts to that type exactly like TF1's HParams did
(bool accepts true/false/1/0, ints refuse floats, lists use ``[a,b]``).
"""

import json
import re

_PARAM_RE = re.compile(r"""
  (?P<name>[a-zA-Z][\w\.]*)      # variable name
  (\[\s*(?P<index>\d+)\s*\])?    # (optional) index
  \s*=\s*
  ((?P<val>[^,\[]*)              # single value
   |
   \[(?P<vals>[^\]]*)\])         # list of values
  ($|,\s*)""", re.VERBOSE)


def _parse_bool(value):
    if isinstance(value, bool):
        return value
    v = str(value).strip().lower()
    if v in ("true", "1"):
        return True
    if v in ("false", "0"):
        return False
    raise ValueError("Could not parse {!r} as bool".format(value))


def _cast(name, value, like):
    """Cast ``value`` (str or python object) to the type of ``like``."""
    if isinstance(like, bool):
        return _parse_bool(value)
    if isinstance(like, int):
        if isinstance(value, float) and value != int(value):
            raise ValueError("Could not cast hparam '%s' %r to int" % (name, value))
        if isinstance(value, str):
            try:
                return int(value)
            except ValueError:
                f = float(value)
                if f != int(f):
                    raise ValueError(
                        "Could not cast hparam '%s' %r to int" % (name, value))
                return int(f)
        return int(value)
    if isinstance(like, float):
        return float(value)
    if isinstance(like, str):
        return str(value)
    return value


class HParams(object):
    """Attribute bag with typed ``parse`` -- see module docstring."""

    def __init__(self, **kwargs):
        object.__setattr__(self, "_hparam_types", {})
        for name, value in kwargs.items():
            self.add_hparam(name, value)

    # -- construction -----------------------------------------------------
    def add_hparam(self, name, value):
        if getattr(self, name, None) is not None and name in self._hparam_types:
            raise ValueError("Hyperparameter name is reserved: %s" % name)
        if isinstance(value, (list, tuple)):
            if not value:
                raise ValueError(
                    "Multi-valued hyperparameters cannot be empty: %s" % name)
            self._hparam_types[name] = (type(value[0]), True)
            value = list(value)
        else:
            self._hparam_types[name] = (type(value), False)
        object.__setattr__(self, name, value)

    def set_hparam(self, name, value):
        if name not in self._hparam_types:
            raise ValueError("Unknown hyperparameter: %s" % name)
        _, is_list = self._hparam_types[name]
        cur = getattr(self, name)
        if is_list:
            if not isinstance(value, (list, tuple)):
                raise ValueError(
                    "Must pass a list for multi-valued parameter: %s." % name)
            like = cur[0] if cur else ""
            object.__setattr__(self, name, [_cast(name, v, like) for v in value])
        else:
            if isinstance(value, (list, tuple)):
                raise ValueError(
                    "Must not pass a list for single-valued parameter: %s" % name)
            object.__setattr__(self, name, _cast(name, value, cur))

    def del_hparam(self, name):
        if name in self._hparam_types:
            delattr(self, name)
            del self._hparam_types[name]

    # -- parsing ----------------------------------------------------------
    def parse(self, values):
        """Parse ``"k=v,k2=[a,b]"`` and override matching hparams."""
        if not values:
            return self
        pos = 0
        s = values.strip()
        while pos < len(s):
            m = _PARAM_RE.match(s, pos)
            if not m:
                raise ValueError("Malformed hyperparameter value: %s" % s[pos:])
            pos = m.end()
            name = m.group("name")
            if name not in self._hparam_types:
                raise ValueError("Unknown hyperparameter type for %s" % name)
            if m.group("vals") is not None:
                vals = [v.strip() for v in m.group("vals").split(",") if v.strip() != ""]
                if m.group("index") is not None:
                    raise ValueError("Malformed hyperparameter value: %s" % s)
                self.set_hparam(name, vals)
            else:
                val = m.group("val").strip()
                if m.group("index") is not None:
                    idx = int(m.group("index"))
                    cur = list(getattr(self, name))
                    cur[idx] = _cast(name, val, cur[idx])
                    object.__setattr__(self, name, cur)
                else:
                    _, is_list = self._hparam_types[name]
                    if is_list:
                        self.set_hparam(name, [val])
                    else:
                        self.set_hparam(name, val)
        return self

    def override_from_dict(self, values_dict):
        for name, value in values_dict.items():
            self.set_hparam(name, value)
        return self

    def parse_json(self, values_json):
        return self.override_from_dict(json.loads(values_json))

    def to_json(self, indent=None, separators=None, sort_keys=False):
        def _jsonable(v):
            try:
                json.dumps(v)
                return True
            except TypeError:
                return False
        return json.dumps({k: v for k, v in self.values().items() if _jsonable(v)},
                          indent=indent, separators=separators,
                          sort_keys=sort_keys)

    def values(self):
        return {n: getattr(self, n) for n in self._hparam_types.keys()}

    def get(self, key, default=None):
        if key in self._hparam_types:
            return getattr(self, key)
        return default

    def __contains__(self, key):
        return key in self._hparam_types

    def __copy__(self):
        new = HParams.__new__(HParams)
        object.__setattr__(new, "_hparam_types", dict(self._hparam_types))
        for k, v in self.__dict__.items():
            if k != "_hparam_types":
                object.__setattr__(new, k, v)
        return new

    def __repr__(self):
        return "HParams(%s)" % ", ".join(
            "%s=%r" % kv for kv in sorted(self.values().items()))
