# coding: utf-8
"""Model registry -- the drop-in boundary of the hot path (reference: models/model.py:11-41).

Contract kept from the reference: model modules register a ``(train_fn, score_fn, infer_fn)``
triple under a case-insensitive name when they are imported; looking up an unknown name raises
``Exception("No supported model <name>")``; registering a name twice raises
``Exception("Conflict Model Name: <name>")``; the value handed back exposes the three functions as
attributes ``.train_fn`` / ``.score_fn`` / ``.infer_fn`` (and unpacks like a 3-tuple).
"""

import logging
from typing import Callable, Dict, NamedTuple

log = logging.getLogger("zero_amd")


class ModelWrapper(NamedTuple):
    train_fn: Callable
    score_fn: Callable
    infer_fn: Callable


class _Registry(object):
    """Name -> ModelWrapper, keys normalised to lower case."""

    def __init__(self):
        self._models: Dict[str, ModelWrapper] = {}

    @staticmethod
    def _key(name):
        return str(name).lower()

    def add(self, name, triple):
        key = self._key(name)
        if key in self._models:
            raise Exception("Conflict Model Name: {}".format(key))
        self._models[key] = triple
        log.info("Registering model: %s", key)

    def find(self, name):
        try:
            return self._models[self._key(name)]
        except KeyError:
            raise Exception("No supported model {}".format(self._key(name))) from None

    def names(self):
        return sorted(self._models)


_REGISTRY = _Registry()


def model_register(model_name, train_fn, score_fn, infer_fn):
    _REGISTRY.add(model_name, ModelWrapper(train_fn, score_fn, infer_fn))


def get_model(model_name):
    return _REGISTRY.find(model_name)


def registered_models():
    return _REGISTRY.names()
