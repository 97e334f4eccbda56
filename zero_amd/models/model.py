# coding: utf-8
"""Model registry -- the drop-in boundary of the hot path.

Same surface as the reference's models/model.py:11-41: modules self-register
``(train_fn, score_fn, infer_fn)`` under a lower-cased name at import time,
``get_model`` raises on unknown names, duplicate registration raises.
"""

import logging
from collections import namedtuple

# global models defined in Zero
_total_models = {}


class ModelWrapper(namedtuple("ModelTupleWrapper",
                              ("train_fn", "score_fn", "infer_fn"))):
    pass


def model_register(model_name, train_fn, score_fn, infer_fn):
    model_name = model_name.lower()

    if model_name in _total_models:
        raise Exception("Conflict Model Name: {}".format(model_name))

    logging.getLogger("zero_amd").info("Registering model: %s", model_name)

    _total_models[model_name] = ModelWrapper(
        train_fn=train_fn,
        score_fn=score_fn,
        infer_fn=infer_fn,
    )


def get_model(model_name):
    model_name = model_name.lower()

    if model_name in _total_models:
        return _total_models[model_name]

    raise Exception("No supported model {}".format(model_name))
