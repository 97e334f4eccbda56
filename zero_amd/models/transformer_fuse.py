# coding: utf-8
"""``transformer_fuse`` -- registered under the reference's name (models/transformer_fuse.py:307).

The layer schedule lives in zero_amd/models/_core.py (training) and
zero_amd/models/_decode.py (incremental decoding); this module only binds the
(train_fn, score_fn, infer_fn) triple to the registry, like the reference module does at
import time.
"""

from zero_amd.models import model
from zero_amd.models._factory import build

train_fn, score_fn, infer_fn = build("transformer_fuse")

# register the model, with a unique name
model.model_register("transformer_fuse", train_fn, score_fn, infer_fn)
