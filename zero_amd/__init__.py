"""zero_amd -- MI355X-native (gfx950) hot path of the bzhangGo/zero NMT toolkit.

Only the Transformer encoder-decoder training step and beam-search decode
(reference: models/transformer*.py, func.py, search.py, utils/parallel.py,
utils/cycle.py) live here, behind the reference's own registration surface
(models/model.py) and HParams / run.py contract.  Python host code calls
hand-written HIP kernels in ``zero_amd/csrc`` through the C-ABI declared in
``include/zero_hip.h`` (ctypes); torch tensors are storage only.
"""

import os as _os

# HIP maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4), round-robin.  With several
# decode batches in flight (evalu.decode_many: one stream per execution lane, besides the training / collective streams)
# two lanes that share a hardware queue run one after the other again: 4 lanes decode 2740 sentences/s on 4 queues,
# 3340 on 8 (profiles/r03_bench_decode_models.jsonl).  The runtime reads the variable when it initialises (first HIP call
# of the process), so it is set here, on import, unless the user chose a value.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
