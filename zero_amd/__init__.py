"""zero_amd -- MI355X-native (gfx950) hot path of the bzhangGo/zero NMT toolkit.

Only the Transformer encoder-decoder training step and beam-search decode
(reference: models/transformer*.py, func.py, search.py, utils/parallel.py,
utils/cycle.py) live here, behind the reference's own registration surface
(models/model.py) and HParams / run.py contract.  Python host code calls
hand-written HIP kernels in ``zero_amd/csrc`` through the C-ABI declared in
``include/zero_hip.h`` (ctypes); torch tensors are storage only.
"""

__version__ = "0.1.0"
