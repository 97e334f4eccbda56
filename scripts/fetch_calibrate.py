"""Known-byte-count reads for calibrating rocprofv3's FETCH_SIZE per access pattern (zk_probe_read, zk_probe.hip): every
launch reads each byte of a fresh 1 GiB buffer once (larger than the 256 MB Infinity Cache and never touched before, so the
whole of it crosses the fabric).  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (scripts/fetch_calibrate.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd import hip  # noqa: E402

lib = hip.lib()
BYTES = 1 << 30
sink = torch.zeros(4, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for pattern in range(5):
        buf = torch.empty(BYTES, dtype=torch.uint8, device="cuda")
        buf.fill_(pattern + 1)                        # written by another kernel: nothing of it is in this launch's L2 ... 
        flush = torch.empty(BYTES, dtype=torch.uint8, device="cuda")
        flush.fill_(7)                                # ... and 1 GiB of other lines pushed through L2 / Infinity Cache after it
        torch.cuda.synchronize()
        lib.call("zk_probe_read", buf.data_ptr(), BYTES, pattern, sink.data_ptr(), s)
        torch.cuda.synchronize()
        del buf, flush
print("bytes per launch:", BYTES)
