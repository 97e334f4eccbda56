"""In-kernel timeline of the 256x256 tile (gemm256_acc, `make TRACE=1` library: scripts/build_trace.sh): where a K step of
the weight-gradient form (ta = 1, tb = 0, K = 4096) goes, for wave 0 (first half of the workgroup) and wave 4 (second
half: the partner of wave 0 on its SIMD), under the LDS-DMA issue schedules of tuning key 14.
usage (GPU box): ZERO_HIP_LIB=$PWD/zero_amd/csrc/libzero_hip_trace.so python scripts/trace_gemm256.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat  # noqa: E402

e = Engine("cuda:0")
dll = ctypes.CDLL(os.environ["ZERO_HIP_LIB"])
tune = e.lib.raw("zk_tune")
M = N = K = 4096
for (ta, tb, name) in ((1, 0, "weight-gradient form (X^T dY)"), (0, 1, "logits form (A B^T)")):
    A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
    B = torch.randn((N, K) if tb else (K, N), device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.float32)
    probs = [(Mat(A, *A.shape), Mat(B, *B.shape), Mat(C, M, N), M, N, K, None)]
    for sched in (0, 130):
        tune(14, sched)
        for _ in range(3):
            e.gemm_grouped(probs, ta, tb, tile=(256, 256, 0))
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 1024)()
        assert dll.zk_debug_trace_read(buf, 1024) == 0
        t = np.array(buf[:1024], dtype=np.int64).reshape(64, 2, 8)
        print("%s, schedule %d: shader-clock cycles per K step (median over steps 8..56)" % (name, sched))
        for w, wn in ((0, "wave 0"), (1, "wave 4")):
            x = t[8:56, w]
            d = np.diff(x, axis=1)
            step = np.median(t[9:57, w, 0] - t[8:56, w, 0])
            print("   %s: wait DMA %5.1f | barrier %5.1f | DMA issue %5.1f | slice0 %5.1f | slice1 %5.1f | slice2 %5.1f | slice3 %5.1f | "
                  "loop back %5.1f | step %5.1f cycles" %
                  ((wn,) + tuple(np.median(d, axis=0)) + (np.median(t[9:57, w, 0] - x[:, 7]), step)))
        # offset between the two waves' slice-0 starts
        print("   wave 4 passes the barrier %+.1f cycles after wave 0 (median)" % np.median(t[8:56, 1, 2] - t[8:56, 0, 2]))
tune(14, 0)
