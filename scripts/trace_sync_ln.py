"""In-kernel timeline of zk_gemm_add_ln (make TRACE=1 library: ZERO_HIP_LIB=.../libzero_hip_trace.so): workgroup 301's
s_memtime stamps at the ring prologue, the end of the K loop, the tile in LDS, the slot published, the peers seen, the
statistics combined and the end of the kernel.  Ticks are 10 ns."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat
from zero_amd import hip
e = Engine("cuda:0")
dll = ctypes.CDLL(os.environ.get("ZERO_HIP_LIB") or hip.LIB_PATH)
for (M, N, K) in [(4096, 512, 512), (4096, 512, 2048)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(K, N, device="cuda") * 0.05).bfloat16()
    R = torch.randn(M, N, device="cuda").bfloat16()
    Y, S = torch.empty_like(R), torch.empty_like(R)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    gam, bet, bias = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    rows = []
    for rep in range(6):
        e.ln_epoch_bump()
        e.gemm_add_ln(Mat(A, M, K), Mat(B, K, N), M, N, K, bias, Mat(R, M, N), gam, bet, Mat(Y, M, N), Mat(S, M, N), mean, rstd, 0.1, 3)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 4104)()
        assert dll.zk_debug_trace_read(buf, 4104) == 0
        ev = np.array(buf[4096:4104], dtype=np.int64)
        rows.append([ev[1] - ev[0], ev[2] - ev[1], ev[3] - ev[2], ev[5] - ev[3], ev[6] - ev[5], ev[7] - ev[6], ev[4] - ev[7], ev[4] - ev[0]])
    r = np.median(np.array(rows[2:]), axis=0) * 10
    print("M,N,K=%d,%d,%d  ns: prologue %d | K loop %d | acc->LDS %d | s + statistics + slot out %d | peers seen (poll + barrier) %d | "
          "slots fetched + combined %d | normalise + stores to end %d | total %d" % ((M, N, K) + tuple(int(x) for x in r)))
