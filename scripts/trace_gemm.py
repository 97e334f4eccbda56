import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat
e = Engine("cuda:0")
if len(sys.argv) > 1:
    e.lib.query("zk_tune", 6, int(sys.argv[1]))
dll = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zero_amd/csrc/libzero_hip.so"))
for (M, N, K, tile, tb) in [(4096, 512, 512, 0, 0), (4096, 512, 512, 0, 1), (4096, 2048, 512, 0, 0), (4096, 512, 2048, 0, 0), (4096, 32768, 512, 0, 1)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn((N, K) if tb else (K, N), device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        e.gemm(Mat(A, M, K), Mat(B, *B.shape), Mat(C, M, N), M, N, K, 0, tb, impl=2 | (tile << 8) | (1 << 16))
    torch.cuda.synchronize()
    nk = K // 64
    buf = (ctypes.c_ulonglong * 4104)()
    assert dll.zk_debug_trace_read(buf, 4104) == 0
    ev = np.array(buf[4096:4101], dtype=np.int64)
    t = np.array(buf[:nk * 8], dtype=np.int64).reshape(nk, 8)[:, :5]
    print("  phases (cycles): setup+prologue issue %d | K loop %d (%d steps) | drain+acc->LDS %d | store epilogue %d | total %d; first-step wait %d" %
          (ev[1] - ev[0], ev[2] - ev[1], nk, ev[3] - ev[2], ev[4] - ev[3], ev[4] - ev[0], t[0, 1] - t[0, 0]))
    d = np.diff(t, axis=1)                      # wait-vm, barrier, dma-issue, compute
    nxt = t[1:, 0] - t[:-1, 4]
    print("M,N,K=%d,%d,%d tile=%d tb=%d  ticks(100MHz=10ns) per K step, median over steps 4..: waitvm %.1f  barrier %.1f  dma-issue %.1f  lds+mfma %.1f  loop-back %.1f  total %.1f" %
          (M, N, K, tile, tb, *np.median(d[2:], axis=0), np.median(nxt[2:]), np.median(t[3:, 0] - t[2:-1, 0])))
