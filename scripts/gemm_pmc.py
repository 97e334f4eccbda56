"""A handful of single GEMM launches for hardware-counter collection (rocprofv3 --pmc)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat
e = Engine("cuda:0")
SH = [(4096, 512, 512, 0, 0, 0), (4096, 512, 512, 0, 1, 0), (4096, 512, 2048, 0, 0, 0), (4096, 2048, 512, 0, 0, 0),
      (4096, 32000, 512, 0, 1, 1), (512, 512, 4096, 1, 0, 1), (32000, 512, 4096, 1, 0, 1)]
for (M, N, K, ta, tb, f32) in SH:
    A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
    B = torch.randn((N, K) if tb else (K, N), device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    for _ in range(3):
        e.gemm(Mat(A, *A.shape), Mat(B, *B.shape), Mat(C, M, N), M, N, K, ta, tb)
    torch.cuda.synchronize()
