"""Does a non-power-of-two leading dimension (no L2 channel camping) speed the GEMMs up?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat
e = Engine("cuda:0")
T, H, F = 4096, 512, 2048
SHAPES = [("fwd HxH", T, H, H, 0, 0), ("fwd ffn1", T, F, H, 0, 0), ("fwd ffn2", T, H, F, 0, 0), ("dgrad HxH", T, H, H, 0, 1),
          ("dgrad ffn1", T, H, F, 0, 1), ("wgrad HxH", H, H, T, 1, 0), ("wgrad ffn1", H, F, T, 1, 0)]
def run(M, N, K, ta, tb, pad, reps=30):
    ar, ac = ((K, M) if ta else (M, K)); br, bc = ((N, K) if tb else (K, N))
    A = torch.randn(ar, ac + pad, device="cuda").bfloat16(); B = torch.randn(br, bc + pad, device="cuda").bfloat16()
    f32 = ta == 1
    C = torch.empty(M, N + pad, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    am, bm, cm = Mat(A, ar, ac, ac + pad), Mat(B, br, bc, bc + pad), Mat(C, M, N, N + pad)
    for _ in range(3): e.gemm(am, bm, cm, M, N, K, ta, tb)
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): e.gemm(am, bm, cm, M, N, K, ta, tb)
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / reps * 1e3
print("%-12s %-18s" % ("shape", "M,N,K") + "".join("%14s" % ("pad %d" % p) for p in (0, 8, 64, 72, 136)))
for name, M, N, K, ta, tb in SHAPES:
    print("%-12s %-18s" % (name, "%d,%d,%d" % (M, N, K)) + "".join("%8.1f us   " % run(M, N, K, ta, tb, p) for p in (0, 8, 64, 72, 136)))
