"""Ring-depth sweep: is the K loop bound by latency x bytes-in-flight?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat
e = Engine("cuda:0")
T, H, F = 4096, 512, 2048
def run(M, N, K, ta, tb, impl, reps=30):
    ar, ac = ((K, M) if ta else (M, K)); br, bc = ((N, K) if tb else (K, N))
    A = torch.randn(ar, ac, device="cuda").bfloat16(); B = torch.randn(br, bc, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    am, bm, cm = Mat(A, ar, ac), Mat(B, br, bc), Mat(C, M, N)
    for _ in range(3): e.gemm(am, bm, cm, M, N, K, ta, tb, bias=bias, impl=impl)
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): e.gemm(am, bm, cm, M, N, K, ta, tb, bias=bias, impl=impl)
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / reps * 1e3
cfgs = [("64x64 ns2", 4, 2), ("64x64 ns4", 4, 0), ("64x64 ns6", 4, 6), ("64x64 ns8", 4, 8), ("128x64 ns2", 2, 2), ("128x64 ns3", 2, 0),
        ("128x64 ns6", 2, 6), ("128x128 ns2", 1, 2), ("128x128 ns3", 1, 0), ("128x128 ns4", 1, 4)]
print("%-22s" % "M,N,K,ta,tb" + "".join("%13s" % c[0] for c in cfgs))
for (M, N, K, ta, tb) in [(T, H, H, 0, 0), (T, H, H, 0, 1), (T, H, F, 0, 0), (T, H, F, 0, 1), (T, F, H, 0, 0), (T, 3 * H, H, 0, 0)]:
    print("%-22s" % ("%d,%d,%d,%d,%d" % (M, N, K, ta, tb)) + "".join("%10.1f us" % run(M, N, K, ta, tb, 2 | (t << 8) | (ns << 24)) for _, t, ns in cfgs))
