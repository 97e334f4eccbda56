"""Per-shape cost of the in-launch LayerNorm: zk_gemm_add_ln against zk_gemm + zk_add_ln_fwd, and zk_gemm_ln_bwd against
zk_gemm(tb = 1, residual) + zk_add_ln_bwd, at 4096 x 512 x K, 40 operations inside one hipGraph (us per operation, kernel
boundaries included)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat

e = Engine("cuda:0")
M, N = 4096, 512
REP = 40
ctx = torch.cuda.stream(e.work_stream)
ctx.__enter__()
bf = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()
gam, bet, bias = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
mean, rstd = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")


def timed(body):
    body(); torch.cuda.synchronize()
    g = e.graph_capture(lambda: [body() for _ in range(REP)])
    for _ in range(3):
        e.graph_launch(g)
    torch.cuda.synchronize()
    s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        e.graph_launch(g)
    f.record()
    torch.cuda.synchronize()
    return s.elapsed_time(f) * 1e3 / (5 * REP)


for K in (512, 1536, 2048):
    A, W, Wt = bf(M, K), bf(K, N, sc=0.05), bf(N, K, sc=0.05)
    R, Y, S, O = bf(M, N), bf(M, N), bf(M, N), bf(M, N)
    DS, DY = bf(M, N), bf(M, N)
    ws = torch.empty(e.lib.query("zk_add_ln_bwd_workspace", M, N) // 4 + 64 * 3 * N, device="cuda")
    e.ln_epoch_bump()
    two = timed(lambda: (e.gemm(Mat(A, M, K), Mat(W, K, N), Mat(Y, M, N), M, N, K, 0, 0, bias=bias),
                         e.add_ln_fwd(Mat(R, M, N), Mat(Y, M, N), gam, bet, Mat(O, M, N), Mat(S, M, N), mean, rstd, 0.1, 3)))
    def one_f():
        e._sync_site = 0
        e.gemm_add_ln(Mat(A, M, K), Mat(W, K, N), M, N, K, bias, Mat(R, M, N), gam, bet, Mat(O, M, N), Mat(S, M, N), mean, rstd, 0.1, 3)
    one = timed(one_f)
    twob = timed(lambda: (e.gemm(Mat(A, M, K), Mat(Wt, N, K), Mat(Y, M, N), M, N, K, 0, 1, residual=Mat(R, M, N)),
                          e.add_ln_bwd(Mat(Y, M, N), Mat(S, M, N), mean, rstd, gam, Mat(DS, M, N), Mat(DY, M, N), None, None, None, 0.1, 3,
                                       private_ws=ws)))
    def one_b():
        e._sync_site = 0
        e.gemm_ln_bwd(Mat(A, M, K), Mat(Wt, N, K), M, N, K, Mat(R, M, N), Mat(S, M, N), mean, rstd, gam, Mat(DS, M, N), Mat(DY, M, N), ws, 0.1, 3)
    oneb = timed(one_b)
    print("K=%4d  forward: two launches %5.1f us, one %5.1f us   backward: two launches %5.1f us, one %5.1f us" % (K, two, one, twob, oneb))
