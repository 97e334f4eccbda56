"""zk_adam + zk_l2norm on a parameter buffer of the Transformer-base size (76.9 M fp32): microseconds and
achieved HBM bandwidth (30 B per parameter).  usage: python scripts/adam_bench.py   (GPU box)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine
e = Engine("cuda:0")
n = 76_900_000 // 4 * 4
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 0.01
m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
sh = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
hyper = torch.zeros(12, device="cuda"); hyper[:6] = torch.tensor([1e-4, 0.9, 0.98, 1e-8, 1.0, 0.0]); hyper[6] = 1.0
pn = torch.zeros(1, device="cuda")
ws = torch.empty(e.lib.query("zk_norm_workspace") * 2, dtype=torch.uint8, device="cuda")
def run():
    e.lib.call("zk_adam", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), n, hyper.data_ptr(),
               pn.data_ptr(), ws.data_ptr(), ws.numel(), e.stream)
for _ in range(3): run()
torch.cuda.synchronize()
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
t.record(); torch.cuda.synchronize()
us = s.elapsed_time(t) / 20 * 1e3
print("zk_adam: %.1f us  -> %.2f TB/s (30 B/param)" % (us, n * 30 / us / 1e6))

# the single-pass norm-free update of the training step (zk_adam_step), float4 per thread and round = tuning key 9
ws2 = torch.empty(e.lib.query("zk_adam_step_workspace"), dtype=torch.uint8, device="cuda")
seed = torch.zeros(1, dtype=torch.int64, device="cuda")
def run2():
    e.lib.call("zk_adam_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), n, hyper.data_ptr(),
               pn.data_ptr(), seed.data_ptr(), 1, None, ws2.data_ptr(), ws2.numel(), e.stream)
for u, gb in ((0, 0), (1, 0), (2, 0), (4, 0), (2, 1024), (4, 1024), (4, 512), (2, 512), (1, 1024), (0, 0), (1, 0)):
    e.lib.raw("zk_tune")(9, u)
    e.lib.raw("zk_tune")(10, gb)
    for _ in range(3): run2()
    torch.cuda.synchronize()
    s.record()
    for _ in range(20): run2()
    t.record(); torch.cuda.synchronize()
    us = s.elapsed_time(t) / 20 * 1e3
    print("zk_adam_step order/unroll=%d blocks=%d: %.1f us  -> %.2f TB/s (30 B/param)" % (u, gb or 2048, us, n * 30 / us / 1e6))
e.lib.raw("zk_tune")(9, 0)
e.lib.raw("zk_tune")(10, 0)
