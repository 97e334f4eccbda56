"""Would two half-batch chains on two streams beat one full-batch chain?  (round 6 probe)
The launches of the 4096-row chain are bound by ~1-us memory round trips and by the burst each phase puts on the L2 / fabric
(profiles/r06_attn_out_ln_timeline_*): both workgroups of a CU are in the same phase at the same time.  Sentences are
independent all the way down, so the chain could run as two chains of 32 sentences on two streams (two branches of the step's
hipGraph), whose phases drift apart.  This times a stand-in encoder stack -- per layer: projection + attention + o_map +
LayerNorm (one launch), enlarge + ReLU, output product + residual + LayerNorm -- for 64 sentences on one stream against
2 x 32 sentences on two streams, inside one hipGraph each.
usage: python scripts/half_batch_probe.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat  # noqa: E402

nh, L, d, F, NL = 8, 64, 64, 2048, 6
H = nh * d
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.3).to(torch.bfloat16)
f32 = lambda *s: torch.randn(*s, device="cuda") * 0.1


class Stack(object):
    def __init__(self, B):
        self.e = Engine("cuda:0")
        self.B, self.T = B, B * L
        T = self.T
        self.layers = [dict(Wp=bf(H, 3 * H) * 0.1, bp=f32(3 * H), Wo=bf(H, H) * 0.1, bo=f32(H), W1=bf(H, F) * 0.1, b1=f32(F),
                            W2=bf(F, H) * 0.1, b2=f32(H), g=torch.ones(H, device="cuda"), b=torch.zeros(H, device="cuda"))
                       for _ in range(NL)]
        self.x0 = bf(T, H)
        self.qkv, self.att, self.h = bf(T, 3 * H), bf(T, H), bf(T, F)
        self.lse = torch.zeros(B * nh * L, device="cuda")
        self.xs = [torch.empty(T, H, dtype=torch.bfloat16, device="cuda") for _ in range(2 * NL)]

    def run(self):
        e, B, T = self.e, self.B, self.T
        e.ln_epoch_bump()
        x = self.x0
        for i, w in enumerate(self.layers):
            y = self.xs[2 * i]
            ok = e.attn_out_ln(Mat(self.qkv, T, H, 3 * H, 0), Mat(self.qkv, T, H, 3 * H, H), Mat(self.qkv, T, H, 3 * H, 2 * H),
                               Mat(self.att, T, H), self.lse, B, nh, L, L, d, None, False, 0.1, 7, Mat(w["Wo"], H, H), w["bo"],
                               Mat(x, T, H), w["g"], w["b"], Mat(y, T, H), None, None, None, 0.1, 8,
                               proj=(Mat(x, T, H), Mat(w["Wp"], H, 3 * H), w["bp"], 3))
            assert ok
            e.gemm(Mat(y, T, H), Mat(w["W1"], H, F), Mat(self.h, T, F), T, F, H, 0, 0, bias=w["b1"], act=1, drop_p=0.1, sid=9)
            z = self.xs[2 * i + 1]
            e.gemm_add_ln(Mat(self.h, T, F), Mat(w["W2"], F, H), T, H, F, w["b2"], Mat(y, T, H), w["g"], w["b"], Mat(z, T, H),
                          None, None, None, 0.1, 10)
            x = z


def timed(body, e, reps=20):
    with torch.cuda.stream(e.work_stream):
        body()
        torch.cuda.synchronize()
        g = e.graph_capture(body)
        for _ in range(3):
            e.graph_launch(g)
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(reps):
            e.graph_launch(g)
        s1.record()
        torch.cuda.synchronize()
    return s0.elapsed_time(s1) / reps * 1e3


full = Stack(64)
print("one chain of 64 sentences (%d launches): %.1f us" % (3 * NL, timed(full.run, full.e)), flush=True)
a, b = Stack(32), Stack(32)
side = torch.cuda.Stream()


def both():
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        b.run()
        ev2 = torch.cuda.Event()
        ev2.record()
    a.run()
    torch.cuda.current_stream().wait_event(ev2)


print("two chains of 32 sentences on two streams (2 x %d launches): %.1f us" % (3 * NL, timed(both, a.e)), flush=True)
print("one chain of 32 sentences alone: %.1f us" % timed(a.run, a.e), flush=True)
for s in (full, a, b):
    assert s.e.sync_ln_errors() == 0
