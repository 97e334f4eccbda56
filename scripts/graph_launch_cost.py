"""Host cost of one hipGraphLaunch against the node count (is a lane's replay loop bound by the launch call?).
usage: python scripts/graph_launch_cost.py   (GPU box)"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat  # noqa: E402

e = Engine("cuda:0")
M, H = 128, 512
bf = torch.bfloat16


def make(n_nodes, stream):
    x = (torch.randn(M, H, device="cuda") * 0.05).to(bf)
    w = (torch.randn(H, H, device="cuda") * 0.05).to(bf)
    y = torch.empty(M, H, device="cuda", dtype=bf)

    def body():
        for i in range(n_nodes):
            e.gemm(Mat(x, M, H), Mat(w, H, H), Mat(y, M, H), M, H, H, 0, 0)
    with torch.cuda.stream(stream):
        body()
        torch.cuda.synchronize()
        g = e.graph_capture(body)
    return g, (x, w, y)


def run(g, stream, reps, out, i):
    with torch.cuda.stream(stream):
        e.graph_launch(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            e.graph_launch(g)
        t1 = time.perf_counter()
        stream.synchronize()
        t2 = time.perf_counter()
    out[i] = ((t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6)


for nodes in (1, 10, 42, 84):
    for lanes in (1, 4):
        streams = [torch.cuda.Stream() for _ in range(lanes)]
        gs = [make(nodes, s) for s in streams]
        out = [None] * lanes
        th = [threading.Thread(target=run, args=(gs[i][0], streams[i], 200, out, i)) for i in range(lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        print("nodes %3d lanes %d: host us per hipGraphLaunch %s, wall us per replay %s" % (
            nodes, lanes, ["%.1f" % o[0] for o in out], ["%.1f" % o[1] for o in out]), flush=True)
