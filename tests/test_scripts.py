"""CPU tests of the measurement scripts whose numbers DESIGN.md quotes: the lower-bound budget of the step
(scripts/step_budget.py, needs only the host function zk_gemm_plan of the library) and the PMC aggregation of the decode
leg (scripts/pmc_traffic.py --decode)."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_step_budget_enumerates_the_step():
    sb = _load("step_budget")
    b, nparam = sb.build(64)
    assert abs(nparam - 76.8e6) < 0.2e6                      # Transformer-base, V = 32000, tied target / softmax table
    rows = b.rows
    n = sum(r["n"] for r in rows.values())
    assert 140 <= n <= 152, n                                # the launches of the step (round 4: 58 LayerNorm and 18 attention-forward launches run inside GEMM launches)
    old_rows = sb.build(64, sync_ln=False)[0].rows
    assert 215 <= sum(r["n"] for r in old_rows.values()) <= 230     # rounds 1-3: the hipGraph held 219 nodes
    assert abs(sum(r["flops"] for r in old_rows.values()) - sum(r["flops"] for r in rows.values())) < 1e6
    flops = sum(r["flops"] for r in rows.values())
    assert abs(flops - 1.519e12) < 0.02e12                   # == bench.train_flops_per_step of the same batch
    wg = next(r for k, r in rows.items() if k.startswith("grouped weight gradients"))
    # operands of the one weight-gradient launch: every X [T, in] and dY [T, out] once + the fp32 gradients
    assert 1.45e9 < wg["hbm"] < 1.52e9 and abs(wg["flops"] - 495e9) < 2e9
    floor = sum(r["floor"] for r in rows.values())
    assert 2.7e-3 < floor < 3.1e-3                            # with the measured 1.5-us kernel boundary as the fixed cost
    assert floor < sum(r["floor"] for r in old_rows.values()) < 3.4e-3
    for r in rows.values():                                  # a floor can never be below the fixed cost of its launches
        assert r["floor"] >= r["fixed"]
    # four times the batch: the floor must grow by less than 4x (launch floors amortised) and more than 2x
    b4, _ = sb.build(256)
    f4 = sum(r["floor"] for r in b4.rows.values())
    assert 2.0 * floor < f4 < 4.0 * floor


def test_step_budget_reads_a_kernel_table():
    sb = _load("step_budget")
    path = os.path.join(ROOT, "profiles", "r03_rocprof_kernel_stats_final.txt")
    meas = sb.measured(path)
    total = sum(v[1] for v in meas.values())
    assert 4.0e3 < total < 5.2e3                              # us per step of the kernels of the step
    assert round(meas["Adam"][0]) == 1 and round(meas["grouped weight gradients"][0]) == 1
    assert round(meas["attention forward / backward"][0]) == 36


def test_pmc_traffic_decode_aggregation(tmp_path):
    """two fake counter passes: 10 decode steps (k_beam_prepare x 10) of two kernels"""
    for counter, vals in (("FETCH_SIZE", {"k_a": 100.0, "k_beam_prepare": 1.0}), ("WRITE_SIZE", {"k_a": 40.0, "k_beam_prepare": 0.0})):
        d = tmp_path / counter
        d.mkdir()
        with open(d / "x_counter_collection.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writeheader()
            for step in range(10):
                for k, v in vals.items():
                    for rep in range(3 if k == "k_a" else 1):
                        w.writerow({"Kernel_Name": "void %s(int)" % k, "Counter_Name": counter, "Counter_Value": v})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_traffic.py"), str(tmp_path / "FETCH_SIZE"),
                          str(tmp_path / "WRITE_SIZE"), "--decode"], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout)
    assert res["decode_steps"] == 10
    # per step: 3 launches of k_a x (100 KB x 2 + 40 KB) + k_beam_prepare 1 KB x 2
    assert abs(res["bytes_per_step"] - (3 * (200 + 40) + 2) * 1024.0) < 1e-6
    assert abs(res["fetch_bytes_per_step"] - (3 * 200 + 2) * 1024.0) < 1e-6
